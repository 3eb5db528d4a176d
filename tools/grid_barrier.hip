// Micro-benchmark: what would fusing the step's GEMM launches into ONE persistent kernel with grid-wide barriers buy?
// hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.hip -o /tmp/gb && /tmp/gb
// A "phase" has the memory shape of one 2048x2048 forward layer at B = 128 on 256 workgroups of 512 threads: every
// workgroup streams a 256 KB column block of that phase's weights (a different 16 MB matrix per phase, 4 workgroups per block), reads a 256 KB
// column block of the previous phase's 1 MB output (written by 64 other workgroups, on all XCDs), and writes its own
// 4 KB output tile.  No MFMA work: the difference between the variants is what the boundary costs.
//   chain        P launches of k_phase on one stream (what the engine does today)
//   persistent   one launch, an atomic-counter grid barrier (agent-scope release / acquire) between phases
//   persistent+W the same, and the weights of phase p+1 are fetched BEFORE the barrier (they do not depend on phase p)
// Every spin loop is bounded (a barrier that is not reached sets a flag and the kernel still drains).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NWG = 256, NT = 512, K = 2048, BF = 128;  // X: [K][BF] floats = 1 MB

template <int NL> struct WRegT { float4 v[NL]; };

template <int NL> __device__ __forceinline__ void load_w(WRegT<NL> &w, const float4 *__restrict__ W, int wg) {
    const float4 *p = W + (size_t)(wg >> 2) * 16384;  // 256 KB column block, shared by the 4 frame tiles
#pragma unroll
    for (int i = 0; i < NL; i++) w.v[i] = p[threadIdx.x + NT * i];
}
template <int NL> __device__ __forceinline__ float sum_w(const WRegT<NL> &w) {
    float s = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) s += w.v[i].x + w.v[i].y + w.v[i].z + w.v[i].w;
    return s;
}
// reads X[:, 32 frames of tile column tm] (2048 rows x 128 B), returns this thread's partial sum
template <int NL> __device__ __forceinline__ float read_x(const float *__restrict__ X, int tm) {
    float s = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int q = threadIdx.x + NT * i;  // 16384 float4: row = q >> 3, c4 = q & 7
        const float4 v = *(const float4 *)(X + (size_t)(q >> 3) * BF + tm * 32 + (q & 7) * 4);
        s += v.x + v.y + v.z + v.w;
    }
    return s;
}
__device__ __forceinline__ void write_tile(float *__restrict__ Xo, int tm, int tn, float val) {
    // 32 rows (k of the next phase) x 32 frames; 256 threads write one float4 each
    if (threadIdx.x < 256) {
        const int r = threadIdx.x >> 3, c4 = threadIdx.x & 7;
        *(float4 *)(Xo + (size_t)(tn * 32 + r) * BF + tm * 32 + c4 * 4) = make_float4(val, val, val, val);
    }
}
__device__ __forceinline__ float wg_sum(float s, float *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0;
    for (int i = 0; i < NT / 64; i++) t += sh[i];
    __syncthreads();
    return t;
}
__device__ __forceinline__ float phase_value(float sw, float sx) { return (sw + sx) * (1.0f / 524288.0f) + 1.0f; }

template <int NL>
__global__ __launch_bounds__(NT) void k_phase(const float4 *__restrict__ W, const float *__restrict__ Xi, float *__restrict__ Xo) {
    __shared__ float sh[NT / 64];
    const int wg = blockIdx.x, tm = wg & 3, tn = wg >> 2;
    WRegT<NL> w; load_w(w, W, wg);
    const float sx = read_x<NL>(Xi, tm);
    const float t = wg_sum(sum_w(w) + sx, sh);
    write_tile(Xo, tm, tn, phase_value(t, 0.0f));
}

template <int NL, bool PREFETCH, bool FENCE = true>
__global__ __launch_bounds__(NT) void k_persistent(const float4 *__restrict__ Wall, float *__restrict__ X0, float *__restrict__ X1,
                                                   int phases, unsigned *ctr, int *timeout_flag) {
    __shared__ float sh[NT / 64];
    const int wg = blockIdx.x, tm = wg & 3, tn = wg >> 2;
    WRegT<NL> w;
    if (PREFETCH) load_w(w, Wall, wg);
    for (int p = 0; p < phases; p++) {
        const float *Xi = (p & 1) ? X1 : X0; float *Xo = (p & 1) ? X0 : X1;
        if (!PREFETCH) load_w(w, Wall + (size_t)p * (NWG / 4) * 16384, wg);
        const float sx = read_x<NL>(Xi, tm);
        const float t = wg_sum(sum_w(w) + sx, sh);
        write_tile(Xo, tm, tn, phase_value(t, 0.0f));
        if (p + 1 == phases) break;
        if (PREFETCH) load_w(w, Wall + (size_t)(p + 1) * (NWG / 4) * 16384, wg);  // independent of this phase's output
        // grid barrier: release our tile, count in, wait for everybody, acquire
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(p + 1) * NWG;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 20)) { *timeout_flag = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main() {
    const int P = 7;  // 4 forward + 3 dX launches per step
    float4 *W; float *X0, *X1; unsigned *ctr; int *flag;
    const size_t w4 = (size_t)P * (NWG / 4) * 16384;  // 16 MB per phase
    CK(hipMalloc(&W, w4 * 16)); CK(hipMalloc(&X0, K * BF * 4)); CK(hipMalloc(&X1, K * BF * 4));
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&flag, 4));
    std::vector<float> hw(w4 * 4);
    for (size_t i = 0; i < hw.size(); i++) hw[i] = (float)((i * 2654435761u >> 20) & 7) * 0.125f;
    CK(hipMemcpy(W, hw.data(), w4 * 16, hipMemcpyHostToDevice));
    std::vector<float> x0(K * BF, 1.0f), ref(K * BF), got(K * BF);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    for (int mode = 0; mode < 8; mode++) {
        const bool full = mode < 3 || mode == 6; const int m3 = mode >= 6 ? 1 : mode % 3;
        auto kc = full ? k_phase<32> : k_phase<1>;
        auto kp0 = mode == 6 ? k_persistent<32, false, false> : mode == 7 ? k_persistent<1, false, false>
                   : full ? k_persistent<32, false> : k_persistent<1, false>;
        auto kp1 = full ? k_persistent<32, true> : k_persistent<1, true>;
        float sum = 0, best = 1e9f;
        for (int r = 0; r < reps + 3; r++) {
            CK(hipMemcpy(X0, x0.data(), K * BF * 4, hipMemcpyHostToDevice));
            CK(hipMemset(X1, 0, K * BF * 4)); CK(hipMemset(ctr, 0, 4)); CK(hipMemset(flag, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (m3 == 0) {
                for (int p = 0; p < P; p++)
                    hipLaunchKernelGGL(kc, dim3(NWG), dim3(NT), 0, 0, (const float4 *)W + (size_t)p * (NWG / 4) * 16384,
                                       (const float *)((p & 1) ? X1 : X0), (p & 1) ? X0 : X1);
            } else if (m3 == 1) {
                hipLaunchKernelGGL(kp0, dim3(NWG), dim3(NT), 0, 0, (const float4 *)W, X0, X1, P, ctr, flag);
            } else {
                hipLaunchKernelGGL(kp1, dim3(NWG), dim3(NT), 0, 0, (const float4 *)W, X0, X1, P, ctr, flag);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        int hflag = 0; CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
        float *Xf = (P & 1) ? X1 : X0;
        CK(hipMemcpy(m3 == 0 ? ref.data() : got.data(), Xf, K * BF * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        if (m3 != 0) for (size_t i = 0; i < ref.size(); i++) bad += ref[i] != got[i];
        const char *names[3] = {"chain of 7 launches", mode >= 6 ? "persistent, barrier WITHOUT the fences (timing only)" : "persistent, grid barrier", "persistent, grid barrier, next W fetched before it"};
        printf("%s %-52s mean %7.2f us  best %7.2f us  = %5.2f us per phase%s%s\n", full ? "512 KB per workgroup and phase:" : "16 KB per workgroup and phase: ", names[m3], sum / reps * 1e3, best * 1e3,
               sum / reps * 1e3 / P, m3 && bad ? "  RESULT DIFFERS" : m3 ? "  (same result)" : "", hflag ? "  BARRIER TIMED OUT" : "");
    }
    return 0;
}
