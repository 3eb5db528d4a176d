// Micro-benchmark: what does a 2-way cross-workgroup split-K seam cost INSIDE a forward / dX launch?
//   hipcc --offload-arch=gfx950 -O3 tools/splitk_seam.hip -o /tmp/seam && /tmp/seam
// The variant under discussion: a 64(n) x 32(b) output tile per PAIR of workgroups (two accumulators per wave sharing
// each Yt fragment: -25 % operand bytes per CU), the K range split over the two workgroups of the pair, the later
// arriver adding the earlier one's partial tile in a fixed order (pair member 0 + member 1: deterministic) before
// the epilogue.  Here: 256 workgroups of 256 threads = 128 pairs with equal blockIdx % 8 (one XCD under round-robin
// placement), each workgroup runs `mfmas` dependent v_mfma_f32_32x32x2_f32 per wave (the 2048^2 forward main loop is
// 256 per wave = 8 us), then
//   plain   every workgroup stores its own 4 KB tile (what a launch without a seam does)
//   seam    member stores its 4 KB partial with sc1 (write-through) 16-byte stores -> s_waitcnt vmcnt(0) -> barrier ->
//           one agent-scope ticket add; the workgroup that draws 1 loads the partner's partial with sc1 loads, adds
//           (member 0 first), stores the tile, resets the ticket  (cdna_hip_programming.md, in-launch split-K recipe)
// Launch time by the dispatch's own begin / end timestamps (hipExtLaunchKernelGGL events), median of 200.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
constexpr int NWG = 256, NT = 256;

template <bool SEAM>
__global__ __launch_bounds__(NT) void k_tile(float *__restrict__ slab, unsigned *__restrict__ ticket, float *__restrict__ out,
                                             int mfmas, float seed, unsigned *bad) {
    __shared__ unsigned drew;
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int pair = ((bid >> 4) << 3) | (bid & 7), member = (bid >> 3) & 1;  // partners: bid and bid ^ 8
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float a = seed + tid * 1e-6f, b = 1.0f + member;
    for (int i = 0; i < mfmas; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float4 v = make_float4(acc[0] + member + 1.0f, acc[1] + tid, acc[2], acc[3]);  // this thread's 16 bytes of the 4 KB tile
    if (!SEAM) {
        *reinterpret_cast<float4 *>(out + (size_t)bid * 1024 + tid * 4) = v;
        return;
    }
    const rsrc_t rs = make_rsrc(slab, NWG * 4096u);
    // sc1 = bit 4 of the aux / cache-policy operand: write-through past the XCD's L2 (no release fence needed)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs,
                                           (pair * 2 + member) * 4096 + tid * 16, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) drew = __hip_atomic_fetch_add(&ticket[pair * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (drew == 0) return;  // the partner combines
    // later arriver: both partials with sc1 loads (its own too: one code path, fixed order member 0 + member 1)
    const float4 p0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (pair * 2 + 0) * 4096 + tid * 16, 0, 16));
    const float4 p1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (pair * 2 + 1) * 4096 + tid * 16, 0, 16));
    const float4 s = make_float4(p0.x + p1.x, p0.y + p1.y, p0.z + p1.z, p0.w + p1.w);
    if (mfmas == 0 && (p0.x != 1.0f || p1.x != 2.0f)) atomicAdd(bad, 1u);  // a stale partial would show here
    *reinterpret_cast<float4 *>(out + (size_t)pair * 1024 + tid * 4) = s;
    if (tid == 0) __hip_atomic_store(&ticket[pair * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
}

template <bool SEAM>
static int run(const char *name, int mfmas, float *slab, unsigned *ticket, float *out, unsigned *bad, double *med_out) {
    const int reps = 200;
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto &e : ev) CK(hipEventCreate(&e));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_tile<SEAM>, dim3(NWG), dim3(NT), 0, st, slab, ticket, out, mfmas, 0.0f, bad);
    for (int i = 0; i < reps; i++)
        hipExtLaunchKernelGGL(k_tile<SEAM>, dim3(NWG), dim3(NT), 0, st, ev[2 * i], ev[2 * i + 1], 0, slab, ticket, out, mfmas, 0.0f, bad);
    CK(hipStreamSynchronize(st));
    std::vector<float> us(reps);
    for (int i = 0; i < reps; i++) {
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        us[i] = ms * 1000.0f;
    }
    std::sort(us.begin(), us.end());
    printf("  %-6s %3d MFMAs per wave: median %6.2f us  p10 %6.2f  p90 %6.2f\n", name, mfmas, us[reps / 2], us[reps / 10], us[reps * 9 / 10]);
    *med_out = us[reps / 2];
    for (auto &e : ev) hipEventDestroy(e);
    hipStreamDestroy(st);
    return 0;
}

int main() {
    float *slab, *out;
    unsigned *ticket, *bad;
    CK(hipMalloc((void **)&slab, NWG * 4096));
    CK(hipMalloc((void **)&out, NWG * 4096));
    CK(hipMalloc((void **)&ticket, 128 * 32 * sizeof(unsigned)));  // one 128-byte line per pair
    CK(hipMalloc((void **)&bad, sizeof(unsigned)));
    CK(hipMemset(ticket, 0, 128 * 32 * sizeof(unsigned)));
    CK(hipMemset(bad, 0, sizeof(unsigned)));
    printf("2-way cross-workgroup split-K seam on a 4 KB tile, 256 workgroups (128 same-XCD pairs), sc1 slabs + ticket:\n");
    for (int mf : {0, 128, 256}) {
        double a = 0, b = 0;
        if (run<false>("plain", mf, slab, ticket, out, bad, &a)) return 1;
        if (run<true>("seam", mf, slab, ticket, out, bad, &b)) return 1;
        printf("  -> the seam adds %.2f us to a launch whose main loop is %d MFMAs per wave\n", b - a, mf);
    }
    unsigned hb = 0;
    CK(hipMemcpy(&hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    printf("stale partials seen by a combiner (mfmas = 0 runs): %u\n", hb);
    return 0;
}
