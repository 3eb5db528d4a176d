// Micro-benchmark: what does the momentum update's memory traffic cost by itself on this chip?
// Reads W and delta, writes both (16 B per weight), over the baseline net's 14.8 M weights
// (118 MB working set), back to back.  hipcc --offload-arch=gfx950 -O3 tools/update_stream.hip -o /tmp/us && /tmp/us
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_update(float4 *__restrict__ w, float4 *__restrict__ d, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = w[i], b = d[i];
        b.x = 0.9f * b.x - 0.1f * (1e-5f * a.x); b.y = 0.9f * b.y - 0.1f * (1e-5f * a.y);
        b.z = 0.9f * b.z - 0.1f * (1e-5f * a.z); b.w = 0.9f * b.w - 0.1f * (1e-5f * a.w);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        d[i] = b; w[i] = a;
    }
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ w, const float4 *__restrict__ d, size_t n4, float *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = w[i], b = d[i];
        s += a.x + b.x + a.y + b.y + a.z + b.z + a.w + b.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_spoil(float4 *__restrict__ p, size_t n4) {  // other traffic between two updates
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { float4 a = p[i]; a.x += 1; p[i] = a; }
}
// the same update over a [K][N] row-major matrix walked in 64x64 tiles (the dW kernel's footprint):
// PAT 0: a wave instruction covers 4 rows x 256 B; PAT 1: 32 rows x 32 B (accumulator layout)
template <int PAT>
__global__ __launch_bounds__(256) void k_update_tiles(float *__restrict__ w, float *__restrict__ d, int K, int N) {
    const int n_wg = N / 64, tiles = (K / 64) * n_wg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int k0 = (t / n_wg) * 64, n0 = (t % n_wg) * 64;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            size_t o;
            if (PAT == 0) o = (size_t)(k0 + (tid >> 4) + 16 * q) * N + n0 + 4 * (tid & 15);
            else o = (size_t)(k0 + 32 * (wave >> 1) + (lane & 31)) * N + n0 + 32 * (wave & 1) + 8 * q + 4 * (lane >> 5);
            float4 a = *reinterpret_cast<float4 *>(w + o), b = *reinterpret_cast<float4 *>(d + o);
            b.x = 0.9f * b.x - 0.1f * (1e-5f * a.x); b.y = 0.9f * b.y - 0.1f * (1e-5f * a.y);
            b.z = 0.9f * b.z - 0.1f * (1e-5f * a.z); b.w = 0.9f * b.w - 0.1f * (1e-5f * a.w);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            *reinterpret_cast<float4 *>(d + o) = b; *reinterpret_cast<float4 *>(w + o) = a;
        }
    }
}
int main() {
    const size_t n = 14800000 / 4 * 4, n4 = n / 4;
    float4 *w, *d, *junk; float *out;
    CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&out, 64));
    const size_t junk4 = (size_t)64 << 20;  // 1 GiB
    CK(hipMalloc(&junk, junk4 * 16));
    CK(hipMemset(w, 0, n * 4)); CK(hipMemset(d, 0, n * 4)); CK(hipMemset(junk, 0, junk4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9f, sum = 0; const int reps = 20;
            for (int r = 0; r < reps + 2; r++) {
                if (mode == 2) hipLaunchKernelGGL(k_spoil, dim3(4096), dim3(256), 0, 0, junk, junk4);
                CK(hipEventRecord(e0));
                if (mode == 1) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, w, d, n4, out);
                else hipLaunchKernelGGL(k_update, dim3(grid), dim3(256), 0, 0, w, d, n4);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) { sum += ms; if (ms < best) best = ms; }
            }
            const double bytes = (mode == 1 ? 8.0 : 16.0) * n;
            printf("grid %5d %-28s mean %7.2f us  best %7.2f us  -> %.2f TB/s (best)\n", grid,
                   mode == 0 ? "update back-to-back" : mode == 1 ? "read-only back-to-back" : "update after 1 GiB of traffic",
                   sum / reps * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    {   // tile walks over a 7168 x 2048 matrix (14.7 M weights)
        const int K = 7168, N = 2048;
        for (int grid : {512, 1024, 2048, 7168 / 64 * 32}) {
            for (int pat = 0; pat < 2; pat++) {
                float best = 1e9f, sum = 0; const int reps = 20;
                for (int r = 0; r < reps + 2; r++) {
                    CK(hipEventRecord(e0));
                    if (pat == 0) hipLaunchKernelGGL(k_update_tiles<0>, dim3(grid), dim3(256), 0, 0, (float *)w, (float *)d, K, N);
                    else hipLaunchKernelGGL(k_update_tiles<1>, dim3(grid), dim3(256), 0, 0, (float *)w, (float *)d, K, N);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) { sum += ms; if (ms < best) best = ms; }
                }
                printf("grid %5d tiles 64x64 pattern %d back-to-back  mean %7.2f us  best %7.2f us -> %.2f TB/s\n", grid, pat,
                       sum / reps * 1e3, best * 1e3, 16.0 * K * N / (best * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
