// Micro-benchmark: does a line a kernel left in an XCD's L2 survive the kernel boundary?  (Decides whether the
// epilogue of kernel l can usefully pre-load the first weight chunks of kernel l+1.)
// hipcc --offload-arch=gfx950 -O3 tools/l2_persist.hip -o /tmp/l2p && /tmp/l2p
// 256 blocks x 64 KB = 16 MB (2 MB per XCD if blocks are dealt round-robin): kernel "touch" loads its slice,
// kernel "read" loads the slice of block (b + shift) % 256 and is timed:
//   cold          after 1 GiB of other traffic
//   same block    right after touch, shift 0   (same XCD if the block -> XCD deal repeats between launches)
//   other XCD     right after touch, shift 1   (a slice another XCD touched: L2 miss, Infinity-Cache hit)
//   same XCD      right after touch, shift 8   (another block's slice on the same XCD)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ buf, int shift, float *out) {
    const int nb = gridDim.x, b = (blockIdx.x + shift) % nb;
    const float4 *p = buf + (size_t)b * 4096;  // 64 KB = 4096 float4
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { float4 v = p[threadIdx.x + 256 * i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_spoil(float4 *__restrict__ p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { float4 a = p[i]; a.x += 1; p[i] = a; }
}
int main() {
    const int nb = 256; const size_t n4 = (size_t)nb * 4096;
    float4 *buf, *junk; float *out;
    CK(hipMalloc(&buf, n4 * 16)); CK(hipMalloc(&out, 64));
    const size_t junk4 = (size_t)64 << 20;
    CK(hipMalloc(&junk, junk4 * 16));
    CK(hipMemset(buf, 0, n4 * 16)); CK(hipMemset(junk, 0, junk4 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[4] = {"cold (after 1 GiB of other traffic)", "same block's slice, right after touch", "slice touched on ANOTHER XCD (shift 1)", "another block's slice, SAME XCD (shift 8)"};
    for (int mode = 0; mode < 4; mode++) {
        float best = 1e9f, sum = 0; const int reps = 20;
        for (int r = 0; r < reps + 2; r++) {
            hipLaunchKernelGGL(k_spoil, dim3(4096), dim3(256), 0, 0, junk, junk4);
            if (mode != 0) hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, buf, 0, out);  // touch
            const int shift = mode == 2 ? 1 : mode == 3 ? 8 : 0;
            hipExtLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, 0, e0, e1, 0, (const float4 *)buf, shift, out);
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-46s mean %6.2f us  best %6.2f us  (16 MB -> %.1f TB/s best)\n", names[mode], sum / reps * 1e3, best * 1e3, 16.777216e6 / (best * 1e-3) / 1e12);
    }
    return 0;
}
