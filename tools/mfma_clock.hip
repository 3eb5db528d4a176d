// Micro-benchmark (diagnostic): sustained rate and in-kernel clock of fp32 MFMA chains on all CUs.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void chain32(float *out, long long *clk, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
        asm volatile("" : "+v"(x), "+v"(y));
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void chain16(float *out, long long *clk, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 4; r++) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; u++)   // 32 x (16x16x4) = 16 x (32x32x2) in FLOPs
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
        asm volatile("" : "+v"(x), "+v"(y));
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 4; r++) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <typename K>
void run(const char *name, K kern, int blocks, int iters, double flop_per_iter_per_wave) {
    float *out; long long *clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk, iters); hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(2 * blocks); hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    double ghz = cyc / rt * 0.1;  // realtime ticks are 100 MHz
    double tf = flop_per_iter_per_wave * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;
    printf("%-22s blocks %4d: %8.1f us  %6.1f TFLOP/s  in-kernel clock %.2f GHz  cycles/iter %.1f\n", name, blocks,
           ms * 1e3, tf, ghz, cyc / blocks / iters);
    hipFree(out); hipFree(clk);
}
int main() {
    const int iters = 2000;  // x16 MFMA(32x32x2) per iteration
    const double fl = 16 * 4096.0;
    for (int blocks : {256, 512}) {
        run("32x32x2 1 acc", chain32<1>, blocks, iters, fl);
        run("32x32x2 4 acc", chain32<4>, blocks, iters, fl);
        run("16x16x4 1 acc", chain16<1>, blocks, iters, fl);
        run("16x16x4 4 acc", chain16<4>, blocks, iters, fl);
        run("16x16x4 8 acc", chain16<8>, blocks, iters, fl);
    }
    // short kernels like the trainer's (256 MFMAs per wave)
    run("32x32x2 1 acc short", chain32<1>, 256, 16, fl);
    run("16x16x4 4 acc short", chain16<4>, 256, 16, fl);
    return 0;
}
