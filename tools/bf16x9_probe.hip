// Probe ("what comes next"): an fp32-accurate GEMM inner loop on the bf16 MFMA pipe.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/bf16x9_probe.hip -o /tmp/bf && /tmp/bf
// Every fp32 operand x is split EXACTLY into three bf16 pieces, x = h + m + l (8 + 8 + 8 mantissa bits, by truncation),
// and a product a*b becomes up to nine exact bf16 x bf16 products accumulated in fp32 by v_mfma_f32_32x32x16_bf16
// (2.5 PFLOP/s dense on MI355X against 157 TFLOP/s for v_mfma_f32_32x32x2_f32: 16 x the rate, so 9 products cost 9/16
// of the fp32 MFMA time) -- at the price of the split: ~5 VALU instructions per operand element, every chunk.
// Two questions, both answered with the forward kernel's own loop shape (one wave per SIMD, a 32 x 32 accumulator per
// wave, operand chunks of 32 k-rows read from a wave-private LDS tile with ds_read_b32):
//   1. accuracy: C = A.B over K = 2048 against float64, for the fp32 FMA chain (what the engine runs), bf16 x 6 and x 9
//      products into one accumulator, and x 9 into three accumulators by magnitude class (summed at the end);
//   2. speed: the loop alone (LDS-resident operands, no global traffic), 256 workgroups x 4 waves, 64 chunks = K 2048.
// Nothing here is used by the engine; it prices an option.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
// hi halves of (x0, x1) packed: low 16 bits = bf16(x0), high 16 bits = bf16(x1)   (v_perm_b32)
__device__ __forceinline__ unsigned pack_hi(unsigned u0, unsigned u1) { return __builtin_amdgcn_perm(u1, u0, 0x07060302u); }

// x[0..7] (consecutive k of one row / column) -> three bf16x8 fragments, x = h + m + l exactly
__device__ __forceinline__ void split8(const float *x, bf16x8 &H, bf16x8 &M, bf16x8 &L) {
    u32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned u0 = fbits(x0), u1 = fbits(x1);
        const float r0 = x0 - bfloat(u0 & 0xFFFF0000u), r1 = x1 - bfloat(u1 & 0xFFFF0000u);  // exact
        const unsigned v0 = fbits(r0), v1 = fbits(r1);
        const float s0 = r0 - bfloat(v0 & 0xFFFF0000u), s1 = r1 - bfloat(v1 & 0xFFFF0000u);  // exact, <= 8 bits left
        h[p] = pack_hi(u0, u1);
        m[p] = pack_hi(v0, v1);
        l[p] = pack_hi(fbits(s0), fbits(s1));
    }
    H = __builtin_bit_cast(bf16x8, h);
    M = __builtin_bit_cast(bf16x8, m);
    L = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// MODE 0: fp32 MFMA chain; 1: bf16 x 6 products, one accumulator; 2: x 9, one accumulator; 3: x 9, three accumulators
// tile: LDS [32 k][32] of A^T (A[i][k] at k*32+i) then [32 k][32] of B (B[k][j] at k*32+j), like the forward kernel's wave tile
template <int MODE>
__device__ __forceinline__ void chunk(const float *tA, const float *tB, int lane, f32x16 &acc, f32x16 &accm, f32x16 &accl) {
    const int i = lane & 31, h = lane >> 5;
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tA[(2 * u + h) * 32 + i], tB[(2 * u + h) * 32 + i], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 2; s++) {  // two k-steps of 16
            float xa[8], xb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xa[j] = tA[(16 * s + 8 * h + j) * 32 + i];
                xb[j] = tB[(16 * s + 8 * h + j) * 32 + i];
            }
            bf16x8 ah, am, al, bh, bm, bl;
            split8(xa, ah, am, al);
            split8(xb, bh, bm, bl);
            if (MODE == 3) {
                accl = mfma_bf(al, bl, accl);
                accl = mfma_bf(al, bm, accl);
                accl = mfma_bf(am, bl, accl);
                accl = mfma_bf(am, bm, accl);
                accl = mfma_bf(al, bh, accl);
                accl = mfma_bf(ah, bl, accl);
                accm = mfma_bf(am, bh, accm);
                accm = mfma_bf(ah, bm, accm);
                acc = mfma_bf(ah, bh, acc);
            } else {
                if (MODE == 2) {
                    acc = mfma_bf(al, bl, acc);
                    acc = mfma_bf(al, bm, acc);
                    acc = mfma_bf(am, bl, acc);
                }
                acc = mfma_bf(am, bm, acc);
                acc = mfma_bf(al, bh, acc);
                acc = mfma_bf(ah, bl, acc);
                acc = mfma_bf(am, bh, acc);
                acc = mfma_bf(ah, bm, acc);
                acc = mfma_bf(ah, bh, acc);
            }
        }
    }
}

// accuracy: one wave, C[32][32] = A[32][K] . B[K][32]; A is given as At[K][32]
template <int MODE>
__global__ __launch_bounds__(64) void k_acc(const float *__restrict__ At, const float *__restrict__ B, float *__restrict__ C, int K) {
    __shared__ float tile[2048];
    const int lane = threadIdx.x;
    f32x16 acc, accm, accl;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = accm[r] = accl[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        for (int e = lane; e < 1024; e += 64) {
            tile[e] = At[(size_t)k0 * 32 + e];
            tile[1024 + e] = B[(size_t)k0 * 32 + e];
        }
        __syncthreads();
        chunk<MODE>(tile, tile + 1024, lane, acc, accm, accl);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; r++) C[acc_row(r, lane) * 32 + (lane & 31)] = MODE == 3 ? (accl[r] + accm[r]) + acc[r] : acc[r];
}

// speed: NW waves per workgroup, each with its own LDS tile, `chunks` chunks each
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_speed(const float *__restrict__ src, float *__restrict__ out, int chunks) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = lds + wave * 2048;
    for (int e = lane; e < 2048; e += 64) tile[e] = src[e];
    __syncthreads();
    f32x16 acc, accm, accl;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = accm[r] = accl[r] = 0.0f;
    for (int c = 0; c < chunks; c++) {
        chunk<MODE>(tile, tile + 1024, lane, acc, accm, accl);
        asm volatile("" ::: "memory");  // re-read the fragments every chunk, as the real loop does
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r] + accm[r] + accl[r];
    if (s == 12345.678f) out[blockIdx.x] = s;  // never true: keeps the loop alive
}

template <int MODE, int NW>
static int speed(const char *name, const float *src, float *out) {
    const int reps = 50, chunks = 64;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto &e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_speed<MODE, NW>), dim3(256), dim3(64 * NW), NW * 8192, st, src, out, chunks);
    for (int i = 0; i < reps; i++)
        hipExtLaunchKernelGGL((k_speed<MODE, NW>), dim3(256), dim3(64 * NW), NW * 8192, st, ev[2 * i], ev[2 * i + 1], 0, src, out, chunks);
    CK(hipStreamSynchronize(st));
    std::vector<float> us(reps);
    for (int i = 0; i < reps; i++) {
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        us[i] = ms * 1000.f;
    }
    std::sort(us.begin(), us.end());
    printf("  %-44s %d waves/CU: median %7.2f us for %d chunks of 32 k per wave\n", name, NW, us[reps / 2], chunks);
    for (auto &e : ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(st);
    return 0;
}

template <int MODE>
static int accuracy(const char *name, const float *dAt, const float *dB, float *dC, int K, const std::vector<double> &ref,
                    const std::vector<double> &mag) {
    hipLaunchKernelGGL(k_acc<MODE>, dim3(1), dim3(64), 0, 0, dAt, dB, dC, K);
    std::vector<float> C(1024);
    CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
    double worst = 0, mean = 0;
    for (int e = 0; e < 1024; e++) {
        const double err = std::fabs((double)C[e] - ref[e]) / mag[e];  // relative to sum |a||b|: the scale rounding acts on
        worst = std::max(worst, err);
        mean += err / 1024;
    }
    printf("  %-44s max %.2e  mean %.2e   (units of sum_k |a||b|; 2^-24 = 5.96e-08)\n", name, worst, mean);
    return 0;
}

int main() {
    const int K = 2048;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uw(-0.034f, 0.034f), uy(0.0f, 1.0f);  // Gen_rand_net-rule weights, sigmoid outputs
    std::vector<float> At((size_t)K * 32), B((size_t)K * 32);
    for (auto &v : At) v = uw(rng);
    for (auto &v : B) v = uy(rng);
    std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
    for (int k = 0; k < K; k++)
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                const double p = (double)At[(size_t)k * 32 + i] * (double)B[(size_t)k * 32 + j];
                ref[i * 32 + j] += p;
                mag[i * 32 + j] += std::fabs(p);
            }
    float *dAt, *dB, *dC, *dout;
    CK(hipMalloc((void **)&dAt, At.size() * 4));
    CK(hipMalloc((void **)&dB, B.size() * 4));
    CK(hipMalloc((void **)&dC, 4096));
    CK(hipMalloc((void **)&dout, 256 * 4));
    CK(hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    printf("accuracy, C[32][32] = A.B over K = %d (weights U(+-0.034), activations U(0,1)) against float64:\n", K);
    if (accuracy<0>("fp32 MFMA 32x32x2 chain (the engine)", dAt, dB, dC, K, ref, mag)) return 1;
    if (accuracy<1>("bf16 x 6 products, one accumulator", dAt, dB, dC, K, ref, mag)) return 1;
    if (accuracy<2>("bf16 x 9 products, one accumulator", dAt, dB, dC, K, ref, mag)) return 1;
    if (accuracy<3>("bf16 x 9 products, three accumulators", dAt, dB, dC, K, ref, mag)) return 1;
    printf("speed of the loop alone (operands in LDS, 256 workgroups, K = 2048 per wave = the 2048^2 forward main loop x 4):\n");
    if (speed<0, 4>("fp32 MFMA 32x32x2 chain", dAt, dout)) return 1;
    if (speed<1, 4>("bf16 x 6, split on the fly", dAt, dout)) return 1;
    if (speed<2, 4>("bf16 x 9, split on the fly", dAt, dout)) return 1;
    if (speed<3, 4>("bf16 x 9, three accumulators", dAt, dout)) return 1;
    if (speed<0, 8>("fp32 MFMA 32x32x2 chain", dAt, dout)) return 1;
    if (speed<1, 8>("bf16 x 6, split on the fly", dAt, dout)) return 1;
    if (speed<2, 8>("bf16 x 9, split on the fly", dAt, dout)) return 1;
    return 0;
}
