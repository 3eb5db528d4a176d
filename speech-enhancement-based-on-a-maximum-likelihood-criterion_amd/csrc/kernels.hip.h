// kernels.hip.h -- hand-written gfx950 (CDNA4) kernels for the ML-GGD DNN trainer.
//
// Replaces the reference's DevFunc.cu kernels + cuBLAS sgemm calls on the train/CV path
// (Train_code_ML_GGD/DevFunc.cu, DevFunc.h:49-87, call sites BP_GPU.cu:334-438,467-509).
//
// Device data layout (see DESIGN.md "HBM layout"): every unit dimension is padded to a
// multiple of 32 (Kp, Np) and the minibatch to Bp = ceil32(B); pad regions are kept at
// exactly 0 so the GEMM main loops need no masks.  Activations are kept in BOTH layouts:
//   Yt [units][Bp]  (frame index contiguous)  -> operand of the forward / dX MFMAs
//   Y  [Bp][units]  (unit index contiguous)   -> operand of the dW MFMAs
// so that every MFMA operand fragment (lane = non-reduction index) is a coalesced 128-byte
// row segment.  Weights stay in the reference's [in][out] row-major order (padded).
//
// All GEMMs use v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).
// Fragment maps (cdna_hip_programming.md section 3): A: lane l holds A[i=l&31][k=l>>5];
// B: lane l holds B[k=l>>5][j=l&31]; C/D reg r: row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Workgroups that share a weight tile (the b_tiles frame-blocks of one unit-tile) are given
// ids that are equal mod 8 and adjacent in dispatch order, so they land on one XCD and the
// tile is fetched from HBM/MALL once and re-read from that XCD's L2 (speed only; any
// placement is correct).
__device__ __forceinline__ void tile_of_block(int id, int u_tiles, int b_tiles, int &ut, int &bt) {
    if ((u_tiles & 7) == 0) {
        const int xcd = id & 7, slot = id >> 3;
        ut = xcd + 8 * (slot / b_tiles);
        bt = slot % b_tiles;
    } else {
        ut = id / b_tiles;
        bt = id % b_tiles;
    }
}

template <int U>
struct FragU {
    float a[U];
    float b[U];
};

// Buffer-resource loads: per-lane byte offset in a VGPR (fixed for the whole kernel), the
// moving part of the address in an SGPR, out-of-range reads return 0 (hardware range check).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, size_t bytes) {
    const unsigned n = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 bload4(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// wave index as a provably wave-uniform value (scalar branches, exact s_waitcnt counts)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---------------------------------------------------------------------------------------
// Forward GEMM + bias + sigmoid:   X^T[n][b] = sum_k W[k][n] * Yt_in[k][b]  (+ bias[n])
// replaces kernMultiCopy + cublasSgemm(N,N) + kernSigmoid (BP_GPU.cu:360-364, DevFunc.cu:36-51,
// 134-149).  One workgroup = one 32(n) x 32(b) output tile; its 4 waves (one per SIMD) split
// the reduction K four ways and are summed through LDS in fixed order (deterministic).
// MODE FWD_SIGMOID: fused epilogue writes y=1/(1+expf(-x)) (0 for pad units) to Yt_out and Y_out.
// MODE FWD_SLAB   : inter-workgroup K split S (small output layers); writes raw partial sums
//                   slab[s][n][b]; the consumer adds the bias and the S slabs in order.
// ---------------------------------------------------------------------------------------
enum { FWD_SIGMOID = 0, FWD_SLAB = 1 };

template <int MODE>
__global__ __launch_bounds__(256) void k_fwd(const float *__restrict__ W, const float *__restrict__ Yt_in,
                                             const float *__restrict__ bias, float *__restrict__ Yt_out,
                                             float *__restrict__ Y_out, float *__restrict__ slab, int Kp, int Np,
                                             int Bp, int N, int n_tiles, int b_tiles, int S) {
    __shared__ float red[4][1024];
    __shared__ float tileT[32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, h = lane >> 5;
    int id = blockIdx.x, s = 0;
    if (MODE == FWD_SLAB) {
        s = id % S;
        id /= S;
    }
    int nt, bt;
    tile_of_block(id, n_tiles, b_tiles, nt, bt);
    const int n0 = nt * 32, b0 = bt * 32;

    // k-pairs of this wave: slot = s*4+wave of S*4 slots over Kp/2 pairs
    const int P = Kp >> 1;
    const int slot = s * 4 + wave, nslots = S * 4;
    const int p0 = (int)((long)P * slot / nslots), p1 = (int)((long)P * (slot + 1) / nslots);

    const rsrc_t rW = make_rsrc(W, (size_t)Kp * Np * 4), rY = make_rsrc(Yt_in, (size_t)Kp * Bp * 4);
    const int voW = (h * Np + n0 + i) * 4, voY = (h * Bp + b0 + i) * 4;
    const int wstep = 2 * Np * 4, ystep = 2 * Bp * 4;  // bytes per k-pair

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    // 3-deep register ring of U k-pairs: 2*U*3 = 48 loads in flight (vmcnt is 6 bits),
    // i.e. ~2 chunks = 16 MFMAs = 1024 cycles of lookahead per wave.
    constexpr int U = 8;
    const int nfull = (p1 - p0) / U;
    FragU<U> f0, f1, f2;
#define FWD_LOAD(F, C)                                              \
    {                                                               \
        const int pb = p0 + (C)*U;                                  \
        _Pragma("unroll") for (int u = 0; u < U; u++) {             \
            F.a[u] = bload(rW, voW, (pb + u) * wstep);              \
            F.b[u] = bload(rY, voY, (pb + u) * ystep);              \
        }                                                           \
    }
#define FWD_COMPUTE(F) \
    { _Pragma("unroll") for (int u = 0; u < U; u++) acc = mfma32(F.a[u], F.b[u], acc); }

    if (nfull > 0) FWD_LOAD(f0, 0);
    if (nfull > 1) FWD_LOAD(f1, 1);
    int c = 0;
    // steady state: every load below is unconditional, so the compiler's s_waitcnt vmcnt(N)
    // before each MFMA leaves the two younger chunks in flight
    for (; c + 5 <= nfull; c += 3) {
        FWD_LOAD(f2, c + 2);
        FWD_COMPUTE(f0);
        FWD_LOAD(f0, c + 3);
        FWD_COMPUTE(f1);
        FWD_LOAD(f1, c + 4);
        FWD_COMPUTE(f2);
    }
    // drain: at most 4 chunks left; f0 = chunk c, f1 = chunk c+1
    if (c < nfull) {
        if (c + 2 < nfull) FWD_LOAD(f2, c + 2);
        FWD_COMPUTE(f0);
    }
    if (c + 1 < nfull) {
        if (c + 3 < nfull) FWD_LOAD(f0, c + 3);
        FWD_COMPUTE(f1);
    }
    if (c + 2 < nfull) FWD_COMPUTE(f2);
    if (c + 3 < nfull) FWD_COMPUTE(f0);
    // tail pairs (< U)
    for (int p = p0 + nfull * U; p < p1; p++) {
        const float a = bload(rW, voW, p * wstep), b = bload(rY, voY, p * ystep);
        acc = mfma32(a, b, acc);
    }
#undef FWD_LOAD
#undef FWD_COMPUTE

    // cross-wave reduction through LDS (fixed order w = 0,1,2,3)
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][acc_row(r, lane) * 32 + i] = acc[r];
    __syncthreads();
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = tid + 256 * q;
        v[q] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
    if (MODE == FWD_SLAB) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 256 * q, row = e >> 5, col = e & 31;
            slab[((size_t)s * Np + n0 + row) * Bp + b0 + col] = v[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 256 * q, row = e >> 5, col = e & 31;
            const int n = n0 + row;
            const float x = v[q] + bias[n];
            const float y = (n < N) ? 1.0f / (1.0f + expf(-x)) : 0.0f;  // kernSigmoid, DevFunc.cu:48
            Yt_out[(size_t)n * Bp + b0 + col] = y;
            tileT[col][row] = y;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + 256 * q, bl = e >> 5, nl = e & 31;
            Y_out[(size_t)(b0 + bl) * Np + n0 + nl] = tileT[bl][nl];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Backward-data GEMM + sigmoid derivative:
//   dEdY^T[k][b] = sum_n W[k][n] * dEdXt[n][b] ;  dEdX_prev = (1-y)*y*dEdY
// replaces cublasSgemm(T,N) (BP_GPU.cu:430, DevFunc.h:49-63) + kernDsigmoid of the layer
// below (BP_GPU.cu:402, DevFunc.cu:53-71).  W is read with OLD values (launched before the
// update of the same layer).  The reduction index n is the contiguous index of W, so each
// wave stages its own [32 k][64 n] piece of W through LDS in full 256-byte row segments and
// reads the A fragments back transposed (row stride 66 floats: conflict-free ds_read_b64).
// ---------------------------------------------------------------------------------------
#define DX_LDW 66
__global__ __launch_bounds__(256) void k_dx(const float *__restrict__ W, const float *__restrict__ dEdXt,
                                            const float *__restrict__ Yt_prev, float *__restrict__ dEdXt_prev,
                                            float *__restrict__ dEdX_prev, int Kp, int Np, int Bp, int k_tiles,
                                            int b_tiles) {
    __shared__ __attribute__((aligned(16))) float wbuf[4][32 * DX_LDW];
    __shared__ float red[4][1024];
    __shared__ float tileT[32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, h = lane >> 5;
    int kt, bt;
    tile_of_block(blockIdx.x, k_tiles, b_tiles, kt, bt);
    const int k0 = kt * 32, b0 = bt * 32;

    const int Q = Np >> 2;   // quads of 4 consecutive n
    const int qw = Q >> 2;   // quads per wave (Np % 32 == 0 -> exact)
    const int q0 = wave * qw;
    const int nch = (qw + 15) >> 4;

    const int c4 = lane & 15, r0 = lane >> 4;
    float *wb = wbuf[wave];

    const rsrc_t rW = make_rsrc(W, (size_t)Kp * Np * 4), rD = make_rsrc(dEdXt, (size_t)Np * Bp * 4);
    const int voW = ((k0 + r0) * Np + 4 * c4) * 4;  // + quad0*16 (scalar) + it*4*Np*4 (imm/scalar)
    const int voD = (2 * h * Bp + b0 + i) * 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    float4 wr[8];
    float bA[32], bB[32];

    // Quads past this wave's range (tail chunk) read in-range or zero data (range check) and
    // are never multiplied (the j < jv guards below).
#define DX_LOAD_W(C)                                                                       \
    {                                                                                      \
        const int sq = (q0 + (C)*16) * 16;                                                 \
        _Pragma("unroll") for (int it = 0; it < 8; it++)                                   \
            wr[it] = bload4(rW, voW, sq + it * (16 * Np));                                 \
    }
#define DX_LOAD_B(BF, C)                                                                   \
    {                                                                                      \
        const int sb = 4 * (q0 + (C)*16) * Bp * 4;                                         \
        _Pragma("unroll") for (int j = 0; j < 16; j++) {                                   \
            BF[2 * j] = bload(rD, voD, sb + (4 * j) * Bp * 4);                             \
            BF[2 * j + 1] = bload(rD, voD, sb + (4 * j + 1) * Bp * 4);                     \
        }                                                                                  \
    }
#define DX_STORE_LDS()                                                                     \
    {                                                                                      \
        _Pragma("unroll") for (int it = 0; it < 8; it++) {                                 \
            float *dst = wb + (r0 + 4 * it) * DX_LDW + 4 * c4;                             \
            *reinterpret_cast<float2 *>(dst) = make_float2(wr[it].x, wr[it].y);            \
            *reinterpret_cast<float2 *>(dst + 2) = make_float2(wr[it].z, wr[it].w);        \
        }                                                                                  \
    }
#define DX_COMPUTE_FULL(BF)                                                                \
    {                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 16; j++) {                                   \
            const float2 av = *reinterpret_cast<const float2 *>(wb + i * DX_LDW + 4 * j + 2 * h); \
            acc = mfma32(av.x, BF[2 * j], acc);                                            \
            acc = mfma32(av.y, BF[2 * j + 1], acc);                                        \
        }                                                                                  \
    }
#define DX_COMPUTE(BF, C)                                                                  \
    {                                                                                      \
        const int jv = qw - (C)*16;                                                        \
        _Pragma("unroll") for (int j = 0; j < 16; j++) {                                   \
            if (j < jv) {                                                                  \
                const float2 av = *reinterpret_cast<const float2 *>(wb + i * DX_LDW + 4 * j + 2 * h); \
                acc = mfma32(av.x, BF[2 * j], acc);                                        \
                acc = mfma32(av.y, BF[2 * j + 1], acc);                                    \
            }                                                                              \
        }                                                                                  \
    }

    DX_LOAD_W(0);
    DX_LOAD_B(bA, 0);
    int c = 0;
    // steady state: chunks c and c+1 are not the last one, hence full (16 quads); all loads
    // unconditional (rows past this wave's range are valid memory or range-checked zeros and
    // are only ever multiplied under the j < jv guard of the drain below)
    for (; c + 2 < nch; c += 2) {
        DX_STORE_LDS();
        __syncthreads();
        DX_LOAD_W(c + 1);
        DX_LOAD_B(bB, c + 1);
        DX_COMPUTE_FULL(bA);
        __syncthreads();
        DX_STORE_LDS();
        __syncthreads();
        DX_LOAD_W(c + 2);
        DX_LOAD_B(bA, c + 2);
        DX_COMPUTE_FULL(bB);
        __syncthreads();
    }
    // drain: one or two chunks left, the last may be partial
    DX_STORE_LDS();
    __syncthreads();
    if (c + 1 < nch) {
        DX_LOAD_W(c + 1);
        DX_LOAD_B(bB, c + 1);
    }
    DX_COMPUTE(bA, c);
    __syncthreads();
    if (c + 1 < nch) {
        DX_STORE_LDS();
        __syncthreads();
        DX_COMPUTE(bB, c + 1);
        __syncthreads();
    }
#undef DX_COMPUTE_FULL
#undef DX_LOAD_W
#undef DX_LOAD_B
#undef DX_STORE_LDS
#undef DX_COMPUTE

#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][acc_row(r, lane) * 32 + i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = tid + 256 * q, row = e >> 5, col = e & 31;
        const float dedy = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const size_t o = (size_t)(k0 + row) * Bp + b0 + col;
        const float y = Yt_prev[o];
        const float g = (1.0f - y) * y * dedy;  // kernDsigmoid, DevFunc.cu:67-68
        dEdXt_prev[o] = g;
        tileT[col][row] = g;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e = tid + 256 * q, bl = e >> 5, kl = e & 31;
        dEdX_prev[(size_t)(b0 + bl) * Kp + k0 + kl] = tileT[bl][kl];
    }
}

// ---------------------------------------------------------------------------------------
// Weight-gradient GEMM with the SGD update as its epilogue:
//   G[k][n] = sum_b Y[b][k] * dEdX[b][n]
//   delta = mom*delta - lr*(G/n_frames + wc*W) ;  W = delta + 1.0f*W
// replaces cublasSgemm(N,T) + kernUpdatedelta + kernAccSum (BP_GPU.cu:432-436,
// DevFunc.cu:490-507,427-443): one pass over W/delta instead of seven.  Both operands are
// row-major activations, so fragments are direct coalesced loads (no LDS).  4 waves = 2x2,
// each TM x TN tiles of 32x32.  FUSED=false writes G instead (data-parallel path: the
// gradient is all-reduced before k_apply_update).
// ---------------------------------------------------------------------------------------
template <int TM, int TN, bool FUSED>
__global__ __launch_bounds__(256) void k_dw(const float *__restrict__ Yrow, int ldA, int Aclamp,
                                            const float *__restrict__ dEdX, float *__restrict__ Wt,
                                            float *__restrict__ delta, float *__restrict__ G, int K, int N, int Np,
                                            int Bp, int n_wg, float nf, float mom, float lr, float wc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int kt = blockIdx.x / n_wg, ntile = blockIdx.x % n_wg;
    const int k0 = kt * (64 * TM) + wm * (32 * TM), n0 = ntile * (64 * TN) + wn * (32 * TN);

    const rsrc_t rA = make_rsrc(Yrow, (size_t)Bp * ldA * 4), rB = make_rsrc(dEdX, (size_t)Bp * Np * 4);
    int va[TM], vb[TN];
#pragma unroll
    for (int t = 0; t < TM; t++) {
        int col = k0 + 32 * t + i;
        col = col < Aclamp ? col : Aclamp - 1;
        va[t] = (h * ldA + col) * 4;
    }
#pragma unroll
    for (int t = 0; t < TN; t++) {
        int col = n0 + 32 * t + i;
        col = col < Np ? col : Np - 1;
        vb[t] = (h * Np + col) * 4;
    }
    const int astep = 2 * ldA * 4, bstep = 2 * Np * 4;  // bytes per frame pair

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.0f;

    constexpr int U = 8;
    const int nch = (Bp >> 1) / U;  // Bp % 32 == 0 -> exact
    float a0[TM][U], b0[TN][U], a1[TM][U], b1[TN][U];
#define DW_LOAD(A, B, C)                                                                         \
    {                                                                                            \
        _Pragma("unroll") for (int u = 0; u < U; u++) {                                          \
            _Pragma("unroll") for (int t = 0; t < TM; t++) A[t][u] = bload(rA, va[t], ((C)*U + u) * astep); \
            _Pragma("unroll") for (int t = 0; t < TN; t++) B[t][u] = bload(rB, vb[t], ((C)*U + u) * bstep); \
        }                                                                                        \
    }
#define DW_COMPUTE(A, B)                                                                         \
    {                                                                                            \
        _Pragma("unroll") for (int u = 0; u < U; u++)                                            \
            _Pragma("unroll") for (int tm = 0; tm < TM; tm++)                                    \
                _Pragma("unroll") for (int tn = 0; tn < TN; tn++)                                \
                    acc[tm][tn] = mfma32(A[tm][u], B[tn][u], acc[tm][tn]);                       \
    }
    DW_LOAD(a0, b0, 0);
    int c = 0;
    for (; c + 2 < nch; c += 2) {  // steady state: unconditional loads
        DW_LOAD(a1, b1, c + 1);
        DW_COMPUTE(a0, b0);
        DW_LOAD(a0, b0, c + 2);
        DW_COMPUTE(a1, b1);
    }
    if (c + 1 < nch) {
        DW_LOAD(a1, b1, c + 1);
        DW_COMPUTE(a0, b0);
        DW_COMPUTE(a1, b1);
    } else {
        DW_COMPUTE(a0, b0);
    }
#undef DW_LOAD
#undef DW_COMPUTE

#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
            const int n = n0 + 32 * tn + i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = k0 + 32 * tm + acc_row(r, lane);
                if (k < K && n < N) {
                    const size_t idx = (size_t)k * Np + n;
                    const float g = acc[tm][tn][r];
                    if (FUSED) {
                        const float w = Wt[idx];
                        const float d = mom * delta[idx] - lr * (g / nf + wc * w);  // kernUpdatedelta, DevFunc.cu:502
                        delta[idx] = d;
                        Wt[idx] = d + 1.0f * w;                                    // kernAccSum, DevFunc.cu:440
                    } else {
                        G[idx] = g;
                    }
                }
            }
        }
}

// Elementwise update from an (all-reduced) gradient, data-parallel path.  Pad entries have
// G = delta = W = 0 and stay 0.  kernUpdatedelta + kernAccSum, DevFunc.cu:490-507,427-443.
__global__ __launch_bounds__(256) void k_apply_update(float *__restrict__ Wt, float *__restrict__ delta,
                                                      const float *__restrict__ G, size_t n4, float nf, float mom,
                                                      float lr, float wc) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += stride) {
        const float4 g = reinterpret_cast<const float4 *>(G)[idx];
        float4 w = reinterpret_cast<float4 *>(Wt)[idx];
        float4 d = reinterpret_cast<float4 *>(delta)[idx];
        d.x = mom * d.x - lr * (g.x / nf + wc * w.x);
        d.y = mom * d.y - lr * (g.y / nf + wc * w.y);
        d.z = mom * d.z - lr * (g.z / nf + wc * w.z);
        d.w = mom * d.w - lr * (g.w / nf + wc * w.w);
        w.x = d.x + 1.0f * w.x;
        w.y = d.y + 1.0f * w.y;
        w.z = d.z + 1.0f * w.z;
        w.w = d.w + 1.0f * w.w;
        reinterpret_cast<float4 *>(delta)[idx] = d;
        reinterpret_cast<float4 *>(Wt)[idx] = w;
    }
}

// ---------------------------------------------------------------------------------------
// Bias gradient + update for every layer in one launch: thread per unit, frames summed
// sequentially in fp32 (kernAccSumrow order, DevFunc.cu:267-285 <- BP_GPU.cu:434), then
// kernUpdatedelta with weightcost 0 and kernAccSum (BP_GPU.cu:435,437).
// FUSED=false stores the local sum to gb (data-parallel path).
// ---------------------------------------------------------------------------------------
struct BiasJob {
    const float *dEdX;  // [Bp][Np]
    float *bias, *dbias, *gb;
    int N, Np, first;   // first = prefix offset of this layer in the flattened unit index
};
struct BiasJobs {
    BiasJob job[10];
    int njobs, total;
};

template <bool FUSED>
__global__ __launch_bounds__(256) void k_bias(BiasJobs jobs, int B, float nf, float mom, float lr) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= jobs.total) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 10; q++)
        if (q < jobs.njobs && g >= jobs.job[q].first) j = q;
    const BiasJob jb = jobs.job[j];
    const int n = g - jb.first;
    if (n >= jb.N) return;
    const float *p = jb.dEdX + n;
    float s = p[0];
    for (int b = 1; b < B; b++) s += p[(size_t)b * jb.Np];
    if (FUSED) {
        const float bv = jb.bias[n];
        const float d = mom * jb.dbias[n] - lr * (s / nf + 0.0f * bv);
        jb.dbias[n] = d;
        jb.bias[n] = d + 1.0f * bv;
    } else {
        jb.gb[n] = s;
    }
}

__global__ __launch_bounds__(256) void k_bias_apply(BiasJobs jobs, float nf, float mom, float lr) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= jobs.total) return;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 10; q++)
        if (q < jobs.njobs && g >= jobs.job[q].first) j = q;
    const BiasJob jb = jobs.job[j];
    const int n = g - jb.first;
    if (n >= jb.N) return;
    const float bv = jb.bias[n];
    const float d = mom * jb.dbias[n] - lr * (jb.gb[n] / nf + 0.0f * bv);
    jb.dbias[n] = d;
    jb.bias[n] = d + 1.0f * bv;
}

// ---------------------------------------------------------------------------------------
// Input staging: in[b][k] (row stride ld, the caller's chunk layout) -> inT[k][b] padded
// with zeros (k >= K or b >= B).  32x32 tiles through LDS; both sides coalesced.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_in(const float *__restrict__ in, int ld, int B, int K,
                                                      float *__restrict__ inT, int Bp, int b_tiles) {
    __shared__ float t[32][33];
    const int kt = blockIdx.x / b_tiles, bt = blockIdx.x % b_tiles;
    const int k0 = kt * 32, b0 = bt * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int b = b0 + ty + 8 * q, k = k0 + tx;
        t[ty + 8 * q][tx] = (b < B && k < K) ? in[(size_t)b * ld + k] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int k = k0 + ty + 8 * q;
        inT[(size_t)k * Bp + b0 + tx] = t[tx][ty + 8 * q];
    }
}

// ---------------------------------------------------------------------------------------
// Output-layer loss, phase A: out = bias + sum_s slab[s]; e = out - targ; per-dimension
// sum_b |e|^beta in the reference's order (kernerror, kernabsolutevalus, kernindex2,
// kernSumcol: DevFunc.cu:399-409,186-191,219-227,167-185 <- BP_GPU.cu:413-416).
// One workgroup per 32 output dims; writes outT, eT ([Dp][Bp], 0 in pads) and colsum[d].
// Dynamic LDS: 32*(Bp+1) floats.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_err(const float *__restrict__ slab, int S, const float *__restrict__ bias,
                                                  const float *__restrict__ targ, int B, int D, int Dp, int Bp,
                                                  float beta, int want_colsum, float *__restrict__ outT,
                                                  float *__restrict__ eT, float *__restrict__ colsum) {
    extern __shared__ __attribute__((aligned(16))) float a2[];  // [32][Bp+1]
    const int d0 = blockIdx.x * 32;
    const int total = 32 * Bp;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int dl = idx / Bp, b = idx - dl * Bp;
        const int d = d0 + dl;
        const size_t o = (size_t)d * Bp + b;
        float x = slab[o];
        for (int s = 1; s < S; s++) x += slab[(size_t)s * Dp * Bp + o];
        x = x + bias[d];
        float e = 0.0f, p = 0.0f;
        if (b < B && d < D) {
            e = x - targ[(size_t)b * D + d];  // kernerror
            if (want_colsum) p = powf(fabsf(e), beta);  // kernabsolutevalus + kernindex2
        } else {
            x = 0.0f;
        }
        outT[o] = x;
        eT[o] = e;
        a2[dl * (Bp + 1) + b] = p;
    }
    if (!want_colsum) return;
    __syncthreads();
    if (threadIdx.x < 32) {
        const float *col = a2 + threadIdx.x * (Bp + 1);
        float s = col[0];  // kernSumcol: (*top) = (*fromp); then += in row order
        for (int b = 1; b < B; b++) s += col[b];
        colsum[d0 + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------
// Output-layer loss, phase B: the gradient.  MLflag != 1: beta-norm gradient
// (kernSubClean2 + kernVecMulNum, DevFunc.cu:376-398,287-293 <- BP_GPU.cu:408-409).
// MLflag == 1: alpha_d = (beta * colsum_d / n)^(1/beta) (kernDivide, kernVecMulNum,
// kernindex2 <- BP_GPU.cu:417-420), g = sgn(e)|e|^(beta-1) * beta / alpha^beta / n
// (kernfunc2 + kernVecMulNum, DevFunc.cu:468-489 <- BP_GPU.cu:422-423).
// colsum is the GLOBAL minibatch sum (all-reduced in data-parallel runs), nf the global
// minibatch size.  One workgroup per 32(d) x 32(b) tile; writes dEdXt and dEdX.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_grad(const float *__restrict__ eT, const float *__restrict__ colsum,
                                                   int B, int D, int Dp, int Bp, float beta, int MLflag, float nf,
                                                   float inv_n, float *__restrict__ scalefactor,
                                                   float *__restrict__ dEdXt, float *__restrict__ dEdX, int b_tiles) {
    __shared__ float tileT[32][33];
    __shared__ float denom[32];
    const int dt = blockIdx.x / b_tiles, bt = blockIdx.x % b_tiles;
    const int d0 = dt * 32, b0 = bt * 32;
    const int tid = threadIdx.x;
    if (MLflag == 1) {
        if (tid < 32) {
            const int d = d0 + tid;
            float q = 1.0f;
            if (d < D) {
                const float v1 = colsum[d] / nf;          // kernDivide
                const float v2 = v1 * beta;               // kernVecMulNum
                const float alpha = powf(v2, 1.0f / beta);  // kernindex2 with ppp = 1.0f/shapefactor
                if (bt == 0) scalefactor[d] = alpha;
                q = powf(alpha, beta);                    // pow(vec[j], alpha) in kernfunc2
            }
            denom[tid] = q;
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int el = tid + 256 * q, dl = el >> 5, bl = el & 31;
        const int d = d0 + dl, b = b0 + bl;
        const size_t o = (size_t)d * Bp + b;
        const float e = eT[o];
        float g = 0.0f;
        if (b < B && d < D) {
            if (MLflag == 1) {
                if (e > 0) g = powf(e, beta - 1.0f) * beta / denom[dl];
                else if (e == 0) g = 0;
                else g = -powf(-e, beta - 1.0f) * beta / denom[dl];
            } else {
                if (e > 0) g = beta * powf(e, beta - 1);
                else if (e == 0) g = 0;
                else g = -beta * powf(-e, beta - 1);
            }
            g = g * inv_n;
        }
        dEdXt[o] = g;
        tileT[bl][dl] = g;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int el = tid + 256 * q, bl = el >> 5, dl = el & 31;
        dEdX[(size_t)(b0 + bl) * Dp + d0 + dl] = tileT[bl][dl];
    }
}

// Forward-only output (cv_bunch_single, BP_GPU.cu:442-512): out[b][d] = bias + sum_s slab,
// compact row-major [B][D] for the D2H copy.
__global__ __launch_bounds__(256) void k_out_rowmajor(const float *__restrict__ slab, int S,
                                                      const float *__restrict__ bias, int B, int D, int Dp, int Bp,
                                                      float *__restrict__ out, int b_tiles) {
    __shared__ float t[32][33];
    const int dt = blockIdx.x / b_tiles, bt = blockIdx.x % b_tiles;
    const int d0 = dt * 32, b0 = bt * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int d = d0 + ty + 8 * q;
        const size_t o = (size_t)d * Bp + b0 + tx;
        float x = slab[o];
        for (int s = 1; s < S; s++) x += slab[(size_t)s * Dp * Bp + o];
        t[ty + 8 * q][tx] = x + bias[d];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int b = b0 + ty + 8 * q, d = d0 + tx;
        if (b < B && d < D) out[(size_t)b * D + d] = t[tx][ty + 8 * q];
    }
}

// Dropout on the transposed activations (kernDropout, DevFunc.cu:26-34 <- BP_GPU.cu:344-355):
// zero where uniform < p, no rescale.  The reference draws from cuRAND's default generator;
// this engine uses a counter-based hash (documented deviation: streams cannot be matched).
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void k_dropout(float *__restrict__ Yt, float *__restrict__ Yrow, int units,
                                                 int unitsp, int ldrow, int B, int Bp, float p, unsigned seed,
                                                 unsigned step) {
    const size_t n = (size_t)unitsp * Bp;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int u = (int)(idx / Bp), b = (int)(idx % Bp);
    if (u >= units || b >= B) return;
    const unsigned hsh = mix32(mix32(seed ^ (step * 0x9e3779b9u)) ^ (unsigned)idx);
    const float r = (float)(hsh >> 8) * (1.0f / 16777216.0f);
    if (r < p) {
        Yt[idx] = 0.0f;
        if (Yrow) Yrow[(size_t)b * ldrow + u] = 0.0f;
    }
}

// kernWeightMultiP (DevFunc.cu:19-25 <- BP_GPU.cu:484-501): CV-time weight scaling.
__global__ __launch_bounds__(256) void k_scale(float *__restrict__ x, size_t n, float p) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) x[idx] = x[idx] * p;
}
