// kernels64.hip.h -- forward and dX GEMMs with a 64 x 64 output tile per workgroup, for minibatches large enough to
// give every CU such a tile (B >= 256 at 4096-wide layers, B >= 512 at 2048: BASELINE.json config 5 is 512 x 4096).
// Included at the end of kernels.hip.h (uses its mfma32 / acc_row / buffer-resource helpers, FwdArgs, DxArgs).
//
// Why a second pair of GEMM kernels.  k_fwd / k_dx give a workgroup ONE 32 x 32 output tile and
// split the reduction over its 4 waves: at B = 128 that is what fills 256 CUs (exactly 256 tiles per 2048-wide layer),
// but every wave stages its own operands -- 8 FLOP per byte brought into the CU -- and every tile ends in an LDS
// reduction + epilogue.  With >= 256 tiles of 64 x 64 there is a better shape, the one k_dwp already has:
//   * 4 waves as 2 x 2, each owning a 32 x 32 accumulator over the FULL reduction: no cross-wave reduction, no
//     reduction epilogue, one ascending chain per output element;
//   * the two operand pieces of a chunk ([32 k][64 n] of W and [32 k][64 b] of Yt: 16 KB) are staged ONCE per
//     workgroup and read by the wave pair that needs each half: 16 FLOP per byte into the CU;
//   * operands by LDS-DMA (buffer_load ... lds) into a ring of 4 chunks, each wave issuing a quarter of a chunk's
//     16 instructions THREE chunks ahead (a weight row first touched comes from HBM: its latency is longer than one
//     chunk's 16 MFMAs last); one s_barrier per chunk, placed in the middle of the chunk's MFMA block, between the
//     counted vmcnt wait for the NEXT chunk and the fragment reads of that chunk -- the MFMAs of the current chunk
//     run on registers on either side of it;
//   * 64 KB of LDS and ~110 VGPRs per workgroup: two workgroups per CU, so a SIMD always has a second wave to issue
//     MFMAs from while the first sits in the barrier.
// Summation order: ONE chain per output element over the reduction index ascending (dX: in quads, {4j, 4j+2} then
// {4j+1, 4j+3}, as k_dx), each v_mfma_f32_32x32x2_f32 two chained fused multiply-adds -- the oracle's MFMA-order twin
// restates it with `waves = 1` (oracle/mlggd_oracle.c, tests/test_gpu_mfma_order.py).
#pragma once

constexpr int T64_NB = 4;     // chunks in the operand ring
constexpr int T64_CH = 4096;  // floats per chunk: [32][64] of each operand
constexpr int T64_SCR = 1024 + 32 * 36;  // per-wave epilogue scratch: the 32 x 32 tile + its transposition tile
constexpr int t64_lds_floats() { return T64_NB * T64_CH; }
static_assert(4 * T64_SCR <= T64_NB * T64_CH, "the epilogue scratch aliases the operand ring");
#define T64_LDSP(p) ((__attribute__((address_space(3))) void *)(p))

// ---------------------------------------------------------------------------------------
// The main loop of the forward kernel (the weight-gradient experiments of round 4 ran on it too): C[64 x 64] += A^T B over `nch` chunks of 32 reduction
// rows, both operands row-major with the reduction index as the ROW ([32][64] pieces: forward: W[k][n], Yt[k][b];
// dW: Y[b][k], dEdX[b][n]).  4 waves as 2 x 2 (wm: half of A's columns, wn: half of B's); ring of 4 chunks, DMA three
// chunks ahead, one barrier per chunk in the middle of its MFMA block.
// after_prologue(): called once between the DMAs of chunks 0..2 and the first wait; it may issue EXTRA (template) more
// vector-memory loads per lane (used by the dW experiments of round 4, profiles/r04_dw64_ab.txt; the forward kernel passes
// 0): vmcnt counts in issue order, so bodies 0 and 1 -- whose awaited chunks 1, 2 are OLDER than those loads -- leave
// them in flight, and from body 2 on they have arrived.
// endA / endB: a byte offset at or past the end of the resource's range -- where the chunks past the last one are
// "loaded" from (out of range for every lane: no memory request).
// ---------------------------------------------------------------------------------------
template <int EXTRA, class F>
__device__ __forceinline__ void t64_rowmajor_loop(const rsrc_t rA, const rsrc_t rB, const int voA, const int voB,
                                                  const int rowA_bytes, const int rowB_bytes, const int endA, const int endB,
                                                  const int nch, float *smem, const int wave, const int wm, const int wn,
                                                  const int fo, f32x16 &acc, F after_prologue) {
    float fa[16], fb[16], ga[16], gb[16];
    // instruction Q (0..3) of this wave's share of chunk C: rows 4t..4t+3 of A (Q < 2) or of B (Q >= 2), t = 2 wave + (Q & 1)
#define T64_DMA(C, BUF, Q)                                                                         \
    {                                                                                              \
        const int c_ = (C), t_ = 2 * wave + ((Q) & 1);                                             \
        const bool live_ = c_ < nch;                                                               \
        float *dst_ = smem + (BUF) * T64_CH + t_ * 256;                                            \
        if ((Q) < 2)                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, T64_LDSP(dst_), 16, voA,                  \
                                                     live_ ? (32 * c_ + 4 * t_) * rowA_bytes : endA, 0, 0); \
        else                                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, T64_LDSP(dst_ + 2048), 16, voB,           \
                                                     live_ ? (32 * c_ + 4 * t_) * rowB_bytes : endB, 0, 0); \
    }
#define T64_RD4(BUF, NA, NB, Q)                                                                    \
    {                                                                                              \
        const float *b_ = smem + (BUF) * T64_CH + fo;                                              \
        _Pragma("unroll") for (int u = 4 * (Q); u < 4 * (Q) + 4; u++) {                           \
            NA[u] = b_[u * 128 + 32 * wm];                                                         \
            NB[u] = b_[2048 + u * 128 + 32 * wn];                                                  \
        }                                                                                          \
    }
    // chunk C on (FA, FB), read from buffer CUR one body ago; chunk C+1 is in (or landing in) buffer CUR+1; chunk C+3
    // goes to buffer CUR+3 = CUR-1, whose chunk C-1 every wave has finished reading (it passed the barrier of body C-1
    // after the lgkmcnt wait at the top of that body).  VM: vector-memory operations that may still be in flight at the
    // wait: the 8 DMAs of chunks C+2, C+3 (+ EXTRA in bodies 0 and 1).
#define T64_BODY(FA, FB, NA, NB, CUR, C, VM)                                                       \
    {                                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): the fragments of chunk C have arrived */ \
        _Pragma("unroll") for (int g = 0; g < 4; g++) {                                            \
            acc = mfma32(FA[2 * g], FB[2 * g], acc);                                               \
            acc = mfma32(FA[2 * g + 1], FB[2 * g + 1], acc);                                       \
            T64_DMA((C) + 3, ((CUR) + 3) & 3, g);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((VM) & 15) | (((VM) >> 4) << 14)); /* vmcnt(VM): this wave's part of chunk C+1 has landed */ \
        __builtin_amdgcn_s_barrier();            /* ... and so has everybody else's */             \
        _Pragma("unroll") for (int g = 4; g < 8; g++) {                                            \
            acc = mfma32(FA[2 * g], FB[2 * g], acc);                                               \
            acc = mfma32(FA[2 * g + 1], FB[2 * g + 1], acc);                                       \
            T64_RD4(((CUR) + 1) & 3, NA, NB, g - 4);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
#pragma unroll
    for (int q = 0; q < 4; q++) T64_DMA(0, 0, q);
#pragma unroll
    for (int q = 0; q < 4; q++) T64_DMA(1, 1, q);
#pragma unroll
    for (int q = 0; q < 4; q++) T64_DMA(2, 2, q);
    after_prologue();
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((8 + EXTRA) & 15) | (((8 + EXTRA) >> 4) << 14));
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) T64_RD4(0, fa, fb, q);
    __builtin_amdgcn_sched_barrier(0);
    // (wave-uniform branches BETWEEN bodies only: a body is one basic block)
    T64_BODY(fa, fb, ga, gb, 0, 0, 8 + EXTRA);
    if (nch > 1) {
        T64_BODY(ga, gb, fa, fb, 1, 1, 8 + EXTRA);
        if (nch > 2) {
            T64_BODY(fa, fb, ga, gb, 2, 2, 8);
            if (nch > 3) {
                T64_BODY(ga, gb, fa, fb, 3, 3, 8);
                int c = 4;
                for (; c + 4 <= nch; c += 4) {
                    T64_BODY(fa, fb, ga, gb, 0, c, 8);
                    T64_BODY(ga, gb, fa, fb, 1, c + 1, 8);
                    T64_BODY(fa, fb, ga, gb, 2, c + 2, 8);
                    T64_BODY(ga, gb, fa, fb, 3, c + 3, 8);
                }
                if (c < nch) {  // 1..3 chunks left
                    T64_BODY(fa, fb, ga, gb, 0, c, 8);
                    if (c + 1 < nch) {
                        T64_BODY(ga, gb, fa, fb, 1, c + 1, 8);
                        if (c + 2 < nch) T64_BODY(fa, fb, ga, gb, 2, c + 2, 8);
                    }
                }
            }
        }
    }
#undef T64_DMA
#undef T64_RD4
#undef T64_BODY
    __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing may still be landing in LDS (the out-of-range chunks past the last)
}

// ---------------------------------------------------------------------------------------
// Forward + bias + sigmoid (hidden layers):  X^T[n][b] = sum_k W[k][n] Yt_in[k][b] + bias[n];  y = 1/(1+expf(-x))
// replaces kernMultiCopy + cublasSgemm(N,N) + kernSigmoid (BP_GPU.cu:360-364).  A.n_tiles = Np/64, A.b_tiles = Bp/64.
// MODE FWD_SLAB: an output layer wide enough for this tiling (one slab: the raw sums go to slab[0][n][b]; the loss
// kernel adds the bias) -- also what lets a test read this kernel's sums back bit for bit.
// ---------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void fwd64_body(const FwdArgs &A, const int bid, float *smem) {
    const float *__restrict__ W = A.W, *__restrict__ Yt_in = A.Yt_in, *__restrict__ bias = A.bias;
    float *__restrict__ Yt_out = A.Yt_out, *__restrict__ Y_out = A.Y_out;
    const int Kp = A.Kp, Np = A.Np, Bp = A.Bp, N = A.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int nt, bt;
    tile_of_block(bid, A.n_tiles, A.b_tiles, A.b_shift, nt, bt, A.map);
    const int n0 = nt * 64, b0 = bt * 64;
    const int nch = Kp >> 5;  // chunks of 32 k rows (Kp is a multiple of 32)

    const rsrc_t rW = make_rsrc(W, (size_t)Kp * Np * 4), rY = make_rsrc(Yt_in, (size_t)Kp * Bp * 4);
    const int endW = Kp * Np * 4, endY = Kp * Bp * 4;
    // DMA: one wave-instruction = 4 rows x 16 slots of 16 B = 1 KB, lane-linear in LDS = 4 rows of the [32][64] piece
    const int r4 = lane >> 4, q16 = lane & 15;
    const int voW = (r4 * Np + n0 + 4 * q16) * 4, voY = (r4 * Bp + b0 + 4 * q16) * 4;
    const int fo = h * 64 + i;  // fragment reads: k-pair u of the chunk -> row 2u + h, columns 32 wm + i (W) / 32 wn + i (Yt)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    // the epilogue's bias values, fetched now (see fwd_body)
    const int er = lane >> 3, ec = lane & 7;
    float bias_pre[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bias_pre[q] = MODE == FWD_SIGMOID ? bias[n0 + 32 * wm + er + 8 * q] : 0.0f;
    asm volatile("" ::: "memory");

    t64_rowmajor_loop<0>(rW, rY, voW, voY, Np * 4, Bp * 4, endW, endY, nch, smem, wave, wm, wn, fo, acc, [] {});
    __syncthreads();                     // every wave is done with the operand ring: the scratch below aliases it

    // epilogue, wave-private: accumulators -> S[n][b] -> bias + sigmoid, 16-byte stores of Yt [unit][frame]; the same
    // values transposed through T[b][n] -> 16-byte stores of Y [frame][unit] (the dW operand)
    float *S = smem + wave * T64_SCR;
    float(*T)[36] = reinterpret_cast<float(*)[36]>(S + 1024);
#pragma unroll
    for (int r = 0; r < 16; r++) S[acc_row(r, lane) * 32 + i] = acc[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = er + 8 * q, col4 = 4 * ec, n = n0 + 32 * wm + row;
        const float4 v4 = *reinterpret_cast<const float4 *>(&S[row * 32 + col4]);
        if (MODE == FWD_SLAB) {
            *reinterpret_cast<float4 *>(&A.slab[(size_t)n * Bp + b0 + 32 * wn + col4]) = v4;
            continue;
        }
        const float bn = bias_pre[q];
        const float4 y4 = sigmoid_det4(v4, bn, n < N);  // kernSigmoid, DevFunc.cu:48
        *reinterpret_cast<float4 *>(&Yt_out[(size_t)n * Bp + b0 + 32 * wn + col4]) = y4;
        T[col4][row] = y4.x;
        T[col4 + 1][row] = y4.y;
        T[col4 + 2][row] = y4.z;
        T[col4 + 3][row] = y4.w;
    }
    if (MODE == FWD_SLAB) return;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int brow = er + 8 * q;
        const size_t yo = A.yblk ? y_blocked_base(n0 + 32 * wm, A.yblk, Bp) + (size_t)(b0 + 32 * wn + brow) * A.yblk + 4 * ec
                                 : (size_t)(b0 + 32 * wn + brow) * Np + n0 + 32 * wm + 4 * ec;
        *reinterpret_cast<float4 *>(&Y_out[yo]) = *reinterpret_cast<const float4 *>(&T[brow][4 * ec]);
    }
}

// ---------------------------------------------------------------------------------------
// dX + sigmoid derivative:  dEdY^T[k][b] = sum_n W[k][n] dEdXt[n][b];  dEdX_prev = (1 - y) y dEdY
// replaces cublasSgemm(T,N) + kernDsigmoid (BP_GPU.cu:430,402).  A.k_tiles = Kp/64, A.b_tiles = Bp/64.
// The reduction index n is the CONTIGUOUS index of W: a chunk is [64 k][32 n] of W (8 slots of 16 B per row) and
// [32 n][64 b] of dEdXt.  The W piece is stored with slot s of row k holding quad s ^ ((k >> 1) & 7) (the permutation
// sits in the per-lane global offset of the DMA), so that the 32 lanes of a transposed fragment read (one row k each,
// the same quad, ds_read_b64) spread over all 16 (row parity, slot) positions of the 256-byte bank row instead of
// one: a 2-way conflict is left (lanes k and k + 16), as in k_dx.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void dx64_body(const DxArgs &A, const int bid, float *smem) {
    const float *__restrict__ W = A.W, *__restrict__ dEdXt = A.dEdXt, *__restrict__ Yt_prev = A.Yt_prev;
    float *__restrict__ dEdXt_prev = A.dEdXt_prev, *__restrict__ dEdX_prev = A.dEdX_prev;
    const int Kp = A.Kp, Np = A.Np, Bp = A.Bp;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    int kt, bt;
    tile_of_block(bid, A.k_tiles, A.b_tiles, A.b_shift, kt, bt, A.map);
    const int k0 = kt * 64, b0 = bt * 64;
    const int nch = Np >> 5;  // chunks of 32 n

    const rsrc_t rW = make_rsrc(W, (size_t)Kp * Np * 4), rD = make_rsrc(dEdXt, (size_t)Np * Bp * 4);
    const int endW = Kp * Np * 4, endD = Np * Bp * 4;
    // W DMA: 8 rows x 8 slots per instruction; instruction t covers rows 8t..8t+7, this wave issues t = 2 wave + {0, 1}
    const int rr = lane >> 3, s8 = lane & 7;
    int voWd[2];
#pragma unroll
    for (int q = 0; q < 2; q++) voWd[q] = ((k0 + rr) * Np + 4 * (s8 ^ ((4 * q + (rr >> 1)) & 7))) * 4;
    // dEdXt DMA: 4 rows x 16 slots per instruction (natural layout)
    const int r4 = lane >> 4, q16 = lane & 15;
    const int voD = (r4 * Bp + b0 + 4 * q16) * 4;
    const int krow = 32 * wm + i;  // this lane's W row of the piece
    int aoff[8];                   // quad j of row krow sits in slot j ^ ((krow >> 1) & 7)
#pragma unroll
    for (int j = 0; j < 8; j++) aoff[j] = krow * 32 + 4 * (j ^ ((krow >> 1) & 7)) + 2 * h;
    const int boff = 2048 + 32 * wn + i;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    const int er = lane >> 3, ec = lane & 7;
    float4 y_pre[4];  // y of this lane's output elements (rows er + 8 q of the wave tile, 4 consecutive frames)
#pragma unroll
    for (int q = 0; q < 4; q++)
        y_pre[q] = *reinterpret_cast<const float4 *>(&Yt_prev[(size_t)(k0 + 32 * wm + er + 8 * q) * Bp + b0 + 32 * wn + 4 * ec]);
    asm volatile("" ::: "memory");

    float fa[16], fb[16], ga[16], gb[16];
#define X64_DMA(C, BUF, Q)                                                                         \
    {                                                                                              \
        const int c_ = (C), t_ = 2 * wave + ((Q) & 1);                                             \
        const bool live_ = c_ < nch;                                                               \
        float *dst_ = smem + (BUF) * T64_CH + t_ * 256;                                            \
        if ((Q) < 2)                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, T64_LDSP(dst_), 16, voWd[(Q) & 1],        \
                                                     live_ ? (8 * t_ * Np + 32 * c_) * 4 : endW, 0, 0); \
        else                                                                                       \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rD, T64_LDSP(dst_ + 2048), 16, voD,           \
                                                     live_ ? (32 * c_ + 4 * t_) * Bp * 4 : endD, 0, 0); \
    }
    // quads 2Q, 2Q+1 of the chunk: MFMA 2j takes n = 4j + {0, 2} (lane half 0 / 1), MFMA 2j+1 n = 4j + {1, 3}
#define X64_RD2(BUF, NA, NB, Q)                                                                    \
    {                                                                                              \
        const float *b_ = smem + (BUF) * T64_CH;                                                   \
        _Pragma("unroll") for (int j = 2 * (Q); j < 2 * (Q) + 2; j++) {                            \
            NA[2 * j] = b_[aoff[j]]; /* two float loads the compiler merges into one ds_read_b64 (a float2 cast */ \
            NA[2 * j + 1] = b_[aoff[j] + 1]; /* makes it assume the read may alias the DMA: s_waitcnt vmcnt(0)) */ \
            NB[2 * j] = b_[boff + (4 * j + 2 * h) * 64];                                           \
            NB[2 * j + 1] = b_[boff + (4 * j + 2 * h + 1) * 64];                                   \
        }                                                                                          \
    }
#define X64_BODY(FA, FB, NA, NB, CUR, C)                                                           \
    {                                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
        _Pragma("unroll") for (int g = 0; g < 4; g++) {                                            \
            acc = mfma32(FA[2 * g], FB[2 * g], acc);                                               \
            acc = mfma32(FA[2 * g + 1], FB[2 * g + 1], acc);                                       \
            X64_DMA((C) + 3, ((CUR) + 3) & 3, g);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);                                                    \
        __builtin_amdgcn_s_barrier();                                                              \
        _Pragma("unroll") for (int g = 4; g < 8; g++) {                                            \
            acc = mfma32(FA[2 * g], FB[2 * g], acc);                                               \
            acc = mfma32(FA[2 * g + 1], FB[2 * g + 1], acc);                                       \
            X64_RD2(((CUR) + 1) & 3, NA, NB, g - 4);                                               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }
#pragma unroll
    for (int q = 0; q < 4; q++) X64_DMA(0, 0, q);
#pragma unroll
    for (int q = 0; q < 4; q++) X64_DMA(1, 1, q);
#pragma unroll
    for (int q = 0; q < 4; q++) X64_DMA(2, 2, q);
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) X64_RD2(0, fa, fb, q);
    __builtin_amdgcn_sched_barrier(0);
    int c = 0;
    for (; c + 4 <= nch; c += 4) {
        X64_BODY(fa, fb, ga, gb, 0, c);
        X64_BODY(ga, gb, fa, fb, 1, c + 1);
        X64_BODY(fa, fb, ga, gb, 2, c + 2);
        X64_BODY(ga, gb, fa, fb, 3, c + 3);
    }
    if (c < nch) {
        X64_BODY(fa, fb, ga, gb, 0, c);
        if (c + 1 < nch) {
            X64_BODY(ga, gb, fa, fb, 1, c + 1);
            if (c + 2 < nch) X64_BODY(fa, fb, ga, gb, 2, c + 2);
        }
    }
#undef X64_DMA
#undef X64_RD2
#undef X64_BODY
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    float *S = smem + wave * T64_SCR;
    float(*T)[36] = reinterpret_cast<float(*)[36]>(S + 1024);
#pragma unroll
    for (int r = 0; r < 16; r++) S[acc_row(r, lane) * 32 + i] = acc[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = er + 8 * q, col4 = 4 * ec;
        const float4 d4 = *reinterpret_cast<const float4 *>(&S[row * 32 + col4]);
        const float4 y4 = y_pre[q];
        float4 g4;  // kernDsigmoid, DevFunc.cu:67-68
        g4.x = (1.0f - y4.x) * y4.x * d4.x;
        g4.y = (1.0f - y4.y) * y4.y * d4.y;
        g4.z = (1.0f - y4.z) * y4.z * d4.z;
        g4.w = (1.0f - y4.w) * y4.w * d4.w;
        *reinterpret_cast<float4 *>(&dEdXt_prev[(size_t)(k0 + 32 * wm + row) * Bp + b0 + 32 * wn + col4]) = g4;
        T[col4][row] = g4.x;
        T[col4 + 1][row] = g4.y;
        T[col4 + 2][row] = g4.z;
        T[col4 + 3][row] = g4.w;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int brow = er + 8 * q;
        *reinterpret_cast<float4 *>(&dEdX_prev[(size_t)(b0 + 32 * wn + brow) * Kp + k0 + 32 * wm + 4 * ec]) =
            *reinterpret_cast<const float4 *>(&T[brow][4 * ec]);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_fwd64(FwdArgs A) { fwd64_body<MODE>(A, (int)blockIdx.x, g_dyn_lds); }
__global__ __launch_bounds__(256, 2) void k_dx64(DxArgs A) { dx64_body(A, (int)blockIdx.x, g_dyn_lds); }
