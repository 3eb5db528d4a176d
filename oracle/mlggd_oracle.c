/*
 * mlggd_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference device path, kernel for kernel, in the order the
 * reference issues them.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/Train_code_ML_GGD/).  See mlggd_oracle.h for the
 * parity status ("parity unpinned" at the cuBLAS / CUDA-libm boundary).
 *
 * Decisions for ambiguities no reference run can settle (SURVEY.md 8a i-vii):
 *  (i)   pow/fabs/abs/log on float arguments inside the .cu files are the float
 *        overloads (powf/fabsf/logf), as nvcc resolves them.
 *  (ii)  cublasSgemm summation order is unspecified; this oracle fixes one order per GEMM
 *        (documented at each gemm_* below) and computes C = alpha*(A.B) + beta*C, i.e. the
 *        dot product is formed from 0.0f and the old C (the broadcast bias) is added last.
 *  (iii) dX uses the weights from before this step's update ("old weights").
 *  (iv)  the gradient is divided by n_frames twice (loss gradient and update).
 *  (vi)  CrossValid2 uses the alpha of the last training minibatch.
 *  (vii) delta buffers start at zero.
 *  (viii) FMA contraction.  The reference is built with plain `nvcc -g` (Makefile:30-33), i.e. the default
 *        --fmad=true: inside its elementwise kernels an a*b+c such as kernUpdatedelta's
 *        `momentum*delta - lr*(g/n + wc*w)` (DevFunc.cu:502) or kernAccSum's `a*x + y` (DevFunc.cu:440) is
 *        contracted into one fused multiply-add wherever the compiler sees fit, and cuBLAS sgemm uses FMAs
 *        throughout.  Which products are fused is a compiler decision no source reading can settle.  This
 *        oracle (and the HIP build, csrc/Makefile) takes the UNFUSED reading: built with -ffp-contract=off.
 *        The same source built with -ffp-contract=fast -mfma (libmlggd_oracle_fma.so) gives the fused
 *        reading; tests/test_oracle.py::test_fma_contraction_variant_is_inside_the_gpu_tolerances measures
 *        the distance between the two: weights and biases 0.7-4.4e-7 of their maximum after 3 steps on the
 *        tiny net and after 2 steps at 2827-2048^3-257 (MMSE, ML beta 1.2, ML beta 0.9), CV metrics within 1e-5
 *        relative -- the test requires 2e-6 / 1e-5, ten times inside the GPU tolerances (weights 2e-5, CV 1e-4),
 *        so either reading of the reference passes the same parity tests.
 * OpenMP only splits independent output elements across threads, so results do not depend on the
 * thread count.
 */
#include "mlggd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct ora_net {
    int numlayers;
    int layersizes[ORA_MAXLAYER];
    int bunchsize;
    float lrate, momentum, weightcost, shapefactor;
    int MLflag;
    /* BP_WorkSpace, BP_GPU.h:17-43 (only the members the train step uses) */
    float *out, *realerror, *errorabsolute, *errorabsolute2, *newobj;
    float *vec1, *vec2, *scalefactor;
    float *weights[ORA_MAXLAYER], *bias[ORA_MAXLAYER];
    float *delta_weights[ORA_MAXLAYER], *delta_bias[ORA_MAXLAYER];
    float *layer_x[ORA_MAXLAYER], *layer_y[ORA_MAXLAYER];
    float *layer_dedy[ORA_MAXLAYER], *layer_dedx[ORA_MAXLAYER];
    float *layer_ydedx[ORA_MAXLAYER], *layer_sumdedx[ORA_MAXLAYER];
    int cap_frames; /* rows allocated for the per-bunch buffers */
    /* dropout (BP_GPU.cu:344-355, 484-501); the generator is the HIP engine's counter hash, not cuRAND (ora_set_dropout) */
    int dropoutflag;
    float visible_omit, hid_omit;
    unsigned seed, step_counter;
    float *in_drop; /* the minibatch's input rows with the visible units dropped (the reference masks `in` in place) */
};

static float *zalloc(size_t n) { /* devnew_vf zero-fills, BP_GPU.cu:528-543 */
    float *p = (float *)calloc(n ? n : 1, sizeof(float));
    if (!p) abort();
    return p;
}

void ora_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void alloc_bunch_buffers(ora_net *net, int frames) {
    const int L = net->numlayers, D = net->layersizes[L - 1];
    free(net->out); free(net->realerror); free(net->errorabsolute);
    free(net->errorabsolute2); free(net->newobj);
    net->out = zalloc((size_t)frames * D);
    net->realerror = zalloc((size_t)frames * D);
    net->errorabsolute = zalloc((size_t)frames * D);
    net->errorabsolute2 = zalloc((size_t)frames * D);
    net->newobj = zalloc((size_t)frames * D);
    for (int l = 1; l < L; l++) {
        const size_t sz = (size_t)frames * net->layersizes[l];
        free(net->layer_x[l]); free(net->layer_y[l]);
        free(net->layer_dedy[l]); free(net->layer_dedx[l]);
        net->layer_x[l] = zalloc(sz);
        net->layer_y[l] = zalloc(sz);
        net->layer_dedy[l] = zalloc(sz);
        net->layer_dedx[l] = zalloc(sz);
    }
    net->cap_frames = frames;
}

/* BP_GPU::BP_GPU, BP_GPU.cu:9-111 */
ora_net *ora_create(int numlayers, const int *layersizes, int bunchsize, float lrate,
                    float momentum, float weightcost, float shapefactor, int MLflag,
                    const float *const *weights, const float *const *bias) {
    if (numlayers < 2 || numlayers > ORA_MAXLAYER || bunchsize < 1) return NULL;
    ora_net *net = (ora_net *)calloc(1, sizeof(ora_net));
    net->numlayers = numlayers;
    for (int i = 0; i < numlayers; i++) net->layersizes[i] = layersizes[i];
    net->bunchsize = bunchsize;
    net->lrate = lrate; net->momentum = momentum; net->weightcost = weightcost;
    net->shapefactor = shapefactor; net->MLflag = MLflag;
    const int D = layersizes[numlayers - 1];
    net->vec1 = zalloc(D); net->vec2 = zalloc(D); net->scalefactor = zalloc(D);
    for (int l = 1; l < numlayers; l++) {
        const size_t wsz = (size_t)layersizes[l] * layersizes[l - 1];
        net->weights[l] = zalloc(wsz);
        net->delta_weights[l] = zalloc(wsz);
        net->layer_ydedx[l] = zalloc(wsz);
        net->bias[l] = zalloc(layersizes[l]);
        net->delta_bias[l] = zalloc(layersizes[l]);
        net->layer_sumdedx[l] = zalloc(layersizes[l]);
        memcpy(net->weights[l], weights[l], wsz * sizeof(float)); /* :106 */
        memcpy(net->bias[l], bias[l], layersizes[l] * sizeof(float)); /* :107 */
    }
    alloc_bunch_buffers(net, bunchsize);
    return net;
}

void ora_destroy(ora_net *net) {
    if (!net) return;
    free(net->out); free(net->realerror); free(net->errorabsolute);
    free(net->errorabsolute2); free(net->newobj);
    free(net->vec1); free(net->vec2); free(net->scalefactor);
    for (int l = 1; l < net->numlayers; l++) {
        free(net->weights[l]); free(net->bias[l]);
        free(net->delta_weights[l]); free(net->delta_bias[l]);
        free(net->layer_x[l]); free(net->layer_y[l]);
        free(net->layer_dedy[l]); free(net->layer_dedx[l]);
        free(net->layer_ydedx[l]); free(net->layer_sumdedx[l]);
    }
    free(net->in_drop);
    free(net);
}

static void ensure_frames(ora_net *net, int frames) {
    if (frames > net->cap_frames) alloc_bunch_buffers(net, frames);
}

/* ---- GEMMs (DevFunc.h:49-87).  Each fixes ONE summation order (cuBLAS leaves it open). */

/* SgemmNN, DevFunc.h:65-75 <- BP_GPU.cu:361,494 : X[B][N] = 1*(Y[B][K] . W[K][N]) + 1*X.
 * Order: dot product over k = 0..K-1 ascending from 0.0f, then + old X (the bias). */
/* Summation-order twin (ambiguity ii made measurable).  split = 1 (default): the orders documented at each gemm_*
 * below -- the oracle every parity test compares against.  split = S > 1: the reduction index of the forward and dX
 * GEMMs is cut into S contiguous ranges, each summed ascending from 0.0f, the S partial sums added in range order --
 * the shape of order a split-K GEMM has (the HIP kernels split K over 4 waves; cuBLAS picks its own, unspecified
 * split).  Both are equally valid readings of `cublasSgemm`; the distance between the two after N steps is what a
 * mere change of GEMM summation order does to a trajectory, which is what tests bound the HIP path against. */
static int g_gemm_split = 1;
void ora_set_gemm_split(int s) { g_gemm_split = s < 1 ? 1 : s; }

/* MFMA-order twin: the summation order -- and the fused multiply-adds -- of the HIP kernels,
 * restated on the CPU, so that a test can pin the three GEMMs of the HIP path to a CPU model BIT FOR BIT on ordinary
 * data and what is left between the HIP path and this oracle is libm alone (expf in the sigmoid, powf in the loss).
 * order 0 (default): the orders documented at each gemm_* -- what every parity test compares against.
 * order 1: csrc/kernels.hip.h --
 *   forward (fwd_body): v_mfma_f32_32x32x2_f32 consumes two consecutive k per instruction as two chained fused
 *     multiply-adds (k, then k+1: established by tests/test_gpu_mfma_order.py, not assumed); a workgroup's 4 waves
 *     take the 4 contiguous ranges of k-PAIRS [P*w/4, P*(w+1)/4), P = ceil32(K)/2 (pad rows are exact zeros), each
 *     chains its range ascending from 0.0f, and the four partial sums are added in wave order; then + bias.
 *     The output layer is split over s_out slabs x 4 waves the same way (slot = 4*s + w of 4*s_out slots); a slab
 *     holds its 4 waves' sum, the loss kernel adds the slabs in order, then the bias (slab_sum).
 *   dX (dx_body): the reduction index n in QUADS of four; the 4 waves take ceil(Q/4) quads each (Q = ceil32(N)/4);
 *     within a quad the two MFMAs consume {4j, 4j+2} then {4j+1, 4j+3}; partial sums added in wave order.
 *   dW (dwp_body): one chain per weight over the frames 0..B-1 ascending (pairs of consecutive frames per MFMA).
 * The elementwise operations around the GEMMs are the same IEEE operations in both orders. */
static int g_gemm_order = 0, g_hip_s_out = 1;
/* waves that share a layer's forward / dX reduction: 4 = k_fwd / k_dx (one 32 x 32 tile per workgroup, reduction over 4
 * waves), 1 = k_fwd64 / k_dx64 (64 x 64 tile, one chain per output element; csrc/kernels64.hip.h).  The engine reports
 * its choice per layer (mlggd_debug_gemm_plan). */
static int g_hip_fwd_waves[ORA_MAXLAYER], g_hip_dx_waves[ORA_MAXLAYER];
void ora_set_gemm_order(int order, int s_out) {
    g_gemm_order = order == 1 ? 1 : 0;
    g_hip_s_out = s_out < 1 ? 1 : s_out;
    for (int l = 0; l < ORA_MAXLAYER; l++) g_hip_fwd_waves[l] = g_hip_dx_waves[l] = 4;
}
void ora_set_gemm_plan(int layer, int fwd_waves, int dx_waves) {
    if (layer < 1 || layer >= ORA_MAXLAYER) return;
    g_hip_fwd_waves[layer] = fwd_waves == 1 ? 1 : 4;
    g_hip_dx_waves[layer] = dx_waves == 1 ? 1 : 4;
}
/* Data-parallel form of the MFMA-order twin (order 1 only): what `world` ranks with B = n / world frames each leave when
 * their partial results meet in RANK ORDER -- which is what the one-GPU emulation of a world does
 * (mlggd_debug_fake_world; a real all-reduce's order is RCCL's own).  The ML statistic: every rank sums its frames by
 * the wavefront reduction of csrc/kernels.hip.h k_colsum (lane l takes frames l, l + 64, ...; six butterfly steps),
 * the ranks' sums are added in rank order.  allreduce = 1 (gradient all-reduce): dW and the bias gradient are
 * per-rank chains over the rank's own frames, added in rank order; 0 (factor exchanges): one chain over all frames. */
static int g_dp_world = 1, g_dp_allreduce = 0;
void ora_set_dp_twin(int world, int allreduce) {
    g_dp_world = world < 1 ? 1 : world;
    g_dp_allreduce = allreduce != 0;
}
#define ORA_FMA __attribute__((target("fma"))) /* fmaf as ONE instruction; -ffp-contract=off still keeps every other a*b+c unfused */
static int ceil32i(int x) { return (x + 31) & ~31; }


/* ---- register-blocked forms of the three GEMMs (speed only).
 * The plain loops in gemm_fwd / gemm_dx / gemm_dw (and their *_hip twins) DEFINE the oracle: one chain per output
 * element, operands and order as documented there.  The blocked forms run the SAME chains -- same operands, same
 * order, same unfused (or, for the twins, fused) operations -- only laid out so that the independent chains of a
 * tile of outputs sit side by side in vector registers instead of going through memory after every term.  Nothing
 * is re-associated: tests/test_oracle.py::test_blocked_gemms_equal_the_plain_loops_bit_for_bit compares the two
 * forms on ragged shapes, and ora_set_gemm_blocked(0) switches back to the plain loops. */
static int g_gemm_blocked = 1;
void ora_set_gemm_blocked(int on) { g_gemm_blocked = on != 0; }

#include <immintrin.h>
typedef float v8f __attribute__((vector_size(32)));
typedef float v8fu __attribute__((vector_size(32), aligned(4), may_alias));
static inline v8f ld8(const float *p) { return *(const v8fu *)p; }
static inline void st8(float *p, v8f v) { *(v8fu *)p = v; }
static inline v8f bc8(float x) { return (v8f){x, x, x, x, x, x, x, x}; }
#define ORA_MAC_PLAIN(c, a, b) ((c) + (a) * (b))                          /* -ffp-contract=off: a multiply, then an add */
#define ORA_MAC_FUSED(c, a, b) ((v8f)_mm256_fmadd_ps((__m256)(a), (__m256)(b), (__m256)(c))) /* one rounding */

/* out[i][0..15] = sum over t = t_lo .. t_hi-1 (ascending, from 0.0f) of a_i[t] * bp[t][0..15], i = 0..3: the 64 chains
 * of a 4 x 16 tile in 8 registers.  Forward: a_i = four frames' rows of Y, t = k, bp = a 16-column panel of W.
 * dW: a_i = four units' rows of Y^T, t = frame, bp = a 16-column panel of dEdX. */
#define ORA_DEF_TILE_4X16(NAME, ATTR, MAC)                                                                             \
    ATTR static inline void NAME(int t_lo, int t_hi, const float *a0, const float *a1, const float *a2,                \
                                 const float *a3, const float *bp, float *out) {                                       \
        const v8f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                        \
        v8f c00 = z, c01 = z, c10 = z, c11 = z, c20 = z, c21 = z, c30 = z, c31 = z;                                    \
        for (int t = t_lo; t < t_hi; t++) {                                                                            \
            const v8f b0 = ld8(bp + (size_t)16 * t), b1 = ld8(bp + (size_t)16 * t + 8);                                \
            v8f a = bc8(a0[t]);                                                                                        \
            c00 = MAC(c00, a, b0); c01 = MAC(c01, a, b1);                                                              \
            a = bc8(a1[t]);                                                                                            \
            c10 = MAC(c10, a, b0); c11 = MAC(c11, a, b1);                                                              \
            a = bc8(a2[t]);                                                                                            \
            c20 = MAC(c20, a, b0); c21 = MAC(c21, a, b1);                                                              \
            a = bc8(a3[t]);                                                                                            \
            c30 = MAC(c30, a, b0); c31 = MAC(c31, a, b1);                                                              \
        }                                                                                                              \
        st8(out, c00); st8(out + 8, c01); st8(out + 16, c10); st8(out + 24, c11);                                      \
        st8(out + 32, c20); st8(out + 40, c21); st8(out + 48, c30); st8(out + 56, c31);                                \
    }
ORA_DEF_TILE_4X16(tile_4x16, , ORA_MAC_PLAIN)
ORA_DEF_TILE_4X16(tile_4x16_fused, ORA_FMA, ORA_MAC_FUSED)

/* columns [j0, j0+16) of the rows [0, R) of M[R][N] as a dense [R][16] panel, zero beyond column N */
static void pack_panel16(int R, int N, int j0, const float *M, float *panel) {
    const int jw = N - j0 < 16 ? N - j0 : 16;
    for (int r = 0; r < R; r++) {
        const float *src = M + (size_t)r * N + j0;
        float *dst = panel + (size_t)16 * r;
        if (jw == 16) { st8(dst, ld8(src)); st8(dst + 8, ld8(src + 8)); }
        else for (int j = 0; j < 16; j++) dst[j] = j < jw ? src[j] : 0.0f;
    }
}

/* Forward, blocked.  The reduction index is cut into n_slab x n_per contiguous ranges [k_lo[s], k_hi[s]); every range
 * is one chain from 0.0f, the n_per partial sums of a slab are added in order, the slabs are added in order, then
 * + old X.  (1 x 1: the documented order; 1 x S: the split twin; s_out x waves: the MFMA-order twin.) */
#define ORA_MAX_RANGES 64
#define ORA_DEF_FWD_BLOCKED(NAME, ATTR, TILE)                                                                          \
    ATTR static void NAME(int B, int K, int N, const float *Y, const float *W, float *X, int n_slab, int n_per,        \
                          const int *k_lo, const int *k_hi) {                                                          \
        const int nblk = (N + 15) / 16;                                                                                \
        _Pragma("omp parallel")                                                                                        \
        {                                                                                                              \
            float *wp = (float *)malloc((size_t)K * 16 * sizeof(float));                                               \
            float part[64], slab[64], tot[64];                                                                         \
            _Pragma("omp for schedule(dynamic, 1)")                                                                    \
            for (int jb = 0; jb < nblk; jb++) {                                                                        \
                const int j0 = jb * 16, jw = N - j0 < 16 ? N - j0 : 16;                                                \
                pack_panel16(K, N, j0, W, wp);                                                                         \
                for (int b0 = 0; b0 < B; b0 += 4) {                                                                    \
                    const float *a[4];                                                                                 \
                    for (int i = 0; i < 4; i++) a[i] = Y + (size_t)(b0 + i < B ? b0 + i : B - 1) * K;                  \
                    for (int sl = 0; sl < n_slab; sl++) {                                                              \
                        for (int w = 0; w < n_per; w++) {                                                              \
                            const int s = sl * n_per + w;                                                              \
                            TILE(k_lo[s], k_hi[s], a[0], a[1], a[2], a[3], wp, part);                                  \
                            for (int i = 0; i < 64; i++) slab[i] = w == 0 ? part[i] : slab[i] + part[i];               \
                        }                                                                                              \
                        for (int i = 0; i < 64; i++) tot[i] = sl == 0 ? slab[i] : tot[i] + slab[i];                    \
                    }                                                                                                  \
                    for (int i = 0; i < 4 && b0 + i < B; i++) {                                                        \
                        float *x = X + (size_t)(b0 + i) * N + j0;                                                      \
                        for (int j = 0; j < jw; j++) x[j] = tot[16 * i + j] + x[j];                                    \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            free(wp);                                                                                                  \
        }                                                                                                              \
    }
ORA_DEF_FWD_BLOCKED(gemm_fwd_blocked, , tile_4x16)
ORA_DEF_FWD_BLOCKED(gemm_fwd_blocked_fused, ORA_FMA, tile_4x16_fused)

static float *transpose_rows(int R, int C, const float *x) { /* x[R][C] -> t[C][R] */
    float *t = (float *)malloc((size_t)R * C * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; c++)
        for (int r = 0; r < R; r++) t[(size_t)c * R + r] = x[(size_t)r * C + c];
    return t;
}

/* dW, blocked: G[k][j] = chain over the frames b = 0..B-1 ascending from 0.0f of Y[b][k] * dEdX[b][j] */
#define ORA_DEF_DW_BLOCKED(NAME, ATTR, TILE)                                                                           \
    ATTR static void NAME(int B, int K, int N, const float *Y, const float *dEdX, float *G) {                          \
        enum { KC = 128 };                                                                                             \
        float *yt = transpose_rows(B, K, Y); /* [K][B]: a unit's frames contiguous */                                  \
        const int nblk = (N + 15) / 16, nkc = (K + KC - 1) / KC;                                                       \
        _Pragma("omp parallel")                                                                                        \
        {                                                                                                              \
            float *dp = (float *)malloc((size_t)B * 16 * sizeof(float));                                               \
            float out[64];                                                                                             \
            _Pragma("omp for schedule(dynamic, 1) collapse(2)")                                                        \
            for (int jb = 0; jb < nblk; jb++)                                                                          \
                for (int kc = 0; kc < nkc; kc++) {                                                                     \
                    const int j0 = jb * 16, jw = N - j0 < 16 ? N - j0 : 16;                                            \
                    const int k_end = (kc + 1) * KC < K ? (kc + 1) * KC : K;                                           \
                    pack_panel16(B, N, j0, dEdX, dp);                                                                  \
                    for (int k0 = kc * KC; k0 < k_end; k0 += 4) {                                                      \
                        const float *a[4];                                                                             \
                        for (int i = 0; i < 4; i++) a[i] = yt + (size_t)(k0 + i < K ? k0 + i : K - 1) * B;             \
                        TILE(0, B, a[0], a[1], a[2], a[3], dp, out);                                                   \
                        for (int i = 0; i < 4 && k0 + i < k_end; i++) {                                                \
                            float *g = G + (size_t)(k0 + i) * N + j0;                                                  \
                            for (int j = 0; j < jw; j++) g[j] = out[16 * i + j];                                       \
                        }                                                                                              \
                    }                                                                                                  \
                }                                                                                                      \
            free(dp);                                                                                                  \
        }                                                                                                              \
        free(yt);                                                                                                      \
    }
ORA_DEF_DW_BLOCKED(gemm_dw_blocked, , tile_4x16)
ORA_DEF_DW_BLOCKED(gemm_dw_blocked_fused, ORA_FMA, tile_4x16_fused)

/* dX in the documented order, blocked: per output element the eight interleaved partial sums q = j mod 8 ARE the
 * eight lanes of one register; a tile of 2 frames x 4 units keeps its 8 such registers live over the whole row. */
static void gemm_dx_blocked(int B, int K, int N, const float *dEdX, const float *W, float *dEdY) {
    const int n8 = N / 8;
#pragma omp parallel for schedule(static)
    for (int k0 = 0; k0 < K; k0 += 4) {
        const float *w[4];
        for (int m = 0; m < 4; m++) w[m] = W + (size_t)(k0 + m < K ? k0 + m : K - 1) * N;
        for (int b0 = 0; b0 < B; b0 += 2) {
            const float *d[2];
            for (int i = 0; i < 2; i++) d[i] = dEdX + (size_t)(b0 + i < B ? b0 + i : B - 1) * N;
            const v8f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            v8f c[2][4] = {{z, z, z, z}, {z, z, z, z}};
            for (int j8 = 0; j8 < n8; j8++) {
                const v8f d0 = ld8(d[0] + 8 * j8), d1 = ld8(d[1] + 8 * j8);
                for (int m = 0; m < 4; m++) {
                    const v8f wv = ld8(w[m] + 8 * j8);
                    c[0][m] = c[0][m] + d0 * wv;
                    c[1][m] = c[1][m] + d1 * wv;
                }
            }
            for (int i = 0; i < 2 && b0 + i < B; i++)
                for (int m = 0; m < 4 && k0 + m < K; m++) {
                    float s[8];
                    st8(s, c[i][m]);
                    for (int q = 0, j = 8 * n8; j + q < N; q++) s[q] += d[i][j + q] * w[m][j + q];
                    dEdY[(size_t)(b0 + i) * K + k0 + m] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                }
        }
    }
}

/* dX with the reduction cut into ranges (the split twin and the MFMA-order twin), blocked: the B frames' chains of a
 * unit are independent, so they run in vector lanes (dEdX transposed once to [unit][frame], frames padded to 16 with
 * zeros); a tile of 16 frames x 4 units keeps its 8 registers live over a whole range.  jseq lists the reduction
 * indices of range r in the order they are consumed, [start[r], start[r+1]); every range is one chain from 0.0f and
 * the ranges' partial sums are added in range order. */
#define ORA_DEF_DXT_BLOCKED(NAME, ATTR, MAC)                                                                           \
    ATTR static void NAME(int B, int K, int N, const float *dEdX, const float *W, float *dEdY, int nranges,            \
                          const int *start, const int *jseq) {                                                         \
        const int B16 = (B + 15) & ~15;                                                                                \
        float *dT = (float *)calloc((size_t)N * B16, sizeof(float));                                                   \
        _Pragma("omp parallel for schedule(static)")                                                                   \
        for (int j = 0; j < N; j++)                                                                                    \
            for (int b = 0; b < B; b++) dT[(size_t)j * B16 + b] = dEdX[(size_t)b * N + j];                             \
        _Pragma("omp parallel for schedule(static)")                                                                   \
        for (int k0 = 0; k0 < K; k0 += 4) {                                                                            \
            const float *w[4];                                                                                         \
            for (int m = 0; m < 4; m++) w[m] = W + (size_t)(k0 + m < K ? k0 + m : K - 1) * N;                          \
            for (int b0 = 0; b0 < B16; b0 += 16) {                                                                     \
                const v8f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                \
                v8f tot[4][2];                                                                                         \
                for (int r = 0; r < nranges; r++) {                                                                    \
                    v8f c00 = z, c01 = z, c10 = z, c11 = z, c20 = z, c21 = z, c30 = z, c31 = z;                        \
                    for (int t = start[r]; t < start[r + 1]; t++) {                                                    \
                        const int j = jseq[t];                                                                         \
                        const v8f d0 = ld8(dT + (size_t)j * B16 + b0), d1 = ld8(dT + (size_t)j * B16 + b0 + 8);        \
                        v8f a = bc8(w[0][j]);                                                                          \
                        c00 = MAC(c00, a, d0); c01 = MAC(c01, a, d1);                                                  \
                        a = bc8(w[1][j]);                                                                              \
                        c10 = MAC(c10, a, d0); c11 = MAC(c11, a, d1);                                                  \
                        a = bc8(w[2][j]);                                                                              \
                        c20 = MAC(c20, a, d0); c21 = MAC(c21, a, d1);                                                  \
                        a = bc8(w[3][j]);                                                                              \
                        c30 = MAC(c30, a, d0); c31 = MAC(c31, a, d1);                                                  \
                    }                                                                                                  \
                    if (r == 0) {                                                                                      \
                        tot[0][0] = c00; tot[0][1] = c01; tot[1][0] = c10; tot[1][1] = c11;                            \
                        tot[2][0] = c20; tot[2][1] = c21; tot[3][0] = c30; tot[3][1] = c31;                            \
                    } else {                                                                                           \
                        tot[0][0] += c00; tot[0][1] += c01; tot[1][0] += c10; tot[1][1] += c11;                        \
                        tot[2][0] += c20; tot[2][1] += c21; tot[3][0] += c30; tot[3][1] += c31;                        \
                    }                                                                                                  \
                }                                                                                                      \
                for (int m = 0; m < 4 && k0 + m < K; m++) {                                                            \
                    float o[16];                                                                                       \
                    st8(o, tot[m][0]); st8(o + 8, tot[m][1]);                                                          \
                    for (int i = 0; i < 16 && b0 + i < B; i++) dEdY[(size_t)(b0 + i) * K + k0 + m] = o[i];             \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
        free(dT);                                                                                                      \
    }
ORA_DEF_DXT_BLOCKED(gemm_dxT_blocked, , ORA_MAC_PLAIN)
ORA_DEF_DXT_BLOCKED(gemm_dxT_blocked_fused, ORA_FMA, ORA_MAC_FUSED)

ORA_FMA static void gemm_fwd_hip(int B, int K, int N, const float *Y, const float *W, float *X, int S, int NW) {
    enum { JB = 128 };
    const int nblk = (N + JB - 1) / JB;
    const int P = ceil32i(K) / 2, nslots = S * NW;
    if (g_gemm_blocked && nslots <= ORA_MAX_RANGES) {
        int lo[ORA_MAX_RANGES], hi[ORA_MAX_RANGES];
        for (int slot = 0; slot < nslots; slot++) {
            lo[slot] = 2 * (int)((unsigned)(P * slot) / (unsigned)nslots);
            hi[slot] = 2 * (int)((unsigned)(P * (slot + 1)) / (unsigned)nslots);
            if (hi[slot] > K) hi[slot] = K; /* rows k >= K are zero pads */
        }
        gemm_fwd_blocked_fused(B, K, N, Y, W, X, S, NW, lo, hi);
        return;
    }
#pragma omp parallel
    {
        float *tot = (float *)malloc((size_t)B * JB * sizeof(float));
        float *slab = (float *)malloc((size_t)B * JB * sizeof(float));
        float *part = (float *)malloc((size_t)B * JB * sizeof(float));
#pragma omp for schedule(dynamic, 1)
        for (int jb = 0; jb < nblk; jb++) {
            const int j0 = jb * JB, jw = (N - j0 < JB) ? N - j0 : JB;
            for (int sl = 0; sl < S; sl++) {
                for (int w = 0; w < NW; w++) {
                    const int slot = sl * NW + w;
                    const int p0 = (int)((unsigned)(P * slot) / (unsigned)nslots);
                    const int p1 = (int)((unsigned)(P * (slot + 1)) / (unsigned)nslots);
                    const int k_lo = 2 * p0, k_hi = 2 * p1 < K ? 2 * p1 : K; /* rows k >= K are zero pads */
                    memset(part, 0, (size_t)B * JB * sizeof(float));
                    for (int k = k_lo; k < k_hi; k++) {
                        const float *wr = W + (size_t)k * N + j0;
                        for (int b = 0; b < B; b++) {
                            const float yv = Y[(size_t)b * K + k];
                            float *a = part + (size_t)b * JB;
                            for (int j = 0; j < jw; j++) a[j] = __builtin_fmaf(wr[j], yv, a[j]);
                        }
                    }
                    for (size_t i = 0; i < (size_t)B * JB; i++) slab[i] = w == 0 ? part[i] : slab[i] + part[i];
                }
                for (size_t i = 0; i < (size_t)B * JB; i++) tot[i] = sl == 0 ? slab[i] : tot[i] + slab[i];
            }
            for (int b = 0; b < B; b++)
                for (int j = 0; j < jw; j++) {
                    float *x = X + (size_t)b * N + j0 + j;
                    *x = tot[(size_t)b * JB + j] + *x;
                }
        }
        free(tot);
        free(slab);
        free(part);
    }
}

/* (dEdX is transposed once to [unit][frame] so that the chains of a unit's B frames -- independent of one another -- run
 * side by side in vector lanes: the order WITHIN every chain is untouched) */
static float *transpose_bn(int B, int N, const float *x) {
    float *t = (float *)malloc((size_t)N * B * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int j = 0; j < N; j++)
        for (int b = 0; b < B; b++) t[(size_t)j * B + b] = x[(size_t)b * N + j];
    return t;
}
ORA_FMA static void gemm_dx_hip(int B, int K, int N, const float *dEdX, const float *W, float *dEdY, int NW) {
    const int Q = ceil32i(N) / 4, qw = (Q + NW - 1) / NW;
    if (g_gemm_blocked && NW <= ORA_MAX_RANGES) {
        int start[ORA_MAX_RANGES + 1], n = 0;
        int *jseq = (int *)malloc((size_t)4 * Q * sizeof(int));
        for (int wv = 0; wv < NW; wv++) {
            const int q0 = wv * qw, qend = q0 + qw < Q ? q0 + qw : Q;
            static const int ord[4] = {0, 2, 1, 3};
            start[wv] = n;
            for (int q = q0; q < qend; q++)
                for (int t = 0; t < 4; t++)
                    if (4 * q + ord[t] < N) jseq[n++] = 4 * q + ord[t]; /* columns j >= N are zero pads */
        }
        start[NW] = n;
        gemm_dxT_blocked_fused(B, K, N, dEdX, W, dEdY, NW, start, jseq);
        free(jseq);
        return;
    }
    float *dT = transpose_bn(B, N, dEdX);
#pragma omp parallel
    {
        float *part = (float *)malloc((size_t)B * sizeof(float)), *tot = (float *)malloc((size_t)B * sizeof(float));
#pragma omp for schedule(static)
        for (int k = 0; k < K; k++) {
            const float *w = W + (size_t)k * N;
            for (int wv = 0; wv < NW; wv++) {
                const int q0 = wv * qw, qend = q0 + qw < Q ? q0 + qw : Q;
                for (int b = 0; b < B; b++) part[b] = 0.0f;
                for (int q = q0; q < qend; q++) {
                    static const int ord[4] = {0, 2, 1, 3};
                    for (int t = 0; t < 4; t++) {
                        const int j = 4 * q + ord[t];
                        if (j >= N) continue; /* columns j >= N are zero pads */
                        const float wj = w[j];
                        const float *d = dT + (size_t)j * B;
                        for (int b = 0; b < B; b++) part[b] = __builtin_fmaf(wj, d[b], part[b]);
                    }
                }
                for (int b = 0; b < B; b++) tot[b] = wv == 0 ? part[b] : tot[b] + part[b];
            }
            for (int b = 0; b < B; b++) dEdY[(size_t)b * K + k] = tot[b];
        }
        free(part);
        free(tot);
    }
    free(dT);
}

ORA_FMA static void gemm_dw_hip(int B, int K, int N, const float *Y, const float *dEdX, float *G) {
    if (g_gemm_blocked) { gemm_dw_blocked_fused(B, K, N, Y, dEdX, G); return; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        float *g = G + (size_t)k * N;
        for (int j = 0; j < N; j++) g[j] = 0.0f;
        for (int b = 0; b < B; b++) {
            const float yv = Y[(size_t)b * K + k];
            const float *d = dEdX + (size_t)b * N;
            for (int j = 0; j < N; j++) g[j] = __builtin_fmaf(yv, d[j], g[j]);
        }
    }
}

static void gemm_fwd(int B, int K, int N, const float *Y, const float *W, float *X) {
    enum { JB = 128 };
    const int nblk = (N + JB - 1) / JB;
    const int S = g_gemm_split;
    if (g_gemm_blocked && S <= ORA_MAX_RANGES) {
        int lo[ORA_MAX_RANGES], hi[ORA_MAX_RANGES];
        for (int sp = 0; sp < S; sp++) {
            lo[sp] = (int)((long)K * sp / S);
            hi[sp] = (int)((long)K * (sp + 1) / S);
        }
        gemm_fwd_blocked(B, K, N, Y, W, X, 1, S, lo, hi);
        return;
    }
#pragma omp parallel
    {
        float *acc = (float *)malloc((size_t)B * JB * sizeof(float));
        float *part = S > 1 ? (float *)malloc((size_t)B * JB * sizeof(float)) : NULL;
#pragma omp for schedule(dynamic, 1)
        for (int jb = 0; jb < nblk; jb++) {
            const int j0 = jb * JB, jw = (N - j0 < JB) ? N - j0 : JB;
            memset(acc, 0, (size_t)B * JB * sizeof(float));
            for (int sp = 0; sp < S; sp++) {
                float *dst = S > 1 ? part : acc;
                if (S > 1) memset(part, 0, (size_t)B * JB * sizeof(float));
                const int k_lo = (int)((long)K * sp / S), k_hi = (int)((long)K * (sp + 1) / S);
                for (int k = k_lo; k < k_hi; k++) {
                    const float *w = W + (size_t)k * N + j0;
                    for (int b = 0; b < B; b++) {
                        const float yv = Y[(size_t)b * K + k];
                        float *a = dst + (size_t)b * JB;
                        for (int j = 0; j < jw; j++) a[j] += yv * w[j];
                    }
                }
                if (S > 1)
                    for (size_t i = 0; i < (size_t)B * JB; i++) acc[i] = sp == 0 ? part[i] : acc[i] + part[i];
            }
            for (int b = 0; b < B; b++)
                for (int j = 0; j < jw; j++) {
                    float *x = X + (size_t)b * N + j0 + j;
                    *x = acc[(size_t)b * JB + j] + *x;
                }
        }
        free(acc);
        free(part);
    }
}

/* SgemmTN, DevFunc.h:49-63 <- BP_GPU.cu:430 : dEdY[B][K] = dEdX[B][N] . W[K][N]^T (beta 0).
 * Order: eight interleaved partial sums q = j mod 8, each over ascending j, combined as
 * ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)). */
static void gemm_dx(int B, int K, int N, const float *dEdX, const float *W, float *dEdY) {
    const int S = g_gemm_split;
    if (S > 1 && g_gemm_blocked && S <= ORA_MAX_RANGES) {
        int start[ORA_MAX_RANGES + 1];
        int *jseq = (int *)malloc((size_t)(N > 0 ? N : 1) * sizeof(int));
        for (int j = 0; j < N; j++) jseq[j] = j;
        for (int sp = 0; sp <= S; sp++) start[sp] = (int)((long)N * sp / S);
        gemm_dxT_blocked(B, K, N, dEdX, W, dEdY, S, start, jseq);
        free(jseq);
        return;
    }
    if (S > 1) { /* order twin: S contiguous ranges of j, each ascending from 0.0f, added in range order (the frames'
                  * independent chains side by side in vector lanes: dEdX transposed once) */
        float *dT = transpose_bn(B, N, dEdX);
#pragma omp parallel
        {
            float *part = (float *)malloc((size_t)B * sizeof(float)), *tot = (float *)malloc((size_t)B * sizeof(float));
#pragma omp for schedule(static)
            for (int k = 0; k < K; k++) {
                const float *w = W + (size_t)k * N;
                for (int sp = 0; sp < S; sp++) {
                    const int j_lo = (int)((long)N * sp / S), j_hi = (int)((long)N * (sp + 1) / S);
                    for (int b = 0; b < B; b++) part[b] = 0.0f;
                    for (int j = j_lo; j < j_hi; j++) {
                        const float wj = w[j];
                        const float *d = dT + (size_t)j * B;
                        for (int b = 0; b < B; b++) part[b] += d[b] * wj;
                    }
                    for (int b = 0; b < B; b++) tot[b] = sp == 0 ? part[b] : tot[b] + part[b];
                }
                for (int b = 0; b < B; b++) dEdY[(size_t)b * K + k] = tot[b];
            }
            free(part);
            free(tot);
        }
        free(dT);
        return;
    }
    if (g_gemm_blocked) { gemm_dx_blocked(B, K, N, dEdX, W, dEdY); return; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        const float *w = W + (size_t)k * N;
        for (int b = 0; b < B; b++) {
            const float *d = dEdX + (size_t)b * N;
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int j = 0;
            for (; j + 8 <= N; j += 8)
                for (int q = 0; q < 8; q++) s[q] += d[j + q] * w[j + q];
            for (int q = 0; j + q < N; q++) s[q] += d[j + q] * w[j + q];
            dEdY[(size_t)b * K + k] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
    }
}

/* SgemmNT, DevFunc.h:77-87 <- BP_GPU.cu:432 : G[K][N] = Y[B][K]^T . dEdX[B][N] (beta 0).
 * Order: b = 0..B-1 ascending from 0.0f. */
static void gemm_dw(int B, int K, int N, const float *Y, const float *dEdX, float *G) {
    if (g_gemm_blocked) { gemm_dw_blocked(B, K, N, Y, dEdX, G); return; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        float *g = G + (size_t)k * N;
        for (int j = 0; j < N; j++) g[j] = 0.0f;
        for (int b = 0; b < B; b++) {
            const float yv = Y[(size_t)b * K + k];
            const float *d = dEdX + (size_t)b * N;
            for (int j = 0; j < N; j++) g[j] += yv * d[j];
        }
    }
}

/* The sigmoid's exponential as the HIP kernels evaluate it (csrc/kernels.hip.h exp_det: the SAME statements, IEEE
 * operations only, no fused multiply-add on either side) -- used by the MFMA-order twin, so that a whole training run
 * of a net whose loss needs no powf equals the HIP path bit for bit.  The documented-order oracle keeps libm's expf:
 * which bits CUDA's expf returns no source reading can settle, and both are <= 1-ulp readings of it. */
__attribute__((optimize("fp-contract=off"))) float ora_exp_det(float x0) {
    float x = x0 < -85.5f ? -85.5f : x0; /* n >= -123 below: y * 2^(n-1) stays a normal number */
    x = x > 88.72283f ? 88.72283f : x;   /* beyond it the result is +inf (selected at the end) */
    x = x0 != x0 ? 0.0f : x;             /* a NaN comes back as it is (selected at the end) */
    const float fn = floorf(1.44269504f * x + 0.5f);
    float r = x - fn * 0.693359375f;
    r = r - fn * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float y = p * z + r;
    y = y + 1.0f;
    union { int i; float f; } s1;
    s1.i = ((int)fn + 126) << 23;
    const float e = (y * s1.f) * 2.0f; /* 2^n as 2^(n-1) * 2: n = 128 has no float of its own */
    return x0 != x0 ? x0 : x0 > 88.72283f ? INFINITY : e;
}
/* x^y (x >= 0) as the HIP loss kernels evaluate it (csrc/kernels.hip.h pow_det: the SAME statements -- IEEE double
 * operations only) -- used by the MFMA-order twin in place of powf, so that the loss chain, too, equals the HIP path bit
 * for bit at every beta.  The documented-order oracle keeps libm's powf. */
__attribute__((optimize("fp-contract=off"))) float ora_pow_det(float xf, float yf) {
    if (xf != xf || yf != yf) return xf + yf;
    if (yf == 0.0f) return 1.0f;
    if (xf == 0.0f) return yf > 0 ? 0.0f : INFINITY;
    if (xf == INFINITY) return yf > 0 ? INFINITY : 0.0f;
    const double x = (double)xf, y = (double)yf; /* exact; a float denormal is a normal double */
    union { double d; long long i; } u;
    u.d = x;
    const long long bits = u.i;
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    u.i = (bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL;
    double m = u.d;                                          /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; } /* [sqrt(1/2), sqrt 2) */
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 21.0; /* log m = 2 s (1 + z/3 + z^2/5 + ...), |s| <= 0.1716: z^10 / 21 < 2e-17 */
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double lg = (2.0 * s) * p + (double)e * 0.6931471805599453;
    const double t = y * lg;
    if (t > 89.0) return INFINITY;
    if (t < -104.0) return 0.0f;
    const double fn = floor(t * 1.4426950408889634 + 0.5);
    double r = t - fn * 0.6931471803691238; /* ln 2, high part (its trailing bits are zero: fn * high is exact) */
    r = r - fn * 1.9082149292705877e-10;    /* ln 2, low part */
    double q = 1.0 / 6227020800.0;          /* 1 / 13! */
    q = q * r + 1.0 / 479001600.0;
    q = q * r + 1.0 / 39916800.0;
    q = q * r + 1.0 / 3628800.0;
    q = q * r + 1.0 / 362880.0;
    q = q * r + 1.0 / 40320.0;
    q = q * r + 1.0 / 5040.0;
    q = q * r + 1.0 / 720.0;
    q = q * r + 1.0 / 120.0;
    q = q * r + 1.0 / 24.0;
    q = q * r + 1.0 / 6.0;
    q = q * r + 0.5;
    q = q * r + 1.0;
    q = q * r + 1.0;
    const int n = (int)fn; /* in [-151, 129]: 2^n is a normal double */
    u.i = (long long)(n + 1023) << 52;
    return (float)(q * u.d);
}
void ora_pow_det_array(const float *x, float y, float *out, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) out[i] = ora_pow_det(x[i], y);
}
/* the loss chain's pow: libm's powf in the documented-order oracle; in the MFMA-order twin the HIP kernels' pow_or_self
 * (x for y = 1, 1 for y = 0 -- what powf returns there as well -- and pow_det otherwise) */
static float lpow(float x, float y) {
    if (g_gemm_order != 1) return powf(x, y);
    return y == 1.0f ? x : y == 0.0f ? 1.0f : ora_pow_det(x, y);
}

__attribute__((optimize("fp-contract=off"))) void ora_exp_det_array(const float *x, float *out, long n, int sigmoid) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) out[i] = sigmoid ? 1.0f / (1.0f + ora_exp_det(-x[i])) : ora_exp_det(x[i]);
}

/* ---- forward: BP_GPU.cu:334-369 (train) and :467-509 (cv) ---- */
/* Dropout.  kernDropout (DevFunc.cu:26-34 <- BP_GPU.cu:344-355): in[i] = 0 where u_i < p, no rescale; CV scales the
 * weights by the keep-probability around each GEMM instead (kernWeightMultiP, DevFunc.cu:19-25 <- BP_GPU.cu:484-501 --
 * W * keep, then W * (1 / keep): a round trip that does not restore every bit, as in the reference).
 * DOCUMENTED DEVIATION: the reference draws u from cuRAND's default generator, whose stream cannot be matched; the HIP
 * engine draws it from a counter-based hash of (seed, step, layer, element) -- csrc/kernels.hip.h k_dropout -- and the
 * oracle restates THAT generator, in both of its readings, so dropout runs can be compared at all.  The element index is
 * the engine's: unit * ceil32(bunchsize) + frame. */
void ora_set_dropout(ora_net *net, int dropoutflag, float visible_omit, float hid_omit, int random_seed) {
    net->dropoutflag = dropoutflag == 1;
    net->visible_omit = visible_omit;
    net->hid_omit = hid_omit;
    net->seed = (unsigned)random_seed;
}
static unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static void dropout_rows(const ora_net *net, float *y, int n, int N, float p, int layer) {
    const unsigned Bp = (unsigned)((net->bunchsize + 31) & ~31), step = net->step_counter * 16u + (unsigned)layer;
    const unsigned base = mix32(net->seed ^ (step * 0x9e3779b9u));
    for (int b = 0; b < n; b++)
        for (int u = 0; u < N; u++) {
            const unsigned hsh = mix32(base ^ ((unsigned)u * Bp + (unsigned)b));
            const float r = (float)(hsh >> 8) * (1.0f / 16777216.0f);
            if (r < p) y[(size_t)b * N + u] = 0.0f;
        }
}
static void scale_weights(float *w, size_t n, float p) {
    for (size_t i = 0; i < n; i++) w[i] = w[i] * p;
}

/* mode 0: training forward (dropout masks when enabled); 1: CV forward (weight scaling when enabled) */
static void forward_impl(ora_net *net, int n, const float *in, int cv) {
    ensure_frames(net, n);
    const int L = net->numlayers;
    const int drop = net->dropoutflag && !cv, cvscale = net->dropoutflag && cv;
    if (drop) { /* :344-347: the visible units of this minibatch */
        const size_t sz = (size_t)n * net->layersizes[0];
        net->in_drop = (float *)realloc(net->in_drop, (sz ? sz : 1) * sizeof(float));
        memcpy(net->in_drop, in, sz * sizeof(float));
        dropout_rows(net, net->in_drop, n, net->layersizes[0], net->visible_omit, 0);
        in = net->in_drop;
    }
    for (int l = 1; l < L; l++) {
        const int N = net->layersizes[l], K = net->layersizes[l - 1];
        if (drop && l > 1) dropout_rows(net, net->layer_y[l - 1], n, K, net->hid_omit, l - 1); /* :350-353 */
        const float keep = 1.0f - (l == 1 ? net->visible_omit : net->hid_omit);
        if (cvscale) scale_weights(net->weights[l], (size_t)K * N, keep); /* :484-489 */
        const float *prev_y = (l == 1) ? in : net->layer_y[l - 1];
        float *x = net->layer_x[l];
        /* kernMultiCopy, DevFunc.cu:134-149 <- BP_GPU.cu:360 */
        for (int b = 0; b < n; b++) memcpy(x + (size_t)b * N, net->bias[l], N * sizeof(float));
        if (g_gemm_order == 1) gemm_fwd_hip(n, K, N, prev_y, net->weights[l], x, l == L - 1 ? g_hip_s_out : 1, g_hip_fwd_waves[l]);
        else gemm_fwd(n, K, N, prev_y, net->weights[l], x); /* :361 */
        if (cvscale) scale_weights(net->weights[l], (size_t)K * N, 1.0f / keep); /* :496-501 */
        if (l != L - 1) {
            /* kernSigmoid, DevFunc.cu:36-51 <- :364 */
            float *y = net->layer_y[l];
            const size_t sz = (size_t)n * N;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < sz; i++) y[i] = 1.0f / (1.0f + (g_gemm_order == 1 ? ora_exp_det(-x[i]) : expf(-x[i])));
        } else {
            memcpy(net->out, x, (size_t)n * N * sizeof(float)); /* cudaMemcpy D2D, :367 */
        }
    }
}
void ora_forward(ora_net *net, int n, const float *in) { forward_impl(net, n, in, 0); }

/* Deverror + Devabsolutevalus + Devindex2 + DevSumcol: BP_GPU.cu:413-416
 * (kernerror DevFunc.cu:399-409, kernabsolutevalus :186-191, kernindex2 :219-227,
 *  kernSumcol :167-185: thread per column, rows summed sequentially in fp32). */
void ora_loss_colsum(ora_net *net, int n, const float *targ, float *colsum) {
    const int D = net->layersizes[net->numlayers - 1];
    const float beta = net->shapefactor;
    for (int b = 0; b < n; b++)
        for (int d = 0; d < D; d++) {
            const size_t i = (size_t)b * D + d;
            net->realerror[i] = net->out[i] - targ[i];
            net->errorabsolute[i] = fabsf(net->realerror[i]);
            net->errorabsolute2[i] = lpow(net->errorabsolute[i], beta);
        }
    if (g_gemm_order == 1 && g_dp_world > 1 && n % g_dp_world == 0) { /* k_colsum per rank, ranks in order */
        const int B = n / g_dp_world;
        for (int d = 0; d < D; d++) {
            float tot = 0.0f;
            for (int r = 0; r < g_dp_world; r++) {
                float lane[64];
                for (int l = 0; l < 64; l++) {
                    float s = 0.0f;
                    for (int b = l; b < B; b += 64) s += net->errorabsolute2[(size_t)(r * B + b) * D + d];
                    lane[l] = s;
                }
                for (int off = 32; off > 0; off >>= 1) {
                    float nx[64];
                    for (int l = 0; l < 64; l++) nx[l] = lane[l] + lane[l ^ off];
                    memcpy(lane, nx, sizeof(lane));
                }
                tot = r == 0 ? lane[0] : tot + lane[0];
            }
            colsum[d] = tot;
        }
        return;
    }
    for (int d = 0; d < D; d++) {
        float s = net->errorabsolute2[d];
        for (int b = 1; b < n; b++) s += net->errorabsolute2[(size_t)b * D + d];
        colsum[d] = s;
    }
}

/* Output-layer gradient, BP_GPU.cu:408-424.  colsum_global: sum over the whole (global)
 * minibatch of |e|^beta per dimension (only read when MLflag==1). */
void ora_loss_grad(ora_net *net, int n, int n_global, const float *targ, const float *colsum_global) {
    const int L = net->numlayers, D = net->layersizes[L - 1];
    const float beta = net->shapefactor;
    float *dedx = net->layer_dedx[L - 1];
    /* kernSubClean2, DevFunc.cu:376-398 <- :408 */
    for (int b = 0; b < n; b++)
        for (int d = 0; d < D; d++) {
            const size_t i = (size_t)b * D + d;
            const float o = net->out[i], t = targ[i];
            float g;
            if (o > t) g = beta * lpow(o - t, beta - 1);
            else if (o == t) g = 0;
            else g = -beta * lpow(t - o, beta - 1);
            dedx[i] = g;
        }
    /* kernVecMulNum, DevFunc.cu:287-293 <- :409 */
    const float inv_n = 1.0f / n_global;
    for (size_t i = 0; i < (size_t)n * D; i++) dedx[i] = dedx[i] * inv_n;
    if (net->MLflag == 1) {
        /* :413-415 recomputed so the phase is self-contained (same values as colsum pass) */
        for (size_t i = 0; i < (size_t)n * D; i++) net->realerror[i] = net->out[i] - targ[i];
        /* kernDivide, DevFunc.cu:445-450 <- :417 ; kernVecMulNum <- :418 ; kernindex2 <- :420 */
        const float nf = (float)n_global;
        const float ppp = 1.0f / beta;
        for (int d = 0; d < D; d++) {
            net->vec1[d] = colsum_global[d] / nf;
            net->vec2[d] = net->vec1[d] * beta;
            net->scalefactor[d] = lpow(net->vec2[d], ppp);
        }
        /* kernfunc2, DevFunc.cu:468-489 <- :422 */
        for (int b = 0; b < n; b++)
            for (int d = 0; d < D; d++) {
                const size_t i = (size_t)b * D + d;
                const float e = net->realerror[i];
                float g;
                if (e > 0) g = lpow(e, beta - 1.0f) * beta / lpow(net->scalefactor[d], beta);
                else if (e == 0) g = 0;
                else g = -lpow(-e, beta - 1.0f) * beta / lpow(net->scalefactor[d], beta);
                net->newobj[i] = g;
            }
        /* kernVecMulNum <- :423 */
        for (size_t i = 0; i < (size_t)n * D; i++) dedx[i] = net->newobj[i] * inv_n;
    }
}

/* Backward without the weight update: BP_GPU.cu:371-438.  Because dX(l) reads W_l before
 * the update of W_l is applied (old-weights semantics) and nothing else in
 * the backward pass reads weights, "all gradients first, then all updates" is the same
 * computation as the reference's per-layer interleaving. */
void ora_backward(ora_net *net, int n, const float *in) {
    const int L = net->numlayers;
    if (net->dropoutflag && net->in_drop) in = net->in_drop; /* the reference masked `in` itself (:346): dW_1 sees the dropped rows */
    for (int l = L - 1; l > 0; l--) {
        const int N = net->layersizes[l], K = net->layersizes[l - 1];
        const float *prev_y = (l == 1) ? in : net->layer_y[l - 1];
        float *dedx = net->layer_dedx[l];
        if (l != L - 1) {
            /* kernDsigmoid, DevFunc.cu:53-71 <- :402 */
            const float *y = net->layer_y[l], *dedy = net->layer_dedy[l];
            const size_t sz = (size_t)n * N;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < sz; i++) dedx[i] = (1.0f - y[i]) * y[i] * dedy[i];
        }
        const int ranks = (g_gemm_order == 1 && g_dp_allreduce && g_dp_world > 1 && n % g_dp_world == 0) ? g_dp_world : 1;
        if (g_gemm_order == 1) {
            if (l != 1) gemm_dx_hip(n, K, N, dedx, net->weights[l], net->layer_dedy[l - 1], g_hip_dx_waves[l]);
            if (ranks == 1) {
                gemm_dw_hip(n, K, N, prev_y, dedx, net->layer_ydedx[l]);
            } else { /* gradient all-reduce: every rank's chain over its own frames, the ranks added in order */
                const int B = n / ranks;
                float *part = (float *)malloc((size_t)K * N * sizeof(float)), *g = net->layer_ydedx[l];
                for (int r = 0; r < ranks; r++) {
                    gemm_dw_hip(B, K, N, prev_y + (size_t)r * B * K, dedx + (size_t)r * B * N, r == 0 ? g : part);
                    if (r > 0) {
#pragma omp parallel for schedule(static)
                        for (size_t i = 0; i < (size_t)K * N; i++) g[i] = g[i] + part[i];
                    }
                }
                free(part);
            }
        } else {
            if (l != 1) gemm_dx(n, K, N, dedx, net->weights[l], net->layer_dedy[l - 1]); /* :430 */
            gemm_dw(n, K, N, prev_y, dedx, net->layer_ydedx[l]);                          /* :432 */
        }
        /* kernAccSumrow, DevFunc.cu:267-285 <- :434 (alpha 0, beta 1; rows summed in order) */
        float *sum = net->layer_sumdedx[l];
        if (ranks > 1) { /* the bias gradient likewise: per-rank sums in frame order, added in rank order */
            const int B = n / ranks;
            for (int j = 0; j < N; j++) {
                float tot = 0.0f;
                for (int r = 0; r < ranks; r++) {
                    float s = sum[j] * 0.0f + 1.0f * dedx[(size_t)r * B * N + j];
                    for (int b = 1; b < B; b++) s += 1.0f * dedx[(size_t)(r * B + b) * N + j];
                    tot = r == 0 ? s : tot + s;
                }
                sum[j] = tot;
            }
            continue;
        }
        for (int j = 0; j < N; j++) {
            float s = sum[j] * 0.0f + 1.0f * dedx[j];
            for (int b = 1; b < n; b++) s += 1.0f * dedx[(size_t)b * N + j];
            sum[j] = s;
        }
    }
}

/* kernUpdatedelta (DevFunc.cu:490-507) + kernAccSum (:427-443) <- BP_GPU.cu:433-437 */
void ora_apply_update(ora_net *net, int n_global) {
    const int L = net->numlayers;
    const float mom = net->momentum, lr = net->lrate, wc = net->weightcost;
    const float zero = 0.0f;
    for (int l = L - 1; l > 0; l--) {
        const int N = net->layersizes[l], K = net->layersizes[l - 1];
        const size_t wsz = (size_t)N * K;
        float *dw = net->delta_weights[l], *w = net->weights[l];
        const float *g = net->layer_ydedx[l];
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < wsz; i++) {
            dw[i] = mom * dw[i] - lr * (g[i] / n_global + wc * w[i]);
        }
        float *db = net->delta_bias[l], *bb = net->bias[l];
        const float *gb = net->layer_sumdedx[l];
        for (int j = 0; j < N; j++) db[j] = mom * db[j] - lr * (gb[j] / n_global + zero * bb[j]);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < wsz; i++) w[i] = dw[i] + 1.0f * w[i];
        for (int j = 0; j < N; j++) bb[j] = db[j] + 1.0f * bb[j];
    }
}

/* BP_GPU::train_bunch_single, BP_GPU.cu:308-440 */
void ora_train_bunch(ora_net *net, int n, const float *in, const float *targ) {
    const int D = net->layersizes[net->numlayers - 1];
    ora_forward(net, n, in);
    float *colsum = net->vec1; /* scratch; ora_loss_grad overwrites vec1 after reading */
    float *tmp = NULL;
    if (net->MLflag == 1) {
        tmp = (float *)malloc(D * sizeof(float));
        ora_loss_colsum(net, n, targ, tmp);
        colsum = tmp;
    }
    ora_loss_grad(net, n, n, targ, colsum);
    free(tmp);
    ora_backward(net, n, in);
    ora_apply_update(net, n);
    net->step_counter++;
}

/* BP_GPU::train, BP_GPU.cu:152-185 */
int ora_train(ora_net *net, int n_frames, const float *in, const float *targ) {
    const int K0 = net->layersizes[0], D = net->layersizes[net->numlayers - 1];
    const int B = net->bunchsize;
    int trained = 0;
    for (int i = 0; i < n_frames; i += B) {
        const int fb = (B > n_frames - i) ? (n_frames - i) : B;
        if (fb == B) {
            ora_train_bunch(net, fb, in, targ);
            trained++;
        } /* else: "this bunch has only %d samples and is ignored", :177-180 */
        in += (size_t)K0 * fb;
        targ += (size_t)D * fb;
    }
    return trained;
}

/* BP_GPU::cv_bunch_single, BP_GPU.cu:442-512 */
void ora_cv_bunch(ora_net *net, int n, const float *in, float *out) {
    const int D = net->layersizes[net->numlayers - 1];
    forward_impl(net, n, in, 1);
    memcpy(out, net->out, (size_t)n * D * sizeof(float));
}

/* BP_GPU::CrossValid, BP_GPU.cu:187-219 */
float ora_cv_sqerr(ora_net *net, int n_frames, const float *in, const float *targ) {
    const int K0 = net->layersizes[0], D = net->layersizes[net->numlayers - 1], B = net->bunchsize;
    float squared_err = 0.0f;
    float *out = (float *)malloc((size_t)B * D * sizeof(float));
    for (int i = 0; i < n_frames; i += B) {
        const int fb = (B > n_frames - i) ? (n_frames - i) : B;
        ora_cv_bunch(net, fb, in, out);
        for (int j = 0; j < fb; j++)
            for (int d = 0; d < D; d++)
                squared_err = squared_err + (out[j * D + d] - targ[j * D + d]) * (out[j * D + d] - targ[j * D + d]);
        in += (size_t)K0 * fb;
        targ += (size_t)D * fb;
    }
    free(out);
    return squared_err;
}

/* BP_GPU::CrossValiddB, BP_GPU.cu:220-253 */
float ora_cv_abserr(ora_net *net, int n_frames, const float *in, const float *targ) {
    const int K0 = net->layersizes[0], D = net->layersizes[net->numlayers - 1], B = net->bunchsize;
    float squared_err = 0.0f;
    float *out = (float *)malloc((size_t)B * D * sizeof(float));
    for (int i = 0; i < n_frames; i += B) {
        const int fb = (B > n_frames - i) ? (n_frames - i) : B;
        ora_cv_bunch(net, fb, in, out);
        for (int j = 0; j < fb; j++)
            for (int d = 0; d < D; d++)
                squared_err = squared_err + fabsf(out[j * D + d] - targ[j * D + d]);
        in += (size_t)K0 * fb;
        targ += (size_t)D * fb;
    }
    squared_err = squared_err / D;
    free(out);
    return squared_err;
}

/* BP_GPU::CrossValid2, BP_GPU.cu:254-306 */
float ora_cv_loglik(ora_net *net, int n_frames, const float *in, const float *targ) {
    const int K0 = net->layersizes[0], D = net->layersizes[net->numlayers - 1], B = net->bunchsize;
    const float shapefactor = net->shapefactor;
    float *err = (float *)malloc((size_t)n_frames * D * sizeof(float));
    float *out = (float *)malloc((size_t)B * D * sizeof(float));
    int h = 0;
    for (int i = 0; i < n_frames; i += B) {
        const int fb = (B > n_frames - i) ? (n_frames - i) : B;
        ora_cv_bunch(net, fb, in, out);
        for (int j = 0; j < fb; j++)
            for (int d = 0; d < D; d++)
                err[(size_t)(j + h) * D + d] = targ[j * D + d] - out[j * D + d];
        in += (size_t)K0 * fb;
        targ += (size_t)D * fb;
        h = h + fb;
    }
    float density1, density2 = 0, density3 = 0, density;
    const float *scalefac = net->scalefactor; /* fromdev_vf_vf(dev.scalefactor), :287 */
    density1 = n_frames * D * logf(shapefactor / (2 * ora_gamma((float)(1.0 / shapefactor)))); /* :288 */
    for (int u = 0; u < D; u++) density2 += logf(scalefac[u]);
    density2 = density2 * n_frames;
    for (int uu = 0; uu < n_frames; uu++)
        for (int uuu = 0; uuu < D; uuu++)
            density3 += powf(fabsf(err[(size_t)uu * D + uuu]) / scalefac[uuu], shapefactor);
    density = density1 - density2 - density3;
    free(out);
    free(err);
    return density;
}

/* BP_GPU::Gamma, BP_GPU.cu:593-640.  The polynomial terms are double expressions
 * (x-2.0 and pow(double,double)); each statement rounds its sum to the float temp. */
float ora_gamma(float x) {
    if (x > 2 && x <= 3) {
        const float c0 = 0.0000677106, c1 = -0.0003442342, c2 = 0.0015397681, c3 = -0.0024467480,
                    c4 = 0.0109736958, c5 = -0.0002109075, c6 = 0.0742379071, c7 = 0.0815782188,
                    c8 = 0.4118402518, c9 = 0.4227843370, c10 = 1.0000000000;
        float temp = 0;
        temp = temp + c0 * pow(x - 2.0, 10.0) + c1 * pow(x - 2.0, 9.0);
        temp = temp + c2 * pow(x - 2.0, 8.0) + c3 * pow(x - 2.0, 7.0);
        temp = temp + c4 * pow(x - 2.0, 6.0) + c5 * pow(x - 2.0, 5.0);
        temp = temp + c6 * pow(x - 2.0, 4.0) + c7 * pow(x - 2.0, 3.0);
        temp = temp + c8 * pow(x - 2.0, 2.0) + c9 * (x - 2.0) + c10;
        return temp;
    } else if (x > 0 && x <= 1) {
        return ora_gamma(x + 2) / (x * (x + 1));
    } else if (x > 1 && x <= 2) {
        return ora_gamma(x + 1) / x;
    } else if (x > 3) {
        int i = 1;
        float temp = 1;
        while (((x - i) > 2 && (x - i) <= 3) == 0) {
            temp = (x - i) * temp;
            i++;
        }
        temp = temp * (x - i);
        return temp * ora_gamma(x - i);
    }
    return 0;
}

/* BP_GPU::returnWeights, BP_GPU.cu:514-525 */
void ora_get_weights(ora_net *net, float *const *weights, float *const *bias) {
    for (int l = 1; l < net->numlayers; l++) {
        memcpy(weights[l], net->weights[l], (size_t)net->layersizes[l] * net->layersizes[l - 1] * sizeof(float));
        memcpy(bias[l], net->bias[l], net->layersizes[l] * sizeof(float));
    }
}

void ora_set_scalefactor(ora_net *net, const float *alpha) {
    memcpy(net->scalefactor, alpha, net->layersizes[net->numlayers - 1] * sizeof(float));
}

const float *ora_tensor(ora_net *net, const char *name, int layer, long *count) {
    const int L = net->numlayers, D = net->layersizes[L - 1];
    long c = 0;
    const float *p = NULL;
    if (!strcmp(name, "scalefactor")) { p = net->scalefactor; c = D; }
    else if (!strcmp(name, "out")) { p = net->out; c = (long)net->cap_frames * D; }
    else if (layer >= 1 && layer < L) {
        const long N = net->layersizes[layer], K = net->layersizes[layer - 1];
        if (!strcmp(name, "x")) { p = net->layer_x[layer]; c = net->cap_frames * N; }
        else if (!strcmp(name, "y")) { p = net->layer_y[layer]; c = net->cap_frames * N; }
        else if (!strcmp(name, "dedx")) { p = net->layer_dedx[layer]; c = net->cap_frames * N; }
        else if (!strcmp(name, "dedy")) { p = net->layer_dedy[layer]; c = net->cap_frames * N; }
        else if (!strcmp(name, "grad_w")) { p = net->layer_ydedx[layer]; c = N * K; }
        else if (!strcmp(name, "grad_b")) { p = net->layer_sumdedx[layer]; c = N; }
        else if (!strcmp(name, "delta_w")) { p = net->delta_weights[layer]; c = N * K; }
        else if (!strcmp(name, "delta_b")) { p = net->delta_bias[layer]; c = N; }
        else if (!strcmp(name, "weights")) { p = net->weights[layer]; c = N * K; }
        else if (!strcmp(name, "bias")) { p = net->bias[layer]; c = N; }
    }
    if (count) *count = c;
    return p;
}
