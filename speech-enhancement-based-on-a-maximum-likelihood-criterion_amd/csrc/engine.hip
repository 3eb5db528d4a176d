// engine.hip -- host side of the MI355X training engine behind the C-ABI of include/mlggd.h.
//
// Mirrors what the reference's class BP_GPU does on the host (Train_code_ML_GGD/BP_GPU.cu):
// device workspace (BP_WorkSpace, BP_GPU.h:17-43), chunk upload, the bunch loop, CV metric
// accumulation and weight read-back -- re-designed for one HIP stream of fused gfx950 kernels
// (kernels.hip.h) instead of ~50 launches + cuBLAS on two racing streams.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/mlggd.h"
#include "kernels.hip.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(MLGGD_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)                 \
    do {                          \
        int _rc = (expr);         \
        if (_rc != MLGGD_OK) return _rc; \
    } while (0)

static inline int ceil32(int x) { return (x + 31) & ~31; }

// ------------------------------------------------------------------ RCCL (loaded on demand)
// Only the data-parallel path needs RCCL; it is dlopen'ed at mlggd_comm_init so a single-GPU process never maps
// it.  The function-pointer types are taken from rccl.h itself (decltype of the declarations; the header is only
// compiled against, nothing is linked), so a signature change in the library breaks the build, not a multi-GPU run.
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == MLGGD_UNIQUE_ID_BYTES, "include/mlggd.h MLGGD_UNIQUE_ID_BYTES != sizeof(ncclUniqueId)");
typedef ncclUniqueId RcclUniqueId;
typedef ncclComm_t RcclComm;
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce_ = nullptr;
    decltype(&ncclAllGather) AllGather_ = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter_ = nullptr;
    decltype(&ncclSend) Send_ = nullptr;  // optional: the all-to-all form of the sharded update
    decltype(&ncclRecv) Recv_ = nullptr;
    decltype(&ncclCommSplit) CommSplit = nullptr;        // optional: MLGGD_DP_STAT_COMM
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString_ = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;        // optional: mlggd_comm_info
    decltype(&ncclCommUserRank) CommUserRank = nullptr;  // optional: mlggd_comm_info
    // fp32 sum / fp32 gather, the only forms the engine uses (the int arguments of the call sites are ignored: they
    // date from the hand-written prototypes and are kept so the call sites read like the NCCL API)
    int AllReduce(const void *s, void *r, size_t n, int, int, RcclComm c, hipStream_t st) const {
        return (int)AllReduce_(s, r, n, ncclFloat32, ncclSum, c, st);
    }
    int AllGather(const void *s, void *r, size_t n, int, RcclComm c, hipStream_t st) const {
        return (int)AllGather_(s, r, n, ncclFloat32, c, st);
    }
    // r receives this rank's block of n floats of the sum over the ranks of the world*n floats at s
    int ReduceScatter(const void *s, void *r, size_t n, RcclComm c, hipStream_t st) const {
        return (int)ReduceScatter_(s, r, n, ncclFloat32, ncclSum, c, st);
    }
    const char *GetErrorString(int rc) const { return GetErrorString_ ? GetErrorString_((ncclResult_t)rc) : "rccl error"; }
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return MLGGD_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return fail(MLGGD_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(&ncclGetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(&ncclCommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce_ = (decltype(&ncclAllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.AllGather_ = (decltype(&ncclAllGather))dlsym(lib, "ncclAllGather");
    g_rccl.ReduceScatter_ = (decltype(&ncclReduceScatter))dlsym(lib, "ncclReduceScatter");
    g_rccl.CommSplit = (decltype(&ncclCommSplit))dlsym(lib, "ncclCommSplit");
    g_rccl.Send_ = (decltype(&ncclSend))dlsym(lib, "ncclSend");
    g_rccl.Recv_ = (decltype(&ncclRecv))dlsym(lib, "ncclRecv");
    g_rccl.CommDestroy = (decltype(&ncclCommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString_ = (decltype(&ncclGetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.GroupStart = (decltype(&ncclGroupStart))dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(&ncclGroupEnd))dlsym(lib, "ncclGroupEnd");
    g_rccl.CommCount = (decltype(&ncclCommCount))dlsym(lib, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(&ncclCommUserRank))dlsym(lib, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce_ || !g_rccl.AllGather_ || !g_rccl.ReduceScatter_ || !g_rccl.CommDestroy ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return fail(MLGGD_ERR_COMM, "librccl is missing required symbols");
    g_rccl.lib = lib;
    return MLGGD_OK;
}
#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        int _r = (expr);                                                                           \
        if (_r != 0)                                                                               \
            return fail(MLGGD_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));        \
    } while (0)

// ------------------------------------------------------------------ ROCTX ranges (loaded on demand)
// MLGGD_ROCTX=1: the phases of every training step (forward, loss, backward, weight update, exchange) are bracketed
// with roctxRangePush / Pop so that `rocprofv3 --marker-trace --kernel-trace` shows them around the kernels they
// enqueue (the reference has no profiling hooks at all, BP_GPU.h:8-16 is an unused macro).
// rocprofv3's own roctx library is tried first, the roctracer one second; off (and never loaded) by default.
struct RoctxApi {
    bool tried = false, on = false;
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
static RoctxApi g_roctx;
static void roctx_load() {
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    const char *v = getenv("MLGGD_ROCTX");
    if (!v || !atoi(v)) return;
    for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void *lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) continue;
        g_roctx.push = (int (*)(const char *))dlsym(lib, "roctxRangePushA");
        g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
        if (g_roctx.push && g_roctx.pop) {
            g_roctx.on = true;
            return;
        }
    }
    fprintf(stderr, "mlggd: MLGGD_ROCTX is set but no roctx library could be loaded\n");
}
struct RoctxRange {  // scope guard; a no-op unless MLGGD_ROCTX=1
    bool on;
    explicit RoctxRange(const char *name) : on(g_roctx.on) {
        if (on) g_roctx.push(name);
    }
    ~RoctxRange() {
        if (on) g_roctx.pop();
    }
};

// ------------------------------------------------------------------ engine state
enum KernelClass { KC_TRANSPOSE = 0, KC_FWD, KC_LOSS, KC_DX, KC_DW, KC_UPDATE, KC_COUNT };
static const char *kKernelClassName[KC_COUNT] = {"transpose", "fwd", "loss", "dx", "dw", "update"};

struct mlggd_engine {
    mlggd_config cfg;
    int L = 0;
    int ls[MLGGD_MAXLAYER] = {0};   // true units
    int lsp[MLGGD_MAXLAYER] = {0};  // padded to 32
    int B = 0, Bp = 0, D = 0, Dp = 0, K0 = 0;
    int device = 0;
    hipStream_t stream = nullptr, comm_stream = nullptr;
    // weight-gradient/update kernels (HBM-bound) run on their own stream beside the MFMA-bound
    // dX chain of the main stream; ev_dx[l] orders dw(l) after dX(l) (old-weights semantics),
    // ev_upd orders the next step after the last update
    hipStream_t dw_stream = nullptr;
    hipEvent_t ev_dx[MLGGD_MAXLAYER] = {0}, ev_upd = nullptr;
    int two_streams = 0;  // measured slower on MI355X (cross-stream event waits + contention): off

    float *W[MLGGD_MAXLAYER] = {0}, *dW[MLGGD_MAXLAYER] = {0};
    float *bias[MLGGD_MAXLAYER] = {0}, *dbias[MLGGD_MAXLAYER] = {0};
    float *G[MLGGD_MAXLAYER] = {0}, *gb[MLGGD_MAXLAYER] = {0};
    float *gb_all = nullptr;
    size_t gb_all_count = 0;
    // one-GPU emulation of a world (mlggd_debug_fake_world): sums over the emulated ranks 0..world-2
    float *Gpre[MLGGD_MAXLAYER] = {0}, *gbpre = nullptr, *colsum_tot = nullptr;
    float *Yt[MLGGD_MAXLAYER] = {0}, *Y[MLGGD_MAXLAYER] = {0};
    float *dEdXt[MLGGD_MAXLAYER] = {0}, *dEdX[MLGGD_MAXLAYER] = {0};
    float *slab = nullptr, *outT = nullptr, *eT = nullptr, *pT = nullptr, *colsum = nullptr, *scalefactor = nullptr;
    int S_out = 1;
    float *chunk_in = nullptr, *chunk_targ = nullptr, *chunk_out = nullptr;
    // CV metrics: 0 = outputs copied back, host fp32 accumulation in the reference's order (default: it is what
    // the log lines are compared on); 1 = sums formed on the device (k_cv_reduce), nothing of size n x D leaves it
    int cv_device = 0;
    double *cv_partial = nullptr;
    size_t cv_partial_cap = 0;
    size_t chunk_cap = 0, out_cap = 0;
    int chunk_frames = 0;
    // indexed chunk: raw frame streams + first frame of every sample row
    float *raw_feat = nullptr, *raw_targ = nullptr, *in_bunch = nullptr;
    float *in_bunch_buf[2] = {nullptr, nullptr};  // in_bunch alternates between them when bunches are staged ahead
    int *first_frame = nullptr;
    size_t raw_cap = 0, first_cap = 0;
    // Frame-stream chunks ping-pong between two device buffer sets: the next chunk is uploaded on copy_stream
    // while the kernels of the current one are still running (raw_feat / raw_targ / first_frame / *_cap above
    // always describe the CURRENT set).
    struct RawSet {
        float *feat = nullptr, *targ = nullptr;
        int *first = nullptr;
        size_t raw_cap = 0, first_cap = 0;
        hipEvent_t last_use = nullptr;  // recorded on the main stream after the last kernel that reads the set
    } raw[2];
    int raw_cur = 0;
    hipStream_t copy_stream = nullptr;
    bool indexed = false;
    int fdim = 0, toff = 0, raw_frames = 0;
    unsigned step_counter = 0;
    // launch-plan knobs (defaults chosen from measurements; env overrides for A/B runs)
    int fwd_pipe = 4, dx_pipe = 4;  // main loops software-pipelined inside the wave (forward: operands by LDS-DMA; 1: through staging registers; 0: the round-1 loops, for A/B)
    // 4 waves per workgroup (one per SIMD) since the main loops are pipelined inside the wave: a wave no longer needs a
    // partner on its SIMD to fill its chunk-boundary gaps, and four partial tiles reduce faster than eight
    // 64 x 64-tile forward / dX kernels (kernels64.hip.h): 1 = where the shape gives every CU such a tile (units and
    // frames multiples of 64, tiles >= CUs), 0 = never, 2 = wherever the shape divides (tests, A/B); MLGGD_TILE64
    int tile64 = 1, n_cus = 256;
    int fwd_nw = 4, dx_nw = 4, dw_tile = 1, dw_persist = 1, dwp_per_cu = 2, dw_merge = 1, loss_fuse = 1, tile_map = 0, stage_ahead = 1;  // dw_tile 0 = auto

    // data parallel
    int world = 1, rank = 0;
    // data-parallel exchange: 0 = all-reduce of the weight gradients, 1 = all-gather of their FACTORS (the
    // activations Y_{l-1} and dEdX_l of every rank; each rank then forms the global-minibatch gradient itself)
    int dp_mode = 0;  // 2 = gather + SHARDED update: each rank updates its block of weight rows, then W is all-gathered
    int shard_rows[MLGGD_MAXLAYER] = {0};  // 64-row tile rows of layer l owned by each rank
    // Sharded update with the ACTIVATIONS exchanged by all-to-all (MLGGD_DP_MODE=shard_a2a; dp_mode stays 2): rank o
    // forms the gradient of ITS block of weight rows of every layer, for which it needs every rank's Y_{l-1} only in
    // the units of that block -- 1/world of what the all-gather of the whole Y_{l-1} delivers.  The producers write Y
    // (and the staged input rows) blocked by owner (kernels.hip.h y_blocked_base), each block is one message, and
    // what arrives is a plain [world x frames][block width] matrix.  dEdX_l is still all-gathered (every rank needs
    // all of its units).  Per rank and step at 8 x 128 frames, 2827-2048^3-257: 4 + 23 MB of factors + 51 MB of W
    // blocks = 78 MB instead of 106 -- the one built exchange whose link arithmetic fits the >= 6x target.
    int dp_a2a = 0;
    std::vector<float *> Ysrc[MLGGD_MAXLAYER];  // emulated world only: source rank s's blocked Y_l (all owners' blocks)
    hipEvent_t ev_W[MLGGD_MAXLAYER] = {0}, ev_dw_done = nullptr;
    bool ev_W_pending[MLGGD_MAXLAYER] = {false};
    bool fake_world = false;  // test hook: `world` ranks emulated one after the other on this GPU, no communicator
    float *Yall[MLGGD_MAXLAYER] = {0}, *dEdXall[MLGGD_MAXLAYER] = {0};
    hipEvent_t ev_ready = nullptr, ev_gathered = nullptr;
    // fine-grained factor exchange (few ranks: cheap collectives, few links): every factor is sent as soon as it
    // exists, and ev_layer[l] marks the moment both factors of layer l have arrived
    int dp_fine = 1;
    hipEvent_t ev_layer[MLGGD_MAXLAYER] = {0};
    // Collectives that sit ON the step's critical path (the last factor dEdX_1 in front of the dW launch; W_1 in front
    // of the next forward pass) are issued on the MAIN stream: a hop to the communication stream and back costs two
    // event hand-offs of 6-10 us each plus ~5 us of queue bubble per record / wait (1-rank rehearsal, round 3:
    // 30 us between dX_2 and k_dwp, 21 us between k_dwp and the next forward_1).  MLGGD_DP_MAINLINE=0: everything on
    // the communication stream, as in round 2 (A/B).
    int dp_mainline = 1;
    bool comm_after_dw = false;  // the communication stream has already waited on an event recorded after the last dW launch
    RcclComm comm = nullptr;
    // MLGGD_DP_STAT_COMM=1: the 257-float all-reduce of the ML statistic gets a communicator of its own (ncclCommSplit
    // of `comm`), so that it does not queue behind the factor / weight collectives of `comm` that are still in flight
    // (collectives of ONE communicator execute in issue order whatever stream they are on).  Off by default: untested
    // between two GPUs.
    RcclComm stat_comm = nullptr;
    // all-reduce exchange (dp_mode 0): 1 = reduce-scatter of G_l over weight-row blocks, update of this rank's block
    // only, all-gather of the W blocks (same link bytes as the all-reduce, 1/world of the update traffic);
    // 0 (MLGGD_DP_AR_SHARD=0) = all-reduce of G_l and the full update on every rank (rounds 1-3, A/B)
    int ar_shard = 1;
    hipEvent_t ev_grad[MLGGD_MAXLAYER] = {0}, ev_red[MLGGD_MAXLAYER] = {0}, ev_bias = nullptr, ev_bias_red = nullptr;

    // timing
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    int last_steps = 0;
    bool timing_valid = false;
    // per-kernel-class profiling
    int prof_class = -1, prof_layer = 0, prof_stride = 1;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    hipEvent_t *prof_attach = nullptr;  // event pair waiting to be attached to the next dW launch
    bool prof_attached = false;
    // Data parallel: an event the communication stream is going to wait for can ride on the producing kernel's own
    // dispatch packet (hipExtLaunchKernelGGL stop event = the packet's completion signal) instead of being a packet of
    // its own behind it: a hipEventRecord on the main queue is a ~5 us bubble between two kernels (round 3, 1-rank
    // rehearsal).  stop_ev_next: attach this event to the next GEMM launch; stop_ev_attached: it was.
    hipEvent_t stop_ev_next = nullptr;
    bool stop_ev_attached = false;
    int dp_stopev = 1;  // MLGGD_DP_STOPEV=0: explicit records (A/B)
    double prof_flops = 0, prof_bytes = 0;

    // diagnostic in-kernel phase stamps (one launch of one (class, layer))
    int stamp_class = -1, stamp_layer = 0, stamp_blocks = 0;
    long long *stamp_buf = nullptr;
    size_t stamp_cap = 0;

    struct DwpTable {  // one cached launch plan of the persistent dW kernel (dwp_table)
        DwpJobs key;
        bool fused;
        int grid;
        DwpDesc *dev;
    };
    std::vector<DwpTable> dwp_tables;
    std::vector<void *> allocs;
    std::vector<const void *> lds_attr_done;  // kernels whose dynamic-LDS limit has been raised on this device
};

// buffer for the selected launch, nullptr otherwise; one-shot
static long long *stamps_for(mlggd_engine *e, int cls, int layer, int blocks) {
    if (e->stamp_class != cls || e->stamp_layer != layer || !e->stamp_buf) return nullptr;
    if ((size_t)blocks * 8 > e->stamp_cap) return nullptr;
    e->stamp_class = -1;
    e->stamp_blocks = blocks;
    return e->stamp_buf;
}

static int dev_alloc(mlggd_engine *e, float **p, size_t count) {
    // +64 floats of slack; zero-filled like the reference's devnew_vf (BP_GPU.cu:528-543)
    const size_t bytes = (count + 64) * sizeof(float);
    HIPCHK(hipMalloc((void **)p, bytes));
    // on the engine's own (non-blocking) stream: a null-stream memset would not be ordered
    // before later uploads on e->stream
    HIPCHK(hipMemsetAsync(*p, 0, bytes, e->stream));
    e->allocs.push_back(*p);
    return MLGGD_OK;
}

static int upload_padded(float *dst, int Np, const float *src, int K, int N, hipStream_t st) {
    // [K][N] compact -> [Kp][Np] padded rows
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)Np * 4, src, (size_t)N * 4, (size_t)N * 4, K, hipMemcpyHostToDevice, st));
    return MLGGD_OK;
}
static int download_padded(float *dst, const float *src, int Np, int K, int N, hipStream_t st) {
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)N * 4, src, (size_t)Np * 4, (size_t)N * 4, K, hipMemcpyDeviceToHost, st));
    return MLGGD_OK;
}

// ------------------------------------------------------------------ kernel launch plan
// Times the launches of one kernel class with a pair of HIP events per launch (mlggd_profile_select).
// The GEMM kernels (classes "dw", "fwd", "dx") take the pair INTO the launch (hipExtLaunchKernelGGL start / stop
// events = the dispatch's own begin / end timestamps, what rocprofv3 reports as the kernel's duration); the other
// classes are bracketed by events recorded on the stream before and after, which adds the cost of the bracket.
struct ProfScope {
    mlggd_engine *e;
    hipStream_t st;
    bool on, attach;
    ProfScope(mlggd_engine *eng, int cls, int layer, hipStream_t s = nullptr)
        : e(eng), st(s ? s : eng->stream), on(false), attach(cls == KC_DW || cls == KC_FWD || cls == KC_DX) {
        if (e->prof_class == cls && (e->prof_layer == 0 || e->prof_layer == layer) &&
            e->step_counter % (unsigned)e->prof_stride == 0 && e->prof_used + 2 <= e->prof_ev.size()) {
            on = true;
            if (attach) {
                e->prof_attach = &e->prof_ev[e->prof_used];  // consumed by the launch itself
                e->prof_attached = false;
            } else if (hipEventRecord(e->prof_ev[e->prof_used], st) != hipSuccess) {
                on = false;
            }
        }
    }
    ~ProfScope() {
        if (!on) return;
        if (attach) {
            e->prof_attach = nullptr;
            if (e->prof_attached) e->prof_used += 2;
        } else if (hipEventRecord(e->prof_ev[e->prof_used + 1], st) == hipSuccess) {
            e->prof_used += 2;
        }
    }
};
// launch a GEMM kernel, with the pending event pair of the profiler attached if there is one
template <typename F, typename... Args>
static void launch_timed(mlggd_engine *e, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    if (e->prof_attach && !e->prof_attached) {
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, st, e->prof_attach[0], e->prof_attach[1], 0u, args...);
        e->prof_attached = true;
    } else if (e->stop_ev_next && !e->stop_ev_attached) {
        hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, st, (hipEvent_t) nullptr, e->stop_ev_next, 0u, args...);
        e->stop_ev_attached = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    }
}

static int launch_check(const char *what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(MLGGD_ERR_DEVICE, "launch of %s failed: %s", what, hipGetErrorString(err));
    return MLGGD_OK;
}

template <typename F>
static int ensure_lds(mlggd_engine *e, F fn, size_t bytes) {
    // kernels with > 64 KB of dynamic LDS need the attribute.  It is a per-device property of the loaded code
    // object, so the "already set" list lives in the engine (one engine = one device), not in the process.
    const void *key = (const void *)fn;
    for (const void *d : e->lds_attr_done)
        if (d == key) return MLGGD_OK;
    HIPCHK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    e->lds_attr_done.push_back(key);
    return MLGGD_OK;
}

static int log2_or_minus1(int d) {  // for divmod_by in the kernels
    for (int sft = 0; sft < 31; sft++)
        if (d == (1 << sft)) return sft;
    return -1;
}
static FwdArgs fwd_args(mlggd_engine *e, int l, float *Yrow_out) {
    FwdArgs a{};
    a.W = e->W[l];
    a.Yt_in = e->Yt[l - 1];
    a.bias = e->bias[l];
    a.Yt_out = e->Yt[l];
    a.Y_out = Yrow_out;
    a.slab = e->slab;
    a.Kp = e->lsp[l - 1];
    a.Np = e->lsp[l];
    a.Bp = e->Bp;
    a.N = e->ls[l];
    a.n_tiles = e->lsp[l] / 32;
    a.b_tiles = e->Bp / 32;
    a.S = (l == e->L - 1) ? e->S_out : 1;
    a.map = e->tile_map;
    a.b_shift = log2_or_minus1(a.b_tiles);
    a.s_shift = log2_or_minus1(a.S);
    a.yblk = (e->dp_a2a && l < e->L - 1) ? e->shard_rows[l + 1] * 64 : 0;  // Y_l is layer l+1's K-side factor
    return a;
}

static DxArgs dx_args(mlggd_engine *e, int l) {
    DxArgs a{};
    a.W = e->W[l];
    a.dEdXt = e->dEdXt[l];
    a.Yt_prev = e->Yt[l - 1];
    a.dEdXt_prev = e->dEdXt[l - 1];
    a.dEdX_prev = e->dEdX[l - 1];
    a.Kp = e->lsp[l - 1];
    a.Np = e->lsp[l];
    a.Bp = e->Bp;
    a.k_tiles = e->lsp[l - 1] / 32;
    a.b_tiles = e->Bp / 32;
    a.map = e->tile_map;
    a.b_shift = log2_or_minus1(a.b_tiles);
    return a;
}

static DwpArgs dwp_args(mlggd_engine *e, int l, const float *in_rows, const float *Yrow_prev, float nf) {
    DwpArgs a;
    memset(&a, 0, sizeof(a));  // every byte, tail padding included: dwp_table compares plans with memcmp
    const int Kp = e->lsp[l - 1], Np = e->lsp[l];
    a.Yrow = (l == 1) ? in_rows : Yrow_prev;
    a.dEdX = e->dEdX[l];
    a.Wt = e->W[l];
    a.delta = e->dW[l];
    a.G = e->G[l];
    a.bias = e->bias[l];
    a.dbias = e->dbias[l];
    a.gb = e->gb[l];
    a.ldA = Kp;  // layer 1 reads the staged, padded copy of the minibatch's rows (in_bunch), not the caller's chunk
    a.K = e->ls[l - 1];
    a.N = e->ls[l];
    a.Kp = Kp;
    a.Np = Np;
    a.B = e->B;
    a.n_wg = (Np + 63) / 64;
    a.ntiles = ((Kp + 63) / 64) * a.n_wg;
    a.nf = nf;
    a.mom = e->cfg.momentum;
    a.lr = e->cfg.lrate;
    a.wc = e->cfg.weightcost;
    a.k_first = 0;
    a.wd_off = 0;
    a.do_bias = 1;
    return a;
}

// one minibatch of the resident chunk: expanded rows, or raw streams + per-row first frame
struct Bunch {
    const float *in;    // expanded: first row of the bunch; indexed: the raw feature stream
    const float *targ;  // expanded: first target row;       indexed: the raw target stream
    const int *first;   // indexed: first frame of each sample row of this bunch, else nullptr
};
static Bunch bunch_at(mlggd_engine *e, int sample) {
    Bunch b;
    if (e->indexed) {
        b.in = e->raw_feat;
        b.targ = e->raw_targ;
        b.first = e->first_frame + sample;
    } else {
        b.in = e->chunk_in + (size_t)sample * e->K0;
        b.targ = e->chunk_targ + (size_t)sample * e->D;
        b.first = nullptr;
    }
    return b;
}
// row-major [Bp][lsp[0]] copy of the bunch (16-byte aligned rows, zero pads) written by the input-staging kernel:
// the layer-1 dW operand, the data-parallel input factor and what dropout masks
static const float *bunch_rows(mlggd_engine *e, const Bunch &) { return e->in_bunch; }

static StageArgs stage_args(mlggd_engine *e, const Bunch &bn, int frames, float *rows_out) {
    StageArgs a;
    a.in = bn.in;
    a.ld = e->K0;
    a.B = frames;
    a.K = e->K0;
    a.Kp = e->lsp[0];
    a.inT = e->Yt[0];
    a.Bp = e->Bp;
    a.b_tiles = e->Bp / 32;
    a.first = bn.first;
    a.fdim = e->fdim;
    a.rows_out = rows_out;
    a.yblk = e->dp_a2a ? e->shard_rows[1] * 64 : 0;  // the staged rows are layer 1's K-side factor
    return a;
}
static int stage_blocks(const mlggd_engine *e) { return (e->lsp[0] / 32) * (e->Bp / 32); }
// the other half of the in_bunch double buffer (target of a bunch staged one step ahead)
static float *in_bunch_other(mlggd_engine *e) {
    return e->in_bunch == e->in_bunch_buf[0] ? e->in_bunch_buf[1] : e->in_bunch_buf[0];
}

// Which GEMM form a layer takes.  fwd: layer l's forward (output tile over units of l x frames); dx: the dX launched for
// layer l (output tile over units of l-1 x frames).  The 64 x 64 form needs a whole number of such tiles and -- unless
// forced -- at least one per CU; the output layer always takes the slab form (small N: K split over workgroups).
static bool tile64_ok(const mlggd_engine *e, int units_p) {
    if (e->tile64 == 0 || units_p % 64 != 0 || e->Bp % 64 != 0) return false;
    return e->tile64 == 2 || (long)(units_p / 64) * (e->Bp / 64) >= e->n_cus;
}
static bool fwd64_used(const mlggd_engine *e, int l) {
    if (e->fwd_pipe != 4 || e->fwd_nw != 4 || !tile64_ok(e, e->lsp[l])) return false;
    return l != e->L - 1 || e->S_out == 1;  // an output layer only when it needs no split-K slabs (it is that wide)
}
static bool dx64_used(const mlggd_engine *e, int l) {
    return l >= 2 && e->dx_pipe == 4 && e->dx_nw == 4 && tile64_ok(e, e->lsp[l - 1]);
}

static int run_transpose(mlggd_engine *e, const Bunch &bn, int frames) {
    ProfScope ps(e, KC_TRANSPOSE, 0);
    hipLaunchKernelGGL(k_transpose_in, dim3(stage_blocks(e)), dim3(256), 0, e->stream,
                       stage_args(e, bn, frames, e->in_bunch));
    return launch_check("k_transpose_in");
}

// every consumer of W on the main stream first waits for the all-gathers of the previous step
static int wait_weight_gathers(mlggd_engine *e, int l_lo, int l_hi) {
    for (int l = l_lo; l <= l_hi; l++)
        if (e->ev_W_pending[l]) {
            HIPCHK(hipStreamWaitEvent(e->stream, e->ev_W[l], 0));
            e->ev_W_pending[l] = false;
        }
    return MLGGD_OK;
}

static int run_dropout(mlggd_engine *e, int layer, const float *chunk_rows) {
    // BP_GPU.cu:344-355: visible_omit on the input, hid_omit on hidden activations
    const float p = (layer == 0) ? e->cfg.visible_omit : e->cfg.hid_omit;
    const size_t n = (size_t)e->lsp[layer] * e->Bp;
    float *rows = (layer == 0) ? const_cast<float *>(chunk_rows) : e->Y[layer];
    const int ld = e->lsp[layer];
    hipLaunchKernelGGL(k_dropout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->Yt[layer], rows,
                       e->ls[layer], e->lsp[layer], ld, e->B, e->Bp, p, (unsigned)e->cfg.random_seed,
                       e->step_counter * 16u + (unsigned)layer);
    return launch_check("k_dropout");
}

// prestaged: Yt[0] (and, for frame-stream chunks, the other in_bunch buffer) already hold this
// bunch -- the previous training step staged it alongside its loss kernel
static int gather_begin(mlggd_engine *e, bool already_ordered);
static int gather_end(mlggd_engine *e);
static void gather_arm(mlggd_engine *e);
static int gather_begin_armed(mlggd_engine *e);
static int gather_one(mlggd_engine *e, const float *src, float *dst, size_t count, hipStream_t st);
static int exchange_y(mlggd_engine *e, int l, const float *src, hipStream_t st);
enum { GATHER_INPUT = 1, GATHER_HIDDEN = 2, GATHER_HIDDEN_EACH = 4 };  // data-parallel factor exchange issued from inside the forward pass

static int run_forward(mlggd_engine *e, const Bunch &bn, int frames, bool training, bool prestaged = false,
                       int gather_flags = 0) {
    e->stop_ev_next = nullptr;  // an event armed by a launch sequence that bailed out early must not ride on an unrelated launch
    e->stop_ev_attached = false;
    if (prestaged) {
        e->in_bunch = in_bunch_other(e);
    } else {
        CHK(run_transpose(e, bn, frames));
    }
    const float *in_rows = bunch_rows(e, bn);
    if (gather_flags & GATHER_INPUT) {  // the input rows can travel from the first microsecond of the step
        // prestaged rows were written by the previous step's loss launch and Yall[0] was last read by its dW launch: if
        // the communication stream has already waited on an event recorded after that launch (sharded update: the W
        // gathers did), it needs no new event
        CHK(gather_begin(e, prestaged && e->comm_after_dw && e->dp_mainline));
        CHK(exchange_y(e, 0, in_rows, nullptr));
        CHK(gather_end(e));
    }
    e->comm_after_dw = false;
    const int b_tiles = e->Bp / 32;
    const bool drop = training && e->cfg.dropoutflag == 1;
    const bool cvscale = !training && e->cfg.dropoutflag == 1;
    if (drop) CHK(run_dropout(e, 0, in_rows));
    for (int l = 1; l < e->L; l++) {
        const int Kp = e->lsp[l - 1], Np = e->lsp[l], n_tiles = Np / 32;
        CHK(wait_weight_gathers(e, l, l));  // sharded data parallel: W_l of the previous step may still be arriving
        if (cvscale) {  // DevWeightMultiP before the GEMM, BP_GPU.cu:484-489
            const float keep = 1.0f - ((l == 1) ? e->cfg.visible_omit : e->cfg.hid_omit);
            hipLaunchKernelGGL(k_scale, dim3(1024), dim3(256), 0, e->stream, e->W[l], (size_t)Kp * Np, keep);
        }
        {
            // this layer's activations are sent right after the launch: the event rides on it
            if (((gather_flags & GATHER_HIDDEN_EACH) && l < e->L - 1) || ((gather_flags & GATHER_HIDDEN) && l == e->L - 2))
                gather_arm(e);
            ProfScope ps(e, KC_FWD, l);
            FwdArgs fa = fwd_args(e, l, e->Y[l]);
            if (fwd64_used(e, l)) {
                fa.n_tiles = Np / 64;
                fa.b_tiles = e->Bp / 64;
                fa.b_shift = log2_or_minus1(fa.b_tiles);
                const size_t lds = t64_lds_floats() * sizeof(float);
                if (l != e->L - 1) {
                    CHK(ensure_lds(e, k_fwd64<FWD_SIGMOID>, lds));
                    launch_timed(e, k_fwd64<FWD_SIGMOID>, dim3(fa.n_tiles * fa.b_tiles), dim3(256), lds, e->stream, fa);
                } else {
                    CHK(ensure_lds(e, k_fwd64<FWD_SLAB>, lds));
                    launch_timed(e, k_fwd64<FWD_SLAB>, dim3(fa.n_tiles * fa.b_tiles), dim3(256), lds, e->stream, fa);
                }
            } else if (l != e->L - 1) {
                long long *st = stamps_for(e, KC_FWD, l, n_tiles * b_tiles);
#define LAUNCH_FWD(NW, PIPE)                                                                                \
    {                                                                                                       \
        const size_t lds = fwd_lds_floats<NW, PIPE>() * sizeof(float);                                      \
        CHK(ensure_lds(e, k_fwd<FWD_SIGMOID, NW, PIPE>, lds));                                              \
        launch_timed(e, k_fwd<FWD_SIGMOID, NW, PIPE>, dim3(n_tiles * b_tiles), dim3(64 * NW), lds, e->stream, fa, st); \
    }
                // PIPE 4 (default): main loop software-pipelined inside the wave, operands by LDS-DMA; 1: the same
                // pipeline through staging registers (MLGGD_FWD_PIPE=1); 0: the round-1 loop (MLGGD_FWD_PIPE=0, and the
                // 16-wave form, whose 128-VGPR budget and 160 KB of LDS the pipelined loops do not fit)
                if (e->fwd_nw == 16) LAUNCH_FWD(16, 0)
                else if (e->fwd_nw == 8 && e->fwd_pipe == 4) LAUNCH_FWD(8, 4)
                else if (e->fwd_nw == 8 && e->fwd_pipe) LAUNCH_FWD(8, 1)
                else if (e->fwd_nw == 8) LAUNCH_FWD(8, 0)
                else if (e->fwd_pipe == 4) LAUNCH_FWD(4, 4)
                else if (e->fwd_pipe) LAUNCH_FWD(4, 1)
                else LAUNCH_FWD(4, 0)
#undef LAUNCH_FWD
            } else {
                const int nwg = n_tiles * b_tiles * e->S_out;
                if (e->fwd_pipe == 4) {
                    const size_t lds = fwd_lds_floats<4, 4>() * sizeof(float);
                    CHK(ensure_lds(e, k_fwd<FWD_SLAB, 4, 4>, lds));
                    launch_timed(e, k_fwd<FWD_SLAB, 4, 4>, dim3(nwg), dim3(256), lds, e->stream, fa, (long long *)nullptr);
                } else {
                    const size_t lds = fwd_lds_floats<4, 1>() * sizeof(float);
                    launch_timed(e, k_fwd<FWD_SLAB, 4, 1>, dim3(nwg), dim3(256), lds, e->stream, fa, (long long *)nullptr);
                }
            }
        }
        CHK(launch_check("k_fwd"));
        if (cvscale) {  // BP_GPU.cu:496-501
            const float keep = 1.0f - ((l == 1) ? e->cfg.visible_omit : e->cfg.hid_omit);
            hipLaunchKernelGGL(k_scale, dim3(1024), dim3(256), 0, e->stream, e->W[l], (size_t)Kp * Np, 1.0f / keep);
        }
        if (drop && l != e->L - 1) CHK(run_dropout(e, l, nullptr));
        if ((gather_flags & GATHER_HIDDEN_EACH) && l < e->L - 1) {  // each hidden layer's activations at once
            CHK(gather_begin_armed(e));
            CHK(exchange_y(e, l, e->Y[l], nullptr));
            CHK(gather_end(e));
        }
        if ((gather_flags & GATHER_HIDDEN) && l == e->L - 2) {  // all hidden activations exist: send them
            CHK(gather_begin_armed(e));                         // beside the output layer, the loss and dX
            for (int g = 1; g < e->L - 1; g++) CHK(exchange_y(e, g, e->Y[g], nullptr));
            CHK(gather_end(e));
        }
    }
    return MLGGD_OK;
}

template <int T>
static int launch_dw(mlggd_engine *e, int l, const float *in_rows, bool fused, float nf, hipStream_t st) {
    const int Kp = e->lsp[l - 1], Np = e->lsp[l];
    const int k_wg = (Kp + 64 * T - 1) / (64 * T), n_wg = (Np + 64 * T - 1) / (64 * T);
    const float *A = (l == 1) ? in_rows : e->Y[l - 1];
    const int ldA = Kp;
    const size_t lds = (size_t)2 * (64 * T) * (64 * T) * sizeof(float);
    if (fused) CHK(ensure_lds(e, k_dw<T, true>, lds));
    else CHK(ensure_lds(e, k_dw<T, false>, lds));
    if (fused)
        launch_timed(e, k_dw<T, true>, dim3(k_wg * n_wg), dim3(256), lds, st, A, ldA, (const float *)e->dEdX[l], e->W[l],
                     e->dW[l], (float *)nullptr, e->bias[l], e->dbias[l], (float *)nullptr, e->ls[l - 1], e->ls[l], Np,
                     e->B, e->Bp, n_wg, nf, e->cfg.momentum, e->cfg.lrate, e->cfg.weightcost,
                     stamps_for(e, KC_DW, l, k_wg * n_wg));
    else
        launch_timed(e, k_dw<T, false>, dim3(k_wg * n_wg), dim3(256), lds, st, A, ldA, (const float *)e->dEdX[l], e->W[l],
                     e->dW[l], e->G[l], e->bias[l], e->dbias[l], e->gb[l], e->ls[l - 1], e->ls[l], Np, e->B, e->Bp, n_wg,
                     nf, e->cfg.momentum, e->cfg.lrate, e->cfg.weightcost, (long long *)nullptr);
    return launch_check("k_dw");
}

// persistent pipelined dW (+update) kernel: Bp = 64*H
static int dwp_grid(mlggd_engine *e, int ntiles) {
    int grid = 256 * (e->dwp_per_cu > 0 ? e->dwp_per_cu : 2);
    return grid > ntiles ? ntiles : grid;
}
// jobs for layers lhi, lhi-1, ..., llo in one launch
static_assert(DWP_MAXJOBS >= MLGGD_MAXLAYER, "job table too small");
static DwpJobs dwp_jobs(mlggd_engine *e, int lhi, int llo, const float *in_rows, float nf) {
    DwpJobs J;
    memset(&J, 0, sizeof(J));
    int nj = 0, end = 0;
    for (int l = lhi; l >= llo; l--) {
        J.job[nj] = dwp_args(e, l, in_rows, e->Y[l - 1], nf);
        end += J.job[nj].ntiles;
        J.tile_end[nj++] = end;
    }
    J.njobs = nj;
    J.total = end;
    return J;
}
// The per-tile records the persistent dW kernel walks (kernels.hip.h DwpDesc), built once per launch plan and
// kept on the device: the plan is the job list (which layers / row blocks / operands) and the grid.  A training
// run uses two plans (the layer-1 operand alternates between the two staged-row buffers), a data-parallel run
// a few more.  Hyper-parameters are not part of the plan: they travel in DwpConst with every launch.
static int dwp_table(mlggd_engine *e, const DwpJobs &J, bool fused, int grid, const DwpDesc **out) {
    // The key is rebuilt field by field into zeroed storage, so padding bytes (DwpArgs has 4 at its tail) and the
    // unused job slots never take part in the memcmp.  Hyper-parameters and the frame count are not part of a plan.
    DwpJobs key;
    memset(&key, 0, sizeof(key));
    for (int j = 0; j < J.njobs && j < DWP_MAXJOBS; j++) {
        const DwpArgs &s = J.job[j];
        DwpArgs &a = key.job[j];
        a.Yrow = s.Yrow; a.dEdX = s.dEdX; a.Wt = s.Wt; a.delta = s.delta; a.G = s.G;
        a.bias = s.bias; a.dbias = s.dbias; a.gb = s.gb;
        a.ldA = s.ldA; a.K = s.K; a.N = s.N; a.Kp = s.Kp; a.Np = s.Np; a.n_wg = s.n_wg; a.ntiles = s.ntiles;
        a.k_first = s.k_first; a.wd_off = s.wd_off; a.do_bias = s.do_bias; a.k_base = s.k_base;
        key.tile_end[j] = J.tile_end[j];
    }
    key.njobs = J.njobs;
    key.total = J.total;
    for (const auto &t : e->dwp_tables)
        if (t.fused == fused && t.grid == grid && memcmp(&t.key, &key, sizeof(key)) == 0) {
            *out = t.dev;
            return MLGGD_OK;
        }
    // A run uses two plans on one GPU (the layer-1 operand alternates between the two staged-row buffers) and a
    // handful more in the data-parallel modes.  A miss costs a hipMalloc and a blocking copy in the middle of a
    // step, so a plan count that keeps growing is a bug (a key that never matches): fail loudly, do not leak.
    if (e->dwp_tables.size() >= 64)
        return fail(MLGGD_ERR_STATE, "dW tile table: %zu launch plans cached and still missing -- plan key unstable",
                    e->dwp_tables.size());
    std::vector<DwpDesc> recs((size_t)J.total + 2 * (size_t)grid);
    size_t n = 0;
    for (int j = 0; j < J.njobs; j++) {
        const DwpArgs &a = J.job[j];
        for (int tl = 0; tl < a.ntiles; tl++, n++) {
            const int kt = a.k_first + tl / a.n_wg, nt = tl % a.n_wg;
            const int k0 = kt * 64, n0 = nt * 64;
            DwpDesc &d = recs[n];
            d.A = a.Yrow + (k0 - a.k_base);
            d.Bm = a.dEdX + n0;
            const size_t base = (size_t)k0 * a.Np + n0;
            int rows = a.K - k0;
            rows = rows < 0 ? 0 : rows > 64 ? 64 : rows;
            if (a.wd_off) {  // bias-only tile: W / delta are neither read nor written
                rows = 0;
                d.W = a.Wt;
                d.D = a.delta;
                d.szW = 0;
            } else {
                d.W = (fused ? a.Wt : a.G) + base;
                d.D = fused ? a.delta + base : nullptr;
                d.szW = (unsigned)(((size_t)a.Kp * a.Np - base) * sizeof(float));
            }
            const int colsw = a.Np - n0 < 64 ? a.Np - n0 : 64;
            int nbias = (a.do_bias && kt == 0) ? a.N - n0 : 0;
            nbias = nbias < 0 ? 0 : nbias > 64 ? 64 : nbias;
            d.bias = (fused ? a.bias : a.gb) + n0;
            d.dbias = fused ? a.dbias + n0 : nullptr;
            d.ldA = a.ldA;
            d.Np = a.Np;
            d.packed = (unsigned)rows | ((unsigned)colsw << 8) | ((unsigned)nbias << 16) | (1u << 24);
        }
    }
    if (n != (size_t)J.total) return fail(MLGGD_ERR_STATE, "dW tile table: %zu records for %d tiles", n, J.total);
    for (; n < recs.size(); n++) {  // "no such tile": empty descriptors -- loads return zeros, stores are dropped
        DwpDesc &d = recs[n];
        d = recs[0];
        d.packed = 0;
        d.szW = 0;
    }
    mlggd_engine::DwpTable t;
    t.key = key;
    t.fused = fused;
    t.grid = grid;
    t.dev = nullptr;
    HIPCHK(hipMalloc((void **)&t.dev, recs.size() * sizeof(DwpDesc)));
    // pageable source, synchronous copy: complete when the call returns, so the launch that follows sees it
    HIPCHK(hipMemcpy(t.dev, recs.data(), recs.size() * sizeof(DwpDesc), hipMemcpyHostToDevice));
    e->dwp_tables.push_back(t);
    *out = t.dev;
    return MLGGD_OK;
}

template <int H>
static int launch_dwp_t(mlggd_engine *e, const DwpJobs &J, bool fused, hipStream_t st, int stamp_layer) {
    const size_t lds = dwp_lds_floats() * sizeof(float);
    const int grid = dwp_grid(e, J.total);
    if (J.total < 1) return MLGGD_OK;
    const DwpDesc *table = nullptr;
    DwpConst C;
    C.B = J.job[0].B;
    C.nf = J.job[0].nf;
    C.mom = J.job[0].mom;
    C.lr = J.job[0].lr;
    C.wc = J.job[0].wc;
    bool bias_only = J.njobs > 0, any_bias_only = false;  // only the sharded data-parallel plans hold bias-only jobs
    for (int j = 0; j < J.njobs; j++) {
        bias_only = bias_only && J.job[j].wd_off != 0;
        any_bias_only = any_bias_only || J.job[j].wd_off != 0;
    }
    if (any_bias_only) {
        if (!bias_only || !fused) return fail(MLGGD_ERR_STATE, "bias-only jobs need a launch of their own on the fused path");
        CHK(dwp_table(e, J, fused, grid, &table));
        unsigned nb;
        memcpy(&nb, &C.nf, sizeof(nb));
        if ((nb & 0x007FFFFFu) == 0u && C.nf >= 1.0f) {
            CHK(ensure_lds(e, k_dwp_bias<H, true>, lds));
            hipLaunchKernelGGL((k_dwp_bias<H, true>), dim3(grid), dim3(256), lds, st, table, J.total, C, (long long *)nullptr);
        } else {
            CHK(ensure_lds(e, k_dwp_bias<H, false>, lds));
            hipLaunchKernelGGL((k_dwp_bias<H, false>), dim3(grid), dim3(256), lds, st, table, J.total, C, (long long *)nullptr);
        }
        return launch_check("k_dwp_bias");
    }
    CHK(dwp_table(e, J, fused, grid, &table));
    if constexpr (H == 2 || H == 8) {
        if (e->stamp_class == KC_DW && e->stamp_layer == -1 && e->stamp_buf && fused && C.nf == 128.0f) {
            // diagnostic: the twin kernel with per-phase cycle sums (rows grid .. 2*grid-1 of the stamp buffer)
            long long *sp = stamps_for(e, KC_DW, -1, 2 * grid);
            if (sp) {
                CHK(ensure_lds(e, k_dwp_phases<H>, lds));
                hipLaunchKernelGGL((k_dwp_phases<H>), dim3(grid), dim3(256), lds, st, table, J.total, C, sp);
                return launch_check("k_dwp_phases");
            }
        }
    }
    long long *stamps = stamps_for(e, KC_DW, stamp_layer, grid);
    if constexpr (H == 2) {
        static const int abl = getenv("MLGGD_DWP_ABLATE") ? atoi(getenv("MLGGD_DWP_ABLATE")) : 0;
        if (abl && fused) {  // timing-only diagnostics: results are wrong by construction
#define DWP_ABL(A_)                                                                                       \
    case A_:                                                                                              \
        CHK(ensure_lds(e, k_dwp_ablate<H, A_>, lds));                                                     \
        launch_timed(e, k_dwp_ablate<H, A_>, dim3(grid), dim3(256), lds, st, table, J.total, C, stamps);  \
        return launch_check("k_dwp_ablate");
            switch (abl) {
                DWP_ABL(1) DWP_ABL(2) DWP_ABL(3) DWP_ABL(4) DWP_ABL(7) DWP_ABL(15) DWP_ABL(31) DWP_ABL(63) DWP_ABL(16) DWP_ABL(48) DWP_ABL(256) DWP_ABL(512) DWP_ABL(576)
            default: return fail(MLGGD_ERR_ARG, "MLGGD_DWP_ABLATE=%d is not built", abl);
            }
#undef DWP_ABL
        }
    }
    // G / n is a multiply when n is a power of two (bit-identical, see kernels.hip.h POW2)
    unsigned nfbits;
    memcpy(&nfbits, &C.nf, sizeof(nfbits));
    const bool pow2 = (nfbits & 0x007FFFFFu) == 0u && C.nf >= 1.0f;
#define DWP_LAUNCH(FUSED_, POW2_)                                                                         \
    {                                                                                                     \
        CHK(ensure_lds(e, k_dwp<H, FUSED_, POW2_>, lds));                                                 \
        launch_timed(e, k_dwp<H, FUSED_, POW2_>, dim3(grid), dim3(256), lds, st, table, J.total, C, stamps); \
    }
    if (fused && pow2) DWP_LAUNCH(true, true)
    else if (fused) DWP_LAUNCH(true, false)
    else if (pow2) DWP_LAUNCH(false, true)
    else DWP_LAUNCH(false, false)
#undef DWP_LAUNCH
    return launch_check("k_dwp");
}
static bool dwp_usable(const mlggd_engine *e) {
    const int Hh = e->Bp / 64;
    return e->dw_persist && e->Bp % 64 == 0 && (Hh == 1 || Hh == 2 || Hh == 4 || Hh == 8);
}
// units: 64-frame units per tile = (frames the jobs' operands hold) / 64
static int launch_dwp(mlggd_engine *e, const DwpJobs &J, bool fused, hipStream_t st, int stamp_layer, int units = 0) {
    switch (units > 0 ? units : e->Bp / 64) {
    case 1: return launch_dwp_t<1>(e, J, fused, st, stamp_layer);
    case 2: return launch_dwp_t<2>(e, J, fused, st, stamp_layer);
    case 4: return launch_dwp_t<4>(e, J, fused, st, stamp_layer);
    case 8: return launch_dwp_t<8>(e, J, fused, st, stamp_layer);
    case 16: return launch_dwp_t<16>(e, J, fused, st, stamp_layer);
    default: return fail(MLGGD_ERR_STATE, "no dW kernel for %d units per tile", units);
    }
}

// ---- data-parallel exchange by all-gather of the gradient's factors (dp_mode 1) -------------------
// G_l = sum over ranks of Y_{l-1,r}^T dEdX_{l,r} is a product of [frames x units] matrices that are 8-16 x
// smaller than G_l itself at 128 frames per rank (7.9 MB of factors vs 58.8 MB of gradients per rank and
// step), so the ranks exchange the factors: every rank gathers all ranks' Y_{l-1} and dEdX_l (rank-major
// rows = the rows of ONE minibatch of world*B frames) and runs the same fused dW + update kernel over the
// global minibatch.  All ranks compute bit-identical updates, so no gradient, bias or weight ever crosses
// the links, there is no separate update pass, and the result is exactly the single-device step with
// bunchsize world*B.  The price is world x the dW MFMA work per rank.
static bool gather_usable(const mlggd_engine *e, int world) {
    const long frames = (long)world * e->Bp;
    const long units = frames / 64;
    return e->dw_persist && e->B == e->Bp && frames % 64 == 0 &&
           (units == 1 || units == 2 || units == 4 || units == 8 || units == 16);
}
static int gather_alloc(mlggd_engine *e) {
    const size_t rows = (size_t)e->world * e->Bp;
    for (int l = 0; l < e->L - 1; l++) {
        // all-to-all form: this rank's block of units from every rank ([world x Bp][block width]); shard_alloc ran first
        const size_t width = e->dp_a2a ? (size_t)e->shard_rows[l + 1] * 64 : (size_t)e->lsp[l];
        CHK(dev_alloc(e, &e->Yall[l], rows * width));
        if (e->dp_a2a) {
            // the producers write Y_l (l = 0: the staged rows, two buffers) blocked by owner: world blocks of [Bp][width]
            if (l == 0) {
                CHK(dev_alloc(e, &e->in_bunch_buf[0], rows * width));
                CHK(dev_alloc(e, &e->in_bunch_buf[1], rows * width));
                e->in_bunch = e->in_bunch_buf[0];
            } else {
                CHK(dev_alloc(e, &e->Y[l], rows * width));
            }
            if (e->fake_world)
                for (int s = 0; s + 1 < e->world; s++) {
                    float *q = nullptr;
                    CHK(dev_alloc(e, &q, rows * width));
                    e->Ysrc[l].push_back(q);
                }
        }
    }
    for (int l = 1; l < e->L; l++) CHK(dev_alloc(e, &e->dEdXall[l], rows * e->lsp[l]));
    HIPCHK(hipEventCreateWithFlags(&e->ev_ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_gathered, hipEventDisableTiming));
    for (int l = 1; l < e->L; l++) HIPCHK(hipEventCreateWithFlags(&e->ev_layer[l], hipEventDisableTiming));
    // Default: the GROUPED factor exchange (four exchanges per step) at every world size.  Round 2 chose the
    // fine-grained form (every factor sent the moment it exists, dW launch split in two) up to 5 ranks on a link model;
    // round 3 measured what it costs before any link time -- +15 us of device time per step (seven more hand-offs
    // between the streams) and twice the host enqueue time (143 vs 70 us per step, next to a 170 us device step) --
    // against a modelled gain of ~20 us of hidden link time at 2 ranks.  MLGGD_DP_FINE=1 selects it; bench.py measures
    // both forms in every multi-GPU run (dp_arms.gather_other_granularity).
    if (const char *v = getenv("MLGGD_DP_FINE")) e->dp_fine = atoi(v);
    else e->dp_fine = 0;
    return MLGGD_OK;
}
// Launch-order knobs of every data-parallel mode (A/B switches; the defaults rest on a 1-rank rehearsal and on
// emulated worlds -- nothing has run between two GPUs yet).
static void dp_knobs(mlggd_engine *e) {
    if (const char *v = getenv("MLGGD_DP_MAINLINE")) e->dp_mainline = atoi(v);
    if (const char *v = getenv("MLGGD_DP_STOPEV")) e->dp_stopev = atoi(v);
    if (const char *v = getenv("MLGGD_DP_AR_SHARD")) e->ar_shard = atoi(v) != 0;
}
// one rank's block -> every rank's slot r of dst (on the communication stream, or on `st`)
static int gather_one(mlggd_engine *e, const float *src, float *dst, size_t count, hipStream_t st = nullptr) {
    if (!st) st = e->comm_stream;
    if (e->fake_world) {  // the other ranks' slots were filled by fake_world_prepass
        HIPCHK(hipMemcpyAsync(dst + (size_t)e->rank * count, src, count * sizeof(float), hipMemcpyDeviceToDevice, st));
        return MLGGD_OK;
    }
    NCCLCHK(g_rccl.AllGather(src, dst, count, 7 /* ncclFloat32 */, e->comm, st));
    return MLGGD_OK;
}
// The K-side factor of layer l+1 (Y_l; l = 0: the staged input rows) to the ranks that need it: every rank's rows to every
// rank (all-gather), or -- all-to-all form of the sharded update -- block p of this rank's owner-blocked Y_l to rank p,
// slot s of Yall[l] receiving rank s's block for THIS rank.  Called between gather_begin / gather_end, whose
// ncclGroupStart / End make the send / recv pairs one collective.
static int exchange_y(mlggd_engine *e, int l, const float *src, hipStream_t st = nullptr) {
    if (!e->dp_a2a) return gather_one(e, src, e->Yall[l], (size_t)e->Bp * e->lsp[l], st);
    if (e->fake_world) return MLGGD_OK;  // the emulation assembles Yall[l] per virtual owner from the sources' blocks (run_step)
    if (!st) st = e->comm_stream;
    const size_t cnt = (size_t)e->Bp * e->shard_rows[l + 1] * 64;
    for (int p = 0; p < e->world; p++) {
        NCCLCHK((int)g_rccl.Send_(src + (size_t)p * cnt, cnt, ncclFloat32, p, e->comm, st));
        NCCLCHK((int)g_rccl.Recv_(e->Yall[l] + (size_t)p * cnt, cnt, ncclFloat32, p, e->comm, st));
    }
    return MLGGD_OK;
}
// the communication stream picks up everything the main stream has produced so far (already_ordered: it has, through
// an earlier event, and nothing newer is needed -- no record / wait, i.e. no bubble on the main queue)
static int gather_begin(mlggd_engine *e, bool already_ordered = false) {
    if (!already_ordered) {
        HIPCHK(hipEventRecord(e->ev_ready, e->stream));
        HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_ready, 0));
    }
    if (!e->fake_world) NCCLCHK(g_rccl.GroupStart());
    return MLGGD_OK;
}
static int gather_end(mlggd_engine *e) {
    if (!e->fake_world) NCCLCHK(g_rccl.GroupEnd());
    return MLGGD_OK;
}
// The next GEMM launch on the main stream produces what the communication stream is going to send: let ev_ready ride
// on that launch (see stop_ev_next).  gather_begin_armed() then only records if the launch could not take the event
// (a profiling pass owns the launch's event slots, or no GEMM launch came).
static void arm_stop(mlggd_engine *e, hipEvent_t ev) {
    if (!e->dp_stopev) return;
    e->stop_ev_next = ev;
    e->stop_ev_attached = false;
}
static bool take_stop(mlggd_engine *e) {  // true: the armed event went out with a launch
    const bool attached = e->stop_ev_next != nullptr && e->stop_ev_attached;
    e->stop_ev_next = nullptr;
    e->stop_ev_attached = false;
    return attached;
}
static void gather_arm(mlggd_engine *e) { arm_stop(e, e->ev_ready); }
static int gather_begin_armed(mlggd_engine *e) {
    if (!take_stop(e)) HIPCHK(hipEventRecord(e->ev_ready, e->stream));
    HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_ready, 0));
    if (!e->fake_world) NCCLCHK(g_rccl.GroupStart());
    return MLGGD_OK;
}
// ---- sharded update (dp_mode 2) --------------------------------------------------------------------
// At 8 ranks the replicated dW launch is 8 x the MFMA work (298 us).  Here rank r forms only ITS block of
// weight rows of every layer over the gathered minibatch -- 1/world of the tiles -- keeps delta for that
// block only, and the updated W blocks are all-gathered in place, layer 1 first so that the next forward
// pass starts as soon as W_1 has arrived.  W is allocated with world*shard rows (zero pad rows) so that
// every rank's block has the same size.  Biases: the tiles of row block 0 run on every rank as bias-only
// jobs (no W/delta access), so all ranks apply the identical bias update without another collective.
static int shard_alloc(mlggd_engine *e) {
    for (int l = 1; l < e->L; l++) {
        const int Kp = e->lsp[l - 1], Np = e->lsp[l];
        const int k_wg = (Kp + 63) / 64;
        e->shard_rows[l] = (k_wg + e->world - 1) / e->world;
        const size_t rows = (size_t)e->shard_rows[l] * 64 * e->world;
        if (rows > (size_t)Kp) {  // grow W_l; the pad rows are never touched by a kernel (descriptors end at Kp)
            float *nw = nullptr;
            CHK(dev_alloc(e, &nw, rows * Np));
            HIPCHK(hipMemcpyAsync(nw, e->W[l], (size_t)Kp * Np * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            e->W[l] = nw;  // the old buffer stays on the free list
            for (auto &t : e->dwp_tables) hipFree(t.dev);  // tile records built so far point into the old buffer
            e->dwp_tables.clear();
        }
        HIPCHK(hipEventCreateWithFlags(&e->ev_W[l], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&e->ev_dw_done, hipEventDisableTiming));
    return MLGGD_OK;
}
// jobs of (virtual) rank r.  pass 0: its row blocks of every layer (own_bias: whether its own row-block-0 job
// applies the bias update); pass 1: the bias-only jobs of the layers whose row block 0 it does not own -- a
// separate launch of the bias-only kernel (k_dwp_bias)
static DwpJobs dwp_jobs_shard(mlggd_engine *e, float nf, int r, int pass, bool own_bias) {
    DwpJobs G = dwp_jobs(e, e->L - 1, 1, e->Yall[0], nf), J;
    memset(&J, 0, sizeof(J));
    int nj = 0, end = 0;
    {
        const bool with_bias_only = true;
        for (int j = 0; j < G.njobs; j++) {
            const int l = e->L - 1 - j;
            DwpArgs a = G.job[j];
            a.Yrow = e->Yall[l - 1];
            a.dEdX = e->dEdXall[l];
            a.B = e->world * e->Bp;
            const int k_wg = a.ntiles / a.n_wg;
            const int kf = r * e->shard_rows[l];
            if (e->dp_a2a) {  // Yall[l-1] holds the units of this rank's block only: [world x Bp][block width]
                a.ldA = e->shard_rows[l] * 64;
                a.k_base = pass == 0 ? kf * 64 : 0;
            }
            const int kl = kf + e->shard_rows[l] < k_wg ? kf + e->shard_rows[l] : k_wg;
            if (pass == 0 && kl > kf) {
                a.k_first = kf;
                a.ntiles = (kl - kf) * a.n_wg;
                a.do_bias = (kf == 0 && own_bias) ? 1 : 0;
            } else if (pass == 1 && kf != 0 && with_bias_only) {
                a.k_first = 0;
                a.ntiles = a.n_wg;
                a.wd_off = 1;
                a.do_bias = 1;
            } else {
                continue;
            }
            J.job[nj] = a;
            end += a.ntiles;
            J.tile_end[nj++] = end;
        }
    }
    J.njobs = nj;
    J.total = end;
    return J;
}
static DwpJobs dwp_jobs_global(mlggd_engine *e, float nf, int lhi, int llo) {
    DwpJobs J = dwp_jobs(e, lhi, llo, e->Yall[0], nf);
    for (int j = 0; j < J.njobs; j++) {
        const int l = lhi - j;
        J.job[j].Yrow = e->Yall[l - 1];
        J.job[j].dEdX = e->dEdXall[l];
        J.job[j].B = e->world * e->Bp;
    }
    return J;
}

static BiasJobs make_bias_jobs(mlggd_engine *e) {
    BiasJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    int first = 0, nj = 0;
    for (int l = e->L - 1; l >= 1; l--) {
        BiasJob &j = jobs.job[nj++];
        j.dEdX = e->dEdX[l];
        j.bias = e->bias[l];
        j.dbias = e->dbias[l];
        j.gb = e->gb[l];
        j.N = e->ls[l];
        j.Np = e->lsp[l];
        j.first = first;
        first += e->lsp[l];
    }
    jobs.njobs = nj;
    jobs.total = first;
    return jobs;
}

// ---- pieces of one step ----------------------------------------------------------------------------
// Where the per-dimension sum of |e|^beta of the ML loss (BP_GPU.cu:416) comes from:
enum ColsumSource {
    CS_LOCAL = 0,   // single device: every loss workgroup sums its own 32 columns of the minibatch
    CS_ALLREDUCE,   // data parallel: local sums (k_colsum), all-reduced over the communicator
    CS_GIVEN,       // e->colsum_tot already holds the global sum (one-GPU emulation of a world)
    CS_ACCUMULATE   // emulation, first pass: error + local sums only, added into e->colsum_tot; no gradient
};
// largest dynamic LDS the loss kernels are ever launched with (bunchsize 1152, the cap mlggd_create enforces)
constexpr size_t LOSS_LDS_MAX = (size_t)(32 * (1152 + 1) + 32) * sizeof(float);

// Output-layer loss of one minibatch: BP_GPU.cu:408-423.  sa / n_stage: input-staging blocks of the NEXT
// minibatch that ride along with the first loss launch (n_stage 0: none).
static int run_loss(mlggd_engine *e, const Bunch &bn, float nf, float inv_n, ColsumSource cs, bool first_acc,
                    const StageArgs &sa, int n_stage) {
    const int L = e->L, B = e->B, Bp = e->Bp, b_tiles = Bp / 32, ML = e->cfg.MLflag;
    ProfScope ps(e, KC_LOSS, 0);
    const size_t lds_grad = (size_t)(LOSS_DT * (Bp + 1) + LOSS_DT) * sizeof(float);  // k_loss_grad: LOSS_DT columns
    const int n_loss = (e->Dp / LOSS_DT) * b_tiles;  // 8 (d) x 32 (b) elements per loss workgroup, one per thread
    if (ML != 1 && e->loss_fuse) {
        if (cs == CS_ACCUMULATE) return MLGGD_OK;  // no minibatch statistic without the ML loss
        LossNormArgs la;
        la.slab = e->slab; la.S = e->S_out; la.bias = e->bias[L - 1]; la.targ = bn.targ;
        la.B = B; la.D = e->D; la.Dp = e->Dp; la.Bp = Bp; la.beta = e->cfg.shapefactor; la.inv_n = inv_n;
        la.outT = e->outT; la.eT = e->eT; la.dEdXt = e->dEdXt[L - 1]; la.dEdX = e->dEdX[L - 1];
        la.b_tiles = b_tiles; la.first = bn.first; la.toff = e->toff;
        hipLaunchKernelGGL(k_loss_norm, dim3(n_loss + n_stage), dim3(256), 0, e->stream, la, n_loss, sa);
        return launch_check("k_loss_norm");
    }
    if (ML == 1 && cs == CS_LOCAL && e->loss_fuse) {  // single device: the whole ML loss in one launch
        LossMlArgs la;
        la.slab = e->slab; la.S = e->S_out; la.bias = e->bias[L - 1]; la.targ = bn.targ;
        la.B = B; la.D = e->D; la.Dp = e->Dp; la.Bp = Bp; la.beta = e->cfg.shapefactor; la.nf = nf; la.inv_n = inv_n;
        la.outT = e->outT; la.eT = e->eT; la.scalefactor = e->scalefactor; la.dEdXt = e->dEdXt[L - 1]; la.dEdX = e->dEdX[L - 1];
        la.first = bn.first; la.toff = e->toff;
        const size_t lds_ml = (size_t)2 * LOSS_DT * (Bp + 1) * sizeof(float);
        CHK(ensure_lds(e, k_loss_ml, (size_t)2 * LOSS_DT * (1152 + 1) * sizeof(float)));
        hipLaunchKernelGGL(k_loss_ml, dim3(e->Dp / LOSS_DT + (n_stage + 3) / 4), dim3(1024), lds_ml, e->stream, la,
                           e->Dp / LOSS_DT, sa, n_stage);
        return launch_check("k_loss_ml");
    }
    LossErrArgs la;
    la.slab = e->slab; la.S = e->S_out; la.bias = e->bias[L - 1]; la.targ = bn.targ;
    la.B = B; la.D = e->D; la.Dp = e->Dp; la.Bp = Bp; la.beta = e->cfg.shapefactor; la.want_pow = ML == 1 ? 1 : 0;
    la.outT = e->outT; la.eT = e->eT; la.pT = e->pT;
    la.b_tiles = b_tiles; la.first = bn.first; la.toff = e->toff;
    hipLaunchKernelGGL(k_loss_err, dim3(n_loss + n_stage), dim3(256), 0, e->stream, la, n_loss, sa);
    CHK(launch_check("k_loss_err"));
    const float *colsum_in = nullptr;
    if (ML == 1 && cs == CS_GIVEN) {
        colsum_in = e->colsum_tot;
    } else if (ML == 1 && cs != CS_LOCAL) {
        hipLaunchKernelGGL(k_colsum, dim3(e->Dp / 32), dim3(256), 0, e->stream, e->pT, B, Bp, e->colsum);  // wavefront reductions
        CHK(launch_check("k_colsum"));
        if (cs == CS_ACCUMULATE) {
            hipLaunchKernelGGL(k_accum, dim3(1), dim3(256), 0, e->stream, e->colsum_tot,
                               first_acc ? (const float *)nullptr : (const float *)e->colsum_tot, (const float *)e->colsum,
                               (size_t)e->Dp);
            return launch_check("k_accum");
        }
        NCCLCHK(g_rccl.AllReduce(e->colsum, e->colsum, (size_t)e->Dp, 7, 0, e->stat_comm ? e->stat_comm : e->comm, e->stream));
        colsum_in = e->colsum;
    }
    if (cs == CS_ACCUMULATE) return MLGGD_OK;
    CHK(ensure_lds(e, k_loss_grad, LOSS_LDS_MAX));
    hipLaunchKernelGGL(k_loss_grad, dim3(n_loss), dim3(256), lds_grad, e->stream, e->eT, e->pT, colsum_in, B, e->D, e->Dp, Bp,
                       e->cfg.shapefactor, ML, nf, inv_n, e->scalefactor, e->dEdXt[L - 1], e->dEdX[L - 1], b_tiles);
    return launch_check("k_loss_grad");
}

// dEdX_{l-1} from dEdX_l and the OLD W_l (+ sigmoid derivative of layer l-1): BP_GPU.cu:402,430
static int run_dx(mlggd_engine *e, int l) {
    const int Kp = e->lsp[l - 1], b_tiles = e->Bp / 32;
    ProfScope ps(e, KC_DX, l);
    long long *st = stamps_for(e, KC_DX, l, (Kp / 32) * b_tiles);
    DxArgs xa = dx_args(e, l);
    if (dx64_used(e, l)) {
        xa.k_tiles = Kp / 64;
        xa.b_tiles = e->Bp / 64;
        xa.b_shift = log2_or_minus1(xa.b_tiles);
        const size_t lds = t64_lds_floats() * sizeof(float);
        CHK(ensure_lds(e, k_dx64, lds));
        launch_timed(e, k_dx64, dim3(xa.k_tiles * xa.b_tiles), dim3(256), lds, e->stream, xa);
        return launch_check("k_dx64");
    }
#define LAUNCH_DX(NW, PIPE)                                                                                  \
    {                                                                                                        \
        const size_t lds = dx_lds_floats<NW, PIPE>() * sizeof(float);                                        \
        CHK(ensure_lds(e, k_dx<NW, PIPE>, lds));                                                             \
        launch_timed(e, k_dx<NW, PIPE>, dim3((Kp / 32) * b_tiles), dim3(64 * NW), lds, e->stream, xa, st);      \
    }
    // PIPE 4 (default where one workgroup per CU is all there is: <= 256 workgroups): main loop software-pipelined
    // inside the wave, operands by LDS-DMA (136 KB of LDS); 1: the same pipeline through staging registers (71 KB: two
    // workgroups per CU for larger minibatches; MLGGD_DX_PIPE=1); 0: the round-1 loop (MLGGD_DX_PIPE=0, A/B)
    const bool dma = e->dx_pipe == 4 && (Kp / 32) * b_tiles <= 256;
    if (e->dx_nw == 8 && e->dx_pipe) LAUNCH_DX(8, 1)
    else if (e->dx_nw == 8) LAUNCH_DX(8, 0)
    else if (dma) LAUNCH_DX(4, 4)
    else if (e->dx_pipe) LAUNCH_DX(4, 1)
    else LAUNCH_DX(4, 0)
#undef LAUNCH_DX
    return launch_check("k_dx");
}

// the weight-gradient kernel of ONE layer (fused: with the update as its epilogue; else G_l, gb_l are written)
static int launch_dw_layer(mlggd_engine *e, int l, const float *in_rows, bool fused, float nf, hipStream_t st) {
    const int Kp = e->lsp[l - 1], Np = e->lsp[l];
    const long tiles128 = (long)((Kp + 127) / 128) * ((Np + 127) / 128);
    const bool big = e->dw_tile == 0 ? tiles128 >= 192 : e->dw_tile == 2;
    ProfScope ps(e, KC_DW, l, st);
    if (dwp_usable(e)) return launch_dwp(e, dwp_jobs(e, l, l, in_rows, nf), fused, st, l);
    if (big) return launch_dw<2>(e, l, in_rows, fused, nf, st);
    return launch_dw<1>(e, l, in_rows, fused, nf, st);
}

static int launch_accum(float *dst, const float *a, const float *b, size_t n, hipStream_t st) {
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_accum, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, dst, a, b, n);
    return launch_check("k_accum");
}

// One-GPU emulation of a world of ranks (mlggd_debug_fake_world).  The global minibatch is rows
// [sample0, sample0 + world*B) of the resident chunk and emulated rank r owns rows [sample0 + r*B, +B) of it:
// the partition of SURVEY 8e, with DIFFERENT rows on every rank.  Ranks 0 .. world-2 run here, one after the
// other, and leave behind what the collectives would have delivered: the ML statistic of ALL ranks in
// colsum_tot, their factors in slot r of Yall / dEdXall (gather modes), or their gradients summed into
// Gpre / gbpre (all-reduce mode).  The last rank then takes the real data-parallel path of run_step with
// device copies / adds in place of the RCCL calls.
static int fake_world_prepass(mlggd_engine *e, int sample0, float nf, float inv_n) {
    const int L = e->L, B = e->B, Bp = e->Bp, W = e->world;
    StageArgs none;
    memset(&none, 0, sizeof(none));
    if (e->cfg.MLflag == 1)
        for (int r = 0; r < W; r++) {
            const Bunch bn = bunch_at(e, sample0 + r * B);
            CHK(run_forward(e, bn, B, true));
            CHK(run_loss(e, bn, nf, inv_n, CS_ACCUMULATE, r == 0, none, 0));
        }
    for (int r = 0; r + 1 < W; r++) {
        const Bunch bn = bunch_at(e, sample0 + r * B);
        CHK(run_forward(e, bn, B, true));
        const float *in_rows = bunch_rows(e, bn);
        CHK(run_loss(e, bn, nf, inv_n, CS_GIVEN, false, none, 0));
        for (int l = L - 1; l >= 2; l--) CHK(run_dx(e, l));
        if (e->dp_mode >= 1) {
            auto put = [&](const float *src, float *dst, size_t count) -> int {
                HIPCHK(hipMemcpyAsync(dst + (size_t)r * count, src, count * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
                return MLGGD_OK;
            };
            if (e->dp_a2a) {  // keep source r's owner-blocked factors whole: every virtual owner picks its block later
                for (int l = 0; l < L - 1; l++) {
                    const size_t all = (size_t)e->world * Bp * e->shard_rows[l + 1] * 64;
                    HIPCHK(hipMemcpyAsync(e->Ysrc[l][r], l == 0 ? in_rows : e->Y[l], all * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
                }
            } else {
                CHK(put(in_rows, e->Yall[0], (size_t)Bp * e->lsp[0]));
                for (int l = 1; l < L - 1; l++) CHK(put(e->Y[l], e->Yall[l], (size_t)Bp * e->lsp[l]));
            }
            for (int l = 1; l < L; l++) CHK(put(e->dEdX[l], e->dEdXall[l], (size_t)Bp * e->lsp[l]));
        } else {
            for (int l = L - 1; l >= 1; l--) {
                CHK(launch_dw_layer(e, l, in_rows, false, nf, e->stream));
                CHK(launch_accum(e->Gpre[l], r == 0 ? nullptr : e->Gpre[l], e->G[l], (size_t)e->lsp[l - 1] * e->lsp[l], e->stream));
            }
            CHK(launch_accum(e->gbpre, r == 0 ? nullptr : e->gbpre, e->gb_all, e->gb_all_count, e->stream));
        }
    }
    return MLGGD_OK;
}

// One SGD step on `frames` (= bunchsize) resident frames: BP_GPU::train_bunch_single,
// BP_GPU.cu:308-440.
// next != nullptr: also stage that bunch's input for the following step (its blocks ride along
// with the loss kernel; see k_loss_norm).  prestaged: this bunch was staged that way.
// sample0: index of the minibatch's first row in the resident chunk (only the emulated world needs it).
static int run_step(mlggd_engine *e, int sample0, bool prestaged = false, const Bunch *next = nullptr) {
    const int L = e->L, B = e->B, Bp = e->Bp;
    e->stop_ev_next = nullptr;  // see run_forward
    e->stop_ev_attached = false;
    const bool dp = e->comm != nullptr || e->fake_world;  // a 1-rank communicator still takes the exchange path (tests)
    const bool gather = dp && e->dp_mode >= 1;
    const int n_global = B * e->world;
    const float nf = (float)n_global;
    const float inv_n = 1.0f / n_global;  // DevVecMulNum(..., 1.0f/n_frames, ...), BP_GPU.cu:409,423
    const int ML = e->cfg.MLflag;
    if (e->fake_world) {
        CHK(fake_world_prepass(e, sample0, nf, inv_n));
        sample0 += (e->world - 1) * B;  // the last emulated rank runs below
    }
    const Bunch bn = bunch_at(e, sample0);
    RoctxRange step_range("mlggd step");

    // With the ML loss the hidden activations are sent AFTER the loss kernels: the 257-float all-reduce of the
    // loss statistic uses the same communicator, and collectives of one communicator run one after the other
    // whatever stream they are on -- it must not queue behind megabytes of factors (the input rows, sent at
    // the start of the step, have long arrived by then).
    // fine: replicated update on few ranks -- send every factor as soon as it exists (the links are the
    // bottleneck there, so they should never idle) and start the upper layers' dW while dEdX_1 still travels
    const bool fine = gather && e->dp_mode == 1 && e->dp_fine != 0;
    {
        RoctxRange r("forward");
        CHK(run_forward(e, bn, B, true, prestaged,
                        gather ? (GATHER_INPUT | (ML != 1 ? (fine ? GATHER_HIDDEN_EACH : GATHER_HIDDEN) : 0)) : 0));
    }
    const float *in_rows = bunch_rows(e, bn);  // after run_forward: it may have switched in_bunch
    // input of the next step: Yt[0] is free from here on (forward_1 has been enqueued); frame-stream
    // rows go to the OTHER in_bunch buffer because this step's dW(1) still reads the current one
    StageArgs sa;
    memset(&sa, 0, sizeof(sa));
    int n_stage = 0;
    if (next) {
        sa = stage_args(e, *next, B, in_bunch_other(e));
        n_stage = stage_blocks(e);
    }
    {
        RoctxRange r("loss (+ staging of the next minibatch)");
        CHK(run_loss(e, bn, nf, inv_n, !dp ? CS_LOCAL : e->fake_world ? CS_GIVEN : CS_ALLREDUCE, false, sa, n_stage));
    }
    RoctxRange back_range("backward: dX, dW + update, exchange");
    if (gather && ML == 1 && L > 2) {
        CHK(gather_begin(e));
        for (int g = 1; g < L - 1; g++) CHK(exchange_y(e, g, e->Y[g], nullptr));
        CHK(gather_end(e));
    }
    const bool two = e->two_streams != 0;
    hipStream_t dws = two ? e->dw_stream : e->stream;
    // single GPU: every dW(l) only needs dEdX_l and Y_{l-1}, so one persistent launch walks the
    // tiles of all layers after the last dX (the data-parallel path keeps one launch per layer so
    // that the all-reduce of layer l overlaps the rest of the backward pass)
    const bool merged = (!dp && !two && e->dw_merge && dwp_usable(e)) || gather;
    int pending_hi = L - 1;  // gather mode: dEdX_l for l in [l .. pending_hi] are final and not yet sent
    bool mainline_done = false;  // the last factor group went out on the main stream: the dW launch needs no further event
    for (int l = L - 1; l >= 1; l--) {
        const int Kp = e->lsp[l - 1], Np = e->lsp[l];
        if (fine) {  // dEdX_l is final here (loss or dX_{l+1} has been enqueued): send it, layer l is then complete
            CHK(gather_begin_armed(e));  // the event rode on dX_{l+1}'s launch (armed below), else it is recorded here
            CHK(gather_one(e, e->dEdX[l], e->dEdXall[l], (size_t)Bp * e->lsp[l]));
            CHK(gather_end(e));
            HIPCHK(hipEventRecord(e->ev_layer[l], e->comm_stream));
        } else if (gather && l <= 2) {  // two groups: everything down to dEdX_2 beside dX_2, dEdX_1 at the end
            if (l == 1 && e->dp_mainline && !two) {
                // dEdX_1 is the one factor nothing can hide: the dW launch waits for it.  On the MAIN stream it costs
                // its own time; by way of the communication stream it cost two event hand-offs more (30 us between
                // dX_2 and k_dwp in the 1-rank rehearsal).  The earlier groups are awaited here, where they have
                // long arrived.
                HIPCHK(hipEventRecord(e->ev_gathered, e->comm_stream));
                HIPCHK(hipStreamWaitEvent(e->stream, e->ev_gathered, 0));
                if (!e->fake_world) NCCLCHK(g_rccl.GroupStart());
                for (int g = pending_hi; g >= l; g--)
                    CHK(gather_one(e, e->dEdX[g], e->dEdXall[g], (size_t)Bp * e->lsp[g], e->stream));
                if (!e->fake_world) NCCLCHK(g_rccl.GroupEnd());
                mainline_done = true;
            } else {
                CHK(gather_begin_armed(e));
                for (int g = pending_hi; g >= l; g--) CHK(gather_one(e, e->dEdX[g], e->dEdXall[g], (size_t)Bp * e->lsp[g]));
                CHK(gather_end(e));
            }
            pending_hi = l - 1;
        }
        // dX_l produces dEdX_{l-1}; if the next iteration sends it (or a group ending with it) on the communication
        // stream, the event that stream waits for rides on this launch
        if (l != 1 && gather && !two &&
            (fine || l - 1 == 2 || (l - 1 == 1 && !(e->dp_mainline))))
            gather_arm(e);
        if (l != 1) CHK(run_dx(e, l));
        if (two) {  // dw(l) after dX(l): dEdX_l is final and W_l has been read (old weights)
            HIPCHK(hipEventRecord(e->ev_dx[l], e->stream));
            HIPCHK(hipStreamWaitEvent(dws, e->ev_dx[l], 0));
        }
        if (merged) continue;  // all layers' dW + update run as one launch after the last dX
        if (dp && !gather && !two) arm_stop(e, e->ev_grad[l]);  // "G_l written" rides on the launch that writes it
        CHK(launch_dw_layer(e, l, in_rows, !dp, nf, dws));
        if (dp && !gather) {
            if (!take_stop(e)) HIPCHK(hipEventRecord(e->ev_grad[l], dws));
            HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_grad[l], 0));
            if (e->fake_world) {  // sum over the emulated ranks: what the all-reduce (or the union of the reduce-scatter blocks) delivers
                if (e->world > 1) CHK(launch_accum(e->G[l], e->Gpre[l], e->G[l], (size_t)Kp * Np, e->comm_stream));
            } else if (e->ar_shard) {
                // reduce-scatter over blocks of weight rows (shard_rows[l] tile rows of 64 per rank; G_l is allocated
                // with world x block rows, the rows past Kp stay zero): this rank receives the sum of ITS block, in place
                const size_t blk = (size_t)e->shard_rows[l] * 64 * Np;
                NCCLCHK(g_rccl.ReduceScatter(e->G[l], e->G[l] + (size_t)e->rank * blk, blk, e->comm, e->comm_stream));
            } else {
                NCCLCHK(g_rccl.AllReduce(e->G[l], e->G[l], (size_t)Kp * Np, 7, 0, e->comm, e->comm_stream));
            }
            HIPCHK(hipEventRecord(e->ev_red[l], e->comm_stream));
        }
    }
    if (gather && e->dp_mode == 2) {
        if (!mainline_done) {
            HIPCHK(hipEventRecord(e->ev_gathered, e->comm_stream));
            HIPCHK(hipStreamWaitEvent(dws, e->ev_gathered, 0));
        }
        const int units = e->world * Bp / 64;
        if (e->fake_world) {
            // all virtual ranks in turn on this GPU: their row blocks are disjoint, so the union is the whole
            // update; the bias update is taken from the LAST rank's bias-only jobs so that path is exercised
            const char *only = getenv("MLGGD_FAKE_ONLY_RANK");  // timing: run one rank's share only (tools/dp_sim.py)
            for (int r = 0; r < e->world; r++) {
                if (only && atoi(only) != r) continue;
                if (e->dp_a2a)  // what the all-to-all delivers to virtual owner r: its block of every source's Y_l
                    for (int l = 0; l < L - 1; l++) {
                        const size_t cnt = (size_t)Bp * e->shard_rows[l + 1] * 64;
                        for (int sr = 0; sr < e->world; sr++) {
                            const float *src = sr + 1 < e->world ? e->Ysrc[l][sr] : (l == 0 ? in_rows : e->Y[l]);
                            HIPCHK(hipMemcpyAsync(e->Yall[l] + (size_t)sr * cnt, src + (size_t)r * cnt, cnt * sizeof(float),
                                                  hipMemcpyDeviceToDevice, dws));
                        }
                    }
                if (only || r == e->world - 1) {  // the bias update comes from the LAST rank's bias-only tiles
                    DwpJobs Jb = dwp_jobs_shard(e, nf, r, 1, false);
                    if (Jb.total > 0) CHK(launch_dwp(e, Jb, true, dws, 1, units));
                }
                DwpJobs J = dwp_jobs_shard(e, nf, r, 0, only ? true : e->world == 1);
                ProfScope ps(e, KC_DW, 1, dws);
                if (J.total > 0) CHK(launch_dwp(e, J, true, dws, 1, units));
            }
        } else {
            {
                DwpJobs Jb = dwp_jobs_shard(e, nf, e->rank, 1, false);  // bias-only tiles: no W access, so they go first
                if (Jb.total > 0) CHK(launch_dwp(e, Jb, true, dws, 1, units));
                ProfScope ps(e, KC_DW, 1, dws);
                DwpJobs J = dwp_jobs_shard(e, nf, e->rank, 0, true);
                if (!two) arm_stop(e, e->ev_dw_done);  // the event the W gathers wait for rides on this launch
                if (J.total > 0) CHK(launch_dwp(e, J, true, dws, 1, units));
            }
            const bool dw_done_attached = take_stop(e);
            // W_1 is what the next forward pass waits for first: its all-gather goes on the MAIN stream, right behind
            // the update (no hand-off to the communication stream and back: 21 us between k_dwp and forward_1 in the
            // 1-rank rehearsal); the upper layers' blocks travel on the communication stream beside forward_1
            const bool w1_main = e->dp_mainline && !two;
            if (!dw_done_attached) HIPCHK(hipEventRecord(e->ev_dw_done, dws));
            HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_dw_done, 0));
            e->comm_after_dw = true;
            for (int l = 1; l < L; l++) {  // layer 1 first: the next forward pass needs it first
                const size_t count = (size_t)e->shard_rows[l] * 64 * e->lsp[l];
                hipStream_t st = (l == 1 && w1_main) ? e->stream : e->comm_stream;
                NCCLCHK(g_rccl.AllGather(e->W[l] + (size_t)e->rank * count, e->W[l], count, 7, e->comm, st));
                if (st == e->comm_stream) {
                    HIPCHK(hipEventRecord(e->ev_W[l], e->comm_stream));
                    e->ev_W_pending[l] = true;
                }
            }
        }
    } else if (fine) {
        // two launches: layers L-1..2 as soon as dEdX_2 has arrived (after dX_2 in stream order, which still reads
        // the old W_2), layer 1 when dEdX_1 has
        const int units = e->world * Bp / 64;
        if (L - 1 >= 2) {
            HIPCHK(hipStreamWaitEvent(dws, e->ev_layer[2], 0));
            ProfScope ps(e, KC_DW, L - 1, dws);
            CHK(launch_dwp(e, dwp_jobs_global(e, nf, L - 1, 2), true, dws, L - 1, units));
        }
        HIPCHK(hipStreamWaitEvent(dws, e->ev_layer[1], 0));
        ProfScope ps(e, KC_DW, 1, dws);
        CHK(launch_dwp(e, dwp_jobs_global(e, nf, 1, 1), true, dws, 1, units));
    } else if (gather) {
        if (!mainline_done) {
            HIPCHK(hipEventRecord(e->ev_gathered, e->comm_stream));
            HIPCHK(hipStreamWaitEvent(dws, e->ev_gathered, 0));
        }
        ProfScope ps(e, KC_DW, 1, dws);
        CHK(launch_dwp(e, dwp_jobs_global(e, nf, L - 1, 1), true, dws, 1, e->world * Bp / 64));
    } else if (merged) {
        ProfScope ps(e, KC_DW, 1, dws);
        CHK(launch_dwp(e, dwp_jobs(e, L - 1, 1, in_rows, nf), true, dws, 1));
    }
    if (dp && !gather) {
        // bias gradients were written by the dw kernels; ev_grad[1] is the last of them
        HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_grad[1], 0));
        if (e->fake_world) {
            if (e->world > 1) CHK(launch_accum(e->gb_all, e->gbpre, e->gb_all, e->gb_all_count, e->comm_stream));
        } else {
            NCCLCHK(g_rccl.AllReduce(e->gb_all, e->gb_all, e->gb_all_count, 7, 0, e->comm, e->comm_stream));
        }
        HIPCHK(hipEventRecord(e->ev_bias_red, e->comm_stream));
        for (int l = L - 1; l >= 1; l--) {
            HIPCHK(hipStreamWaitEvent(dws, e->ev_red[l], 0));
            ProfScope ps(e, KC_UPDATE, l, dws);
            const int Kp = e->lsp[l - 1], Np = e->lsp[l];
            // kernUpdatedelta + kernAccSum (DevFunc.cu:490-507,427-443) on the rows [row0, row0 + rows) of W_l: the
            // whole matrix (MLGGD_DP_AR_SHARD=0), or the block of weight rows this rank received the summed gradient
            // of -- delta is kept for that block only.  The emulated world has every block's sum in G_l and applies
            // the blocks of all its ranks one after the other (they are disjoint: the union is the whole update).
            const int r_lo = !e->ar_shard ? 0 : e->fake_world ? 0 : e->rank;
            const int r_hi = !e->ar_shard ? 0 : e->fake_world ? e->world - 1 : e->rank;
            const char *only = e->fake_world ? getenv("MLGGD_FAKE_ONLY_RANK") : nullptr;  // timing: one rank's share only (tools/dp_sim.py)
            for (int r = r_lo; r <= r_hi; r++) {
                if (only && e->ar_shard && atoi(only) != r) continue;
                const int row0 = e->ar_shard ? r * e->shard_rows[l] * 64 : 0;
                const int row1 = e->ar_shard ? (row0 + e->shard_rows[l] * 64 < Kp ? row0 + e->shard_rows[l] * 64 : Kp) : Kp;
                if (row1 <= row0) continue;  // a block that lies entirely in the pad rows
                const size_t off = (size_t)row0 * Np, n4 = (size_t)(row1 - row0) * Np / 4;
                const size_t blocks = (n4 + 255) / 256;
                hipLaunchKernelGGL(k_apply_update, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, dws,
                                   e->W[l] + off, e->dW[l] + off, e->G[l] + off, n4, nf, e->cfg.momentum, e->cfg.lrate,
                                   e->cfg.weightcost);
                CHK(launch_check("k_apply_update"));
            }
        }
        HIPCHK(hipStreamWaitEvent(dws, e->ev_bias_red, 0));
        BiasJobs jobs = make_bias_jobs(e);
        hipLaunchKernelGGL(k_bias_apply, dim3((jobs.total + 255) / 256), dim3(256), 0, dws, jobs, nf,
                           e->cfg.momentum, e->cfg.lrate);
        CHK(launch_check("k_bias_apply"));
        if (e->ar_shard && !e->fake_world) {
            // the updated W blocks travel back: all-gather in place, layer 1 first and -- like the sharded factor
            // mode -- on the MAIN stream (the next forward pass waits for it first; no hand-off to the communication
            // stream and back), the upper layers on the communication stream beside forward_1, each with its event
            const bool w1_main = e->dp_mainline && !two;
            HIPCHK(hipEventRecord(e->ev_dw_done, dws));
            HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_dw_done, 0));
            for (int l = 1; l < L; l++) {
                const size_t count = (size_t)e->shard_rows[l] * 64 * e->lsp[l];
                hipStream_t st = (l == 1 && w1_main) ? e->stream : e->comm_stream;
                NCCLCHK(g_rccl.AllGather(e->W[l] + (size_t)e->rank * count, e->W[l], count, 7, e->comm, st));
                if (st == e->comm_stream) {
                    HIPCHK(hipEventRecord(e->ev_W[l], e->comm_stream));
                    e->ev_W_pending[l] = true;
                }
            }
        }
    }
    if (two) {  // the next forward pass (and any host read-back on the main stream) waits for the updates
        HIPCHK(hipEventRecord(e->ev_upd, dws));
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_upd, 0));
    }
    e->step_counter++;
    return MLGGD_OK;
}

static int create_concurrent_stream(mlggd_engine *e, hipStream_t *out, const char *what);

// ------------------------------------------------------------------ C-ABI
extern "C" {

const char *mlggd_last_error(void) { return g_err; }

int mlggd_device_count(int *count) {
    if (!count) return fail(MLGGD_ERR_ARG, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return MLGGD_OK;
}

int mlggd_create(const mlggd_config *cfg, const float *const *weights, const float *const *bias, mlggd_handle *out) {
    if (!cfg || !weights || !bias || !out) return fail(MLGGD_ERR_ARG, "NULL argument");
    if (cfg->struct_size != (int32_t)sizeof(mlggd_config))
        return fail(MLGGD_ERR_ARG, "mlggd_config.struct_size %d != %d", cfg->struct_size, (int)sizeof(mlggd_config));
    if (cfg->numlayers < 2 || cfg->numlayers > MLGGD_MAXLAYER)
        return fail(MLGGD_ERR_ARG, "numlayers %d not in 2..%d", cfg->numlayers, MLGGD_MAXLAYER);
    if (cfg->bunchsize < 1) return fail(MLGGD_ERR_ARG, "bunchsize %d < 1", cfg->bunchsize);
    for (int i = 0; i < cfg->numlayers; i++)
        if (cfg->layersizes[i] < 1) return fail(MLGGD_ERR_ARG, "layersizes[%d] = %d", i, cfg->layersizes[i]);
    for (int i = 1; i < cfg->numlayers; i++) {
        // the kernels address a weight matrix with 32-bit byte offsets into one buffer resource
        const long long bytes = 4ll * ceil32(cfg->layersizes[i - 1]) * ceil32(cfg->layersizes[i]);
        if (bytes >= (1ll << 31))
            return fail(MLGGD_ERR_ARG, "layer %d: %d x %d weights exceed the 2 GiB a kernel can address", i,
                        cfg->layersizes[i - 1], cfg->layersizes[i]);
    }
    if (ceil32(cfg->bunchsize) > 1152)  // LOSS_LDS_MAX: 32 columns x (1152 + 1) frames of the loss kernels' LDS tile
        return fail(MLGGD_ERR_ARG, "bunchsize %d too large for the loss kernel's LDS tile (max 1152)", cfg->bunchsize);
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev < 1)
        return fail(MLGGD_ERR_DEVICE, "no HIP device available (%s)", de == hipSuccess ? "count 0" : hipGetErrorString(de));
    // BP_GPU.cu:17-21 "GPU Num %d Not In Range"
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(MLGGD_ERR_ARG, "GPU Num %d Not In Range %d-%d", cfg->device, 0, ndev - 1);

    mlggd_engine *e = new mlggd_engine();
    e->cfg = *cfg;
    e->L = cfg->numlayers;
    e->device = cfg->device;
    for (int i = 0; i < e->L; i++) {
        e->ls[i] = cfg->layersizes[i];
        e->lsp[i] = ceil32(cfg->layersizes[i]);
    }
    e->B = cfg->bunchsize;
    e->Bp = ceil32(e->B);
    e->K0 = e->ls[0];
    e->D = e->ls[e->L - 1];
    e->Dp = e->lsp[e->L - 1];
    if (const char *v = getenv("MLGGD_FWD_NW")) e->fwd_nw = atoi(v);
    if (const char *v = getenv("MLGGD_DX_NW")) e->dx_nw = atoi(v);
    if (const char *v = getenv("MLGGD_FWD_PIPE")) e->fwd_pipe = atoi(v);
    if (const char *v = getenv("MLGGD_DX_PIPE")) e->dx_pipe = atoi(v);
    if (const char *v = getenv("MLGGD_DW_TILE")) e->dw_tile = atoi(v);
    if (const char *v = getenv("MLGGD_TWO_STREAMS")) e->two_streams = atoi(v);
    if (const char *v = getenv("MLGGD_DW_PERSIST")) e->dw_persist = atoi(v);
    if (const char *v = getenv("MLGGD_DWP_PER_CU")) e->dwp_per_cu = atoi(v);
    if (const char *v = getenv("MLGGD_DW_MERGE")) e->dw_merge = atoi(v);
    if (const char *v = getenv("MLGGD_LOSS_FUSE")) e->loss_fuse = atoi(v);
    if (const char *v = getenv("MLGGD_TILE_MAP")) e->tile_map = atoi(v);
    if (const char *v = getenv("MLGGD_STAGE_AHEAD")) e->stage_ahead = atoi(v);
    if (const char *v = getenv("MLGGD_CV_DEVICE")) e->cv_device = atoi(v) ? 1 : 0;
    if (const char *v = getenv("MLGGD_TILE64")) e->tile64 = atoi(v);
    *out = e;  // so the caller can destroy on failure

    roctx_load();
    HIPCHK(hipSetDevice(e->device));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus > 0) e->n_cus = cus;
    }
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // (the second compute stream only exists when MLGGD_TWO_STREAMS asks for it: every HIP stream takes one of the
    // process's few hardware queues, and two streams of one engine that land on the SAME queue serialise -- round 3:
    // the first communicator engine created after other engines had come and gone ran its step in 251 instead of
    // 173 us because its communication stream shared a queue with its main stream)
    if (e->two_streams) HIPCHK(hipStreamCreateWithFlags(&e->dw_stream, hipStreamNonBlocking));
    for (int l = 0; l < e->L; l++) HIPCHK(hipEventCreateWithFlags(&e->ev_dx[l], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_upd, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&e->ev_t0));
    HIPCHK(hipEventCreate(&e->ev_t1));

    const int L = e->L, Bp = e->Bp;
    CHK(dev_alloc(e, &e->Yt[0], (size_t)e->lsp[0] * Bp));
    CHK(dev_alloc(e, &e->in_bunch_buf[0], (size_t)(Bp + 1) * e->lsp[0]));
    CHK(dev_alloc(e, &e->in_bunch_buf[1], (size_t)(Bp + 1) * e->lsp[0]));
    e->in_bunch = e->in_bunch_buf[0];
    for (int l = 1; l < L; l++) {
        const size_t wsz = (size_t)e->lsp[l - 1] * e->lsp[l];
        CHK(dev_alloc(e, &e->W[l], wsz));
        CHK(dev_alloc(e, &e->dW[l], wsz));
        CHK(dev_alloc(e, &e->bias[l], e->lsp[l]));
        CHK(dev_alloc(e, &e->dbias[l], e->lsp[l]));
        const size_t asz = (size_t)e->lsp[l] * Bp;
        if (l != L - 1) {
            CHK(dev_alloc(e, &e->Yt[l], asz));
            CHK(dev_alloc(e, &e->Y[l], asz));
        }
        CHK(dev_alloc(e, &e->dEdXt[l], asz));
        CHK(dev_alloc(e, &e->dEdX[l], asz));
    }
    // output-layer GEMM: split K across workgroups so the small N still fills the chip
    {
        const int tiles = (e->Dp / 32) * (Bp / 32);
        const int pairs = e->lsp[L - 2] / 2;
        // one round of workgroups on the 256 CUs when that still splits K at least four ways (36 tiles: 7 x 36 = 252
        // workgroups; 8 x 36 = 288 left 32 of them for a second round: +1 us per step), else enough to fill the chip
        int S = 256 / tiles >= 4 ? 256 / tiles : (256 + tiles - 1) / tiles;
        if (S > pairs / 16) S = pairs / 16;  // >= 4 k-pairs per wave
        if (S < 1) S = 1;
        if (S > 32) S = 32;
        if (const char *v = getenv("MLGGD_S_OUT")) S = atoi(v) < 1 ? 1 : atoi(v) > 32 ? 32 : atoi(v);  // A/B knob
        e->S_out = S;
    }
    CHK(dev_alloc(e, &e->slab, (size_t)e->S_out * e->Dp * Bp));
    CHK(dev_alloc(e, &e->outT, (size_t)e->Dp * Bp));
    CHK(dev_alloc(e, &e->eT, (size_t)e->Dp * Bp));
    CHK(dev_alloc(e, &e->pT, (size_t)e->Dp * Bp));
    CHK(dev_alloc(e, &e->colsum, e->Dp));
    CHK(dev_alloc(e, &e->scalefactor, e->Dp));

    const int rc = mlggd_set_weights(e, weights, bias);
    if (rc != MLGGD_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_destroy(mlggd_handle e) {
    if (!e) return MLGGD_OK;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->comm_stream) hipStreamSynchronize(e->comm_stream);
    if (e->dw_stream) hipStreamSynchronize(e->dw_stream);
    if (e->stat_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->stat_comm);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    for (void *p : e->allocs) hipFree(p);
    if (e->chunk_in) hipFree(e->chunk_in);
    if (e->chunk_targ) hipFree(e->chunk_targ);
    if (e->chunk_out) hipFree(e->chunk_out);
    if (e->cv_partial) hipFree(e->cv_partial);
    for (auto &t : e->dwp_tables) hipFree(t.dev);
    if (e->copy_stream) hipStreamSynchronize(e->copy_stream);
    for (auto &r : e->raw) {
        if (r.feat) hipFree(r.feat);
        if (r.targ) hipFree(r.targ);
        if (r.first) hipFree(r.first);
        if (r.last_use) hipEventDestroy(r.last_use);
    }
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    for (hipEvent_t ev : e->prof_ev) hipEventDestroy(ev);
    for (int l = 0; l < MLGGD_MAXLAYER; l++) {
        if (e->ev_grad[l]) hipEventDestroy(e->ev_grad[l]);
        if (e->ev_red[l]) hipEventDestroy(e->ev_red[l]);
    }
    for (int l = 0; l < MLGGD_MAXLAYER; l++) {
        if (e->ev_W[l]) hipEventDestroy(e->ev_W[l]);
        if (e->ev_layer[l]) hipEventDestroy(e->ev_layer[l]);
    }
    if (e->ev_dw_done) hipEventDestroy(e->ev_dw_done);
    if (e->ev_ready) hipEventDestroy(e->ev_ready);
    if (e->ev_gathered) hipEventDestroy(e->ev_gathered);
    if (e->ev_bias) hipEventDestroy(e->ev_bias);
    if (e->ev_bias_red) hipEventDestroy(e->ev_bias_red);
    if (e->ev_t0) hipEventDestroy(e->ev_t0);
    if (e->ev_t1) hipEventDestroy(e->ev_t1);
    if (e->comm_stream) hipStreamDestroy(e->comm_stream);
    for (int l = 0; l < MLGGD_MAXLAYER; l++)
        if (e->ev_dx[l]) hipEventDestroy(e->ev_dx[l]);
    if (e->ev_upd) hipEventDestroy(e->ev_upd);
    if (e->dw_stream) hipStreamDestroy(e->dw_stream);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return MLGGD_OK;
}

int mlggd_set_weights(mlggd_handle e, const float *const *weights, const float *const *bias) {
    if (!e || !weights || !bias) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    CHK(wait_weight_gathers(e, 1, e->L - 1));
    for (int l = 1; l < e->L; l++) {
        if (!weights[l] || !bias[l]) return fail(MLGGD_ERR_ARG, "weights[%d] or bias[%d] is NULL", l, l);
        CHK(upload_padded(e->W[l], e->lsp[l], weights[l], e->ls[l - 1], e->ls[l], e->stream));  // BP_GPU.cu:106
        HIPCHK(hipMemcpyAsync(e->bias[l], bias[l], (size_t)e->ls[l] * 4, hipMemcpyHostToDevice, e->stream));  // :107
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_get_weights(mlggd_handle e, float *const *weights, float *const *bias) {
    if (!e || !weights || !bias) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    CHK(wait_weight_gathers(e, 1, e->L - 1));
    for (int l = 1; l < e->L; l++) {
        if (!weights[l] || !bias[l]) return fail(MLGGD_ERR_ARG, "weights[%d] or bias[%d] is NULL", l, l);
        CHK(download_padded(weights[l], e->W[l], e->lsp[l], e->ls[l - 1], e->ls[l], e->stream));  // BP_GPU.cu:522
        HIPCHK(hipMemcpyAsync(bias[l], e->bias[l], (size_t)e->ls[l] * 4, hipMemcpyDeviceToHost, e->stream));  // :523
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_get_scalefactor(mlggd_handle e, float *alpha) {
    if (!e || !alpha) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(alpha, e->scalefactor, (size_t)e->D * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_set_scalefactor(mlggd_handle e, const float *alpha) {
    if (!e || !alpha) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->scalefactor, alpha, (size_t)e->D * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_set_cv_device_reduce(mlggd_handle e, int on) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    e->cv_device = on ? 1 : 0;
    return MLGGD_OK;
}

int mlggd_set_lrate(mlggd_handle e, float lrate) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    e->cfg.lrate = lrate;
    return MLGGD_OK;
}

static int ensure_chunk(mlggd_engine *e, int n_frames) {
    const size_t need = (size_t)n_frames + e->Bp + 32;  // slack rows: padded tiles may read past the last frame
    if (need <= e->chunk_cap) return MLGGD_OK;
    if (e->chunk_in) hipFree(e->chunk_in);
    if (e->chunk_targ) hipFree(e->chunk_targ);
    e->chunk_in = e->chunk_targ = nullptr;
    e->chunk_cap = 0;
    HIPCHK(hipMalloc((void **)&e->chunk_in, need * e->K0 * sizeof(float)));
    HIPCHK(hipMalloc((void **)&e->chunk_targ, need * e->D * sizeof(float)));
    HIPCHK(hipMemsetAsync(e->chunk_in, 0, need * e->K0 * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(e->chunk_targ, 0, need * e->D * sizeof(float), e->stream));
    e->chunk_cap = need;
    return MLGGD_OK;
}

static int check_frames(mlggd_engine *e, int n_frames) {
    if (n_frames < 0) return fail(MLGGD_ERR_ARG, "n_frames %d < 0", n_frames);
    const int cap = e->cfg.max_cache_frames > 0 ? e->cfg.max_cache_frames : MLGGD_MAXCACHEFRAME;
    if (n_frames > cap) return fail(MLGGD_ERR_ARG, "n_frames %d exceeds the chunk capacity %d", n_frames, cap);
    return MLGGD_OK;
}

int mlggd_load_chunk(mlggd_handle e, int n_frames, const float *in, const float *targ) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    CHK(check_frames(e, n_frames));
    if (n_frames > 0 && !in) return fail(MLGGD_ERR_ARG, "in is NULL");
    HIPCHK(hipSetDevice(e->device));
    CHK(ensure_chunk(e, n_frames));
    // todev_vf_vf("in"/"targ"), BP_GPU.cu:163-164
    if (n_frames > 0) {
        HIPCHK(hipMemcpyAsync(e->chunk_in, in, (size_t)n_frames * e->K0 * 4, hipMemcpyHostToDevice, e->stream));
        if (targ)
            HIPCHK(hipMemcpyAsync(e->chunk_targ, targ, (size_t)n_frames * e->D * 4, hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    e->chunk_frames = n_frames;
    e->indexed = false;
    return MLGGD_OK;
}

// the chunk as raw frame streams + the first frame of every sample row.
int mlggd_load_frames(mlggd_handle e, int n_frames, int fea_context, const float *feat, const float *targ,
                      int n_samples, const int32_t *first_frame, int targ_offset) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (fea_context < 1 || e->K0 % fea_context != 0)
        return fail(MLGGD_ERR_ARG, "fea_context %d does not divide layersizes[0] = %d", fea_context, e->K0);
    const int fdim = e->K0 / fea_context;
    if (n_frames < 0 || n_samples < 0) return fail(MLGGD_ERR_ARG, "negative size");
    CHK(check_frames(e, n_samples));
    if (n_samples > 0 && (!feat || !first_frame)) return fail(MLGGD_ERR_ARG, "feat/first_frame is NULL");
    if (targ_offset < 0 || targ_offset >= fea_context)
        return fail(MLGGD_ERR_ARG, "targ_offset %d not in [0, fea_context)", targ_offset);
    for (int s = 0; s < n_samples; s++)
        if (first_frame[s] < 0 || first_frame[s] + fea_context > n_frames)
            return fail(MLGGD_ERR_ARG, "sample %d: window [%d,%d) outside the %d uploaded frames", s, first_frame[s],
                        first_frame[s] + fea_context, n_frames);
    HIPCHK(hipSetDevice(e->device));
    if (!e->copy_stream) {
        CHK(create_concurrent_stream(e, &e->copy_stream, "upload"));  // must not queue behind the chunk's kernels
        for (auto &r : e->raw) HIPCHK(hipEventCreateWithFlags(&r.last_use, hipEventDisableTiming));
    }
    // the set that is NOT current: its last reader (two chunks ago) has finished or is about to; wait for it
    // on the host before touching the allocation, then copy beside whatever the main stream is still running
    mlggd_engine::RawSet &r = e->raw[e->raw_cur ^ 1];
    HIPCHK(hipEventSynchronize(r.last_use));
    const size_t need = (size_t)n_frames + fea_context + 8;
    if (need > r.raw_cap) {
        if (r.feat) hipFree(r.feat);
        if (r.targ) hipFree(r.targ);
        r.feat = r.targ = nullptr;
        r.raw_cap = 0;
        HIPCHK(hipMalloc((void **)&r.feat, need * fdim * sizeof(float)));
        HIPCHK(hipMalloc((void **)&r.targ, need * e->D * sizeof(float)));
        HIPCHK(hipMemsetAsync(r.feat, 0, need * fdim * sizeof(float), e->copy_stream));
        HIPCHK(hipMemsetAsync(r.targ, 0, need * e->D * sizeof(float), e->copy_stream));
        r.raw_cap = need;
    }
    const size_t need_s = (size_t)n_samples + e->Bp + 32;
    if (need_s > r.first_cap) {
        if (r.first) hipFree(r.first);
        r.first = nullptr;
        r.first_cap = 0;
        HIPCHK(hipMalloc((void **)&r.first, need_s * sizeof(int)));
        HIPCHK(hipMemsetAsync(r.first, 0, need_s * sizeof(int), e->copy_stream));
        r.first_cap = need_s;
    }
    if (n_frames > 0) {
        HIPCHK(hipMemcpyAsync(r.feat, feat, (size_t)n_frames * fdim * 4, hipMemcpyHostToDevice, e->copy_stream));
        if (targ) HIPCHK(hipMemcpyAsync(r.targ, targ, (size_t)n_frames * e->D * 4, hipMemcpyHostToDevice, e->copy_stream));
    }
    if (n_samples > 0)
        HIPCHK(hipMemcpyAsync(r.first, first_frame, (size_t)n_samples * sizeof(int), hipMemcpyHostToDevice, e->copy_stream));
    // the caller's buffers are free when this returns; kernels enqueued from now on see the data (the host has
    // observed the copies complete)
    HIPCHK(hipStreamSynchronize(e->copy_stream));
    e->raw_cur ^= 1;
    e->raw_feat = r.feat;
    e->raw_targ = r.targ;
    e->first_frame = r.first;
    e->raw_cap = r.raw_cap;
    e->first_cap = r.first_cap;
    e->chunk_frames = n_samples;
    e->raw_frames = n_frames;
    e->indexed = true;
    e->fdim = fdim;
    e->toff = targ_offset;
    return MLGGD_OK;
}

int mlggd_train_frames(mlggd_handle e, int n_frames, int fea_context, const float *feat, const float *targ,
                       int n_samples, const int32_t *first_frame, int targ_offset, int *bunches_trained) {
    if (n_samples > 0 && !targ) return fail(MLGGD_ERR_ARG, "targ is NULL");
    CHK(mlggd_load_frames(e, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset));
    CHK(mlggd_train_resident(e, 0, n_samples, bunches_trained));
    return mlggd_sync(e);
}

// The same without the final wait: returns once the chunk is on the device (the caller's buffers are free) and
// its steps are enqueued.  The next call uploads the next chunk into the other device buffer set while these
// steps still run; mlggd_sync / mlggd_get_weights / any CV call waits for them.
int mlggd_train_frames_async(mlggd_handle e, int n_frames, int fea_context, const float *feat, const float *targ,
                             int n_samples, const int32_t *first_frame, int targ_offset, int *bunches_trained) {
    if (n_samples > 0 && !targ) return fail(MLGGD_ERR_ARG, "targ is NULL");
    CHK(mlggd_load_frames(e, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset));
    return mlggd_train_resident(e, 0, n_samples, bunches_trained);
}

// Pinned host memory for the caller's chunk buffers (faster, truly asynchronous H2D).
int mlggd_alloc_pinned(size_t bytes, void **out) {
    if (!out) return fail(MLGGD_ERR_ARG, "out is NULL");
    HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return MLGGD_OK;
}
int mlggd_alloc_pinned_on(int device, size_t bytes, void **out) {
    if (!out) return fail(MLGGD_ERR_ARG, "out is NULL");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)  // the same message mlggd_create gives (BP_GPU.cu:17-21)
        return fail(MLGGD_ERR_ARG, "GPU Num %d Not In Range %d-%d", device, 0, ndev - 1);
    // the calling thread may have no current device yet (a host IO thread): pin through `device`'s context, then
    // put the thread's current device back so the call has no side effect on it
    int prev = -1;
    const bool had_prev = hipGetDevice(&prev) == hipSuccess;
    HIPCHK(hipSetDevice(device));
    const hipError_t err = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (had_prev && prev != device) hipSetDevice(prev);
    if (err != hipSuccess) return fail(MLGGD_ERR_DEVICE, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    return MLGGD_OK;
}
int mlggd_free_pinned(void *p) {
    if (p) HIPCHK(hipHostFree(p));
    return MLGGD_OK;
}

int mlggd_train_resident(mlggd_handle e, int first_frame, int n_frames, int *bunches_trained) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (first_frame < 0 || n_frames < 0 || first_frame + n_frames > e->chunk_frames)
        return fail(MLGGD_ERR_ARG, "frames [%d,%d) outside the resident chunk of %d frames", first_frame,
                    first_frame + n_frames, e->chunk_frames);
    HIPCHK(hipSetDevice(e->device));
    int trained = 0;
    HIPCHK(hipEventRecord(e->ev_t0, e->stream));
    // bunch loop of BP_GPU::train, BP_GPU.cu:170-184: full bunches only
    bool prestaged = false;
    // an emulated world consumes one GLOBAL minibatch of world*B rows per step (and stages nothing ahead)
    const int per_step = e->fake_world ? e->B * e->world : e->B;
    for (int i = 0; i + per_step <= n_frames; i += per_step) {
        const bool has_next = e->stage_ahead && !e->fake_world && i + 2 * e->B <= n_frames;
        Bunch next;
        if (has_next) next = bunch_at(e, first_frame + i + e->B);
        CHK(run_step(e, first_frame + i, prestaged, has_next ? &next : nullptr));
        prestaged = has_next;
        trained++;
    }
    HIPCHK(hipEventRecord(e->ev_t1, e->stream));
    if (e->indexed && e->copy_stream)  // the next-but-one upload reuses this buffer set: it waits for this point
        HIPCHK(hipEventRecord(e->raw[e->raw_cur].last_use, e->stream));
    e->last_steps = trained;
    e->timing_valid = true;
    if (bunches_trained) *bunches_trained = trained;
    return MLGGD_OK;
}

int mlggd_sync(mlggd_handle e) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->dw_stream) HIPCHK(hipStreamSynchronize(e->dw_stream));
    if (e->comm_stream) HIPCHK(hipStreamSynchronize(e->comm_stream));
    return MLGGD_OK;
}

int mlggd_train_chunk(mlggd_handle e, int n_frames, const float *in, const float *targ, int *bunches_trained) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (n_frames > 0 && (!in || !targ)) return fail(MLGGD_ERR_ARG, "in/targ is NULL");
    CHK(mlggd_load_chunk(e, n_frames, in, targ));
    CHK(mlggd_train_resident(e, 0, n_frames, bunches_trained));
    return mlggd_sync(e);  // BP_GPU.cu:439: train() returns after the last step completed
}

int mlggd_last_train_ms(mlggd_handle e, float *ms, int *steps) {
    if (!e || !ms) return fail(MLGGD_ERR_ARG, "NULL argument");
    if (!e->timing_valid) return fail(MLGGD_ERR_STATE, "no mlggd_train_resident call to time");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->ev_t1));
    HIPCHK(hipEventElapsedTime(ms, e->ev_t0, e->ev_t1));
    if (steps) *steps = e->last_steps;
    return MLGGD_OK;
}

// Forward over a chunk whose inputs are resident: outputs into chunk_out[n][D].
static int forward_resident(mlggd_engine *e, int n_frames) {
    if ((size_t)n_frames > e->out_cap) {
        if (e->chunk_out) hipFree(e->chunk_out);
        e->chunk_out = nullptr;
        e->out_cap = 0;
        HIPCHK(hipMalloc((void **)&e->chunk_out, ((size_t)n_frames + 32) * e->D * sizeof(float)));
        e->out_cap = n_frames;
    }
    const int b_tiles = e->Bp / 32;
    // bunch loop of CrossValid*, BP_GPU.cu:202-216: INCLUDES the trailing partial bunch
    for (int i = 0; i < n_frames; i += e->B) {
        const int fb = (e->B > n_frames - i) ? (n_frames - i) : e->B;
        CHK(run_forward(e, bunch_at(e, i), fb, false));
        hipLaunchKernelGGL(k_out_rowmajor, dim3((e->Dp / 32) * b_tiles), dim3(256), 0, e->stream, e->slab, e->S_out,
                           e->bias[e->L - 1], fb, e->D, e->Dp, e->Bp, e->chunk_out + (size_t)i * e->D, b_tiles);
        CHK(launch_check("k_out_rowmajor"));
    }
    return MLGGD_OK;
}

int mlggd_forward(mlggd_handle e, int n_frames, const float *in, float *out) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (n_frames > 0 && (!in || !out)) return fail(MLGGD_ERR_ARG, "in/out is NULL");
    CHK(mlggd_load_chunk(e, n_frames, in, nullptr));
    if (n_frames == 0) return MLGGD_OK;
    CHK(forward_resident(e, n_frames));
    HIPCHK(hipMemcpyAsync(out, e->chunk_out, (size_t)n_frames * e->D * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

// BP_GPU::Gamma, BP_GPU.cu:593-640: 10th-order polynomial on (2,3] in double powers, each
// pair of terms rounded into a float accumulator; recurrences outside that interval.
float mlggd_gamma(float x) {
    static const float coef[11] = {0.0000677106, -0.0003442342, 0.0015397681, -0.0024467480, 0.0109736958,
                                   -0.0002109075, 0.0742379071, 0.0815782188,  0.4118402518,  0.4227843370,
                                   1.0000000000};
    if (x > 2 && x <= 3) {
        const double t = x - 2.0;
        float acc = 0;
        for (int i = 0; i < 8; i += 2) acc = acc + coef[i] * pow(t, 10.0 - i) + coef[i + 1] * pow(t, 9.0 - i);
        acc = acc + coef[8] * pow(t, 2.0) + coef[9] * t + coef[10];
        return acc;
    }
    if (x > 0 && x <= 1) return mlggd_gamma(x + 2) / (x * (x + 1));
    if (x > 1 && x <= 2) return mlggd_gamma(x + 1) / x;
    if (x > 3) {
        int i = 1;
        float prod = 1;
        while (!((x - i) > 2 && (x - i) <= 3)) {
            prod = (x - i) * prod;
            i++;
        }
        prod = prod * (x - i);
        return prod * mlggd_gamma(x - i);
    }
    return 0;
}

// Host accumulation exactly as CrossValid / CrossValiddB / CrossValid2 do it
// (BP_GPU.cu:207-213, 240-250, 271-301): fp32 scalars, frame-major order.  The chunk must be
// resident (expanded or indexed); target of sample i is trow(i).
static int cv_accumulate(mlggd_engine *e, int n_frames, const std::function<const float *(int)> &trow, float *sqerr,
                         float *abserr, float *loglik) {
    const int D = e->D;
    std::vector<float> out((size_t)n_frames * D);
    if (n_frames > 0) {
        CHK(forward_resident(e, n_frames));
        HIPCHK(hipMemcpyAsync(out.data(), e->chunk_out, (size_t)n_frames * D * 4, hipMemcpyDeviceToHost, e->stream));
    }
    std::vector<float> scalefac(D);
    if (loglik)
        HIPCHK(hipMemcpyAsync(scalefac.data(), e->scalefactor, (size_t)D * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (sqerr) {
        float s = 0.0f;
        for (int i = 0; i < n_frames; i++) {
            const float *t = trow(i), *o = &out[(size_t)i * D];
            for (int d = 0; d < D; d++) s = s + (o[d] - t[d]) * (o[d] - t[d]);
        }
        *sqerr = s;
    }
    if (abserr) {
        float s = 0.0f;
        for (int i = 0; i < n_frames; i++) {
            const float *t = trow(i), *o = &out[(size_t)i * D];
            for (int d = 0; d < D; d++) s = s + fabsf(o[d] - t[d]);
        }
        *abserr = s / D;
    }
    if (loglik) {
        const float beta = e->cfg.shapefactor;
        float density1, density2 = 0, density3 = 0;
        density1 = n_frames * D * logf(beta / (2 * mlggd_gamma((float)(1.0 / beta))));
        for (int u = 0; u < D; u++) density2 += logf(scalefac[u]);
        density2 = density2 * n_frames;
        for (int i = 0; i < n_frames; i++) {
            const float *t = trow(i), *o = &out[(size_t)i * D];
            for (int d = 0; d < D; d++) {
                const float err = t[d] - o[d];
                density3 += powf(fabsf(err) / scalefac[d], beta);
            }
        }
        *loglik = density1 - density2 - density3;
    }
    return MLGGD_OK;
}

// The same three numbers with the sums formed on the device (mlggd_set_cv_device_reduce): one
// forward pass, k_cv_reduce per bunch, 3 doubles per 32x32 tile copied back -- no n x D transfer and no host
// loop.  The chunk (inputs AND targets) must be resident.
static int cv_device_reduce(mlggd_engine *e, int n_frames, float *sqerr, float *abserr, float *loglik) {
    const int D = e->D, b_tiles = e->Bp / 32, tiles = (e->Dp / 32) * b_tiles;
    const int nb = (n_frames + e->B - 1) / e->B;
    const size_t need = (size_t)(nb > 0 ? nb : 1) * tiles * 3;
    if (need > e->cv_partial_cap) {
        if (e->cv_partial) hipFree(e->cv_partial);
        e->cv_partial = nullptr;
        e->cv_partial_cap = 0;
        HIPCHK(hipMalloc((void **)&e->cv_partial, need * sizeof(double)));
        e->cv_partial_cap = need;
    }
    // bunch loop of CrossValid*, BP_GPU.cu:202-216: INCLUDES the trailing partial bunch
    for (int i = 0, k = 0; i < n_frames; i += e->B, k++) {
        const int fb = (e->B > n_frames - i) ? (n_frames - i) : e->B;
        const Bunch bn = bunch_at(e, i);
        CHK(run_forward(e, bn, fb, false));
        CvArgs a;
        a.slab = e->slab; a.S = e->S_out; a.bias = e->bias[e->L - 1]; a.targ = bn.targ;
        a.B = fb; a.D = D; a.Dp = e->Dp; a.Bp = e->Bp; a.beta = e->cfg.shapefactor;
        a.alpha = loglik ? e->scalefactor : nullptr;
        a.first = bn.first; a.toff = e->toff; a.b_tiles = b_tiles;
        a.partial = e->cv_partial + (size_t)k * tiles * 3;
        hipLaunchKernelGGL(k_cv_reduce, dim3(tiles), dim3(256), 0, e->stream, a);
        CHK(launch_check("k_cv_reduce"));
    }
    std::vector<double> part((size_t)nb * tiles * 3);
    std::vector<float> scalefac(D);
    if (nb > 0)
        HIPCHK(hipMemcpyAsync(part.data(), e->cv_partial, part.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    if (loglik) HIPCHK(hipMemcpyAsync(scalefac.data(), e->scalefactor, (size_t)D * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    double s[3] = {0, 0, 0};
    for (size_t i = 0; i < part.size(); i += 3)
        for (int j = 0; j < 3; j++) s[j] += part[i + j];
    if (sqerr) *sqerr = (float)s[0];
    if (abserr) *abserr = (float)(s[1] / D);
    if (loglik) {  // density1 - density2 - density3, BP_GPU.cu:271-301
        const float beta = e->cfg.shapefactor;
        const double density1 = (double)n_frames * D * logf(beta / (2 * mlggd_gamma((float)(1.0 / beta))));
        double density2 = 0;
        for (int u = 0; u < D; u++) density2 += logf(scalefac[u]);
        *loglik = (float)(density1 - density2 * n_frames - s[2]);
    }
    return MLGGD_OK;
}

static int cv_metrics(mlggd_engine *e, int n_frames, const float *in, const float *targ, float *sqerr, float *abserr,
                      float *loglik) {
    if (n_frames > 0 && (!in || !targ)) return fail(MLGGD_ERR_ARG, "in/targ is NULL");
    if (e->cv_device) {
        CHK(mlggd_load_chunk(e, n_frames, in, targ));
        return cv_device_reduce(e, n_frames, sqerr, abserr, loglik);
    }
    CHK(mlggd_load_chunk(e, n_frames, in, nullptr));
    const int D = e->D;
    return cv_accumulate(e, n_frames, [&](int i) { return targ + (size_t)i * D; }, sqerr, abserr, loglik);
}

int mlggd_cv_sqerr(mlggd_handle e, int n, const float *in, const float *targ, float *out) {
    if (!e || !out) return fail(MLGGD_ERR_ARG, "NULL argument");
    return cv_metrics(e, n, in, targ, out, nullptr, nullptr);
}
int mlggd_cv_abserr(mlggd_handle e, int n, const float *in, const float *targ, float *out) {
    if (!e || !out) return fail(MLGGD_ERR_ARG, "NULL argument");
    return cv_metrics(e, n, in, targ, nullptr, out, nullptr);
}
int mlggd_cv_loglik(mlggd_handle e, int n, const float *in, const float *targ, float *out) {
    if (!e || !out) return fail(MLGGD_ERR_ARG, "NULL argument");
    return cv_metrics(e, n, in, targ, nullptr, nullptr, out);
}
int mlggd_cv_all(mlggd_handle e, int n, const float *in, const float *targ, float *sqerr, float *abserr,
                 float *loglik) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    return cv_metrics(e, n, in, targ, sqerr, abserr, (e->cfg.MLflag == 1) ? loglik : nullptr);
}

int mlggd_cv_all_frames(mlggd_handle e, int n_frames, int fea_context, const float *feat, const float *targ,
                        int n_samples, const int32_t *first_frame, int targ_offset, float *sqerr, float *abserr,
                        float *loglik) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (n_samples > 0 && !targ) return fail(MLGGD_ERR_ARG, "targ is NULL");
    if (e->cv_device) {
        CHK(mlggd_load_frames(e, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset));
        return cv_device_reduce(e, n_samples, sqerr, abserr, (e->cfg.MLflag == 1) ? loglik : nullptr);
    }
    CHK(mlggd_load_frames(e, n_frames, fea_context, feat, nullptr, n_samples, first_frame, targ_offset));
    const int D = e->D;
    return cv_accumulate(
        e, n_samples, [&](int i) { return targ + (size_t)(first_frame[i] + targ_offset) * D; }, sqerr, abserr,
        (e->cfg.MLflag == 1) ? loglik : nullptr);
}

int mlggd_forward_frames(mlggd_handle e, int n_frames, int fea_context, const float *feat, int n_samples,
                         const int32_t *first_frame, float *out) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (n_samples > 0 && !out) return fail(MLGGD_ERR_ARG, "out is NULL");
    CHK(mlggd_load_frames(e, n_frames, fea_context, feat, nullptr, n_samples, first_frame, 0));
    if (n_samples == 0) return MLGGD_OK;
    CHK(forward_resident(e, n_samples));
    HIPCHK(hipMemcpyAsync(out, e->chunk_out, (size_t)n_samples * e->D * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}

int mlggd_debug_tensor(mlggd_handle e, const char *name, int layer, float *dst, size_t count) {
    if (!e || !name || !dst) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    const int L = e->L, B = e->B, Bp = e->Bp;
    const std::string nm(name);
    auto need = [&](size_t n) -> int {
        return count < n ? fail(MLGGD_ERR_ARG, "dst holds %zu floats, %s needs %zu", count, name, n) : MLGGD_OK;
    };
    if (nm == "scalefactor") {
        CHK(need(e->D));
        return mlggd_get_scalefactor(e, dst);
    }
    if (nm == "out") {  // outT [Dp][Bp] -> [B][D]
        CHK(need((size_t)B * e->D));
        std::vector<float> t((size_t)e->Dp * Bp);
        HIPCHK(hipMemcpyAsync(t.data(), e->outT, t.size() * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (int b = 0; b < B; b++)
            for (int d = 0; d < e->D; d++) dst[(size_t)b * e->D + d] = t[(size_t)d * Bp + b];
        return MLGGD_OK;
    }
    if (layer < 1 || layer >= L) return fail(MLGGD_ERR_ARG, "layer %d not in 1..%d", layer, L - 1);
    const int N = e->ls[layer], Np = e->lsp[layer], K = e->ls[layer - 1];
    if (nm == "y" || nm == "dedx" || nm == "yt" || nm == "dedxt") {
        const bool tr = (nm == "yt" || nm == "dedxt");
        const float *src = (nm == "y") ? e->Y[layer] : (nm == "dedx") ? e->dEdX[layer]
                           : (nm == "yt") ? e->Yt[layer] : e->dEdXt[layer];
        if (!src) return fail(MLGGD_ERR_ARG, "%s not kept for layer %d", name, layer);
        CHK(need((size_t)B * N));
        std::vector<float> t((size_t)Np * Bp);
        HIPCHK(hipMemcpyAsync(t.data(), src, t.size() * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        const int yb = (nm == "y" && e->dp_a2a && layer < L - 1) ? e->shard_rows[layer + 1] * 64 : 0;
        if (yb) {  // owner-blocked layout (all-to-all form): [block][Bp][yb]
            std::vector<float> tb((size_t)e->world * Bp * yb);
            HIPCHK(hipMemcpyAsync(tb.data(), src, tb.size() * 4, hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            for (int b = 0; b < B; b++)
                for (int n = 0; n < N; n++) dst[(size_t)b * N + n] = tb[(size_t)(n / yb) * Bp * yb + (size_t)b * yb + n % yb];
            return MLGGD_OK;
        }
        for (int b = 0; b < B; b++)
            for (int n = 0; n < N; n++)
                dst[(size_t)b * N + n] = tr ? t[(size_t)n * Bp + b] : t[(size_t)b * Np + n];
        return MLGGD_OK;
    }
    if (nm == "weights" || nm == "delta_w" || nm == "grad_w") {
        CHK(wait_weight_gathers(e, 1, L - 1));
        const float *src = (nm == "weights") ? e->W[layer] : (nm == "delta_w") ? e->dW[layer] : e->G[layer];
        if (!src) return fail(MLGGD_ERR_STATE, "tensor %s does not exist on this path", name);
        if (!src) return fail(MLGGD_ERR_ARG, "%s not kept for layer %d", name, layer);
        CHK(need((size_t)K * N));
        CHK(download_padded(dst, src, Np, K, N, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return MLGGD_OK;
    }
    if (nm == "bias" || nm == "delta_b") {
        CHK(need(N));
        HIPCHK(hipMemcpyAsync(dst, nm == "bias" ? e->bias[layer] : e->dbias[layer], (size_t)N * 4,
                              hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return MLGGD_OK;
    }
    return fail(MLGGD_ERR_ARG, "unknown tensor name '%s'", name);
}

// ---- data parallel
// gradient buffers (layer_ydedx / layer_sumdedx of BP_WorkSpace) only exist on the all-reduce path
static int shard_alloc(mlggd_engine *e);
static int allreduce_alloc(mlggd_engine *e) {
    // reduce-scatter form: the block table of the sharded factor mode (shard_rows[l] tile rows per rank, W_l grown
    // to world x block rows, one event per layer for the W all-gathers); G_l gets the same rows so that every rank's
    // block of the reduce-scatter has the same size (rows past Kp are never written: they stay zero)
    if (e->ar_shard) CHK(shard_alloc(e));
    size_t gbn = 0;
    for (int l = 1; l < e->L; l++) gbn += e->lsp[l];
    CHK(dev_alloc(e, &e->gb_all, gbn));
    e->gb_all_count = gbn;
    size_t off = 0;
    for (int l = e->L - 1; l >= 1; l--) {
        const size_t rows = e->ar_shard ? (size_t)e->shard_rows[l] * 64 * e->world : (size_t)e->lsp[l - 1];
        CHK(dev_alloc(e, &e->G[l], (rows > (size_t)e->lsp[l - 1] ? rows : (size_t)e->lsp[l - 1]) * e->lsp[l]));
        e->gb[l] = e->gb_all + off;
        off += e->lsp[l];
        HIPCHK(hipEventCreateWithFlags(&e->ev_grad[l], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->ev_red[l], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&e->ev_bias, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_bias_red, hipEventDisableTiming));
    return MLGGD_OK;
}

// A stream of its own is only worth having if it does NOT share a hardware queue with the engine's main stream: the
// HIP runtime multiplexes streams onto a few hardware queues (by reference count, so the mapping depends on every
// stream the process has created and destroyed before), and two streams on one queue serialise.  Round 3, 1-rank
// rehearsal: the first communicator engine created after other engines had come and gone ran its step in 251 instead
// of 173 us for exactly that reason.  (Stream priorities are no way out: a high-priority communication stream made the
// first engine of a process run at 770 us per step.)  So the stream is PROBED: a 300 us spin kernel on the main stream,
// an empty kernel on the candidate; the candidate is kept if its kernel finishes while the spin is still running.
// Rejected candidates are destroyed only afterwards, so that each new one lands on another queue.
static int create_concurrent_stream(mlggd_engine *e, hipStream_t *out, const char *what) {
    hipEvent_t ev_main = nullptr, ev_cand = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_cand, hipEventDisableTiming));
    std::vector<hipStream_t> rejected;
    hipStream_t keep = nullptr;
    int rc = MLGGD_OK, tries = 0;
    static const bool verbose = getenv("MLGGD_VERBOSE") != nullptr;
    for (; tries < 8 && !keep && rc == MLGGD_OK; tries++) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(MLGGD_ERR_DEVICE, "cannot create the %s stream", what);
            break;
        }
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, e->stream, 30000ll);  // 300 us of the 100 MHz wall clock
        hipEventRecord(ev_main, e->stream);
        hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, cand);
        hipEventRecord(ev_cand, cand);
        hipEventSynchronize(ev_cand);
        const bool concurrent = hipEventQuery(ev_main) == hipErrorNotReady;
        hipStreamSynchronize(e->stream);
        if (hipGetLastError() != hipSuccess) rc = fail(MLGGD_ERR_DEVICE, "probing the %s stream failed", what);
        if (concurrent) keep = cand;
        else rejected.push_back(cand);
    }
    if (!keep && !rejected.empty()) {  // no concurrent queue to be had: correct anyway, only slower
        keep = rejected.back();
        rejected.pop_back();
        fprintf(stderr, "mlggd: the %s stream shares a hardware queue with the main stream (no free queue after %d tries)\n",
                what, tries);
    } else if (verbose) {
        fprintf(stderr, "mlggd: %s stream found after %d candidate(s)\n", what, tries);
    }
    for (hipStream_t r : rejected) hipStreamDestroy(r);
    hipEventDestroy(ev_main);
    hipEventDestroy(ev_cand);
    *out = keep;
    return rc;
}
static int create_comm_stream(mlggd_engine *e) { return create_concurrent_stream(e, &e->comm_stream, "communication"); }

int mlggd_comm_unique_id(void *id) {
    if (!id) return fail(MLGGD_ERR_ARG, "id is NULL");
    CHK(rccl_load());
    RcclUniqueId uid;
    NCCLCHK(g_rccl.GetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return MLGGD_OK;
}

int mlggd_comm_init(mlggd_handle e, const void *id, int world_size, int rank) {
    if (!e || !id) return fail(MLGGD_ERR_ARG, "NULL argument");
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return fail(MLGGD_ERR_ARG, "rank %d / world_size %d invalid", rank, world_size);
    if (e->comm) return fail(MLGGD_ERR_STATE, "communicator already initialised");
    if (e->cfg.dropoutflag == 1 && world_size > 1)
        return fail(MLGGD_ERR_ARG, "dropout is not supported on the data-parallel path");
    // the exchange mode first: a request the shape rules out must fail BEFORE a communicator exists (every rank takes
    // the same branch, so nobody is left waiting inside ncclCommInitRank)
    int mode;
    {
        const char *m = getenv("MLGGD_DP_MODE");  // "gather" | "shard" | "allreduce"; default: gather where usable,
        // sharded update from 6 ranks on (the W all-gather costs ~922/world us over world-1 links, the replicated
        // update ~35 us per extra rank: DESIGN.md section 6 -- a cost model, not yet a measurement)
        mode = gather_usable(e, world_size) ? (world_size >= 6 ? 2 : 1) : 0;
        if (m && !strcmp(m, "allreduce")) mode = 0;
        if (m && (!strcmp(m, "gather") || !strcmp(m, "shard") || !strcmp(m, "shard_a2a"))) {
            if (!gather_usable(e, world_size))
                return fail(MLGGD_ERR_ARG, "MLGGD_DP_MODE=%s needs bunchsize %% 32 == 0 and world*bunchsize in {64,128,256,512,1024}", m);
            mode = !strcmp(m, "gather") ? 1 : 2;
            e->dp_a2a = !strcmp(m, "shard_a2a") ? 1 : 0;  // opt-in only: no default or threshold changes before a scaling record exists
            if (e->dp_a2a && e->cfg.dropoutflag == 1)
                return fail(MLGGD_ERR_ARG, "MLGGD_DP_MODE=shard_a2a: dropout masks the row-major activations, which this mode keeps blocked by owner");
        }
    }
    CHK(rccl_load());
    HIPCHK(hipSetDevice(e->device));
    RcclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    NCCLCHK(g_rccl.CommInitRank(&e->comm, world_size, uid, rank));
    e->world = world_size;
    e->rank = rank;
    e->dp_mode = mode;
    dp_knobs(e);
    if (const char *v = getenv("MLGGD_DP_STAT_COMM"))
        if (atoi(v) != 0 && e->cfg.MLflag == 1) {
            if (!g_rccl.CommSplit) return fail(MLGGD_ERR_COMM, "MLGGD_DP_STAT_COMM=1: librccl has no ncclCommSplit");
            NCCLCHK(g_rccl.CommSplit(e->comm, 0, rank, &e->stat_comm, nullptr));  // collective: every rank takes this branch (same env, same MLflag)
        }
    CHK(create_comm_stream(e));
    if (e->dp_a2a && (!g_rccl.Send_ || !g_rccl.Recv_)) return fail(MLGGD_ERR_COMM, "MLGGD_DP_MODE=shard_a2a: librccl has no ncclSend / ncclRecv");
    if (e->dp_mode >= 1) {
        if (e->dp_a2a) CHK(shard_alloc(e));  // the block table first: the all-to-all buffers are sized by it
        CHK(gather_alloc(e));
        return (e->dp_mode == 2 && !e->dp_a2a) ? shard_alloc(e) : MLGGD_OK;
    }
    return allreduce_alloc(e);
}

// Test hook: emulate `world_size` ranks on ONE GPU, without a communicator.  Every training step then
// consumes one GLOBAL minibatch of world_size*bunchsize rows of the resident chunk; emulated rank r owns rows
// [r*bunchsize, (r+1)*bunchsize) of it (the partition of SURVEY 8e; different rows on every rank).  The ranks run
// one after the other (fake_world_prepass, then the real exchange path for the last one) with device copies /
// adds in place of the RCCL calls.  The result must equal the single-device step with bunchsize
// world_size*bunchsize on the same rows -- the data-parallel contract.
// mode: 0 = factor all-gather, replicated update; 1 = factor all-gather, sharded update; 2 = gradient all-reduce.
int mlggd_debug_fake_world(mlggd_handle e, int world_size, int mode) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    if (e->comm || e->fake_world) return fail(MLGGD_ERR_STATE, "communicator already initialised");
    if (mode < 0 || mode > 3) return fail(MLGGD_ERR_ARG, "mode %d not in 0..3", mode);
    if (world_size < 1 || (mode != 2 && !gather_usable(e, world_size)))
        return fail(MLGGD_ERR_ARG, "fake world of %d ranks: needs bunchsize %% 32 == 0 and world*bunchsize in {64,...,1024}", world_size);
    if (e->cfg.dropoutflag == 1 && world_size > 1)
        return fail(MLGGD_ERR_ARG, "dropout is not supported on the data-parallel path");
    HIPCHK(hipSetDevice(e->device));
    e->world = world_size;
    e->rank = world_size - 1;  // the rank that takes the real exchange path; the others are emulated before it
    e->fake_world = true;
    e->dp_mode = mode == 2 ? 0 : (mode == 1 || mode == 3) ? 2 : 1;
    e->dp_a2a = mode == 3 ? 1 : 0;
    if (e->dp_a2a && e->cfg.dropoutflag == 1) return fail(MLGGD_ERR_ARG, "fake world: dropout is not supported with the all-to-all form");
    dp_knobs(e);
    CHK(create_comm_stream(e));
    CHK(dev_alloc(e, &e->colsum_tot, e->Dp));
    if (mode == 2) {
        CHK(allreduce_alloc(e));
        for (int l = 1; l < e->L; l++) CHK(dev_alloc(e, &e->Gpre[l], (size_t)e->lsp[l - 1] * e->lsp[l]));
        CHK(dev_alloc(e, &e->gbpre, e->gb_all_count));
        HIPCHK(hipStreamSynchronize(e->stream));
        return MLGGD_OK;
    }
    if (mode == 3) CHK(shard_alloc(e));
    CHK(gather_alloc(e));
    if (mode == 1) CHK(shard_alloc(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    return MLGGD_OK;
}
// size and rank of the RCCL communicator as RCCL itself reports them (ncclCommCount / ncclCommUserRank);
// 0 / -1 without a communicator
int mlggd_comm_info(mlggd_handle e, int *nranks, int *rank) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    int n = 0, r = -1;
    if (e->comm) {
        if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(MLGGD_ERR_COMM, "librccl lacks ncclCommCount / ncclCommUserRank");
        NCCLCHK(g_rccl.CommCount(e->comm, &n));
        NCCLCHK(g_rccl.CommUserRank(e->comm, &r));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return MLGGD_OK;
}
// number of cached launch plans of the persistent dW kernel (tests: it must not grow with the step count)
int mlggd_debug_plan_count(mlggd_handle e, int *plans) {
    if (!e || !plans) return fail(MLGGD_ERR_ARG, "NULL argument");
    *plans = (int)e->dwp_tables.size();
    return MLGGD_OK;
}
int mlggd_debug_gemm_plan(mlggd_handle e, int layer, int *fwd_waves, int *dx_waves) {
    if (!e || !fwd_waves || !dx_waves) return fail(MLGGD_ERR_ARG, "NULL argument");
    if (layer < 1 || layer >= e->L) return fail(MLGGD_ERR_ARG, "layer %d not in 1..%d", layer, e->L - 1);
    *fwd_waves = fwd64_used(e, layer) ? 1 : e->fwd_nw;  // (the output layer's 4 waves x out_slabs otherwise)
    *dx_waves = dx64_used(e, layer) ? 1 : e->dx_nw;
    return MLGGD_OK;
}
int mlggd_debug_out_slabs(mlggd_handle e, int *slabs) {
    if (!e || !slabs) return fail(MLGGD_ERR_ARG, "NULL argument");
    *slabs = e->S_out;
    return MLGGD_OK;
}
int mlggd_dp_mode(mlggd_handle e, int *mode) {
    if (!e || !mode) return fail(MLGGD_ERR_ARG, "NULL argument");
    // 0 single device, 1 all-reduce, 2 gather, 3 gather + sharded update, 4 = 3 with the activations by all-to-all
    *mode = (e->comm || e->fake_world) ? (e->dp_a2a ? 4 : 1 + e->dp_mode) : 0;
    return MLGGD_OK;
}

// Diagnostic: out[i] = fn(x[i], y) evaluated ON THE DEVICE with the same libm calls the kernels use
// ("powf" "expf" "sigmoid" "div"); tests measure their distance to the oracle's host libm in ulps.
int mlggd_debug_math(mlggd_handle e, const char *fn, const float *x, float y, float *out, size_t n) {
    if (!e || !fn || (n > 0 && (!x || !out))) return fail(MLGGD_ERR_ARG, "NULL argument");
    const int f = !strcmp(fn, "powf") ? 0 : !strcmp(fn, "expf") ? 1 : !strcmp(fn, "sigmoid") ? 2 : !strcmp(fn, "div") ? 3 : !strcmp(fn, "exp_det") ? 4 : !strcmp(fn, "pow_det") ? 5 : !strcmp(fn, "sigmoid4") ? 6 : -1;
    if (f < 0) return fail(MLGGD_ERR_ARG, "unknown function '%s'", fn);
    if (n == 0) return MLGGD_OK;
    HIPCHK(hipSetDevice(e->device));
    float *dx = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&dx, n * sizeof(float)));
    if (hipMalloc((void **)&dout, n * sizeof(float)) != hipSuccess) {
        hipFree(dx);
        return fail(MLGGD_ERR_DEVICE, "hipMalloc of %zu bytes failed", n * sizeof(float));
    }
    int rc = MLGGD_OK;
    if (hipMemcpyAsync(dx, x, n * sizeof(float), hipMemcpyHostToDevice, e->stream) != hipSuccess) rc = MLGGD_ERR_DEVICE;
    if (rc == MLGGD_OK) {
        hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, f, dx, y, dout, n);
        rc = launch_check("k_debug_math");
    }
    if (rc == MLGGD_OK && (hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                           hipStreamSynchronize(e->stream) != hipSuccess))
        rc = fail(MLGGD_ERR_DEVICE, "mlggd_debug_math: copy back failed");
    hipFree(dx);
    hipFree(dout);
    return rc;
}

// ---- per-kernel-class timing inside the timed region (bench.py roofline object)
int mlggd_profile_select(mlggd_handle e, const char *kernel_class, int layer, int max_launches) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    HIPCHK(hipSetDevice(e->device));
    e->prof_class = -1;
    e->prof_used = 0;
    if (!kernel_class || !*kernel_class) return MLGGD_OK;
    for (int c = 0; c < KC_COUNT; c++)
        if (!strcmp(kernel_class, kKernelClassName[c])) e->prof_class = c;
    if (e->prof_class < 0) return fail(MLGGD_ERR_ARG, "unknown kernel class '%s'", kernel_class);
    e->prof_layer = layer;
    if (max_launches < 1) max_launches = 1;
    while (e->prof_ev.size() < (size_t)2 * max_launches) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        e->prof_ev.push_back(ev);
    }
    return MLGGD_OK;
}

int mlggd_profile_stride(mlggd_handle e, int every_nth_step) {
    if (!e) return fail(MLGGD_ERR_ARG, "NULL handle");
    e->prof_stride = every_nth_step < 1 ? 1 : every_nth_step;
    return MLGGD_OK;
}

int mlggd_profile_read(mlggd_handle e, float *mean_usec, int *launches) {
    if (!e || !mean_usec) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    double tot = 0;
    int n = 0;
    for (size_t i = 0; i + 1 < e->prof_used; i += 2) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]));
        tot += ms;
        n++;
    }
    *mean_usec = n ? (float)(tot * 1000.0 / n) : 0.0f;
    if (launches) *launches = n;
    e->prof_used = 0;
    return MLGGD_OK;
}

// Diagnostic: phase stamps of the NEXT launch of (class, layer); see kernels.hip.h stamp().
int mlggd_debug_stamp_select(mlggd_handle e, const char *kernel_class, int layer) {
    if (!e || !kernel_class) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    if (!e->stamp_buf) {
        e->stamp_cap = 8 * 8192;
        HIPCHK(hipMalloc((void **)&e->stamp_buf, e->stamp_cap * sizeof(long long)));
        e->allocs.push_back(e->stamp_buf);
    }
    HIPCHK(hipMemsetAsync(e->stamp_buf, 0, e->stamp_cap * sizeof(long long), e->stream));
    e->stamp_class = -1;
    for (int c = 0; c < KC_COUNT; c++)
        if (!strcmp(kernel_class, kKernelClassName[c])) e->stamp_class = c;
    if (e->stamp_class < 0) return fail(MLGGD_ERR_ARG, "unknown kernel class '%s'", kernel_class);
    e->stamp_layer = layer;
    e->stamp_blocks = 0;
    return MLGGD_OK;
}

int mlggd_debug_stamp_read(mlggd_handle e, long long *out, int cap_blocks, int *blocks) {
    if (!e || !out || !blocks) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int n = e->stamp_blocks < cap_blocks ? e->stamp_blocks : cap_blocks;
    if (n > 0) HIPCHK(hipMemcpy(out, e->stamp_buf, (size_t)n * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    *blocks = n;
    return MLGGD_OK;
}

// Cost of one HIP-event bracket on the engine's stream, calibrated in-process: a kernel
// bracketed once measures overhead + t, bracketed twice back to back overhead + 2t, so
// overhead = 2*T1 - T2.  Uses an empty kernel (a kernel that writes memory would make the second of two
// back-to-back launches wait for the first one's dirty lines and skew T2).
int mlggd_profile_overhead(mlggd_handle e, float *usec) {
    if (!e || !usec) return fail(MLGGD_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(e->device));
    const int reps = 40;
    std::vector<hipEvent_t> ev(4 * reps);
    for (auto &x : ev) HIPCHK(hipEventCreate(&x));
    auto nop = [&]() -> int {
        hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), 0, e->stream);
        return launch_check("k_nop");
    };
    for (int r = 0; r < reps; r++) {
        HIPCHK(hipEventRecord(ev[4 * r], e->stream));
        CHK(nop());
        HIPCHK(hipEventRecord(ev[4 * r + 1], e->stream));
        HIPCHK(hipEventRecord(ev[4 * r + 2], e->stream));
        CHK(nop());
        CHK(nop());
        HIPCHK(hipEventRecord(ev[4 * r + 3], e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    double t1 = 0, t2 = 0;
    for (int r = 4; r < reps; r++) {  // first reps warm up
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, ev[4 * r], ev[4 * r + 1]));
        HIPCHK(hipEventElapsedTime(&b, ev[4 * r + 2], ev[4 * r + 3]));
        t1 += a;
        t2 += b;
    }
    for (auto &x : ev) hipEventDestroy(x);
    const double ov = (2 * t1 - t2) / (reps - 4) * 1000.0;
    *usec = ov > 0 ? (float)ov : 0.0f;
    return MLGGD_OK;
}

// Static description of the launch plan (DESIGN.md "kernels"): algorithmic FLOPs and bytes of
// one launch of the (class, layer) kernel; layer 0 = sum over layers.
int mlggd_dw_launches_per_step(mlggd_handle e, int *launches) {
    if (!e || !launches) return fail(MLGGD_ERR_ARG, "NULL argument");
    const bool dp = e->comm != nullptr || e->fake_world;
    const bool merged = (!dp && !e->two_streams && e->dw_merge && dwp_usable(e)) || (dp && e->dp_mode >= 1);
    const bool fine = dp && e->dp_mode == 1 && e->dp_fine != 0 && e->L - 1 >= 2;
    *launches = fine ? 2 : merged ? 1 : e->L - 1;
    return MLGGD_OK;
}

int mlggd_kernel_work(mlggd_handle e, const char *kernel_class, int layer, double *flops, double *bytes) {
    if (!e || !kernel_class) return fail(MLGGD_ERR_ARG, "NULL argument");
    double f = 0, by = 0;
    // the gather path runs the dW kernel over the global minibatch on every rank
    // the gather paths run the dW kernel over the global minibatch: replicated (dp_mode 1) on all tiles,
    // sharded (dp_mode 2) on this rank's 1/world of the weight rows
    const bool dpx = (e->comm != nullptr || e->fake_world) && e->dp_mode >= 1 && !strcmp(kernel_class, "dw");
    const double W = e->world, B = e->B;
    auto gemm = [&](const char *cls, int l, double &ff, double &bb) {
        if (l < 1 || l >= e->L) return;
        const double K = e->ls[l - 1], N = e->ls[l];
        if (!strcmp(cls, "fwd")) {
            ff += 2.0 * B * K * N;
            bb += 4.0 * (K * N + B * K + 2 * B * N);
        } else if (!strcmp(cls, "dx")) {
            if (l == 1) return;
            ff += 2.0 * B * K * N;
            bb += 4.0 * (K * N + B * N + 3 * B * K);
        } else if (!strcmp(cls, "dw")) {
            if (!dpx) {
                ff += 2.0 * B * K * N;
                bb += 4.0 * (4 * K * N + B * K + B * N);  // W, delta read + written
            } else if (e->dp_mode == 1) {
                ff += 2.0 * B * W * K * N;
                bb += 4.0 * (4 * K * N + B * W * K + B * W * N);
            } else {
                ff += 2.0 * B * W * K * N / W;
                bb += 4.0 * (4 * K * N / W + B * W * K / W + B * W * N);
            }
        }
    };
    for (int l = 1; l < e->L; l++) {
        if (layer != 0 && layer != l) continue;
        gemm(kernel_class, l, f, by);
    }
    if (flops) *flops = f;
    if (bytes) *bytes = by;
    return MLGGD_OK;
}

}  // extern "C"
