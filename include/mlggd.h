/*
 * mlggd.h -- C-ABI of the MI355X (gfx950) training engine: the drop-in boundary for the
 * reference's device engine `class BP_GPU` (Train_code_ML_GGD/BP_GPU.h:45-70), which has
 * no FFI layer of its own.  Plain pointers and sizes only; every entry point returns an int
 * status (0 = MLGGD_OK) and mlggd_last_error() gives the message, where the reference
 * printf()s and exit(0)s (BP_GPU.cu:20,534,578).
 *
 * Conventions shared with the reference:
 *  - host matrices are row-major: in [n_frames][layersizes[0]], targ/out
 *    [n_frames][layersizes[L-1]], weights[l] [layersizes[l-1]][layersizes[l]];
 *  - weights[] / bias[] are arrays of numlayers pointers indexed by layer l = 1..L-1
 *    (slot 0 unused), exactly as BP_GPU's float** arguments (BPtrain.cc:77-78,108);
 *  - the caller owns all host buffers; the engine copies on entry and owns device memory;
 *  - calls are synchronous unless stated otherwise and must come from one thread.
 */
#ifndef MLGGD_H
#define MLGGD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLGGD_MAXLAYER 10          /* BP_GPU.h:6  MAXLAYER */
#define MLGGD_MAXCACHEFRAME 200000 /* BP_GPU.h:7  MAXCACHEFRAME */
#define MLGGD_UNIQUE_ID_BYTES 128  /* sizeof(ncclUniqueId) */

enum {
    MLGGD_OK = 0,
    MLGGD_ERR_ARG = 1,     /* bad argument / shape */
    MLGGD_ERR_DEVICE = 2,  /* HIP runtime error (no device, alloc, launch) */
    MLGGD_ERR_COMM = 3,    /* RCCL error */
    MLGGD_ERR_STATE = 4    /* call not valid in the engine's current state */
};

typedef struct mlggd_engine *mlggd_handle;

/* Constructor arguments of BP_GPU (BP_GPU.h:48-49, BP_GPU.cu:9-11), same meaning. */
typedef struct mlggd_config {
    int32_t struct_size;     /* = sizeof(mlggd_config), ABI guard */
    int32_t random_seed;     /* seeds the dropout generator (BP_GPU.cu:59-60) */
    int32_t device;          /* GPU ordinal, BP_GPU.cu:15-23 (gpu_used=) */
    int32_t numlayers;       /* 2..MLGGD_MAXLAYER */
    int32_t layersizes[MLGGD_MAXLAYER];
    int32_t bunchsize;       /* frames per minibatch on THIS rank */
    float lrate;
    float momentum;
    float weightcost;
    float shapefactor;       /* beta of the GGD / beta-norm */
    int32_t MLflag;          /* 1: ML-GGD objective, else beta-norm (BP_GPU.cu:408-424) */
    int32_t dropoutflag;     /* BP_GPU.cu:344-355,484-501 */
    float visible_omit;
    float hid_omit;
    int32_t max_cache_frames; /* rows of the resident chunk buffers; 0 -> MLGGD_MAXCACHEFRAME */
    int32_t reserved[7];
} mlggd_config;

/* ---- lifetime: BP_GPU::BP_GPU / ~BP_GPU (BP_GPU.cu:9-150) ---- */
int mlggd_create(const mlggd_config *cfg, const float *const *weights, const float *const *bias,
                 mlggd_handle *out);
int mlggd_destroy(mlggd_handle h);
const char *mlggd_last_error(void);
int mlggd_device_count(int *count); /* cudaGetDeviceCount, BP_GPU.cu:15 */

/* ---- training: BP_GPU::train (BP_GPU.cu:152-185) ----
 * Uploads the chunk, runs train_bunch_single (BP_GPU.cu:308-440) on every FULL bunch and
 * skips the trailing partial bunch, returns after the last step has completed.
 * *bunches_trained (optional) receives the number of steps run. */
int mlggd_train_chunk(mlggd_handle h, int n_frames, const float *in, const float *targ,
                      int *bunches_trained);

/* The same, split so a benchmark can time steps on HBM-resident data:
 * mlggd_load_chunk = the two todev_vf_vf calls (BP_GPU.cu:163-164);
 * mlggd_train_resident = the bunch loop (BP_GPU.cu:170-184) over frames
 * [first_frame, first_frame + n_frames) of the resident chunk; asynchronous --
 * mlggd_sync() waits for completion. */
int mlggd_load_chunk(mlggd_handle h, int n_frames, const float *in, const float *targ);
int mlggd_train_resident(mlggd_handle h, int first_frame, int n_frames, int *bunches_trained);
int mlggd_sync(mlggd_handle h);

/* ---- cross-validation: BP_GPU::CrossValid / CrossValiddB / CrossValid2
 * (BP_GPU.cu:187-306) over cv_bunch_single (BP_GPU.cu:442-512).  Each returns the chunk's
 * sum exactly as the reference accumulates it (host fp32 scalar, frame-major order):
 * sqerr = sum (o-t)^2 ; abserr = sum |o-t| / D ; loglik = GGD log-likelihood with the
 * alpha of the last training minibatch. */
int mlggd_cv_sqerr(mlggd_handle h, int n_frames, const float *in, const float *targ, float *out);
int mlggd_cv_abserr(mlggd_handle h, int n_frames, const float *in, const float *targ, float *out);
int mlggd_cv_loglik(mlggd_handle h, int n_frames, const float *in, const float *targ, float *out);
/* All three from ONE forward pass (same accumulation order; loglik only if MLflag==1). */
int mlggd_cv_all(mlggd_handle h, int n_frames, const float *in, const float *targ,
                 float *sqerr, float *abserr, float *loglik);
/* cv_bunch_single over a whole chunk: out[n_frames][D] network outputs (forward only). */
int mlggd_forward(mlggd_handle h, int n_frames, const float *in, float *out);

/* ---- input pipeline on the device (no counterpart in the reference, which
 * context-expands every chunk on one host thread, Interface.cc:778-785, and uploads the 11x
 * larger matrix, BP_GPU.cu:163-164).  A sample is a window of fea_context CONSECUTIVE frames,
 * i.e. a contiguous slice of the normalised frame stream, so the caller uploads the chunk's
 * frames once -- feat [n_frames][layersizes[0]/fea_context], targ [n_frames][D] -- plus, for
 * every sample ROW (in the shuffled row order Readchunk would have produced), the index of its
 * first frame; sample i's target is frame first_frame[i] + targ_offset.  Rows are gathered by
 * the input-staging kernel; results are bit-identical to the expanded path.
 * After mlggd_load_frames, mlggd_train_resident indexes SAMPLES of this chunk. */
int mlggd_load_frames(mlggd_handle h, int n_frames, int fea_context, const float *feat, const float *targ,
                      int n_samples, const int32_t *first_frame, int targ_offset);
int mlggd_train_frames(mlggd_handle h, int n_frames, int fea_context, const float *feat, const float *targ,
                       int n_samples, const int32_t *first_frame, int targ_offset, int *bunches_trained);
/* mlggd_train_frames without the final wait: returns when the chunk is on the device and its steps are enqueued;
 * the next chunk's upload then overlaps them (two device buffer sets).  mlggd_sync() waits. */
int mlggd_train_frames_async(mlggd_handle h, int n_frames, int fea_context, const float *feat, const float *targ,
                       int n_samples, const int32_t *first_frame, int targ_offset, int *bunches_trained);
int mlggd_cv_all_frames(mlggd_handle h, int n_frames, int fea_context, const float *feat, const float *targ,
                        int n_samples, const int32_t *first_frame, int targ_offset, float *sqerr, float *abserr,
                        float *loglik);
int mlggd_forward_frames(mlggd_handle h, int n_frames, int fea_context, const float *feat, int n_samples,
                         const int32_t *first_frame, float *out);
/* pinned host memory for chunk buffers (optional; faster H2D than pageable memory) */
int mlggd_alloc_pinned(size_t bytes, void **out);
/* the same from a thread that has not selected a device (host IO threads): pins through `device`'s context */
int mlggd_alloc_pinned_on(int device, size_t bytes, void **out);
int mlggd_free_pinned(void *p);

/* ---- state: BP_GPU::returnWeights (BP_GPU.cu:514-525) and dev.scalefactor (:287) ---- */
int mlggd_get_weights(mlggd_handle h, float *const *weights, float *const *bias);
int mlggd_set_weights(mlggd_handle h, const float *const *weights, const float *const *bias);
int mlggd_get_scalefactor(mlggd_handle h, float *alpha /* [D] */);
int mlggd_set_scalefactor(mlggd_handle h, const float *alpha /* [D] */);
int mlggd_set_lrate(mlggd_handle h, float lrate);
/* CV metrics: on = the three sums are formed on the device (per-tile partials in double, combined
 * on the host; no n x D copy, no host loop); off (default, or env MLGGD_CV_DEVICE=0) = outputs copied back and
 * accumulated on the host in fp32 in the reference's frame-major order (BP_GPU.cu:207-213), the values the
 * reference's log lines carry.  The two differ by the rounding of that fp32 accumulation, which grows with the size
 * of the CV set: 1e-5 relative at 1,200 frames, 1.4e-3 at 7,920 frames x 257 (the running sum is then ~1e7 times a
 * term); the device sums are the accurate ones, the host-order ones are what the reference prints. */
int mlggd_set_cv_device_reduce(mlggd_handle h, int on);
float mlggd_gamma(float x); /* BP_GPU::Gamma, BP_GPU.cu:593-640 */

/* Copies an internal tensor of the LAST step to the host in the reference's row-major
 * layout (parity tests).  name: "out" "y" "dedx" [bunchsize][units(layer)];
 * "delta_w" "weights" [K][N]; "delta_b" "bias" [N]; "scalefactor" [D].
 * count = capacity of dst in floats; fails if too small. */
int mlggd_debug_tensor(mlggd_handle h, const char *name, int layer, float *dst, size_t count);

/* ---- data parallel over the GPUs of one node (new work): one process per
 * GPU; rank r trains rows [r*bunchsize,(r+1)*bunchsize) of every global minibatch of
 * world*bunchsize frames.  Exchanges per step (RCCL): the per-dimension sum |e|^beta (ML only,
 * all-reduce of 257 floats) and ONE of three forms of the gradient exchange (mlggd_dp_mode below):
 * an all-reduce of the weight/bias gradients; an all-gather of the gradient's FACTORS (every rank's
 * Y_{l-1} and dEdX_l rows) after which each rank forms the global-minibatch gradient itself; or the
 * factor all-gather with each rank updating only its block of weight rows, followed by an all-gather
 * of W.  Every 1/n_frames factor uses the GLOBAL minibatch size, so the run equals a single-device
 * run with bunchsize = world*bunchsize.  rank 0 fills a unique id, the caller broadcasts it out of
 * band.  Status: tested on one GPU only (1-rank communicator; emulated worlds of 2-8 ranks with
 * different rows per rank); never run between two GPUs -- the mode thresholds are a cost model. */
int mlggd_comm_unique_id(void *id /* MLGGD_UNIQUE_ID_BYTES */);
int mlggd_comm_init(mlggd_handle h, const void *id, int world_size, int rank);
/* what RCCL itself reports for the engine's communicator (ncclCommCount / ncclCommUserRank); 0 / -1 without one.
 * bench.py prints it as `rccl_ranks` so a multi-GPU line proves the collectives ran over that many ranks. */
int mlggd_comm_info(mlggd_handle h, int *nranks, int *rank);

/* Per-step device time of the last mlggd_train_resident call, measured with HIP events
 * on the engine's stream: total ms over `steps` steps. */
int mlggd_last_train_ms(mlggd_handle h, float *ms, int *steps);

/* Kernel-class timing INSIDE a timed mlggd_train_resident region (bench.py's roofline
 * object): mlggd_profile_select times every launch of the named class ("transpose" "fwd"
 * "loss" "dx" "dw" "update"; layer 0 = all layers) with a pair of HIP events, up to
 * max_launches; NULL/"" switches it off.  The GEMM classes ("fwd" "dx" "dw") take the pair INTO
 * the launch (hipExtLaunchKernelGGL start/stop events = the dispatch's own begin/end timestamps,
 * what rocprofv3 --kernel-trace reports); the other classes are bracketed by events recorded on
 * the stream before and after, which adds the bracket's cost.  mlggd_profile_read syncs and returns
 * the mean launch duration in microseconds and the number of launches seen.
 * mlggd_kernel_work gives the algorithmic FLOPs / bytes of ONE launch of (class, layer)
 * (layer 0 = summed over layers), the figures DESIGN.md states per kernel. */
int mlggd_profile_select(mlggd_handle h, const char *kernel_class, int layer, int max_launches);
int mlggd_profile_stride(mlggd_handle h, int every_nth_step); /* bracket only every n-th step (default 1) */
int mlggd_profile_read(mlggd_handle h, float *mean_usec, int *launches);
/* cost of one event bracket itself (in-process calibration: 2*T(one kernel) - T(two kernels)) */
int mlggd_profile_overhead(mlggd_handle h, float *usec);
int mlggd_kernel_work(mlggd_handle h, const char *kernel_class, int layer, double *flops, double *bytes);
/* how many launches of the weight-gradient/update kernel one training step issues: 1 when the
 * layers share one persistent launch (single GPU), numlayers-1 otherwise */
int mlggd_dw_launches_per_step(mlggd_handle h, int *launches);
/* 0 = single device, 1 = data parallel by all-reduce of the weight gradients, 2 = by all-gather of their
 * factors with the update replicated on every rank, 3 = the same with the update sharded over the ranks and
 * W all-gathered, 4 = 3 with the activations exchanged by all-to-all, each rank receiving only the units of its block
 * of weight rows (opt-in: MLGGD_DP_MODE=shard_a2a) (defaults: 2 up to 5 ranks, 3 from 6 ranks, 1 where the shape rules out the gather --
 * bunchsize % 32 != 0 or world*bunchsize not in {64,128,256,512,1024}; MLGGD_DP_MODE=allreduce|gather|shard
 * at comm init overrides) */
int mlggd_dp_mode(mlggd_handle h, int *mode);
/* Test hook: emulate `world_size` ranks on one GPU (device copies / adds instead of collectives).  Every
 * training step then consumes world_size*bunchsize rows of the resident chunk, emulated rank r owning rows
 * [r*bunchsize,(r+1)*bunchsize) of them.  mode: 0 = factor all-gather + replicated update, 1 = factor
 * all-gather + sharded update, 2 = gradient all-reduce, 3 = sharded update with the activations by all-to-all. */
int mlggd_debug_fake_world(mlggd_handle h, int world_size, int mode);
/* Test hook: number of launch plans (tile-record tables) the persistent dW kernel has cached; constant after the
 * first steps of a run (2 on one GPU, a few in the data-parallel modes). */
int mlggd_debug_plan_count(mlggd_handle h, int *plans);

/* Test hook: the number of split-K slabs of the output-layer forward GEMM (the loss kernel adds them in order).  The
 * oracle's MFMA-order twin needs it to restate the HIP path's summation order (oracle/mlggd_oracle.c). */
int mlggd_debug_out_slabs(mlggd_handle h, int *slabs);
/* Test hook: into how many waves' contiguous ranges layer `layer`'s forward GEMM and its dX GEMM (the one that produces
 * dEdX of layer - 1) cut their reduction: 4 = one 32 x 32 output tile per workgroup (k_fwd / k_dx), 1 = the 64 x 64-tile
 * kernels for large minibatches (k_fwd64 / k_dx64: one chain per output element).  The MFMA-order twin restates it. */
int mlggd_debug_gemm_plan(mlggd_handle h, int layer, int *fwd_waves, int *dx_waves);

/* Diagnostic: out[i] = fn(x[i], y) evaluated on the device -- fn "pow_det" (the loss chain's power: kernindex2 / kernfunc2 /
 * kernSubClean2, DevFunc.cu:219-227,468-489,376-398), "exp_det", "sigmoid" = 1 / (1 + exp_det(-x)) (kernSigmoid,
 * DevFunc.cu:36-51): what the kernels themselves evaluate, IEEE operations only, restated in the oracle's MFMA-order twin --
 * the parity tests require the SAME BITS on both sides for every argument; "div" = x / y (IEEE); and, for the record only,
 * ocml's own "powf" and "expf", which no kernel calls: how far they sit from the correctly rounded values, in ulps. */
int mlggd_debug_math(mlggd_handle h, const char *fn, const float *x, float y, float *out, size_t n);

/* Diagnostic (not part of the reference surface): in-kernel phase stamps of the NEXT launch of
 * (class "fwd"|"dx"|"dw", layer): 8 int64 slots per workgroup in 100 MHz ticks
 * (s_memrealtime); slot meaning per kernel is documented at stamp() call sites in
 * csrc/kernels.hip.h.  The stamps go to a debug buffer only. */
int mlggd_debug_stamp_select(mlggd_handle h, const char *kernel_class, int layer);
int mlggd_debug_stamp_read(mlggd_handle h, long long *out /* [cap_blocks][8] */, int cap_blocks, int *blocks);

#ifdef __cplusplus
}
#endif
#endif /* MLGGD_H */
