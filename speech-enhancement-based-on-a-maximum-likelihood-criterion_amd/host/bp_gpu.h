// bp_gpu.h -- a `class BP_GPU` with the reference's constructor and public methods
// (Train_code_ML_GGD/BP_GPU.h:45-70) implemented as a thin C++ shim over the C-ABI of
// include/mlggd.h, so a main() written against the reference class compiles against the
// MI355X engine unchanged.  Error behaviour: the reference printf()s and exit(0)s
// (BP_GPU.cu:20,534,578); the shim prints the engine's message and exit(1)s.
#pragma once
#include "../../include/mlggd.h"

class BP_GPU {
  public:
    BP_GPU(int random_seed, int a_GPU_selected, int a_numlayers, int *a_layersizes, int a_bunchsize, float a_lrate,
           float a_momentum, float a_weightcost, float **weights, float **bias, float shapefactor, int MLflag,
           int dropoutflag, float visible_omit, float hid_omit);
    ~BP_GPU();
    void train(int n_frames, float *in, const float *targ);
    // One SGD step on n_frames = bunchsize rows (BP_GPU.h:53, BP_GPU.cu:308-440).  The reference's member is public
    // but takes pointers into its private device workspace (train passes dev.in + offset, BP_GPU.cu:170-184), so no
    // outside caller can use it there; here it takes HOST rows like train() does: upload + one step + wait.
    void train_bunch_single(int n_frames, float *in, const float *targ);
    float CrossValid(int n_frames, const float *in, const float *targ);
    float CrossValiddB(int n_frames, const float *in, const float *targ);
    float CrossValid2(int n_frames, const float *in, const float *targ);
    // one forward pass for all three metrics (same accumulation order as the three above)
    void CrossValidAll(int n_frames, const float *in, const float *targ, float *sqerr, float *abserr, float *loglik);
    void cv_bunch_single(int n_frames, const float *in, float *out);
    // frame-stream chunks (not in the reference): rows gathered on the device, see mlggd_load_frames
    // wait = false: return once the chunk is uploaded (the buffers are free) and its steps are enqueued; the next
    // chunk's upload then overlaps them.  sync() / returnWeights / CrossValid* wait for everything.
    void train_frames(int n_frames, int fea_context, const float *feat, const float *targ, int n_samples,
                      const int *first_frame, int targ_offset, bool wait = true);
    void sync();
    void CrossValidAll_frames(int n_frames, int fea_context, const float *feat, const float *targ, int n_samples,
                              const int *first_frame, int targ_offset, float *sqerr, float *abserr, float *loglik);
    void returnWeights(float **weights, float **bias);
    float Gamma(float x) { return mlggd_gamma(x); }
    // data parallel (not in the reference): join an RCCL communicator of `world` ranks
    void joinComm(const void *unique_id, int world, int rank);
    mlggd_handle handle() { return h_; }

    int numlayers;
    int layersizes[MLGGD_MAXLAYER];
    int bunchsize;
    float lrate, shapefactor, momentum, weightcost;
    int dropoutflag, MLflag;
    float visible_omit, hid_omit;

  private:
    void check(int rc, const char *what);
    mlggd_handle h_ = nullptr;
};
