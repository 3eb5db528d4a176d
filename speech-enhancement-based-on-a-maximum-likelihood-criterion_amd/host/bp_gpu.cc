#include "bp_gpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

void BP_GPU::check(int rc, const char *what) {
    if (rc == MLGGD_OK) return;
    fprintf(stderr, "%s failed: %s\n", what, mlggd_last_error());
    printf("%s failed: %s\n", what, mlggd_last_error());
    exit(1);
}

BP_GPU::BP_GPU(int random_seed, int a_GPU_selected, int a_numlayers, int *a_layersizes, int a_bunchsize, float a_lrate,
               float a_momentum, float a_weightcost, float **weights, float **bias, float a_shapefactor, int a_MLflag,
               int a_dropoutflag, float a_visible_omit, float a_hid_omit)
    : numlayers(a_numlayers), bunchsize(a_bunchsize), lrate(a_lrate), shapefactor(a_shapefactor),
      momentum(a_momentum), weightcost(a_weightcost), dropoutflag(a_dropoutflag), MLflag(a_MLflag),
      visible_omit(a_visible_omit), hid_omit(a_hid_omit) {
    int ndev = 0;
    check(mlggd_device_count(&ndev), "mlggd_device_count");
    printf("Total GPU Device : %d\n", ndev);  // BP_GPU.cu:16
    mlggd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.random_seed = random_seed;
    cfg.device = a_GPU_selected;
    cfg.numlayers = a_numlayers;
    for (int i = 0; i < a_numlayers && i < MLGGD_MAXLAYER; i++) layersizes[i] = cfg.layersizes[i] = a_layersizes[i];
    cfg.bunchsize = a_bunchsize;
    cfg.lrate = a_lrate;
    cfg.momentum = a_momentum;
    cfg.weightcost = a_weightcost;
    cfg.shapefactor = a_shapefactor;
    cfg.MLflag = a_MLflag;
    cfg.dropoutflag = a_dropoutflag;
    cfg.visible_omit = a_visible_omit;
    cfg.hid_omit = a_hid_omit;
    check(mlggd_create(&cfg, weights, bias, &h_), "mlggd_create");
    printf("Use GPU Device : %d\n", a_GPU_selected);                                        // BP_GPU.cu:23
    printf("Created net with %d layers, bunchsize %d.\n", numlayers, bunchsize);            // BP_GPU.cu:110
}

BP_GPU::~BP_GPU() { mlggd_destroy(h_); }

void BP_GPU::train(int n_frames, float *in, const float *targ) {
    int trained = 0;
    check(mlggd_train_chunk(h_, n_frames, in, targ, &trained), "mlggd_train_chunk");
    const int rest = n_frames - trained * bunchsize;
    if (rest > 0) printf("this bunch has only %d samples and is ignored.\n", rest);  // BP_GPU.cu:179
}

void BP_GPU::train_bunch_single(int n_frames, float *in, const float *targ) {
    if (n_frames != bunchsize) {  // the engine's launch plan is built for one minibatch size (the reference's n_frames here is always bunchsize, BP_GPU.cu:175)
        fprintf(stderr, "train_bunch_single: n_frames %d is not the bunchsize %d\n", n_frames, bunchsize);
        printf("train_bunch_single: n_frames %d is not the bunchsize %d\n", n_frames, bunchsize);
        exit(1);
    }
    int trained = 0;
    check(mlggd_load_chunk(h_, n_frames, in, targ), "mlggd_load_chunk");
    check(mlggd_train_resident(h_, 0, n_frames, &trained), "mlggd_train_resident");
    check(mlggd_sync(h_), "mlggd_sync");
}

float BP_GPU::CrossValid(int n, const float *in, const float *targ) {
    float v = 0;
    check(mlggd_cv_sqerr(h_, n, in, targ, &v), "mlggd_cv_sqerr");
    return v;
}
float BP_GPU::CrossValiddB(int n, const float *in, const float *targ) {
    float v = 0;
    check(mlggd_cv_abserr(h_, n, in, targ, &v), "mlggd_cv_abserr");
    return v;
}
float BP_GPU::CrossValid2(int n, const float *in, const float *targ) {
    float v = 0;
    check(mlggd_cv_loglik(h_, n, in, targ, &v), "mlggd_cv_loglik");
    return v;
}
void BP_GPU::CrossValidAll(int n, const float *in, const float *targ, float *sqerr, float *abserr, float *loglik) {
    check(mlggd_cv_all(h_, n, in, targ, sqerr, abserr, loglik), "mlggd_cv_all");
}
void BP_GPU::cv_bunch_single(int n, const float *in, float *out) {
    check(mlggd_forward(h_, n, in, out), "mlggd_forward");
}
void BP_GPU::sync() { check(mlggd_sync(h_), "mlggd_sync"); }
void BP_GPU::train_frames(int n_frames, int fea_context, const float *feat, const float *targ, int n_samples,
                          const int *first_frame, int targ_offset, bool wait) {
    int trained = 0;
    if (wait)
        check(mlggd_train_frames(h_, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset, &trained),
              "mlggd_train_frames");
    else
        check(mlggd_train_frames_async(h_, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset,
                                       &trained),
              "mlggd_train_frames_async");
    const int rest = n_samples - trained * bunchsize;
    if (rest > 0) printf("this bunch has only %d samples and is ignored.\n", rest);  // BP_GPU.cu:179
}
void BP_GPU::CrossValidAll_frames(int n_frames, int fea_context, const float *feat, const float *targ, int n_samples,
                                  const int *first_frame, int targ_offset, float *sqerr, float *abserr, float *loglik) {
    check(mlggd_cv_all_frames(h_, n_frames, fea_context, feat, targ, n_samples, first_frame, targ_offset, sqerr, abserr,
                              loglik),
          "mlggd_cv_all_frames");
}
void BP_GPU::returnWeights(float **weights, float **bias) {
    check(mlggd_get_weights(h_, weights, bias), "mlggd_get_weights");
}
void BP_GPU::joinComm(const void *unique_id, int world, int rank) {
    check(mlggd_comm_init(h_, unique_id, world, rank), "mlggd_comm_init");
}
