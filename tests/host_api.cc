// host_api.cc -- C entry points over trainer_io for the CPU test-suite (ctypes): lets
// tests/ drive the real host code (CLI parse, pfile reader, chunk planner, chunk reader,
// .wts writer) without a GPU and compare it with an independent NumPy restatement.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dp_launch.h"
#include "trainer_io.h"

using namespace mlggd_host;

static thread_local std::string g_err;

extern "C" {

const char *mlggd_host_last_error() { return g_err.c_str(); }

void *mlggd_host_open(int argc, char **argv) {
    Interface *io = new Interface;
    try {
        io->Initial(argc, argv, false);
        io->get_pfile_info();
        return io;
    } catch (const std::exception &e) {
        g_err = e.what();
        delete io;
        return nullptr;
    }
}

void mlggd_host_close(void *h) { delete (Interface *)h; }

int mlggd_host_info(void *h, unsigned *sents, unsigned *frames, int *table, int cap) {
    Interface *io = (Interface *)h;
    *sents = io->total_sents;
    *frames = io->total_frames;
    for (unsigned i = 0; i < io->total_sents && (int)i < cap; i++) table[i] = io->framesBeforeSent[i];
    return io->numlayers;
}

int mlggd_host_norm(void *h, float *mean, float *inv_std, int cap) {
    Interface *io = (Interface *)h;
    const int n = (int)io->mean().size();
    for (int i = 0; i < n && i < cap; i++) {
        mean[i] = io->mean()[i];
        inv_std[i] = io->inv_std()[i];
    }
    return n;
}

// plans chunks for `range`; cv != 0 selects the CV planner.  Returns total_chunks (or -1).
int mlggd_host_plan(void *h, const char *range, int cv, int *starts, int cap, unsigned *total_samples) {
    Interface *io = (Interface *)h;
    try {
        if (cv) io->get_chunk_info_cv(range); else io->get_chunk_info(range);
        const ChunkPlan &pl = cv ? io->cv_plan : io->train_plan;
        for (unsigned i = 0; i < pl.total_chunks && (int)i < cap; i++) starts[i] = pl.frame_st[i];
        *total_samples = pl.total_samples;
        return (int)pl.total_chunks;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// reads chunk `index`; copies samples*K0 / samples*D floats out.  Returns samples (or -1).
int mlggd_host_read_chunk(void *h, int index, int cv, float *in, float *targ) {
    Interface *io = (Interface *)h;
    try {
        const int n = cv ? io->Readchunk_cv(index) : io->Readchunk(index);
        const WorkPara *p = io->para;
        memcpy(in, p->indata[0], (size_t)n * p->layersizes[0] * sizeof(float));
        memcpy(targ, p->targ[0], (size_t)n * p->layersizes[io->numlayers - 1] * sizeof(float));
        return n;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// frame-stream form of the same chunk: raw normalised frames + first frame per row.
// Returns samples (or -1); *nframes receives the number of frames copied.
int mlggd_host_read_chunk_frames(void *h, int index, int cv, float *feat, float *targ, int *first, int *nframes) {
    Interface *io = (Interface *)h;
    try {
        const int n = cv ? io->Readchunk_frames_cv(index) : io->Readchunk_frames(index);
        const WorkPara *p = io->para;
        const int fr = p->chunk_frames[0];
        memcpy(feat, p->frames_in[0], (size_t)fr * p->fea_dim * sizeof(float));
        memcpy(targ, p->frames_targ[0], (size_t)fr * p->layersizes[io->numlayers - 1] * sizeof(float));
        memcpy(first, p->first_frame[0], (size_t)n * sizeof(int));
        *nframes = fr;
        return n;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// One parsed command-line field of WorkPara as text (ints "%d", floats "%.9g": enough digits to round-trip a float),
// by the key=value name Interface::Initial knows it under (Interface.cc:150-315); "numlayers" is the layer count the
// parser DERIVED from layersizes= (the key numlayers= itself is not a key of the reference's parser).
// Returns the length, -1 for an unknown key.
int mlggd_host_para(void *h, const char *key, char *out, int cap) {
    Interface *io = (Interface *)h;
    const WorkPara *p = io->para;
    const std::string k(key);
    std::string v;
    char buf[64];
    auto I = [&](int x) { snprintf(buf, sizeof(buf), "%d", x); v = buf; };
    auto F = [&](float x) { snprintf(buf, sizeof(buf), "%.9g", (double)x); v = buf; };
    if (k == "fea_file") v = p->fea_FN;
    else if (k == "norm_file") v = p->fea_normFN;
    else if (k == "targ_file") v = p->targ_FN;
    else if (k == "outwts_file") v = p->out_weightFN;
    else if (k == "log_file") v = p->log_FN;
    else if (k == "initwts_file") v = p->init_weightFN;
    else if (k == "train_sent_range") v = p->train_sent_range;
    else if (k == "cv_sent_range") v = p->cv_sent_range;
    else if (k == "fea_dim") I(p->fea_dim);
    else if (k == "fea_context") I(p->fea_context);
    else if (k == "targ_offset") I(p->targ_offset);
    else if (k == "dropoutflag") I(p->dropoutflag);
    else if (k == "MLflag") I(p->MLflag);
    else if (k == "traincache") I(p->traincache);
    else if (k == "bunchsize") I(p->bunchsize);
    else if (k == "gpu_used") I(p->gpu_used);
    else if (k == "init_randem_seed") I(p->init_randem_seed);
    else if (k == "momentum") F(p->momentum);
    else if (k == "shapefactor") F(p->shapefactor);
    else if (k == "weightcost") F(p->weightcost);
    else if (k == "lrate") F(p->lrate);
    else if (k == "visible_omit") F(p->visible_omit);
    else if (k == "hid_omit") F(p->hid_omit);
    else if (k == "init_randem_weight_min") F(p->init_randem_weight_min);
    else if (k == "init_randem_weight_max") F(p->init_randem_weight_max);
    else if (k == "init_randem_bias_min") F(p->init_randem_bias_min);
    else if (k == "init_randem_bias_max") F(p->init_randem_bias_max);
    else if (k == "numlayers") I(io->numlayers);
    else if (k == "layersizes") {
        for (int i = 0; i < io->numlayers; i++) {
            snprintf(buf, sizeof(buf), i ? ",%d" : "%d", p->layersizes[i]);
            v += buf;
        }
    } else return -1;
    snprintf(out, cap, "%s", v.c_str());
    return (int)v.size();
}

void mlggd_host_shuffle(void *h, int *vec, int len) { ((Interface *)h)->GetRandIndex(vec, len); }

int mlggd_host_weights(void *h, int layer, float *w, float *b) {
    Interface *io = (Interface *)h;
    const WorkPara *p = io->para;
    if (layer < 1 || layer >= io->numlayers) return -1;
    memcpy(w, p->weights[layer], (size_t)p->layersizes[layer] * p->layersizes[layer - 1] * sizeof(float));
    memcpy(b, p->bias[layer], (size_t)p->layersizes[layer] * sizeof(float));
    return 0;
}

int mlggd_host_write_pfile(const char *path, const int *sent_lengths, int nsent, int num_features, const float *feat) {
    try {
        write_pfile(path, std::vector<int>(sent_lengths, sent_lengths + nsent), num_features, feat);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// data-parallel launch helpers (dp_launch.h)
int mlggd_host_rank_rows(int n_samples, int bunch, int world, int rank, int *rows, int cap) {
    const std::vector<int> r = rank_sample_rows(n_samples, bunch, world, rank);
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) rows[i] = r[i];
    return (int)r.size();
}

// 0 on success, -1 on error / timeout (message in mlggd_host_last_error)
int mlggd_host_rendezvous(const char *path, int world, int rank, unsigned char *id, double timeout_s) {
    try {
        rendezvous(path ? path : "", world, rank, id, timeout_s);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

void mlggd_host_rendezvous_cleanup(const char *path, int world) { rendezvous_cleanup(path, world); }

}  // extern "C"
