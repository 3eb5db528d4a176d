// trainer_io.h -- host-side IO of the trainer: the counterpart of the reference's
// `class Interface` / `struct WorkPara` (Train_code_ML_GGD/Interface.h:30-127), written from
// scratch.  Same public method names and meaning so bptrain_main.cc reads like BPtrain.cc:
//   Initial            key=value CLI, log header, norm file, initial .wts   (Interface.cc:133-482)
//   get_pfile_info     pfile header + sentence table of both pfiles        (Interface.cc:519-585)
//   get_chunk_info[_cv] chunk planner                                       (Interface.cc:588-716)
//   Readchunk[_cv]     read + byte-swap + z-norm + context-expand (+shuffle)(Interface.cc:719-965)
//   GetRandIndex       Fisher-Yates on lrand48                              (Interface.cc:975-986)
//   Writeweights       MATLAB level-4 .wts                                  (Interface.cc:484-516)
// Differences (documented in INTEGRATION.md): errors throw IoError (main prints the
// reference's message and exits 1 instead of 0); buffers are std::vector; the cond-var
// hand-off uses a predicate loop.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace mlggd_host {

constexpr int kMaxLayer = 10;            // Interface.h:6
constexpr int kPfileHeaderBytes = 32768; // Interface.cc:13

struct IoError : std::runtime_error {
    explicit IoError(const std::string &m) : std::runtime_error(m) {}
};

// struct WorkPara, Interface.h:30-69
struct WorkPara {
    std::string fea_FN, fea_normFN, targ_FN, init_weightFN, out_weightFN, log_FN;
    std::string train_sent_range, cv_sent_range;
    int fea_dim = 0, fea_context = 0, targ_offset = 0, dropoutflag = 0, traincache = 0, bunchsize = 0;
    int layersizes[kMaxLayer] = {0};
    float momentum = 0, shapefactor = 0, weightcost = 0, lrate = 0, visible_omit = 0, hid_omit = 0;
    int MLflag = 0, gpu_used = 0, init_randem_seed = 0;
    float init_randem_weight_min = -0.1f, init_randem_weight_max = 0.1f;
    float init_randem_bias_min = -0.1f, init_randem_bias_max = 0.1f;
    // frame-stream chunk (device-side input pipeline): raw normalised frames + first frame of every row
    float *frames_in[2] = {nullptr, nullptr};    // [chunk frames][fea_dim]
    float *frames_targ[2] = {nullptr, nullptr};  // [chunk frames][layersizes[L-1]]
    int *first_frame[2] = {nullptr, nullptr};    // [samples]
    int chunk_frames[2] = {0, 0};
    float *indata[2] = {nullptr, nullptr};  // double buffer, [traincache][layersizes[0]]
    float *targ[2] = {nullptr, nullptr};    //                [traincache][layersizes[L-1]]
    float *weights[kMaxLayer] = {nullptr};  // index 1..L-1, row-major [in][out]
    float *bias[kMaxLayer] = {nullptr};
};

struct ChunkPlan {
    int sent_st = 0, sent_en = 0;
    std::vector<int> frame_st;  // start frame of every chunk
    unsigned total_chunks = 0, total_samples = 0;
};

class Interface {
  public:
    Interface();
    ~Interface();
    void Initial(int argc, char **argv);
    // open_output=false: parse + load only, never touches outwts_file / log_file (tests, tools)
    void Initial(int argc, char **argv, bool open_output);
    void Writeweights();
    void get_pfile_info();
    void get_chunk_info(const std::string &range);
    void get_chunk_info_cv(const std::string &range);
    int Readchunk(int index);     // fills para->indata[0] / targ[0]; returns samples
    int Readchunk_cv(int index);
    // The same chunk WITHOUT host-side context expansion: fills para->frames_in[0] / frames_targ[0]
    // (normalised frames of the chunk) and para->first_frame[0][row] = first frame of the sample
    // that Readchunk would have expanded into that row (same lrand48 draw).  Returns samples.
    int Readchunk_frames(int index);
    int Readchunk_frames_cv(int index);
    // expanded buffers are only allocated on demand (Readchunk/Readchunk_cv)
    void want_expanded_buffers();
    // call after handing slot 0 of the frame-stream buffers to the consumer (pointer swap 0<->1)
    void frames_swapped() { fr_fill_ ^= 1; }
    // Sizes both sets of frame-stream chunk buffers for the LARGEST chunk of `plan`, so that no buffer is
    // re-allocated while chunks of that plan are being trained (re-allocating page-locked memory synchronises
    // the device).  Call after get_chunk_info[_cv], before the first Readchunk_frames[_cv] of the plan.
    void reserve_frame_buffers(const ChunkPlan &plan);
    // allocator of the frame-stream chunk buffers (default malloc/free); set before the first Readchunk_frames
    void set_buffer_allocator(void *(*alloc)(size_t), void (*release)(void *)) {
        buf_alloc_ = alloc;
        buf_free_ = release;
    }
    void GetRandIndex(int *vec, int len);
    void logf(const char *fmt, ...);

    WorkPara *para;
    unsigned total_frames = 0, total_sents = 0;
    unsigned total_chunks = 0, total_samples = 0, cv_total_chunks = 0, cv_total_samples = 0;
    std::vector<int> framesBeforeSent;  // cumulative END frame of sentence i
    ChunkPlan train_plan, cv_plan;
    std::vector<int> chunk_index;
    FILE *fp_log = nullptr;
    int numlayers = 0;
    int cur_chunk_samples = 0;

    const std::vector<float> &mean() const { return mean_; }
    const std::vector<float> &inv_std() const { return dVar_; }

  private:
    void parse_args(int argc, char **argv);
    void write_log_header();
    void load_norm();
    void load_init_weights();
    ChunkPlan plan_chunks(const std::string &range, const char *what);
    int read_chunk(const ChunkPlan &plan, int index, bool shuffle, bool expand);
    static unsigned header_uint(const std::string &hdr, const char *key, FILE *log);
    void read_sentence_table(FILE *fp, long offset, unsigned nsent, std::vector<int> &out);

    FILE *fp_data = nullptr, *fp_targ = nullptr, *fp_out = nullptr;
    std::vector<float> mean_, dVar_;
    std::vector<float> buf_in_[2], buf_targ_[2];
    // frame-stream chunk buffers: grow-only, allocated through a pluggable allocator so that the trainer can
    // hand out page-locked memory (pageable memory made the per-chunk upload the slowest host step)
    struct HostBuf {
        void *p = nullptr;
        size_t bytes = 0;
    };
    HostBuf fr_in_[2], fr_targ_[2], fr_first_[2];
    void *(*buf_alloc_)(size_t) = nullptr;
    void (*buf_free_)(void *) = nullptr;
    void *ensure(HostBuf &b, size_t bytes);
    struct Mapping {
        const unsigned char *base = nullptr;
        size_t size = 0;
        bool tried = false;
    } map_data_, map_targ_;
    const unsigned char *map_file(FILE *fp);
    size_t map_size(FILE *fp);
    bool expanded_ready_ = false;
    int fr_fill_ = 0;  // which frame-stream buffer the next Readchunk_frames fills
    std::vector<unsigned char> raw_;  // fread staging, kept between chunks
    int io_threads_ = 1;              // threads of the byte-swap + normalise loop (MLGGD_IO_THREADS, default: usable CPUs, max 4)
    std::vector<std::vector<float>> w_, b_;
};

// pfile writer used by the synthetic-data tools and tests (format: SURVEY.md 8c)
void write_pfile(const std::string &path, const std::vector<int> &sent_lengths, int num_features,
                 const float *features /* [sum(len)][num_features] */);

}  // namespace mlggd_host
