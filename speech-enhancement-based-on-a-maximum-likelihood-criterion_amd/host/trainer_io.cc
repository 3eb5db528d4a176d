// trainer_io.cc -- see trainer_io.h.  Behaviour follows Train_code_ML_GGD/Interface.cc
// (cited per function); the code is new.
#include "trainer_io.h"

#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>

namespace mlggd_host {

namespace {

std::string format(const char *fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return buf;
}

inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }  // swap32(), Interface.cc:81-97

inline float be_float(const unsigned char *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    v = bswap32(v);
    float f;
    memcpy(&f, &v, 4);
    return f;
}
inline int32_t be_int(const unsigned char *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return (int32_t)bswap32(v);
}

void parse_range(const std::string &range, int &st, int &en, bool &ok) {
    const size_t dash = range.find('-');
    ok = dash != std::string::npos;
    if (!ok) return;
    st = atoi(range.substr(0, dash).c_str());
    en = atoi(range.substr(dash + 1).c_str());
}

}  // namespace

// CPUs this process may really use: affinity mask and cgroup quota (a container often shows every host CPU)
static int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0 && c < n) n = c;
    }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const int c = (int)((atof(quota) / (double)period) + 0.5);
            if (c >= 1 && c < n) n = c;
        }
        fclose(f);
    }
    return n;
}

Interface::Interface() : para(new WorkPara) {
    // a handful of threads is enough (the loop is memory-bound), and idle OpenMP workers spin for a while
    // after every parallel region, next to the thread that feeds the GPU (OMP_WAIT_POLICY=passive in the
    // environment avoids that; it is read when libgomp loads, so it cannot be set from here)
    io_threads_ = usable_cpus();
    if (io_threads_ > 4) io_threads_ = 4;
    if (const char *v = getenv("MLGGD_IO_THREADS")) io_threads_ = atoi(v) > 0 ? atoi(v) : 1;
}

// read-only mapping of a pfile (nullptr if it cannot be mapped); MLGGD_MMAP=0 forces the fread path
const unsigned char *Interface::map_file(FILE *fp) {
    Mapping &m = fp == fp_data ? map_data_ : map_targ_;
    if (!m.tried) {
        m.tried = true;
        const char *v = getenv("MLGGD_MMAP");
        struct stat st;
        if (!(v && atoi(v) == 0) && fstat(fileno(fp), &st) == 0 && st.st_size > 0) {
            void *q = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (q != MAP_FAILED) {
                m.base = static_cast<const unsigned char *>(q);
                m.size = (size_t)st.st_size;
            }
        }
    }
    return m.base;
}
size_t Interface::map_size(FILE *fp) { return (fp == fp_data ? map_data_ : map_targ_).size; }

void *Interface::ensure(HostBuf &b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return b.p;
    if (b.p) (buf_free_ ? buf_free_ : free)(b.p);
    const size_t want = bytes + bytes / 16 + 4096;  // chunks differ a little in length: avoid regrowing every time
    b.p = buf_alloc_ ? buf_alloc_(want) : malloc(want);
    if (!b.p) throw IoError("out of host memory for a chunk buffer");
    b.bytes = want;
    return b.p;
}

void Interface::reserve_frame_buffers(const ChunkPlan &plan) {
    const WorkPara &p = *para;
    int max_frames = 0;
    for (unsigned i = 0; i < plan.total_chunks; i++) {
        const bool last = i + 1 == plan.total_chunks;
        const int frames = (last ? framesBeforeSent[plan.sent_en] : plan.frame_st[i + 1]) - plan.frame_st[i];
        if (frames > max_frames) max_frames = frames;
    }
    const size_t samples = plan.total_chunks > 1 ? (size_t)p.traincache : (size_t)plan.total_samples;
    for (int s = 0; s < 2; s++) {
        ensure(fr_in_[s], (size_t)max_frames * p.fea_dim * sizeof(float));
        ensure(fr_targ_[s], (size_t)max_frames * p.layersizes[numlayers - 1] * sizeof(float));
        ensure(fr_first_[s], (samples > 0 ? samples : 1) * sizeof(int));
    }
}

Interface::~Interface() {
    for (Mapping *m : {&map_data_, &map_targ_})
        if (m->base) munmap(const_cast<unsigned char *>(m->base), m->size);
    for (HostBuf *set : {fr_in_, fr_targ_, fr_first_})
        for (int i = 0; i < 2; i++)
            if (set[i].p) (buf_free_ ? buf_free_ : free)(set[i].p);
    if (fp_data) fclose(fp_data);
    if (fp_targ) fclose(fp_targ);
    if (fp_log) fclose(fp_log);
    if (fp_out) fclose(fp_out);
    delete para;
}

void Interface::logf(const char *fmt, ...) {
    if (!fp_log) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(fp_log, fmt, ap);
    va_end(ap);
}

// key=value parser, Interface.cc:150-315.  Unknown keys are ignored (finetune.pl passes
// numlayers=, which the reference never reads: the layer count comes from layersizes).
void Interface::parse_args(int argc, char **argv) {
    WorkPara &p = *para;
    struct StrKey { const char *k; std::string *v; };
    struct IntKey { const char *k; int *v; };
    struct FltKey { const char *k; float *v; };
    const StrKey skeys[] = {{"fea_file", &p.fea_FN}, {"norm_file", &p.fea_normFN}, {"targ_file", &p.targ_FN},
                            {"outwts_file", &p.out_weightFN}, {"log_file", &p.log_FN},
                            {"initwts_file", &p.init_weightFN}, {"train_sent_range", &p.train_sent_range},
                            {"cv_sent_range", &p.cv_sent_range}};
    const IntKey ikeys[] = {{"fea_dim", &p.fea_dim}, {"fea_context", &p.fea_context}, {"targ_offset", &p.targ_offset},
                            {"dropoutflag", &p.dropoutflag}, {"MLflag", &p.MLflag}, {"traincache", &p.traincache},
                            {"bunchsize", &p.bunchsize}, {"gpu_used", &p.gpu_used},
                            {"init_randem_seed", &p.init_randem_seed}};
    const FltKey fkeys[] = {{"momentum", &p.momentum}, {"shapefactor", &p.shapefactor}, {"weightcost", &p.weightcost},
                            {"lrate", &p.lrate}, {"visible_omit", &p.visible_omit}, {"hid_omit", &p.hid_omit},
                            {"init_randem_weight_min", &p.init_randem_weight_min},
                            {"init_randem_weight_max", &p.init_randem_weight_max},
                            {"init_randem_bias_max", &p.init_randem_bias_max},
                            {"init_randem_bias_min", &p.init_randem_bias_min}};
    for (int a = 1; a < argc; a++) {
        const std::string arg(argv[a]);
        const size_t eq = arg.find('=');
        if (eq == std::string::npos) throw IoError(format("Arg: %s  Format Error", argv[a]));  // Interface.cc:153-157
        const std::string key = arg.substr(0, eq), val = arg.substr(eq + 1);
        bool done = false;
        for (const auto &k : skeys)
            if (key == k.k) { *k.v = val; done = true; }
        for (const auto &k : ikeys)
            if (key == k.k) { *k.v = atoi(val.c_str()); done = true; }
        for (const auto &k : fkeys)
            if (key == k.k) { *k.v = (float)atof(val.c_str()); done = true; }
        if (!done && key == "layersizes") {  // Interface.cc:298-313
            int count = 0;
            size_t pos = 0;
            while (true) {
                const size_t comma = val.find(',', pos);
                if (count >= kMaxLayer) throw IoError(format("layersizes has more than %d layers", kMaxLayer));
                p.layersizes[count++] = atoi(val.substr(pos, comma == std::string::npos ? comma : comma - pos).c_str());
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
            numlayers = count;
        }
    }
}

// log header, Interface.cc:338-371 (the reference prints the two float omit rates with %d,
// which is undefined; they are printed with %f here)
void Interface::write_log_header() {
    const WorkPara &p = *para;
    logf("parameters input:\n");
    logf("fea_file:             %s\n", p.fea_FN.c_str());
    logf("norm_file:            %s\n", p.fea_normFN.c_str());
    logf("targ_file:            %s\n", p.targ_FN.c_str());
    logf("outwts_file:          %s\n", p.out_weightFN.c_str());
    logf("log_file:\t\t          %s\n", p.log_FN.c_str());
    logf("initwts_file:         %s\n", p.init_weightFN.c_str());
    logf("train_sent_range:     %s\n", p.train_sent_range.c_str());
    logf("cv_sent_range:        %s\n", p.cv_sent_range.c_str());
    logf("fea_dim:\t\t          %d\n", p.fea_dim);
    logf("fea_context:\t\t      %d\n", p.fea_context);
    logf("bunchsize:\t\t        %d\n", p.bunchsize);
    logf("gpu_used:\t\t          %d\n", p.gpu_used);
    logf("train_cache:\t\t      %d\n", p.traincache);
    logf("init_randem_seed:\t\t  %d\n", p.init_randem_seed);
    logf("targ_offset:\t\t      %d\n", p.targ_offset);
    logf("dropoutflag:\t\t      %d\n", p.dropoutflag);
    logf("MLflag:\t\t      %d\n", p.MLflag);
    logf("init_randem_weight_max:\t\t  %f\n", p.init_randem_weight_max);
    logf("init_randem_weight_min:\t\t  %f\n", p.init_randem_weight_min);
    logf("init_randem_bias_max:\t\t    %f\n", p.init_randem_bias_max);
    logf("init_randem_bias_min:\t\t    %f\n", p.init_randem_bias_min);
    logf("momentum:\t\t                %f\n", p.momentum);
    logf("shapefactor:\t\t                %f\n", p.shapefactor);
    logf("weightcost:\t\t              %f\n", p.weightcost);
    logf("learnrate:\t\t              %f\n", p.lrate);
    logf("visible_omit:\t\t      %f\n", p.visible_omit);
    logf("hid_omit:\t\t      %f\n", p.hid_omit);
    logf("layersizes:\t\t              ");
    for (int j = 0; j < numlayers; j++) logf("%d,", p.layersizes[j]);
    logf("\n");
    logf("Please check...\n");
}

// norm file, Interface.cc:373-399: a "vec N" line, N means, a "vec N" line, N inverse std-devs
void Interface::load_norm() {
    FILE *fp = fopen(para->fea_normFN.c_str(), "rt");
    if (!fp) throw IoError(format("can not open normalization file: %s", para->fea_normFN.c_str()));
    logf("Loading Norm file...\n");
    const int n = para->fea_dim;
    mean_.assign(n, 0.f);
    dVar_.assign(n, 0.f);
    char line[1024];
    auto next = [&]() -> const char * {
        if (!fgets(line, sizeof(line), fp)) line[0] = '\0';
        return line;
    };
    next();
    for (int j = 0; j < n; j++) mean_[j] = (float)atof(next());
    next();
    for (int j = 0; j < n; j++) dVar_[j] = (float)atof(next());
    fclose(fp);
    logf("Norm file loaded.\n");
}

// initial weights, Interface.cc:401-468: MATLAB level-4 matrices {type,mrows,ncols,imagf,namelen},
// name, float32 data; weights (mrows=out, ncols=in) then bias (1 x out) per layer
void Interface::load_init_weights() {
    WorkPara &p = *para;
    w_.assign(numlayers, {});
    b_.assign(numlayers, {});
    for (int i = 1; i < numlayers; i++) {
        w_[i].assign((size_t)p.layersizes[i] * p.layersizes[i - 1], 0.f);
        b_[i].assign(p.layersizes[i], 0.f);
        p.weights[i] = w_[i].data();
        p.bias[i] = b_[i].data();
    }
    srand48(p.init_randem_seed);  // Interface.cc:411: once, for weights and data index
    if (p.init_weightFN.empty())  // sigmoid build: Interface.cc:424-426
        throw IoError("fatal_error, please set initial weights file");
    FILE *fp = fopen(p.init_weightFN.c_str(), "rb");
    if (!fp) throw IoError(format("can not open initial weights file: %s", p.init_weightFN.c_str()));
    logf("Loading Init weight file...\n");
    auto read_header = [&](int32_t stat[5]) {
        char name[256];
        if (fread(stat, sizeof(int32_t), 5, fp) != 5 || stat[4] < 0 || stat[4] > 255 ||
            fread(name, 1, stat[4], fp) != (size_t)stat[4]) {
            fclose(fp);
            throw IoError("init weights file is truncated");
        }
    };
    for (int i = 1; i < numlayers; i++) {
        int32_t stat[5];
        read_header(stat);
        if (stat[1] != p.layersizes[i] || stat[2] != p.layersizes[i - 1]) {
            fclose(fp);
            logf("%d,%d,%d,%d\n", stat[1], stat[2], p.layersizes[i], p.layersizes[i - 1]);
            throw IoError("init weights node nums do not match");
        }
        if (fread(w_[i].data(), sizeof(float), w_[i].size(), fp) != w_[i].size()) {
            fclose(fp);
            throw IoError("init weights file is truncated");
        }
        read_header(stat);
        if (stat[2] != p.layersizes[i] || stat[1] != 1) {
            fclose(fp);
            throw IoError("init bias node nums do not match");
        }
        if (fread(b_[i].data(), sizeof(float), b_[i].size(), fp) != b_[i].size()) {
            fclose(fp);
            throw IoError("init weights file is truncated");
        }
    }
    fclose(fp);
    logf("Init weight file loaded.\n");
}

void Interface::Initial(int argc, char **argv) { Initial(argc, argv, true); }

// Interface::Initial, Interface.cc:133-482
void Interface::Initial(int argc, char **argv, bool open_output) {
    parse_args(argc, argv);
    WorkPara &p = *para;
    if (open_output) {
        if (!(fp_log = fopen(p.log_FN.c_str(), "wt")))
            throw IoError(format("can not open output log file: %s", p.log_FN.c_str()));
    }
    if (!(fp_data = fopen(p.fea_FN.c_str(), "rb"))) throw IoError(format("can not open feature file: %s", p.fea_FN.c_str()));
    if (!(fp_targ = fopen(p.targ_FN.c_str(), "rb"))) throw IoError(format("can not open target file: %s", p.targ_FN.c_str()));
    if (open_output) {
        // opened (and truncated) up front like the reference, Interface.cc:332
        if (!(fp_out = fopen(p.out_weightFN.c_str(), "wb")))
            throw IoError(format("can not open output weights file: %s", p.out_weightFN.c_str()));
    }
    if (numlayers < 2) throw IoError("layersizes must name at least two layers");
    if (p.fea_dim < 1 || p.fea_context < 1 || p.traincache < 1 || p.bunchsize < 1)
        throw IoError("fea_dim, fea_context, traincache and bunchsize must be positive");
    write_log_header();
    load_norm();
    load_init_weights();
    if (p.fea_dim * p.fea_context != p.layersizes[0])  // Interface.cc:471-475
        throw IoError("feadim times context must be equal to layersizes[0]");
    // Interface.cc:476-480 allocates two [traincache][layersizes[0]] matrices up front; here they are
    // allocated on first use of the expanding reader (the frame-stream reader does not need them)
    if (fp_log) fflush(fp_log);
}

void Interface::want_expanded_buffers() {
    if (expanded_ready_) return;
    WorkPara &p = *para;
    for (int i = 0; i < 2; i++) {
        buf_in_[i].assign((size_t)p.layersizes[0] * p.traincache, 0.f);
        buf_targ_[i].assign((size_t)p.layersizes[numlayers - 1] * p.traincache, 0.f);
        p.indata[i] = buf_in_[i].data();
        p.targ[i] = buf_targ_[i].data();
    }
    expanded_ready_ = true;
}

// Interface::Writeweights, Interface.cc:484-516
void Interface::Writeweights() {
    if (!fp_out) throw IoError("output weights file is not open");
    logf("Saving weights to file...\n");
    const WorkPara &p = *para;
    auto put = [&](const std::string &name, int mrows, int ncols, const float *data) {
        const int32_t stat[5] = {10, mrows, ncols, 0, (int32_t)name.size() + 1};
        fwrite(stat, sizeof(int32_t), 5, fp_out);
        fwrite(name.c_str(), 1, name.size() + 1, fp_out);
        fwrite(data, sizeof(float), (size_t)mrows * ncols, fp_out);
    };
    for (int i = 1; i < numlayers; i++) {
        put(format("weights%d%d", i, i + 1), p.layersizes[i], p.layersizes[i - 1], p.weights[i]);
        put(format("bias%d", i + 1), 1, p.layersizes[i], p.bias[i]);
    }
    fflush(fp_out);
    logf("Saving over.\n");
}

// Interface::get_uint, Interface.cc:988-1009
unsigned Interface::header_uint(const std::string &hdr, const char *key, FILE *) {
    const size_t pos = hdr.find(key);
    if (pos == std::string::npos) throw IoError("pfile header format is Not correct.");
    unsigned val = 0;
    int count = 0;
    sscanf(hdr.c_str() + pos + strlen(key), " %u%n", &val, &count);
    if (count <= 1) throw IoError(format("%s num in pfile header is Not correct.", key));
    return val;
}

// Interface::read_tail, Interface.cc:1011-1024: the sentence table holds S+1 big-endian
// int32 frame offsets; entry 0 (always 0) is skipped, so out[i] = END frame of sentence i
void Interface::read_sentence_table(FILE *fp, long offset, unsigned nsent, std::vector<int> &out) {
    std::vector<unsigned char> raw((size_t)nsent * 4);
    if (fseek(fp, offset + 4, SEEK_SET) != 0 || fread(raw.data(), 4, nsent, fp) != nsent)
        throw IoError("pfile tail is Not correct.");
    out.resize(nsent);
    for (unsigned i = 0; i < nsent; i++) out[i] = be_int(&raw[(size_t)i * 4]);
}

// Interface::get_pfile_info, Interface.cc:519-585
void Interface::get_pfile_info() {
    std::string header(kPfileHeaderBytes, '\0');
    logf("begin to read in_pfile\n");
    fseek(fp_data, 0, SEEK_SET);
    if (fread(&header[0], kPfileHeaderBytes, 1, fp_data) != 1) throw IoError("Failed to read data pfile header.");
    header.resize(strnlen(header.c_str(), kPfileHeaderBytes));
    total_sents = header_uint(header, "-num_sentences", fp_log);
    total_frames = header_uint(header, "-num_frames", fp_log);
    long row_bytes = sizeof(float) * (2 + para->fea_dim);
    read_sentence_table(fp_data, (long)total_frames * row_bytes + kPfileHeaderBytes, total_sents, framesBeforeSent);

    logf("begin to read target_pfile\n");
    std::string theader(kPfileHeaderBytes, '\0');
    fseek(fp_targ, 0, SEEK_SET);
    if (fread(&theader[0], kPfileHeaderBytes, 1, fp_targ) != 1) throw IoError("Failed to read target pfile header.");
    theader.resize(strnlen(theader.c_str(), kPfileHeaderBytes));
    const unsigned tsents = header_uint(theader, "-num_sentences", fp_log);
    const unsigned tframes = header_uint(theader, "-num_frames", fp_log);
    row_bytes = sizeof(float) * (2 + para->layersizes[numlayers - 1]);
    std::vector<int> ttable;
    read_sentence_table(fp_targ, (long)tframes * row_bytes + kPfileHeaderBytes, tsents, ttable);
    logf("tmpsentnum=%d,tmpframenum=%d,total_frames=%d\n", tsents, tframes, total_frames);
    if (tsents != total_sents || tframes != total_frames)
        throw IoError("frames or sentence num in target pfile and data pfile is not consistent.");
    logf("frames or sentence num in target pfile and data pfile is consistent.\n");
    for (unsigned i = 0; i < total_sents; i++)
        if (ttable[i] != framesBeforeSent[i])
            throw IoError(format("tails in target pfile and data pfile is not consistent---%d.", i));
    logf("Get pfile info over: Training data has %u frames, %u sentences.\n", total_frames, total_sents);
}

// Chunk planner, Interface.cc:588-650 (train) == :653-716 (cv).  A sentence of L frames
// yields L-(ctx-1) samples (0 if L < ctx).  Whenever the running sample count reaches
// traincache the chunk is cut: the next chunk starts at frame
//   next_st = sentence_end - (count - traincache)
// and inherits the samples of the current sentence that start at or after next_st and still
// fit, i.e. max(0, sentence_end - next_st - (ctx-1)).
ChunkPlan Interface::plan_chunks(const std::string &range, const char *what) {
    ChunkPlan plan;
    bool ok;
    parse_range(range, plan.sent_st, plan.sent_en, ok);
    if (!ok) throw IoError(format("%ssent range: %s format error.", what, range.c_str()));
    if (plan.sent_en < plan.sent_st || plan.sent_st < 0 || plan.sent_en >= (int)total_sents)
        throw IoError(format("%ssent range: %d to %d number error.", what, plan.sent_st, plan.sent_en));
    const int ctx = para->fea_context, cache = para->traincache;
    int frame = plan.sent_st == 0 ? 0 : framesBeforeSent[plan.sent_st - 1];
    plan.frame_st.push_back(frame);
    int count = 0;
    for (int s = plan.sent_st; s <= plan.sent_en; s++) {
        const int len = framesBeforeSent[s] - frame;
        frame = framesBeforeSent[s];
        count += (len >= ctx) ? len - (ctx - 1) : 0;
        while (count >= cache) {
            const int next_st = frame - (count - cache);
            if (next_st >= (int)total_frames) throw IoError("chunk planner ran past the end of the pfile");
            plan.frame_st.push_back(next_st);
            count = (frame - next_st > ctx - 1) ? frame - next_st - ctx + 1 : 0;
        }
    }
    plan.total_chunks = (unsigned)plan.frame_st.size();
    plan.total_samples = (plan.total_chunks - 1) * cache + count;
    return plan;
}

void Interface::get_chunk_info(const std::string &range) {
    train_plan = plan_chunks(range, "");
    total_chunks = train_plan.total_chunks;
    total_samples = train_plan.total_samples;
    logf("Get chunk info over: Training sentences have %d chunks, %d samples.\n", total_chunks, total_samples);
}

void Interface::get_chunk_info_cv(const std::string &range) {
    cv_plan = plan_chunks(range, "cv ");
    cv_total_chunks = cv_plan.total_chunks;
    cv_total_samples = cv_plan.total_samples;
    logf("Get cv chunk info over: CV sentences have %d chunks, %d samples.\n", cv_total_chunks, cv_total_samples);
}

// Interface::GetRandIndex, Interface.cc:975-986
void Interface::GetRandIndex(int *vec, int len) {
    for (int i = 0; i < len - 1; i++) {
        const int idx = (int)(lrand48() % (len - i));
        const int tmp = vec[idx];
        vec[idx] = vec[len - 1 - i];
        vec[len - 1 - i] = tmp;
    }
}

// Interface::Readchunk / Readchunk_cv, Interface.cc:719-838 / :841-965.
// Reads the chunk's frames of both pfiles (rows are {int32 sent, int32 frame, float x dim},
// big-endian), normalises features AND targets with the feature mean / inverse std-dev
// (Interface.cc:763-764, 807-808), and writes sample s (a window of fea_context frames that
// lies inside one sentence and inside the chunk) to row order[s] of indata[0], its target
// (the frame targ_offset into the window) to row order[s] of targ[0].
int Interface::read_chunk(const ChunkPlan &plan, int index, bool shuffle, bool expand) {
    const WorkPara &p = *para;
    const int dim = p.fea_dim, ctx = p.fea_context, K0 = p.layersizes[0], D = p.layersizes[numlayers - 1];
    const int st = plan.frame_st[index];
    const bool last = index == (int)plan.total_chunks - 1;
    const int frames = (last ? framesBeforeSent[plan.sent_en] : plan.frame_st[index + 1]) - st;
    const int samples = last ? (int)plan.total_samples - p.traincache * index : p.traincache;

    std::vector<int> order(samples);
    for (int i = 0; i < samples; i++) order[i] = i;
    if (shuffle) GetRandIndex(order.data(), samples);  // Interface.cc:754 (train only)

    // fread the chunk's rows, then byte-swap + z-normalise them (Interface.cc:760-776: every value is
    // (x - mean[j]) * inv_std[j] with the NOISY statistics, targets included).  The conversion is what the
    // epoch of the executable was bound by once the device ran at 855 k frames/s, so it is spread over the
    // host cores; each frame is converted by exactly one thread with the same two fp32 operations, so the
    // result does not depend on the thread count.  The staging buffer is kept between chunks (a fresh
    // 100 MB vector per chunk costs more in page faults than the conversion itself).
    auto load = [&](FILE *fp, int ncol, auto ensure_out, int &first_sent) {
        const size_t row_bytes = (size_t)(ncol + 2) * 4;
        const size_t off = (size_t)kPfileHeaderBytes + (size_t)st * row_bytes;
        // Preferred source: the file mapped read-only -- the conversion threads then pull the rows straight out
        // of the page cache in parallel (one thread's fread of ~100 MB per file and chunk was what the epoch of
        // the executable waited for).  fread into a staging buffer (kept between chunks) if it cannot be mapped.
        const unsigned char *mapped = map_file(fp);
        const unsigned char *src0;
        if (mapped && off + row_bytes * frames <= map_size(fp)) {
            src0 = mapped + off;
        } else {
            if (raw_.size() < row_bytes * frames) raw_.resize(row_bytes * frames);
            if (fseek(fp, (long)off, SEEK_SET) != 0) throw IoError(format("pfile cannot fseek to chunk %d.", index));
            if (fread(raw_.data(), row_bytes, frames, fp) != (size_t)frames)
                throw IoError(format("pfile is too short for chunk %d.", index));
            src0 = raw_.data();
        }
        first_sent = frames > 0 ? be_int(src0) : 0;
        const unsigned char *rawp = src0;
        float *out = ensure_out((size_t)frames * ncol);
        const float *mean = mean_.data(), *istd = dVar_.data();
#pragma omp parallel for schedule(static) num_threads(io_threads_)
        for (int f = 0; f < frames; f++) {
            const unsigned char *src = rawp + f * row_bytes + 8;
            float *dst = out + (size_t)f * ncol;
            for (int j0 = 0; j0 < ncol; j0 += dim) {  // ncol is a multiple of dim for every pfile the trainer reads
                const int jn = ncol - j0 < dim ? ncol - j0 : dim;
                for (int j = 0; j < jn; j++) {
                    float v = be_float(src + 4 * (j0 + j));
                    v -= mean[j];
                    v *= istd[j];
                    dst[j0 + j] = v;
                }
            }
        }
    };
    // walk the sentences inside the chunk; calls emit(sample_index, first_frame_in_chunk)
    auto walk = [&](int first_sent, auto emit) {
        int done = 0, sent = first_sent, frame_id = st, sample = 0;
        while (done != frames) {
            const int in_sent = (framesBeforeSent[sent] > frames + st) ? frames - done : framesBeforeSent[sent] - frame_id;
            for (int j = 0; j <= in_sent - ctx; j++) emit(sample++, done + j);
            frame_id = framesBeforeSent[sent];
            sent++;
            done += in_sent;
        }
        return sample;
    };

    int sent0 = 0;
    if (expand) {
        want_expanded_buffers();
        std::vector<float> feat, targ;
        auto into = [](std::vector<float> &v) {
            return [&v](size_t n) {
                v.resize(n);
                return v.data();
            };
        };
        load(fp_data, dim, into(feat), sent0);
        float *in0 = p.indata[0];
        walk(sent0, [&](int s, int f) {
            if (s < samples)
                memcpy(in0 + (size_t)order[s] * K0, feat.data() + (size_t)f * dim, (size_t)ctx * dim * sizeof(float));
        });
        load(fp_targ, D, into(targ), sent0);
        float *tg0 = p.targ[0];
        walk(sent0, [&](int s, int f) {
            if (s < samples)
                memcpy(tg0 + (size_t)order[s] * D, targ.data() + (size_t)(f + p.targ_offset) * D, (size_t)D * sizeof(float));
        });
    } else {
        // frame-stream form: keep the normalised frames, record where each row's window starts
        const int fi = fr_fill_;
        auto into = [this](HostBuf &b) {
            return [this, &b](size_t n) { return static_cast<float *>(ensure(b, n * sizeof(float))); };
        };
        load(fp_data, dim, into(fr_in_[fi]), sent0);
        load(fp_targ, D, into(fr_targ_[fi]), sent0);
        int *first = static_cast<int *>(ensure(fr_first_[fi], (size_t)(samples > 0 ? samples : 1) * sizeof(int)));
        memset(first, 0, (size_t)samples * sizeof(int));
        walk(sent0, [&](int s, int f) {
            if (s < samples) first[order[s]] = f;
        });
        para->frames_in[0] = static_cast<float *>(fr_in_[fi].p);
        para->frames_targ[0] = static_cast<float *>(fr_targ_[fi].p);
        para->first_frame[0] = first;
        para->chunk_frames[0] = frames;
    }
    return samples;
}

int Interface::Readchunk(int index) { return read_chunk(train_plan, index, true, true); }
int Interface::Readchunk_cv(int index) { return read_chunk(cv_plan, index, false, true); }
int Interface::Readchunk_frames(int index) { return read_chunk(train_plan, index, true, false); }
int Interface::Readchunk_frames_cv(int index) { return read_chunk(cv_plan, index, false, false); }

// ---- pfile writer (for synthetic data; layout as parsed above and in SURVEY.md 8c)
void write_pfile(const std::string &path, const std::vector<int> &sent_lengths, int num_features, const float *features) {
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) throw IoError(format("cannot open %s for writing", path.c_str()));
    long nframes = 0;
    for (int l : sent_lengths) nframes += l;
    const int ncol = num_features + 2;
    std::string fmt = "dd";
    fmt.append(num_features, 'f');
    std::string hdr = format("-pfile_header version 0 size %d\n-num_sentences %d\n-num_frames %ld\n"
                             "-first_feature_column 2\n-num_features %d\n-first_label_column %d\n-num_labels 0\n",
                             kPfileHeaderBytes, (int)sent_lengths.size(), nframes, num_features, ncol);
    hdr += "-format " + fmt + "\n";
    hdr += format("-data size %ld offset 0 ndim 2 nrow %ld ncol %d\n", nframes * ncol, nframes, ncol);
    hdr += format("-sent_table_data size %d offset %ld ndim 1\n-end\n", (int)sent_lengths.size() + 1, nframes * ncol);
    std::vector<char> head(kPfileHeaderBytes, 0);
    memcpy(head.data(), hdr.data(), hdr.size());
    fwrite(head.data(), 1, head.size(), fp);
    std::vector<uint32_t> row(ncol);
    long f = 0;
    for (size_t s = 0; s < sent_lengths.size(); s++)
        for (int t = 0; t < sent_lengths[s]; t++, f++) {
            row[0] = bswap32((uint32_t)s);
            row[1] = bswap32((uint32_t)t);
            for (int j = 0; j < num_features; j++) {
                uint32_t v;
                memcpy(&v, &features[f * num_features + j], 4);
                row[2 + j] = bswap32(v);
            }
            fwrite(row.data(), 4, ncol, fp);
        }
    uint32_t acc = 0, be = 0;
    fwrite(&be, 4, 1, fp);
    for (int l : sent_lengths) {
        acc += (uint32_t)l;
        be = bswap32(acc);
        fwrite(&be, 4, 1, fp);
    }
    fclose(fp);
}

}  // namespace mlggd_host
