// host_sanitize.cc -- TEST INFRASTRUCTURE: the trainer's host side (trainer_io.cc: CLI, norm, .wts, pfile reader, chunk
// planner, chunk readers; prefetch.h: the reader thread and its two-slot hand-off) driven WITHOUT a GPU, so that it
// can run under AddressSanitizer + UBSan and under ThreadSanitizer (tests/test_host_sanitizers.py builds it both
// ways).  The reference's counterpart of this code has the races SURVEY.md section 5 lists (a condition-variable wait
// without a predicate loop, Interface.cc:25-30); this is the check that the rewrite has none.
//   host_sanitize <scratch dir>
// Writes a small synthetic corpus, then walks an epoch the way bptrain_main.cc does -- chunk plan, lrand48 shuffle,
// reader thread ahead of a "trainer" that reads every byte of every chunk -- in both chunk forms (frame stream /
// expanded), then the CV chunks and the weight file.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "prefetch.h"
#include "trainer_io.h"

using namespace mlggd_host;

static void write_wts(const std::string &path, const std::vector<int> &ls) {  // MATLAB level-4, Interface.cc:484-516
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) abort();
    for (size_t l = 1; l < ls.size(); l++) {
        for (int kind = 0; kind < 2; kind++) {
            char name[32];
            if (kind == 0) snprintf(name, sizeof(name), "weights%zu%zu", l, l + 1);
            else snprintf(name, sizeof(name), "bias%zu", l + 1);
            const int mrows = kind == 0 ? ls[l] : 1, ncols = kind == 0 ? ls[l - 1] : ls[l];
            const int hdr[5] = {10, mrows, ncols, 0, (int)strlen(name) + 1};
            fwrite(hdr, 4, 5, f);
            fwrite(name, 1, strlen(name) + 1, f);
            std::vector<float> v((size_t)mrows * ncols);
            for (size_t i = 0; i < v.size(); i++) v[i] = 0.01f * (float)((i * 7 + l) % 13) - 0.06f;
            fwrite(v.data(), 4, v.size(), f);
        }
    }
    fclose(f);
}

// slow_first_chunk_ms > 0: the consumer dwells on chunk 0, so a reader that fails on chunk 1 does so while the slot
// is still full (that error used to be lost and the trainer blocked for ever).  expect_error: the reader's
// message is returned through *reader_error instead of ending the process.
static double epoch(Interface *io, bool frames, int slow_first_chunk_ms = 0, std::string *reader_error = nullptr) {
    WorkPara *p = io->para;
    io->get_chunk_info(p->train_sent_range);
    io->chunk_index.resize(io->total_chunks);
    for (unsigned i = 0; i < io->total_chunks; i++) io->chunk_index[i] = (int)i;
    if (slow_first_chunk_ms == 0) io->GetRandIndex(io->chunk_index.data(), (int)io->total_chunks);  // else file order: chunk 0 is readable
    Slot slot;
    std::string err;
    std::atomic<bool> stop{false};
    std::thread fetch(fetch_loop, io, &slot, &err, frames, &stop, (const std::string *)nullptr);
    const int K0 = p->layersizes[0], D = p->layersizes[io->numlayers - 1];
    double sum = 0;
    for (unsigned i = 0; i < io->total_chunks; i++) {
        if (!slot.wait(true)) break;
        if (slow_first_chunk_ms > 0 && i == 0) std::this_thread::sleep_for(std::chrono::milliseconds(slow_first_chunk_ms));
        const int n = io->cur_chunk_samples;
        if (frames) {  // every frame, every first-frame index, and the window each index addresses
            const int nf = p->chunk_frames[1];
            for (size_t j = 0; j < (size_t)nf * p->fea_dim; j++) sum += p->frames_in[1][j];
            for (size_t j = 0; j < (size_t)nf * D; j++) sum += p->frames_targ[1][j];
            for (int s = 0; s < n; s++) {
                const int f0 = p->first_frame[1][s];
                if (f0 < 0 || f0 + p->fea_context > nf) { fprintf(stderr, "window outside the chunk\n"); exit(2); }
                sum += p->frames_in[1][(size_t)(f0 + p->fea_context - 1) * p->fea_dim + p->fea_dim - 1];
            }
        } else {
            for (size_t j = 0; j < (size_t)n * K0; j++) sum += p->indata[1][j];
            for (size_t j = 0; j < (size_t)n * D; j++) sum += p->targ[1][j];
        }
        slot.set(false);
    }
    fetch.join();
    if (reader_error) *reader_error = err;
    else if (!err.empty()) { fprintf(stderr, "reader: %s\n", err.c_str()); exit(2); }
    return sum;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_sanitize <scratch dir>\n"); return 2; }
    const std::string d = argv[1];
    const int dim = 9, ctx = 5;
    // 14 sentences, one shorter than the context window (contributes no sample), ragged lengths
    const std::vector<int> lens = {40, 33, 7, 61, 3, 50, 29, 45, 38, 11, 52, 31, 27, 44};
    int nfr = 0;
    for (int n : lens) nfr += n;
    std::vector<float> noisy((size_t)nfr * dim), clean((size_t)nfr * dim);
    for (size_t i = 0; i < noisy.size(); i++) {
        noisy[i] = 5.0f + 0.001f * (float)((i * 2654435761u) % 4001) - 2.0f;
        clean[i] = 4.0f + 0.001f * (float)((i * 40503u) % 3001) - 1.5f;
    }
    write_pfile(d + "/n.pfile", lens, dim, noisy.data());
    write_pfile(d + "/c.pfile", lens, dim, clean.data());
    {
        FILE *f = fopen((d + "/n.norm").c_str(), "w");
        if (!f) return 2;
        fprintf(f, "vec %d\n", dim);
        for (int i = 0; i < dim; i++) fprintf(f, "%.6f\n", 5.0 + 0.01 * i);
        fprintf(f, "vec %d\n", dim);
        for (int i = 0; i < dim; i++) fprintf(f, "%.6f\n", 0.9 + 0.01 * i);
        fclose(f);
    }
    const std::vector<int> ls = {dim * ctx, 16, 12, dim};
    write_wts(d + "/init.wts", ls);
    std::vector<std::string> a = {"BPtrain_Sigmoid", "gpu_used=0", "numlayers=4", "layersizes=45,16,12,9", "bunchsize=16",
                                  "MLflag=1", "shapefactor=1.2", "momentum=0.9", "weightcost=0.00001", "lrate=0.1",
                                  "fea_dim=9", "fea_context=5", "traincache=64", "init_randem_seed=27870775",
                                  "targ_offset=2", "initwts_file=" + d + "/init.wts", "norm_file=" + d + "/n.norm",
                                  "fea_file=" + d + "/n.pfile", "targ_file=" + d + "/c.pfile",
                                  "outwts_file=" + d + "/out.wts", "log_file=" + d + "/log.txt", "train_sent_range=0-10",
                                  "cv_sent_range=11-13", "dropoutflag=0", "visible_omit=0.1", "hid_omit=0.1"};
    double sums[2] = {0, 0};
    for (int form = 0; form < 2; form++) {  // 0: frame-stream chunks, 1: host-expanded chunks (the reference's form)
        std::vector<char *> av;
        std::vector<std::string> copy = a;  // Initial() may split the strings in place, as the reference does
        for (auto &s : copy) av.push_back(&s[0]);
        Interface *io = new Interface;
        try {
            io->Initial((int)av.size(), av.data(), /*open_output=*/true);
            io->get_pfile_info();
            sums[form] = epoch(io, form == 0);
            io->Writeweights();
            io->get_chunk_info_cv(io->para->cv_sent_range);
            for (unsigned i = 0; i < io->cv_total_chunks; i++) {
                if (form == 0) {
                    if (i == 0) io->reserve_frame_buffers(io->cv_plan);
                    const int n = io->Readchunk_frames_cv((int)i);
                    for (int s = 0; s < n; s++) sums[form] += io->para->first_frame[0][s];
                } else {
                    const int n = io->Readchunk_cv((int)i);
                    for (size_t j = 0; j < (size_t)n * ls[0]; j++) sums[form] += io->para->indata[0][j];
                }
            }
        } catch (const std::exception &e) {
            fprintf(stderr, "host_sanitize: %s\n", e.what());
            return 2;
        }
        delete io;
    }
    // A pfile cut off in the middle of its data (a crashed feacat, a full disk): the header still promises every
    // frame, so the planner lays out all chunks and the READER hits the end of the file on a later chunk -- while
    // the consumer is still busy with chunk 0.  The epoch must end with the reader's message, not hang.
    for (int form = 0; form < 2; form++) {
        FILE *f = fopen((d + "/n.pfile").c_str(), "rb");
        if (!f) return 2;
        std::vector<char> all;
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) all.insert(all.end(), buf, buf + got);
        fclose(f);
        f = fopen((d + "/cut.pfile").c_str(), "wb");
        if (!f) return 2;
        fwrite(all.data(), 1, all.size(), f);
        fclose(f);
        std::vector<std::string> copy = a;
        for (auto &s : copy)
            if (s.rfind("fea_file=", 0) == 0) s = "fea_file=" + d + "/cut.pfile";
        std::vector<char *> av;
        for (auto &s : copy) av.push_back(&s[0]);
        Interface *io = new Interface;
        std::string reader_error;
        try {
            io->Initial((int)av.size(), av.data(), /*open_output=*/false);
            io->get_pfile_info();  // header and sentence table read while the file is whole
            const size_t row = 8 + 4 * (size_t)dim;
            if (truncate((d + "/cut.pfile").c_str(), (off_t)(32768 + row * 150)) != 0) return 2;  // 150 of the 471 frames stay
            epoch(io, form == 0, /*slow_first_chunk_ms=*/200, &reader_error);
        } catch (const std::exception &e) {  // a reader that notices the truncation up front is fine too
            reader_error = e.what();
        }
        delete io;
        if (reader_error.empty()) { fprintf(stderr, "host_sanitize: the truncated pfile went unnoticed (form %d)\n", form); return 2; }
        printf("truncated pfile, form %d: reader error handed over: %s\n", form, reader_error.c_str());
    }
    printf("host_sanitize OK: frame-stream checksum %.6f, expanded checksum %.6f\n", sums[0], sums[1]);
    return 0;
}
