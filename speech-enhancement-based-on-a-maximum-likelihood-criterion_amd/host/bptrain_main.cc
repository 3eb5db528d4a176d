// bptrain_main.cc -- the trainer executable (drop-in for the reference's BPtrain_Sigmoid,
// Train_code_ML_GGD/BPtrain.cc:55-146): same key=value command line as emitted by
// finetune.pl:50-76, same log lines, same .wts output; one epoch per process.
//
// Flow (BPtrain.cc:74-145): Initial -> BP_GPU -> get_pfile_info -> plan training chunks ->
// shuffle the chunk order (one lrand48 stream: chunk order first, then one shuffle per chunk
// read, in read order) -> prefetch thread reads chunk i+1 while the GPU trains on chunk i ->
// write weights -> cross-validate -> three log lines.
//
// Input pipeline (new): by default chunks travel as raw normalised frames plus the
// first frame of every (shuffled) sample row and are gathered on the GPU; MLGGD_EXPANDED=1 selects
// the reference's host-side context expansion (Interface::Readchunk) instead -- same results.
//
// Data parallel (new): when WORLD_SIZE > 1 (torchrun-style env: RANK,
// LOCAL_RANK, WORLD_SIZE; MLGGD_ID_FILE names a file on a shared filesystem used to hand the
// RCCL unique id from rank 0 to the others) every rank reads the same chunks with the same
// seed and trains rows [rank*bunchsize,(rank+1)*bunchsize) of each global minibatch of
// WORLD_SIZE*bunchsize samples; rank 0 alone writes the log, the weights and runs CV.
#include <chrono>
#include <condition_variable>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "bp_gpu.h"
#include "dp_launch.h"
#include "prefetch.h"
#include "trainer_io.h"

using mlggd_host::Interface;
using mlggd_host::IoError;
using mlggd_host::WorkPara;

namespace {

using mlggd_host::Slot;

std::atomic<bool> g_stop_fetch{false};  // set when the trainer gives up before consuming every chunk
int g_pinned_device = 0;                // device whose context owns the page-locked chunk buffers
std::string g_pinned_error;             // engine's message when a page-locked allocation failed (fetch thread only)

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// rank 0 creates the RCCL id and hands it to the other ranks through MLGGD_ID_FILE (dp_launch.h: a handshake
// with per-launch nonces, so a file left by an earlier epoch's launch is never accepted)
void exchange_id(int world, int rank, unsigned char id[MLGGD_UNIQUE_ID_BYTES]) {
    static_assert(MLGGD_UNIQUE_ID_BYTES == mlggd_host::kUniqueIdBytes, "id size");
    const char *path = getenv("MLGGD_ID_FILE");
    if (rank == 0 && mlggd_comm_unique_id(id) != MLGGD_OK) throw IoError(mlggd_last_error());
    mlggd_host::rendezvous(path ? path : "", world, rank, id, (double)env_int("MLGGD_RENDEZVOUS_TIMEOUT", 600));
}

}  // namespace

// Data-parallel runs only: RCCL collectives have no timeout, so a rank whose peer has died would sit in its next
// collective for ever.  The main thread bumps g_progress at every chunk / phase; if it stays unchanged for
// MLGGD_WATCHDOG_S seconds (default 900, 0 = off) this thread says where the rank was and ends the process with
// status 3, so that a launcher (or the batch system) sees a failure instead of a hang.
static std::atomic<long> g_progress{0};
static std::atomic<const char *> g_phase{"start"};
static void progress(const char *phase_name) {
    g_phase = phase_name;
    g_progress++;
}
static void watchdog_loop(int rank, int world, int timeout_s) {
    long seen = -1;
    double since = 0;
    for (;;) {
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        const long now = g_progress.load();
        if (now < 0) return;  // normal end of main
        if (now != seen) {
            seen = now;
            since = 0;
            continue;
        }
        since += 0.5;
        if (since >= timeout_s) {
            fprintf(stderr, "BPtrain watchdog: rank %d of %d made no progress for %d s in phase '%s' (step %ld); a peer has "
                            "probably died inside a collective -- giving up\n", rank, world, timeout_s, g_phase.load(), seen);
            fflush(stderr);
            _exit(3);
        }
    }
}

// MLGGD_TIMING=1: wall-clock of the phases on stderr (where does an epoch of the executable go?)
static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static void phase(const char *name, double &t_last) {
    static const bool on = getenv("MLGGD_TIMING") != nullptr;
    const double t = now_s();
    if (on) fprintf(stderr, "[timing] %-28s %8.3f s\n", name, t - t_last);
    t_last = t;
}

int main(int argc, char *argv[]) {
    const double t_start = (double)time(NULL);
    double t_phase = now_s();
    // BPtrain.cc:71: black on red when stdout is a terminal, as the reference prints it; plain text into pipes / logs
    printf(isatty(STDOUT_FILENO) ? "\033[41;30m--------activation functin is sigmoid--------\033[0m\n"
                                 : "--------activation functin is sigmoid--------\n");
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0);
    const int local_rank = env_int("LOCAL_RANK", rank);
    Interface *io = new Interface;
    try {
        io->Initial(argc, argv, /*open_output=*/rank == 0);
        phase("arguments, norm, init weights", t_phase);
        WorkPara *p = io->para;
        // ---- train (BPtrain.cc:81-102).  The chunk plan and the fetch thread only touch host state, so the
        // first chunk is read while the engine initialises HIP and uploads the weights (the reference creates
        // BP_GPU first, BPtrain.cc:77-78, and reads the first chunk afterwards).
        io->get_pfile_info();
        io->get_chunk_info(p->train_sent_range);
        io->chunk_index.resize(io->total_chunks);
        for (unsigned i = 0; i < io->total_chunks; i++) io->chunk_index[i] = (int)i;
        io->GetRandIndex(io->chunk_index.data(), (int)io->total_chunks);

        Slot slot;
        std::string fetch_error;
        const bool frames = env_int("MLGGD_EXPANDED", 0) == 0;
        // page-locked chunk buffers: the per-chunk upload of ~200 MB runs at DMA speed instead of through
        // the runtime's staging copies
        // (allocated by the fetch thread, which has no current device of its own: the allocator names this
        // rank's device, so no rank pins its buffers through GPU 0's context)
        const int device = world > 1 ? local_rank : p->gpu_used;
        g_pinned_device = device;
        if (frames && env_int("MLGGD_PINNED", 1))
            io->set_buffer_allocator(
                [](size_t n) -> void * {
                    void *q = nullptr;
                    if (mlggd_alloc_pinned_on(g_pinned_device, n, &q) != MLGGD_OK) {
                        g_pinned_error = mlggd_last_error();
                        fprintf(stderr, "page-locked chunk buffer of %zu bytes: %s\n", n, g_pinned_error.c_str());
                        return nullptr;
                    }
                    return q;
                },
                [](void *q) { mlggd_free_pinned(q); });
        std::thread fetch(mlggd_host::fetch_loop, io, &slot, &fetch_error, frames, &g_stop_fetch, &g_pinned_error);
        struct FetchGuard {  // any exception from here on: stop and join the reader before unwinding
            std::thread &t;
            Slot &s;
            ~FetchGuard() {
                if (t.joinable()) {
                    g_stop_fetch = true;
                    s.set(false);
                    t.join();
                }
            }
        } fetch_guard{fetch, slot};
        phase("pfile headers, chunk plan", t_phase);

        std::thread watchdog;
        const int wd_s = env_int("MLGGD_WATCHDOG_S", 900);
        if (world > 1 && wd_s > 0) {
            watchdog = std::thread(watchdog_loop, rank, world, wd_s);
            watchdog.detach();
        }
        progress("engine + communicator");
        BP_GPU *net = new BP_GPU(p->init_randem_seed, device, io->numlayers, p->layersizes, p->bunchsize, p->lrate,
                                 p->momentum, p->weightcost, p->weights, p->bias, p->shapefactor, p->MLflag,
                                 p->dropoutflag, p->visible_omit, p->hid_omit);
        if (world > 1) {
            unsigned char id[MLGGD_UNIQUE_ID_BYTES];
            exchange_id(world, rank, id);
            net->joinComm(id, world, rank);  // collective: returns once every rank has joined
            if (rank == 0) mlggd_host::rendezvous_cleanup(getenv("MLGGD_ID_FILE"), world);
        }
        phase("engine (HIP init, upload)", t_phase);
        const int K0 = p->layersizes[0], D = p->layersizes[io->numlayers - 1], B = p->bunchsize;
        std::vector<float> loc_in, loc_targ;
        for (unsigned i = 0; i < io->total_chunks; i++) {
            progress("waiting for a chunk");
            if (!slot.wait(true)) break;  // the reader failed: its message is raised after the join below
            progress("training a chunk");
            io->logf("Starting chunk %d of %d containing %d samples.\n", i + 1, io->total_chunks, io->cur_chunk_samples);
            if (io->fp_log) fflush(io->fp_log);
            const int ns = io->cur_chunk_samples;
            if (frames) {
                const int *first = p->first_frame[1];
                std::vector<int> loc_first;
                int nloc = ns;
                if (world > 1) {  // this rank's rows of every complete global minibatch
                    const std::vector<int> rows = mlggd_host::rank_sample_rows(ns, B, world, rank);
                    loc_first.resize(rows.size());
                    for (size_t j = 0; j < rows.size(); j++) loc_first[j] = first[rows[j]];
                    first = loc_first.data();
                    nloc = (int)rows.size();
                }
                // no wait: the buffers are free once the chunk is on the device, and the next chunk's upload
                // overlaps these steps (the engine keeps two device buffer sets)
                net->train_frames(p->chunk_frames[1], p->fea_context, p->frames_in[1], p->frames_targ[1], nloc, first,
                                  p->targ_offset, /*wait=*/false);
            } else if (world == 1) {
                net->train(ns, p->indata[1], p->targ[1]);
            } else {
                // this rank's rows of every complete global minibatch, compacted
                const std::vector<int> rows = mlggd_host::rank_sample_rows(ns, B, world, rank);
                loc_in.resize(rows.size() * K0);
                loc_targ.resize(rows.size() * D);
                for (size_t j = 0; j < rows.size(); j++) {
                    memcpy(&loc_in[j * K0], p->indata[1] + (size_t)rows[j] * K0, (size_t)K0 * sizeof(float));
                    memcpy(&loc_targ[j * D], p->targ[1] + (size_t)rows[j] * D, (size_t)D * sizeof(float));
                }
                net->train((int)rows.size(), loc_in.data(), loc_targ.data());
            }
            slot.set(false);
        }
        fetch.join();
        if (!fetch_error.empty()) throw IoError(fetch_error);
        progress("final sync");
        net->sync();
        progress("weights / CV");
        phase("training chunks", t_phase);

        io->logf("Total cost time: %.1f s.\n", (double)time(NULL) - t_start);  // BPtrain.cc:104-105

        if (rank == 0) {
            printf("begin to write weights\n");
            net->returnWeights(p->weights, p->bias);
            io->Writeweights();
            printf("finish to write weights\n\n");
            phase("download + write weights", t_phase);

            // ---- CV (BPtrain.cc:112-140)
            printf("begin to CV\n");
            io->logf("Starting CV.\n");
            io->get_chunk_info_cv(p->cv_sent_range);
            float squared_err = 0.0f, dB_squared_err = 0.0f, likelihood = 0.0f;
            for (unsigned i = 0; i < io->cv_total_chunks; i++) {
                float sq = 0, ab = 0, ll = 0;
                int n;
                progress("cross validation");
                if (frames) {
                    if (i == 0) io->reserve_frame_buffers(io->cv_plan);  // the device is idle here (net->sync() above)
                    n = io->Readchunk_frames_cv((int)i);
                    net->CrossValidAll_frames(p->chunk_frames[0], p->fea_context, p->frames_in[0], p->frames_targ[0], n,
                                              p->first_frame[0], p->targ_offset, &sq, &ab, &ll);
                } else {
                    n = io->Readchunk_cv((int)i);
                    net->CrossValidAll(n, p->indata[0], p->targ[0], &sq, &ab, &ll);
                }
                printf("cur_chunk_samples=%d\n", n);
                squared_err += sq;
                dB_squared_err += ab;
                if (p->MLflag == 1) likelihood += ll;
            }
            const float cvacc = ((float)squared_err / io->cv_total_samples);
            io->logf("CV over. squared error: %f\n", cvacc);
            const float cvacc1 = ((float)dB_squared_err / io->cv_total_samples);
            io->logf("CV over. square root squared error: %f\n", cvacc1);
            if (p->MLflag == 1) {
                const float cvacc2 = ((float)likelihood / io->cv_total_samples);
                io->logf("CV2 over. CV log likelihood: %f\n", cvacc2);
            }
            if (io->fp_log) fflush(io->fp_log);
            phase("cross validation", t_phase);
        }
        printf("all finish!\n");
        g_progress = -1000000;  // the watchdog leaves
        delete net;
        delete io;
        phase("teardown", t_phase);
    } catch (const std::exception &e) {
        // the reference writes the message to the log and exit(0)s (e.g. Interface.cc:320,325,451);
        // same message, non-zero status
        io->logf("%s\n", e.what());
        if (io->fp_log) fflush(io->fp_log);
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
