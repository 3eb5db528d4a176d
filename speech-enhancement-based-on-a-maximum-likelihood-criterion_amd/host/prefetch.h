// prefetch.h -- the chunk prefetch of the trainer: one reader thread fills the "next" chunk buffers while the
// trainer consumes the "current" ones (threadFetch + waitSignal / setSignal of the reference,
// Train_code_ML_GGD/BPtrain.cc:15-54, Interface.cc:14-53 -- there a condition variable behind a single `if`; here a
// predicate loop).  Header-only so that tests/host_sanitize.cc can run exactly this code under ThreadSanitizer /
// AddressSanitizer without a GPU.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <utility>

#include "trainer_io.h"

namespace mlggd_host {

// Two-slot hand-off between the fetch thread and the trainer.  `failed` is the reader's error hand-off: it is part of
// the wait predicate and nothing clears it, so a reader that dies while the trainer is still busy with the previous
// chunk (slot full, a later set(false) by the trainer) cannot be missed -- the trainer's next wait(true) returns
// false instead of blocking for ever on a thread that is gone.
struct Slot {
    std::mutex m;
    std::condition_variable cv;
    bool full = false;
    bool failed = false;
    // false: the reader has failed; nothing further will be handed over
    bool wait(bool want) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return failed || full == want; });
        return !failed;
    }
    void set(bool v) {
        {
            std::lock_guard<std::mutex> lk(m);
            full = v;
        }
        cv.notify_all();
    }
    void fail() {
        {
            std::lock_guard<std::mutex> lk(m);
            failed = true;
        }
        cv.notify_all();
    }
};

inline void swap_buffers(Interface *io, bool frames) {  // BPtrain.cc:25-32
    WorkPara *p = io->para;
    if (frames) {
        std::swap(p->frames_in[0], p->frames_in[1]);
        std::swap(p->frames_targ[0], p->frames_targ[1]);
        std::swap(p->first_frame[0], p->first_frame[1]);
        std::swap(p->chunk_frames[0], p->chunk_frames[1]);
        io->frames_swapped();
    } else {
        std::swap(p->indata[0], p->indata[1]);
        std::swap(p->targ[0], p->targ[1]);
    }
}

// threadFetch, BPtrain.cc:15-54.  stop: set by the trainer when it gives up before consuming every chunk.
// error_suffix: appended to a reader error (why a buffer allocation failed, known to the allocator only).
inline void fetch_loop(Interface *io, Slot *slot, std::string *error, bool frames, std::atomic<bool> *stop,
                       const std::string *error_suffix) {
    try {
        if (frames) io->reserve_frame_buffers(io->train_plan);  // once, for the largest chunk: no re-allocation mid-epoch
        for (unsigned i = 0; i < io->total_chunks && !*stop; i++) {
            const int n = frames ? io->Readchunk_frames(io->chunk_index[i]) : io->Readchunk(io->chunk_index[i]);
            if (i > 0) slot->wait(false);  // trainer done with the [1] buffers
            if (*stop) break;
            io->cur_chunk_samples = n;
            swap_buffers(io, frames);
            slot->set(true);
        }
    } catch (const std::exception &e) {
        // the message is complete before fail() publishes it (the slot's mutex orders the two); cur_chunk_samples is
        // NOT touched here: the trainer may be reading it for the chunk it still holds
        *error = e.what();
        if (error_suffix && !error_suffix->empty()) *error += ": " + *error_suffix;
        slot->fail();
    }
}

}  // namespace mlggd_host
