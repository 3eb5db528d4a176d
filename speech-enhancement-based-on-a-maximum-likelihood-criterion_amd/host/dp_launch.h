// dp_launch.h -- host-side pieces of the data-parallel launch of BPtrain_Sigmoid (new work, SURVEY.md 8e;
// the reference is single-GPU, TC/BPtrain.cc:77-78): which rows of a chunk a rank trains, and how the RCCL
// unique id gets from rank 0 to the other ranks through a file without any of them ever accepting a stale one.
#pragma once
#include <string>
#include <vector>

namespace mlggd_host {

constexpr int kUniqueIdBytes = 128;  // MLGGD_UNIQUE_ID_BYTES, include/mlggd.h

// Sample rows of a chunk of n_samples rows that rank `rank` of `world` trains: rows
// [g*world*bunch + rank*bunch, + bunch) of every COMPLETE global minibatch g (the trailing partial one is
// dropped, as BP_GPU::train drops a partial bunch, TC/BP_GPU.cu:177-180).  Returned in training order.
std::vector<int> rank_sample_rows(int n_samples, int bunch, int world, int rank);

// File rendezvous for the 128-byte id.  finetune.pl starts one process per epoch, so a file left by an
// earlier launch may already sit at `path`; a rank must never join with it.  Protocol (every file written to
// a temporary name and renamed into place):
//   rank 0: removes path, path.go and path.ack.*, draws a nonce N, writes path = {magic, N, id};
//           waits until every path.ack.r = {N_seen, M_r} carries N_seen == N, then writes
//           path.go = {N, M_1 .. M_{world-1}}.
//   rank r: draws M_r; whenever path holds a nonce it has not acknowledged yet it writes
//           path.ack.r = {that nonce, M_r}; it returns the id read with nonce N' once path.go carries N' AND
//           its own M_r -- a go file from an earlier launch cannot contain the M_r drawn just now.
// Throws IoError after timeout_s seconds.  id: input for rank 0, output for the others.
void rendezvous(const std::string &path, int world, int rank, unsigned char id[kUniqueIdBytes], double timeout_s);
// rank 0, once every rank has joined the communicator: remove the rendezvous files
void rendezvous_cleanup(const std::string &path, int world);

}  // namespace mlggd_host
