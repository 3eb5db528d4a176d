#!/usr/bin/env python3
"""Compute-side cost of the data-parallel step on ONE GPU: `world` ranks are emulated one after the other
(device copies instead of RCCL), so only the dW launch over the gathered global minibatch is timed here --
what the factor exchange costs in kernels on ONE rank, not what the links cost."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); NB = 32
inp, targ = synth.make_frames(NB * B, 257, 11)
for world, sharded in ((1, False), (2, False), (4, False), (8, False), (8, True), (4, True)):
    if sharded:
        os.environ["MLGGD_FAKE_ONLY_RANK"] = "3"  # time ONE rank's share (a rank that also runs the bias-only tiles)
    eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
    if world > 1:
        eng.fake_world(world, sharded)
    eng.load_chunk(inp, targ)
    eng.train_resident(0, NB * B); eng.sync()
    eng.profile_select("dw", 0, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read(); eng.profile_select(None)
    steps = NB // world
    print("world %d%s: dW launches over %d gathered frames %.1f us per step (%d launch%s per step, kernel start/stop events)"
          % (world, " sharded update (rank 3's share only; W all-gather not emulated)" if sharded else "", world * B,
             us * n / steps, n // steps, "" if n == steps else "es"), flush=True)
    eng.close()

# The all-reduce arm (BASELINE.json's exchange) per rank at world 8: k_dwp<.,false> writes G_l, the reduce-scatter delivers
# this rank's block of the sum, k_apply_update runs on that block only (1/8 of the update traffic), the W blocks are
# all-gathered.  Emulated world: the collectives are device adds; MLGGD_FAKE_ONLY_RANK times ONE rank's update share.
for ar_shard in ("1", "0"):
    os.environ["MLGGD_DP_AR_SHARD"] = ar_shard
    os.environ["MLGGD_FAKE_ONLY_RANK"] = "3"
    eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
    eng.fake_world(8, allreduce=True)
    eng.load_chunk(inp, targ)
    eng.train_resident(0, NB * B); eng.sync()
    res = {}
    for cls in ("dw", "update"):
        eng.profile_select(cls, 0, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read(); eng.profile_select(None)
        res[cls] = us * n / (NB // 8)
    print("world 8 all-reduce arm, MLGGD_DP_AR_SHARD=%s (rank 3's update share only): update kernels %.1f us per step "
          "(the emulated prepass adds 7 unfused dW passes to the dw class: %.1f us)" % (ar_shard, res["update"], res["dw"]), flush=True)
    eng.close()
