#!/usr/bin/env python3
"""Diagnostic: where a k_dwp workgroup's time goes, per tile (shader-clock cycles summed by wave 0 over its tiles
in the diagnostic twin kernel k_dwp_phases; see kernels.hip.h PHASES)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); inp, targ = synth.make_frames(64 * B, 257, 11)
eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
eng.load_chunk(inp, targ)
for _ in range(8):
    eng.train_resident(0, 64 * B)
eng.sync()
names = ["unit 0: loads + MFMAs", "unit 0: hand-off (vmcnt, LDS write, barrier)", "unit 1: loads + MFMAs",
         "epilogue (transpose, update, stores)", "unit 1: hand-off"]
for rep in range(3):
    eng.stamp_select("dw", -1)
    eng.train_resident(8 * B, 2 * B); eng.sync()
    st = eng.stamp_read().astype(np.float64)
n = len(st) // 2
wall = (st[:n, 2] - st[:n, 0]) * 0.01          # us
clk = (st[:n, 3] - st[:n, 1]) / np.maximum(st[:n, 2] - st[:n, 0], 1) * 100.0   # MHz
tiles = st[:n, 4]
ph = st[n:2 * n, :5]
print("k_dwp_phases: %d WGs, span %.2f us, WG time mean %.2f max %.2f us, clock %.0f MHz, tiles/WG %.2f" %
      (n, (st[:n, 2].max() - st[:n, 0].min()) * 0.01, wall.mean(), wall.max(), clk.mean(), tiles.mean()))
per_tile = ph / tiles[:, None]
tot = per_tile.sum(1)
for k, nm in enumerate(names):
    print("   %-46s %7.0f cycles/tile (%4.1f %%)  = %.2f us" % (nm, per_tile[:, k].mean(), 100 * per_tile[:, k].mean() / tot.mean(),
                                                               per_tile[:, k].mean() / clk.mean()))
print("   %-46s %7.0f cycles/tile          = %.2f us   (MFMA issue floor: 64 MFMAs x 64 cycles = 4096 per wave, "
      "8192 per SIMD with two workgroups per CU)" % ("sum", tot.mean(), tot.mean() / clk.mean()))
