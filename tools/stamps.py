#!/usr/bin/env python3
"""Diagnostic: in-kernel phase timings (100 MHz stamps) of the GEMM kernels on the baseline net."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); inp, targ = synth.make_frames(16 * B, 257, 11)
eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
eng.load_chunk(inp, targ)
eng.train_resident(0, 8 * B); eng.sync()
names = {"fwd": ["mainloop", "reduce-sync", "epilogue"], "dx": ["mainloop", "reduce-sync", "epilogue"],
         "dw": ["stage", "prefetch-issue", "mfma", "acc->lds", "update+store"]}
def clock_mhz(st, a, b):
    """shader clock between stamp pairs (wall, clk) at slots a and b"""
    dt = (st[:, b] - st[:, a]) / 0.01  # back to 100 MHz ticks
    dc = (st[:, b + 1] - st[:, a + 1]) / 0.01
    ok = dt > 0
    return (dc[ok] / dt[ok] * 100.0)

for rep in range(2):
    eng.stamp_select("dw", 1)
    eng.train_resident(8 * B, 2 * B); eng.sync()
    st = eng.stamp_read().astype(np.float64) * 0.01
t0 = st[:, 0].min()
dur = st[:, 2] - st[:, 0]
mhz = clock_mhz(st, 0, 2)
print("dw (all layers, persistent): %d WGs, span %.2f us; WG time mean %.2f max %.2f us; tiles per WG %.2f; "
      "shader clock mean %.0f MHz (min %.0f max %.0f)" % (len(st), st[:, 2].max() - t0, dur.mean(), dur.max(),
      (st[:, 4] / 0.01).mean(), mhz.mean(), mhz.min(), mhz.max()))
for cls, layers in (("fwd", (1, 2)), ("dx", (2, 3))):
    for l in layers:
        for rep in range(2):
            eng.stamp_select(cls, l)
            eng.train_resident(8 * B, 2 * B); eng.sync()
            st = eng.stamp_read().astype(np.float64) * 0.01  # us
        if cls == "fwd":
            mhz = clock_mhz(st, 4, 6)
            print("%s layer %d: shader clock mean %.0f MHz (min %.0f max %.0f)" % (cls, l, mhz.mean(), mhz.min(), mhz.max()))
            st = st[:, :4]
        nz = [j for j in range(st.shape[1]) if st[:, j].any()]
        t0 = st[:, nz[0]].min()
        print("%s layer %d: %d WGs, span %.2f us (first start -> last end); start spread %.2f us" %
              (cls, l, len(st), st[:, nz[-1]].max() - t0, st[:, nz[0]].max() - t0))
        for a, b, nm in zip(nz[:-1], nz[1:], names[cls]):
            d = st[:, b] - st[:, a]
            print("   %-16s mean %6.2f  p10 %6.2f  p90 %6.2f us" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
        d = st[:, nz[-1]] - st[:, nz[0]]
        print("   %-16s mean %6.2f  max %6.2f us" % ("WG total", d.mean(), d.max()))
