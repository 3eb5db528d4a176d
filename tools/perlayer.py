#!/usr/bin/env python3
"""Per-(class, layer) mean kernel time with HIP events on the baseline net."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ml = int(os.environ.get("ML", "0"))
ls = synth.baseline_layersizes(); B = 128; NB = 32
ws, bs = synth.make_weights(ls); inp, targ = synth.make_frames(NB * B, 257, 11)
eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 1.2 if ml else 2.0, ml)
eng.load_chunk(inp, targ); eng.train_resident(0, NB * B); eng.sync()
tot = 0.0
for cls, layers in (("transpose", (0,)), ("fwd", (1, 2, 3, 4)), ("loss", (0,)), ("dx", (4, 3, 2)), ("dw", (4, 3, 2, 1))):
    for l in layers:
        eng.profile_select(cls, l, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read()
        if n:
            # one dW launch covers all layers on a single GPU: its work is the sum over the layers
            f, by = eng.kernel_work(cls, 0 if (cls == "dw" and eng.dw_launches_per_step() == 1) else l)
            print("%-10s layer %d: %7.2f us  (%d launches)%s" % (cls, l, us, n,
                  "  %.1f TFLOP/s  %.0f GB/s alg" % (f / us / 1e6, by / us / 1e3) if f else ""))
            tot += us * n / NB
eng.profile_select(None)
print("sum of kernel times per step: %.1f us" % tot)
