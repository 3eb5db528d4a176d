#!/usr/bin/env python3
"""A/B launch-plan knobs in ONE process on one device (cdna guide rule 24): per config,
ms/step over the whole step and mean kernel time per class (HIP events)."""
import importlib, itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); NB = 32
inp, targ = synth.make_frames(NB * B, 257, 11)
configs = [dict(zip(("MLGGD_FWD_NW", "MLGGD_DX_NW", "MLGGD_DW_TILE"), c)) for c in
           [("4", "4", "2"), ("8", "4", "2"), ("16", "4", "2"), ("8", "8", "2"), ("8", "8", "1"), ("16", "8", "1")]]
if len(sys.argv) > 1:
    configs = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
ref = None
for rnd in range(2):
    for cfg in configs:
        for k in [k for k in os.environ if k.startswith("MLGGD_")]: del os.environ[k]
        os.environ.update(cfg)
        eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
        eng.load_chunk(inp, targ)
        eng.train_resident(0, NB * B); eng.sync()
        w = eng.returnWeights()[0][1]
        if ref is None: ref = w
        err = float(np.abs(w - ref).max() / np.abs(ref).max())
        t0 = time.perf_counter()
        for _ in range(8): eng.train_resident(0, NB * B)
        eng.sync(); dt = (time.perf_counter() - t0) / (8 * NB)
        per = {}
        for cls, layers in (("fwd", range(1, len(ls))), ("dx", range(2, len(ls))), ("dw", (0,))):
            for l in layers:  # kernel-own start/stop events (hipExtLaunchKernelGGL), one (class, layer) at a time
                eng.profile_select(cls, l, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read(); per[cls, l] = us
        eng.profile_select(None)
        print("round %d %s: %.1f us/step  fwd %s  dx %s  dw %.1f us  (W2 relerr vs first %.1e)" %
              (rnd, " ".join("%s=%s" % (k[6:], v) for k, v in cfg.items()), dt * 1e6,
               "/".join("%.2f" % per["fwd", l] for l in range(1, len(ls))), "/".join("%.2f" % per["dx", l] for l in range(2, len(ls))),
               per["dw", 0], err), flush=True)
        eng.close()
