"""A/B of MLGGD_STAGE_AHEAD (the next minibatch's input staged beside the loss kernel) for the MMSE and the ML-GGD step."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); NB = 32
inp, targ = synth.make_frames(NB * B, 257, 11)
for ml, beta in ((0, 2.0), (1, 1.2)):
    for stage in ("1", "0"):
        os.environ["MLGGD_STAGE_AHEAD"] = stage
        eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, beta, ml)
        eng.load_chunk(inp, targ)
        for _ in range(4): eng.train_resident(0, NB * B)
        eng.sync()
        per = {}
        for cls in ("loss", "transpose", "fwd", "dx"):
            eng.profile_select(cls, 0, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read(); per[cls] = (round(us, 2), n)
        eng.profile_select(None)
        t0 = time.perf_counter()
        for _ in range(8): eng.train_resident(0, NB * B)
        eng.sync(); dt = (time.perf_counter() - t0) / (8 * NB)
        print("ml=%d stage_ahead=%s: %.1f us/step; bracketed (incl. ~3-4 us bracket): %s" % (ml, stage, dt * 1e6, per), flush=True)
        eng.close()
