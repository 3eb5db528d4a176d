#!/usr/bin/env python3
"""Timing-only ablations of the persistent dW kernel in ONE process on one device (rule 24): which part of a tile's
work the launch time follows.  MLGGD_DWP_ABLATE bits: 1 no epilogue update/stores, 2 no W/delta loads, 4 no MFMAs,
8 no fragment reads (and no MFMAs), 16 no operand loads, 32 no operand LDS writes.  Results are wrong by construction."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
ls = synth.baseline_layersizes(); B = 128
ws, bs = synth.make_weights(ls); NB = 32
inp, targ = synth.make_frames(NB * B, 257, 11)
names = {0: "full kernel", 1: "no epilogue update/stores", 2: "no W/delta loads", 3: "no epilogue, no W/delta loads",
         4: "no MFMAs", 7: "no epilogue, W/delta, MFMAs", 15: "... and no fragment reads", 31: "... and no operand loads",
         63: "... and no operand LDS writes (barriers + tile walk only)", 16: "no operand loads", 48: "no operand loads / LDS writes"}
# the ablation variant is read once per process (static), so each variant gets its own child process... unless
# we create engines in one process: the static is per process -> use subprocesses but ONE device, interleaved rounds
import subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
    eng.load_chunk(inp, targ)
    for _ in range(6): eng.train_resident(0, NB * B)
    eng.sync()
    eng.profile_select("dw", 0, 4096); eng.train_resident(0, NB * B); us, n = eng.profile_read(); eng.profile_select(None)
    print(json.dumps({"us": us, "n": n})); sys.exit(0)
res = {}
names[512] = "operands staged through registers + ds_write_b128 instead of LDS-DMA (correct results)"
names[576] = "... and written to LDS after the last MFMA instead of inside the block (correct results)"
names[256] = "update in scalar fp32 instructions instead of packed pairs (correct results)"
variants = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3, 4, 16, 48, 7, 15, 31, 63]
for rnd in range(3 if len(variants) <= 3 else 2):
    for abl in variants:
        env = dict(os.environ); env["MLGGD_DWP_ABLATE"] = str(abl)
        if abl == 0: env.pop("MLGGD_DWP_ABLATE")
        out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        try:
            us = json.loads(out.stdout.strip().splitlines()[-1])["us"]
        except Exception:
            print(out.stdout, out.stderr); raise
        res.setdefault(abl, []).append(us)
for abl, v in res.items():
    print("ablate %2d  %-58s %s us" % (abl, names[abl], " ".join("%6.2f" % x for x in v)))
