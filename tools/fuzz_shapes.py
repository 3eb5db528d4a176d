#!/usr/bin/env python3
"""Randomised parity sweep: random layer counts / sizes / bunch sizes / losses, 2-3 steps on the GPU against
the CPU oracle (weights 2e-5 of max|W|, like tests/test_gpu_parity.py).  Expanded and frame-stream chunks,
emulated data-parallel ranks where the shape allows.  SEED / N env vars.  ALIGN64=1: hidden widths in multiples of 64 and
minibatches of 64 / 128 / 256 frames -- with MLGGD_TILE64=2 in the environment every hidden forward / dX then runs through
the 64 x 64-tile kernels (csrc/kernels64.hip.h)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
from oracle import pyoracle

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("N", "40"))
HP = (0.1, 0.9, 1e-5)


def relmax(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


worst = 0.0
n_bits = 0
for case in range(N):
    dim = int(rng.integers(3, 40)); ctx = int(rng.choice([1, 3, 5, 7]))
    nh = int(rng.integers(0, 5))
    ls = [dim * ctx] + [int(rng.integers(1, 300)) for _ in range(nh)] + [dim]
    B = int(rng.choice([1, 7, 32, 50, 64, 100, 128, 192, 256, 300]))
    if os.environ.get("ALIGN64") == "1":
        ls = [dim * ctx] + [64 * int(rng.integers(1, 6)) for _ in range(max(nh, 1))] + [dim]
        B = int(rng.choice([64, 128, 256]))
    ml, beta = [(0, 2.0), (0, 1.0), (1, 2.0), (1, 1.2), (1, 1.0), (1, 0.9)][int(rng.integers(0, 6))]
    if ml == 1 and B < 7:
        ml, beta = 0, 2.0  # the ML gradient is ~1/|e| with a one-frame minibatch: ill-conditioned, not a parity case
    steps = int(rng.integers(1, 4))
    world = 1
    if B % 32 == 0 and rng.random() < 0.4:
        cands = [w for w in (2, 4, 8) if w * B in (64, 128, 256, 512, 1024)]
        if cands:
            world = int(rng.choice(cands))
    dp = ['gather', 'shard', 'allreduce', 'shard_a2a'][int(rng.integers(0, 4))] if world > 1 else ''
    frames_mode = bool(world == 1 and rng.random() < 0.4)
    ws, bs = synth.make_weights(ls, seed=100 + case)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    nfr = steps * world * B + ctx + 40
    feat = rng.standard_normal((nfr, dim), dtype=np.float32)
    targ_fr = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
    first = rng.permutation(nfr - ctx + 1)[:steps * world * B + int(rng.integers(0, 6))].astype(np.int32)
    toff = int(rng.integers(0, ctx))
    steps = len(first) // (world * B)  # a small B turns the ragged tail into extra full bunches
    idx = first[:, None] + np.arange(ctx)[None, :]
    inp = np.ascontiguousarray(feat[idx].reshape(len(first), ctx * dim))
    tg = np.ascontiguousarray(targ_fr[first + toff])
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    if world > 1:
        eng.fake_world(world, sharded=dp == 'shard', allreduce=dp == 'allreduce', a2a=dp == 'shard_a2a')  # rank r: rows [r*B,(r+1)*B) of each global minibatch
    ora = pyoracle.OracleNet(ls, world * B, *HP, beta, ml, ws, bs)
    if frames_mode:
        got = eng.train_frames(feat, targ_fr, first, ctx, toff)
    else:
        got = eng.train(inp, tg)
    exp = ora.train(inp, tg)  # one device, bunchsize world*B, the same rows
    assert got == steps and exp == steps, (got, exp, steps)
    we, be = eng.returnWeights(); wo, bo = ora.get_weights()
    err = max(max(relmax(a, b) for a, b in zip(we, wo)), max(relmax(a, b) for a, b in zip(be, bo)))
    worst = max(worst, err)
    # one-frame minibatches take steps of the order of the weights themselves: rounding differences are amplified
    tol = 2e-5 if B >= 7 else 2e-4
    if beta <= 1.0:  # (beta = 1, either objective: the gradient is a pure sign -- it jumps by 2 / n or 2 / sum|e| at e = 0)
        # the beta < 1 gradient sgn(e)|e|^(beta-1) is largest, and changes sign, where e -> 0: an output that differs
        # in its last bits (GEMM summation order) flips one such element and moves a bias by ~1e-4 of max|b| in a few
        # steps (SEED=22 case 12: out equal to 7e-7, alpha to 1e-7, one of 7,424 gradient elements with the other sign)
        tol = 5e-4
    # ... and against the oracle's MFMA-order twin (the HIP kernels' summation order, exponential and power restated on the
    # CPU) EVERY BIT must agree -- one device, and emulated worlds in the twin's data-parallel form (the ranks' ML
    # statistics, and with the gradient all-reduce their weight / bias gradients, met in rank order)
    bits = ""
    if True:
        pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan(), dp_world=world, dp_allreduce=dp == "allreduce")
        try:
            tw = pyoracle.OracleNet(ls, world * B, *HP, beta, ml, ws, bs)
            assert tw.train(inp, tg) == steps
            wt, bt = tw.get_weights()
            same = all(np.array_equal(a, b) for a, b in zip(list(we) + list(be), list(wt) + list(bt)))
            tw.close()
        finally:
            pyoracle.set_gemm_order("ref")
        bits = "  = twin bitwise" if same else "  DIFFERS FROM THE TWIN"
        n_bits += 1
        assert same, (case, ls, B, ml, beta, world, dp)
    if bits and B < 7:
        # a one-frame minibatch on a deep net: the distance to the documented order is the ORACLE's own order sensitivity
        # (the HIP path equals the twin in every bit) -- SEED=91 case 92, 5 layers, 4 steps of one frame: 3.4e-4 of max|W|
        tol = 2e-3
    tag = "ok " if err < tol else "BAD"
    print("%s case %2d: layers %-28s B %3d  loss (%d,%.1f) steps %d world %d%s%s  err %.1e%s" %
          (tag, case, ls, B, ml, beta, steps, world, " " + dp if dp else "", " frames" if frames_mode else "", err, bits),
          flush=True)
    assert err < tol
    eng.close(); ora.close()
print("all %d cases within tolerance; worst %.1e; %d of them checked against the MFMA-order twin: every bit equal" % (N, worst, n_bits))
