"""GPU parity at the REAL shapes of BASELINE.json's configs: each test drives the HIP
path through the C-ABI and compares it with the CPU oracle on the same seeded inputs.

  config 1  BPtrain_Sigmoid on a tiny synthetic pfile, 257x11 -> 2048x3 -> 257, MMSE (MLflag=0, beta=2)
  config 2/3 epoch horizon: one 102,400-sample chunk (800 steps of 128 frames), ML beta 1.2 and 0.9, MMSE
  config 4  8-rank data parallel at 2827-2048x3-257, B = 128 per rank (emulated world, different rows per
            rank) against one device with bunchsize 1024: gather, shard and all-reduce exchanges
  config 5  2827-4096x6-257, B = 512, ML beta 1.2

Tolerances are those of test_gpu_parity.py (fp32, different but fixed summation orders): weights 2e-5 of
max|W| after a few steps, CV metrics 1e-4 relative (the north_star figure), alpha 1e-5 after a few steps.
After 800 steps two fp32 trajectories with different summation orders have drifted apart by more than a few
steps' rounding; the epoch-horizon test states its own (measured) bounds for weights and alpha and keeps
1e-4 for the three CV numbers the reference logs (BPtrain.cc:131-138) wherever the loss is well-conditioned
(beta >= 1); for beta = 0.9 the bound is derived inside the test from the oracle's own summation-order twins."""
import os
import re
import subprocess

import numpy as np
import pytest

import hostlib

pytestmark = pytest.mark.gpu
HP = (0.1, 0.9, 1e-5)  # lrate, momentum, weightcost of finetune.pl:10,21,27


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def relrms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30))


def test_config5_six_4096_layers_bunch_512(pkg, pyoracle, synth):
    """BASELINE config 5: 2827-4096^6-257, 512-frame minibatch, ML-GGD beta = 1.2: two steps + the CV triple.
    This is k_dwp<8,.> (8 units of 64 frames per tile), k_fwd/k_dx with K = 4096 and 7 jobs in one dW launch at
    real width (TC/BP_GPU.cu:308-440, 187-306)."""
    ls, B = synth.baseline_layersizes(hidden=4096, nhid=6), 512
    ws, bs = synth.make_weights(ls, seed=55)
    rng = np.random.default_rng(56)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(2 * B + 40, 257, 11, seed=57)    # two bunches + a ragged tail (ignored)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 1.2, 1)
    ora = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
    assert eng.train(inp, targ) == 2 and ora.train(inp, targ) == 2
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        dw, dwo = eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)
        assert relmax(dw, dwo) < 5e-4, l
        assert abs(dw.astype(np.float64).sum() - dwo.astype(np.float64).sum()) <= 1e-4 * np.abs(dwo).astype(np.float64).sum()
        assert relmax(eng.debug_tensor("delta_b", l + 1), ora.tensor("delta_b", l + 1)) < 5e-4, l
    assert relmax(eng.debug_tensor("out"), ora.tensor("out", rows=B)) < 2e-4
    assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    cin, ctarg = synth.make_frames(700, 257, 11, seed=58)          # 1 bunch + 188 frames
    sq, ab, ll = eng.cv_all(cin, ctarg)
    assert abs(sq - ora.cv_sqerr(cin, ctarg)) <= 1e-4 * abs(sq)
    assert abs(ab - ora.cv_abserr(cin, ctarg)) <= 1e-4 * abs(ab)
    assert abs(ll - ora.cv_loglik(cin, ctarg)) <= 1e-4 * abs(ll)
    eng.close()
    ora.close()


@pytest.mark.parametrize("mode", ["gather", "shard", "allreduce", "shard_a2a"])
def test_config4_eight_ranks_at_the_real_shape(pkg, pyoracle, synth, mode):
    """BASELINE config 4 (8-GPU data parallel, ML-GGD beta 1.2) at its real shape, 2827-2048^3-257 with 128 frames
    per rank, on ONE GPU: eight ranks emulated one after the other, rank r on rows [128 r, 128 (r+1)) of each
    global minibatch of 1024 rows, against the oracle with bunchsize 1024 on the same rows.
    gather/shard: k_dwp<16,true> over 16 units of gathered frames; shard: 8 uneven row blocks per layer
    (45 = 8 x 6 - 3 tile rows in layer 1, 32 = 8 x 4 in the others); shard_a2a: the same blocks with the activations
    written owner-blocked (block widths 384 and 256 units) and each virtual owner receiving only its block of them."""
    ls, B, world, steps = synth.baseline_layersizes(), 128, 8, 2
    ws, bs = synth.make_weights(ls, seed=41)
    rng = np.random.default_rng(42)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(steps * world * B, 257, 11, seed=43)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 1.2, 1)
    eng.fake_world(world, sharded=mode == "shard", allreduce=mode == "allreduce", a2a=mode == "shard_a2a")
    ora = pyoracle.OracleNet(ls, world * B, *HP, 1.2, 1, ws, bs)
    assert eng.train(inp, targ) == steps and ora.train(inp, targ) == steps
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 5e-4, l
        assert relmax(eng.debug_tensor("delta_b", l + 1), ora.tensor("delta_b", l + 1)) < 5e-4, l
    assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    cin, ctarg = synth.make_frames(300, 257, 11, seed=44)
    sq, ab, ll = eng.cv_all(cin, ctarg)
    assert abs(sq - ora.cv_sqerr(cin, ctarg)) <= 1e-4 * abs(sq)
    assert abs(ll - ora.cv_loglik(cin, ctarg)) <= 1e-4 * abs(ll)
    # and EVERY BIT of the oracle's MFMA-order twin in its data-parallel form: eight ranks' k_colsum wavefront sums of the
    # ML statistic met in rank order, and -- gradient all-reduce -- their weight / bias gradient chains likewise
    pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan(), dp_world=world, dp_allreduce=mode == "allreduce")
    try:
        twin = pyoracle.OracleNet(ls, world * B, *HP, 1.2, 1, ws, bs)
        assert twin.train(inp, targ) == steps
        wt, bt = twin.get_weights()
        for l in range(len(we)):
            assert np.array_equal(we[l], wt[l]) and np.array_equal(be[l], bt[l]), l
        assert np.array_equal(eng.scalefactor(), twin.tensor("scalefactor"))
        twin.close()
    finally:
        pyoracle.set_gemm_order("ref")
    eng.close()
    ora.close()


@pytest.mark.parametrize("mode", ["gather", "shard", "shard_a2a"])
def test_eight_ranks_equal_the_twin_at_bunchsize_1024_in_every_bit(pkg, pyoracle, synth, mode):
    """The data-parallel contract, sharpened: with the factor exchanges the weight gradient is ONE chain per weight over
    the frames of the global minibatch in their order (rank 0's rows, rank 1's, ...), everything before it is per frame,
    and the MMSE loss needs no cross-rank statistic -- so 8 ranks x 128 frames at 2827-2048^3-257 must leave exactly
    the bits of the oracle's MFMA-order twin run with bunchsize 1024 on the same rows (the gradient all-reduce sums
    per-rank partial chains instead and is held to the tolerance above)."""
    ls, B, world, steps = synth.baseline_layersizes(), 128, 8, 3
    ws, bs = synth.make_weights(ls, seed=41)
    inp, targ = synth.make_frames(steps * world * B, 257, 11, seed=43)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 2.0, 0)
    eng.fake_world(world, sharded=mode == "shard", a2a=mode == "shard_a2a")
    pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan())
    try:
        twin = pyoracle.OracleNet(ls, world * B, *HP, 2.0, 0, ws, bs)
        assert eng.train(inp, targ) == steps and twin.train(inp, targ) == steps
        we, be = eng.returnWeights()
        wt, bt = twin.get_weights()
        for l in range(len(we)):
            assert np.array_equal(we[l], wt[l]), ("weights", l + 1)
            assert np.array_equal(be[l], bt[l]), ("bias", l + 1)
        twin.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()


def test_config5_eight_ranks_take_the_allreduce_exchange(pkg, pyoracle, synth):
    """BASELINE config 5 in its 8-GPU form: 2827-4096^6-257, 512 frames per rank, ML-GGD beta 1.2, eight ranks
    emulated on one GPU against the oracle with bunchsize 4096 on the same rows.  At 8 x 512 frames the
    factor exchange is NOT usable (64 units of 64 gathered frames per tile; k_dwp is built for <= 16), so config 5 on
    8 GPUs takes the all-reduce of the weight gradients -- BASELINE.json's own exchange: k_dwp<8,false> writes G_l
    (386 MB over the 7 layers), the gradients are summed over the ranks, k_apply_update / k_bias_apply with
    n = 4096.  The test asserts the mode so that the fallback is visible, not silent."""
    ls, B, world, steps = synth.baseline_layersizes(hidden=4096, nhid=6), 512, 8, 2
    ws, bs = synth.make_weights(ls, seed=61)
    rng = np.random.default_rng(62)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(steps * world * B, 257, 11, seed=63)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 1.2, 1)
    with pytest.raises(pkg.MlggdError, match="fake world"):
        eng.fake_world(world)                       # factor all-gather: refused at this shape
    eng.fake_world(world, allreduce=True)
    assert eng.dp_mode() == 1                       # 1 = all-reduce of the weight gradients
    ora = pyoracle.OracleNet(ls, world * B, *HP, 1.2, 1, ws, bs)
    assert eng.train(inp, targ) == steps and ora.train(inp, targ) == steps
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 5e-4, l
        assert relmax(eng.debug_tensor("delta_b", l + 1), ora.tensor("delta_b", l + 1)) < 5e-4, l
    assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    cin, ctarg = synth.make_frames(700, 257, 11, seed=64)
    sq, ab, ll = eng.cv_all(cin, ctarg)
    assert abs(sq - ora.cv_sqerr(cin, ctarg)) <= 1e-4 * abs(sq)
    assert abs(ll - ora.cv_loglik(cin, ctarg)) <= 1e-4 * abs(ll)
    eng.close()
    ora.close()


def test_launch_plan_cache_does_not_grow(pkg, synth):
    """the tile-record tables of the persistent dW kernel are cached per launch plan; a key that never
    matched would re-allocate and upload a table in the middle of every step.  One GPU: exactly two plans (the
    layer-1 operand alternates between the two staged-row buffers), however many steps run."""
    ls, B = synth.baseline_layersizes(hidden=256, nhid=2), 128
    ws, bs = synth.make_weights(ls, seed=3)
    inp, targ = synth.make_frames(12 * B, 257, 11, seed=4)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 2.0, 0)
    eng.train(inp, targ)
    first = eng.plan_count()
    for _ in range(3):
        eng.train(inp, targ)
    assert eng.plan_count() == first and 1 <= first <= 2
    eng.close()


def test_config1_executable_on_a_tiny_pfile_at_the_real_shape(pkg, pyoracle, tmp_path):
    """BASELINE config 1: BPtrain_Sigmoid with fea_dim=257 fea_context=11 layersizes=2827,2048,2048,2048,257
    MLflag=0 shapefactor=2 on a tiny synthetic pfile (10 sentences of ~190 frames): the weights file
    and the CV log lines against the same epoch driven from Python (the real host IO code for the chunk and
    sample order, the CPU oracle for the math).  TC/BPtrain.cc:74-145, TC/Interface.cc:719-965."""
    exe = os.path.join(hostlib.HOST, "BPtrain_Sigmoid")
    subprocess.check_call(["make", "-C", hostlib.HOST, "-s"])
    rng = np.random.default_rng(31)
    dim, ctx, B = 257, 11, 128
    lens = [int(x) for x in rng.integers(170, 211, 10)]
    nfr = sum(lens)
    noisy = rng.normal(8, 3, (nfr, dim)).astype(np.float32)
    clean = (0.7 * noisy + rng.normal(0, 1.5, (nfr, dim))).astype(np.float32)
    hostlib.write_pfile(str(tmp_path / "n.pfile"), lens, noisy)
    hostlib.write_pfile(str(tmp_path / "c.pfile"), lens, clean)
    hostlib.write_norm(str(tmp_path / "n.norm"), noisy.mean(0), 1.0 / noisy.std(0))
    ls = [dim * ctx, 2048, 2048, 2048, dim]
    subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "5", *map(str, ls), str(tmp_path),
                           str(tmp_path / "init.wts"), "1", "2", "5"], stdout=subprocess.DEVNULL)
    kv = dict(gpu_used=0, numlayers=5, layersizes=",".join(map(str, ls)), bunchsize=B, MLflag=0, shapefactor=2,
              momentum=0.9, weightcost=1e-5, lrate=0.1, fea_dim=dim, fea_context=ctx, traincache=600,
              init_randem_seed=27870775, targ_offset=5, initwts_file=tmp_path / "init.wts", norm_file=tmp_path / "n.norm",
              fea_file=tmp_path / "n.pfile", targ_file=tmp_path / "c.pfile", outwts_file=tmp_path / "mlp.1.wts",
              log_file=tmp_path / "mlp.1.log", train_sent_range="0-7", cv_sent_range="8-9", dropoutflag=0,
              visible_omit=0.1, hid_omit=0.1)
    res = subprocess.run([exe] + ["%s=%s" % (k, v) for k, v in kv.items()], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all finish!" in res.stdout
    log = open(tmp_path / "mlp.1.log").read()

    io = hostlib.HostIO(**kv)
    w0, b0 = hostlib.read_wts(str(tmp_path / "init.wts"), ls)
    ora = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, w0, b0)
    starts, total = io.plan("0-7")
    order = io.shuffle(len(starts))
    assert len(starts) >= 2
    steps = 0
    for n, ci in enumerate(order):
        inp, tg = io.read_chunk(ci, ls[0], dim, 600)
        assert "Starting chunk %d of %d containing %d samples." % (n + 1, len(starts), len(inp)) in log
        steps += ora.train(inp, tg)
    assert steps >= 8
    cvs, cvtotal = io.plan("8-9", cv=True)
    sq = ab = np.float32(0)
    for ci in range(len(cvs)):
        inp, tg = io.read_chunk(ci, ls[0], dim, 600, cv=True)
        sq += np.float32(ora.cv_sqerr(inp, tg))
        ab += np.float32(ora.cv_abserr(inp, tg))
    io.close()
    ws, bs = hostlib.read_wts(str(tmp_path / "mlp.1.wts"), ls)
    wo, bo = ora.get_weights()
    for l in range(4):
        assert relmax(ws[l], wo[l]) < 5e-5, l
        assert relmax(bs[l], bo[l]) < 5e-5, l
    got = [float(re.search(pat + r": (-?[\d.]+)", log).group(1)) for pat in
           ("CV over. squared error", "CV over. square root squared error")]
    want = [float(sq) / cvtotal, float(ab) / cvtotal]
    for g_, w_ in zip(got, want):
        assert abs(g_ - w_) <= 1e-4 * abs(w_) + 1e-6, (got, want)
    assert "CV2 over" not in log      # MLflag=0: no likelihood line (BPtrain.cc:136-139)
    ora.close()


@pytest.mark.parametrize("ml,beta", [(1, 1.2), (1, 1.0), (1, 0.9), (0, 2.0)])
def test_epoch_horizon_one_full_chunk_of_800_steps(pkg, pyoracle, synth, ml, beta):
    """north_star: "per-epoch ML-GGD loss matching the reference to 1e-4 rel".  One traincache-sized chunk of
    102,400 samples = 800 steps of 128 frames at 2827-2048^3-257 (TC/BP_GPU.cu:170-184), then the three numbers
    the reference logs after an epoch (TC/BPtrain.cc:112-139: CV squared error, |error|/D, GGD log-likelihood)
    on a held-out chunk, and alpha, the per-dimension GGD scale the last step leaves behind (BP_GPU.cu:417-420).

    beta > 1 (MMSE, ML beta 1.2): CV numbers within 1e-4 relative -- the north_star figure -- (measured r02:
    1e-6 .. 4e-6), alpha within 5e-4 of its maximum (8.6e-5), weights within 2e-3 of max|W| (1.5e-4 .. 8.4e-4:
    800 steps of two fp32 trajectories with different summation orders).

    beta = 1 (the shipped objective): CV numbers within 1e-4 relative as well; alpha and the weights bounded by the
    oracle's own order twins as for beta = 0.9 below (the sign gradient jumps where an error crosses zero).

    beta = 0.9 -- PARITY UNPINNED, and not reachable at 1e-4 by ANY implementation that is not bit-identical to the
    reference build: the loss itself is ill-conditioned (the gradient sgn(e)|e|^(beta-1) jumps where an error
    crosses zero and grows as |e| -> 0), so a trajectory amplifies rounding-level differences.  The bound is
    therefore DERIVED IN THE TEST from the oracle alone: the same oracle is run with a different but
    equally valid GEMM summation order (cuBLAS leaves it open, SURVEY 8a-ii: `ora_set_gemm_split`, reductions as 4
    resp. 7 contiguous partial sums -- what any split-K GEMM does), and the HIP path must sit within K_TWIN = 4 times
    the larger of the two twin-to-oracle distances, for the CV numbers, alpha (relative rms) and the weights
    (relative rms) alike.  The loss chain itself agrees with the oracle to a few ulp when fed identical
    activations (tests/test_gpu_loss_ulps.py): the distance measured here is the trajectory's, not a kernel's."""
    ls, B, n = synth.baseline_layersizes(), 128, 102400
    ws, bs = synth.make_weights(ls)
    inp, targ = synth.make_frames(n, 257, 11)
    cin, ctarg = synth.make_frames(3000, 257, 11, seed=77)
    eng = pkg.BPGpu(synth.DEFAULT_SEED, 0, ls, B, *HP, ws, bs, beta, ml)
    assert eng.train(inp, targ) == 800

    s_out, plan = eng.out_slabs(), eng.gemm_plan()

    def oracle_run(split, order="ref"):
        pyoracle.set_gemm_split(split)
        pyoracle.set_gemm_order(order, s_out, plan=plan)
        try:
            o = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
            assert o.train(inp, targ) == 800
            r = {"sq": o.cv_sqerr(cin, ctarg), "ab": o.cv_abserr(cin, ctarg),
                 "ll": o.cv_loglik(cin, ctarg) if ml else 0.0, "alpha": o.tensor("scalefactor").copy(), "w": o.get_weights()[0]}
            o.close()
            return r
        finally:
            pyoracle.set_gemm_split(1)
            pyoracle.set_gemm_order("ref")

    ora = oracle_run(1)
    sq, ab, ll = eng.cv_all(cin, ctarg)
    hip = {"sq": sq, "ab": ab, "ll": ll if ml else 0.0, "alpha": eng.scalefactor(), "w": eng.returnWeights()[0]}

    def dist(a, b):
        d = {k: abs(a[k] - b[k]) / max(abs(b[k]), 1e-30) for k in ("sq", "ab", "ll")}
        d["alpha"] = relrms(a["alpha"], b["alpha"]) if ml else 0.0
        d["alpha_max"] = relmax(a["alpha"], b["alpha"]) if ml else 0.0
        d["w"] = max(relrms(x, y) for x, y in zip(a["w"], b["w"]))
        d["w_max"] = max(relmax(x, y) for x, y in zip(a["w"], b["w"]))
        return d

    d_hip = dist(hip, ora)
    print("epoch horizon ml=%d beta=%.1f: HIP vs oracle: sqerr %.1e abserr %.1e loglik %.1e | alpha relrms %.1e relmax %.1e | "
          "weights relrms %.1e relmax %.1e" % (ml, beta, d_hip["sq"], d_hip["ab"], d_hip["ll"], d_hip["alpha"],
                                               d_hip["alpha_max"], d_hip["w"], d_hip["w_max"]))
    if beta > 1.0:
        assert d_hip["sq"] <= 1e-4 and d_hip["ab"] <= 1e-4 and d_hip["ll"] <= 1e-4
        assert d_hip["alpha_max"] < 5e-4 and d_hip["w_max"] < 2e-3
        # all 800 steps equal the oracle's MFMA-order twin (the HIP kernels' summation order and their IEEE-only
        # exponential and power, restated on the CPU) in EVERY bit of the 14.7 M weights, of alpha and of the CV numbers
        tw = oracle_run(1, "hip")
        assert all(np.array_equal(x, y) for x, y in zip(hip["w"], tw["w"]))
        assert hip["sq"] == tw["sq"] and hip["ab"] == tw["ab"] and hip["ll"] == tw["ll"]
        assert np.array_equal(hip["alpha"], tw["alpha"])
        print("   HIP == MFMA-order twin bit for bit after 800 steps (weights, alpha, CV numbers)")
    else:
        K_TWIN = 4.0
        # three twins of the oracle: reductions as 4 / 7 contiguous partial sums, and the MFMA-order twin -- the HIP
        # kernels' own summation order with fused multiply-adds (bit-identical GEMMs, tests/test_gpu_mfma_order.py), so
        # its distance to the oracle is what the HIP path's choices (summation order, IEEE-only exponential and power)
        # do to this trajectory; the HIP path's distance to IT must be zero, in every bit (r04: exp_det, pow_det)
        names = ("split 4", "split 7", "MFMA order") if beta < 1.0 else ("split 4", "MFMA order")  # (an 800-step oracle run takes ~20-40 s)
        runs = [oracle_run(4)] + ([oracle_run(7)] if beta < 1.0 else []) + [oracle_run(1, "hip")]
        twins = [dist(r, ora) for r in runs]
        yard = {k: max(t[k] for t in twins) for k in twins[0]}
        for nm, t in zip(names, twins):
            print("   oracle order twin (%s) vs oracle: sqerr %.1e abserr %.1e loglik %.1e | alpha relrms %.1e relmax %.1e | "
                  "weights relrms %.1e relmax %.1e" % (nm, t["sq"], t["ab"], t["ll"], t["alpha"], t["alpha_max"], t["w"], t["w_max"]))
        t = dist(hip, runs[-1])
        print("   HIP vs the MFMA-order twin (must be zero): sqerr %.1e abserr %.1e loglik %.1e | alpha relrms %.1e relmax %.1e | "
              "weights relrms %.1e relmax %.1e" % (t["sq"], t["ab"], t["ll"], t["alpha"], t["alpha_max"], t["w"], t["w_max"]))
        # beta = 0.9, measured r03: HIP vs oracle 8.7e-5 / 5.2e-5 / 9.4e-5 (all three inside the north_star's 1e-4 this
        # time; r02's build: 1.05e-4 on the first), twins 3.4e-5 / 2.9e-5 / 3.8e-5 and 3.0e-5 / 3.0e-6 / 1.6e-5 -- the twins
        # differ from each other by 10x in one number, so the CV numbers pass at 1e-4 OR inside K_TWIN x the yardstick.
        # alpha (relrms 1.3e-3 vs 1.4e-3 / 1.3e-3) and the weights (6.6e-3 vs 6.9e-3 / 6.7e-3) sit exactly ON the twins'
        # distance: the divergence is the trajectory's own, saturated at the same level whatever perturbs it.
        # beta = 1 (the SHIPPED objective, TC/finetune.pl:25-26; r04): the gradient sgn(e) / sum|e| is bounded but still
        # jumps by 2 / sum|e| where an error crosses zero, so alpha and the weights drift the same way (HIP vs oracle
        # alpha relrms 5.4e-4, weights 2.2e-3: bounded by the twins like beta = 0.9) -- while the three numbers the
        # reference LOGS stay at 2.7e-5 / 2.7e-5 / 1.9e-5 and are held to the north_star's plain 1e-4.
        # bit for bit the MFMA-order twin, all 800 steps, whatever the distance to the documented order
        if True:
            assert all(np.array_equal(x, y) for x, y in zip(hip["w"], runs[-1]["w"]))
            assert np.array_equal(hip["alpha"], runs[-1]["alpha"])
            assert hip["sq"] == runs[-1]["sq"] and hip["ab"] == runs[-1]["ab"] and hip["ll"] == runs[-1]["ll"]
        for k in ("sq", "ab", "ll"):
            assert d_hip[k] <= (1e-4 if beta == 1.0 else max(1e-4, K_TWIN * yard[k])), (k, d_hip[k], yard[k])
        for k in ("alpha", "w"):
            assert d_hip[k] <= K_TWIN * yard[k], (k, d_hip[k], yard[k])
        assert d_hip["sq"] <= 1e-3 and d_hip["ab"] <= 1e-3 and d_hip["ll"] <= 1e-3  # and never an order of magnitude worse
    # the training loss has actually moved (the comparison is not between two untrained nets)
    assert sq / (3000 * 257) < 0.9
    eng.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_cv_sums_formed_on_the_device(pkg, synth, ml, beta):
    """SURVEY 8f2 (TC/BP_GPU.cu:187-306): with mlggd_set_cv_device_reduce the three CV sums are formed on the
    device (k_cv_reduce: fp32 terms as the host loops form them, sums in double) and nothing of size n x D is
    copied back.  They must equal a float64 sum over the engine's own forward outputs to 1e-6 relative, and the
    default reference-order fp32 host accumulation to its own rounding (about sqrt(n*D) * 6e-8; 1e-4 allowed) --
    for expanded chunks and for frame-stream chunks, with a ragged last bunch."""
    dim, ctx, B, toff = 257, 11, 128, 5
    ls = [dim * ctx, 512, 384, dim]
    rng = np.random.default_rng(13)
    nfr = 1500
    feat = rng.standard_normal((nfr, dim), dtype=np.float32)
    targ = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
    first = rng.permutation(nfr - ctx + 1)[:9 * B + 50].astype(np.int32)
    idx = first[:, None] + np.arange(ctx)[None, :]
    inp = np.ascontiguousarray(feat[idx].reshape(len(first), ctx * dim))
    tg = np.ascontiguousarray(targ[first + toff])
    ws, bs = synth.make_weights(ls, seed=14)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    assert eng.train(inp[:3 * B], tg[:3 * B]) == 3            # leaves an alpha behind for the likelihood
    host = eng.cv_all(inp, tg)
    out = eng.forward(inp).astype(np.float32)
    e32 = out - tg
    want_sq = float((e32 * e32).astype(np.float64).sum())
    want_ab = float(np.abs(e32).astype(np.float64).sum() / dim)
    eng.set_cv_device_reduce(True)
    dev = eng.cv_all(inp, tg)
    devf = eng.cv_all_frames(feat, targ, first, ctx, toff)
    assert dev == devf                                        # same kernels, same tiles: bit-identical
    assert abs(dev[0] - want_sq) <= 1e-6 * want_sq and abs(dev[1] - want_ab) <= 1e-6 * want_ab
    assert abs(dev[0] - host[0]) <= 1e-4 * abs(host[0]) and abs(dev[1] - host[1]) <= 1e-4 * abs(host[1])
    assert eng.CrossValid(inp, tg) == dev[0] and eng.CrossValiddB(inp, tg) == dev[1]
    if ml:
        alpha = eng.scalefactor().astype(np.float32)
        d3 = float(np.power(np.abs(tg - out) / alpha[None, :], np.float32(beta)).astype(np.float64).sum())
        d1 = len(first) * dim * float(np.log(np.float32(beta / (2 * pkg.gamma(np.float32(1.0 / beta))))))
        d2 = float(np.log(alpha).astype(np.float64).sum()) * len(first)
        want_ll = d1 - d2 - d3
        assert abs(dev[2] - want_ll) <= 2e-6 * abs(want_ll)
        assert abs(dev[2] - host[2]) <= 1e-4 * abs(host[2])
        assert eng.CrossValid2(inp, tg) == dev[2]
    eng.set_cv_device_reduce(False)
    assert eng.cv_all(inp, tg) == host                        # switching back restores the reference-order path
    eng.close()
