"""CPU: the oracle against (a) the independent float64 model, (b) the committed goldens,
(c) its own phase split (the data-parallel contract), (d) edge cases of BP_GPU::train."""
import math
import os

import numpy as np
import pytest

from ref64 import Ref64

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = [(0, 2.0), (0, 1.0), (1, 2.0), (1, 1.2), (1, 1.0), (1, 0.9)]
HP = (0.1, 0.9, 1e-5)


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("ml,beta", CASES)
def test_oracle_matches_float64_model(pyoracle, synth, ml, beta):
    ls, B = [15, 8, 8, 8, 5], 8
    ws, bs = synth.make_weights(ls, seed=3)
    inp, targ = synth.make_frames(3 * B, 5, 3, seed=4)
    o = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
    r = Ref64(ls, *HP, beta, ml, ws, bs)
    assert o.train(inp, targ) == 3
    for i in range(3):
        r.step(inp[i * B:(i + 1) * B], targ[i * B:(i + 1) * B])
    w, b = o.get_weights()
    for l in range(4):
        assert relmax(w[l], r.W[l]) < 2e-6
        assert relmax(b[l], r.b[l]) < 2e-5
        assert relmax(o.tensor("delta_w", l + 1), r.dW[l]) < 2e-5
    cv = r.cv(inp, targ)
    assert abs(o.cv_sqerr(inp, targ) - cv["sqerr"]) < 1e-5 * cv["sqerr"]
    assert abs(o.cv_abserr(inp, targ) - cv["abserr"]) < 1e-5 * cv["abserr"]
    if ml:
        assert relmax(o.tensor("scalefactor"), r.alpha) < 1e-5
        # the reference's Gamma() is a polynomial (Gamma(1)=1.00001): compare loosely
        assert abs(o.cv_loglik(inp, targ) - cv["loglik"]) < 2e-4 * abs(cv["loglik"])


@pytest.mark.parametrize("ml,beta", CASES)
def test_oracle_matches_tiny_goldens_bit_exact(pyoracle, synth, ml, beta):
    g = np.load(os.path.join(GOLD, "tiny_net.npz"))
    ls, B = [int(x) for x in g["layersizes"]], int(g["bunch"])
    ws, bs = synth.make_weights(ls, seed=int(g["wseed"]))
    inp, targ = synth.make_frames(int(g["frames"]), 5, 3, seed=int(g["dseed"]))
    o = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
    o.train(inp, targ)
    key = "ml%d_b%s" % (ml, beta)
    w, b = o.get_weights()
    for l in range(4):
        assert np.array_equal(w[l], g["%s_W%d" % (key, l + 1)])
        assert np.array_equal(b[l], g["%s_b%d" % (key, l + 1)])
        assert np.array_equal(o.tensor("delta_w", l + 1), g["%s_dW%d" % (key, l + 1)])
    assert np.array_equal(o.tensor("dedx", 4, rows=B), g[key + "_dedx4"])
    if ml:
        assert np.array_equal(o.tensor("scalefactor"), g[key + "_alpha"])
    cv = g[key + "_cv"]
    assert o.cv_sqerr(inp, targ) == np.float32(cv[0])
    assert o.cv_abserr(inp, targ) == np.float32(cv[1])


@pytest.mark.parametrize("beta", [1.2, 1.0])
def test_oracle_matches_baseline_goldens(pyoracle, synth, beta):
    g = np.load(os.path.join(GOLD, "baseline_net.npz"))
    ls, B = synth.baseline_layersizes(), int(g["bunch"])
    ws, bs = synth.make_weights(ls)
    inp, targ = synth.make_frames(2 * B, 257, 11)
    o = pyoracle.OracleNet(ls, B, *HP, beta, 1, ws, bs)
    assert o.train(inp, targ) == 2
    key = "ml1_b%s" % beta
    w, b = o.get_weights()
    for l in range(4):
        idx = g["%s_idx%d" % (key, l + 1)]
        assert np.array_equal(w[l].ravel()[idx], g["%s_Wsamp%d" % (key, l + 1)])
        s = g["%s_Wsum%d" % (key, l + 1)]
        assert abs(w[l].astype(np.float64).sum() - s[0]) <= 1e-9 * s[1]
    assert np.array_equal(o.tensor("scalefactor"), g[key + "_alpha"])


def test_gamma_table(pyoracle):
    g = np.load(os.path.join(GOLD, "gamma.npz"))
    for x, want in zip(g["x"], g["gamma"]):
        got = pyoracle.gamma(float(x))
        assert got == want
        assert abs(got - math.gamma(float(x))) < 1e-4 * math.gamma(float(x))  # polynomial, BP_GPU.cu:597-607
    assert pyoracle.gamma(0.0) == 0.0 and pyoracle.gamma(-1.0) == 0.0


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_phase_split_equals_one_step(pyoracle, synth, ml, beta):
    """Two 'ranks' of 16 frames each, gradients and column sums added, == one 32-frame step
    (SURVEY.md 8e parity definition), up to summation order."""
    ls, Bl = [33, 24, 17, 11], 16
    ws, bs = synth.make_weights(ls, seed=8)
    inp, targ = synth.make_frames(2 * Bl, 11, 3, seed=9)
    single = pyoracle.OracleNet(ls, 2 * Bl, *HP, beta, ml, ws, bs)
    single.train_bunch(inp, targ)
    ranks = [pyoracle.OracleNet(ls, Bl, *HP, beta, ml, ws, bs) for _ in range(2)]
    cols = []
    for r, net in enumerate(ranks):
        net.forward(inp[r * Bl:(r + 1) * Bl])
        cols.append(net.loss_colsum(targ[r * Bl:(r + 1) * Bl]))
    colsum = cols[0] + cols[1]
    for r, net in enumerate(ranks):
        net.loss_grad(targ[r * Bl:(r + 1) * Bl], 2 * Bl, colsum)
        net.backward(inp[r * Bl:(r + 1) * Bl])
    for l in (1, 2, 3):
        g = ranks[0].tensor("grad_w", l) + ranks[1].tensor("grad_w", l)
        assert relmax(g, single.tensor("grad_w", l)) < 1e-5
        gb = ranks[0].tensor("grad_b", l) + ranks[1].tensor("grad_b", l)
        assert relmax(gb, single.tensor("grad_b", l)) < 1e-5
    if ml:
        assert relmax(ranks[0].tensor("scalefactor"), single.tensor("scalefactor")) < 1e-6


def test_train_skips_partial_bunch_and_empty(pyoracle, synth):
    ls, B = [15, 8, 5], 8
    ws, bs = synth.make_weights(ls, seed=1)
    inp, targ = synth.make_frames(2 * B + 5, 5, 3)
    a = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, ws, bs)
    b = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, ws, bs)
    assert a.train(inp, targ) == 2            # trailing 5 frames ignored, BP_GPU.cu:177-180
    assert b.train(inp[:2 * B], targ[:2 * B]) == 2
    for x, y in zip(a.get_weights()[0], b.get_weights()[0]):
        assert np.array_equal(x, y)
    assert a.train(inp[:0], targ[:0]) == 0
    assert a.train(inp[:3], targ[:3]) == 0


def test_zero_error_gradient_is_zero(pyoracle, synth):
    """e == 0 branch of kernSubClean2 / kernfunc2 (DevFunc.cu:388-391,479-482)."""
    ls, B = [6, 4], 4
    ws = [np.zeros((6, 4), np.float32)]
    bs = [np.zeros(4, np.float32)]
    inp = np.ones((B, 6), np.float32)
    targ = np.zeros((B, 4), np.float32)
    targ[1:, :] = 1.0  # row 0 has e == 0 exactly, others e = -1
    for ml, beta in CASES:
        o = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
        o.train_bunch(inp, targ)
        d = o.tensor("dedx", 1, rows=B)
        assert np.all(d[0] == 0) and np.all(np.isfinite(d)) and np.all(d[1:] < 0)


@pytest.mark.parametrize("shape", ["tiny", "baseline"])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2), (1, 1.0), (1, 0.9)])
def test_fma_contraction_variant_is_inside_the_gpu_tolerances(pyoracle, synth, ml, beta, shape):
    """Ambiguity (viii), oracle/mlggd_oracle.c: the reference is built with nvcc's default --fmad=true
    (TC/Makefile:30-33), so its elementwise kernels (kernUpdatedelta, TC/DevFunc.cu:502) and cuBLAS contract
    a*b+c into FMAs where the compiler sees fit; the oracle and the HIP build take the unfused reading
    (-ffp-contract=off).  The same oracle source built with -ffp-contract=fast -mfma is the other reading.
    Both must sit inside the tolerances the GPU parity tests use (weights 2e-5 of max|W|, CV 1e-4 relative) --
    here they are required to agree 10x tighter than that, so whichever reading the reference's binary
    embodies, the same parity verdict follows."""
    if shape == "tiny":
        ls, B, steps, dim, ctx = [15, 8, 8, 8, 5], 8, 3, 5, 3
    else:
        ls, B, steps, dim, ctx = synth.baseline_layersizes(), 128, 2, 257, 11
    ws, bs = synth.make_weights(ls, seed=3)
    rng = np.random.default_rng(103)   # non-zero biases, as in the GPU parity tests (tests/test_gpu_parity.py make_pair)
    bs = [rng.uniform(-0.1, 0.1, x.shape).astype(np.float32) for x in bs]
    inp, targ = synth.make_frames(steps * B, dim, ctx, seed=4)
    a = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
    b = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs, variant="fma")
    assert a.train(inp, targ) == steps and b.train(inp, targ) == steps
    wa, ba = a.get_weights()
    wb, bb = b.get_weights()
    worst = max(max(relmax(x, y) for x, y in zip(wa, wb)), max(relmax(x, y) for x, y in zip(ba, bb)))
    assert any(not np.array_equal(x, y) for x, y in zip(wa, wb))   # the two builds really differ
    assert worst < 2e-6, worst
    cin, ctarg = synth.make_frames(300, dim, ctx, seed=77)
    for f in ("cv_sqerr", "cv_abserr") + (("cv_loglik",) if ml else ()):
        x, y = getattr(a, f)(cin, ctarg), getattr(b, f)(cin, ctarg)
        assert abs(x - y) <= 1e-5 * abs(x), (f, x, y)
    if ml:
        assert relmax(a.tensor("scalefactor"), b.tensor("scalefactor")) < 1e-6
    print("fma vs strict, %s ml=%d beta=%.1f: weights %.1e" % (shape, ml, beta, worst))
    a.close()
    b.close()


@pytest.mark.parametrize("s_out", [1, 3, 7])
def test_mfma_order_twin_is_the_same_function_in_another_summation_order(pyoracle, synth, s_out):
    """oracle `ora_set_gemm_order(1, s_out)`: the HIP kernels' own summation order with fused multiply-adds (forward / dX
    over the 4 waves' contiguous ranges, the output layer over s_out slabs x 4 waves, dW over the frames in order;
    csrc/kernels.hip.h fwd_body / dx_body / dwp_body).  It must be the same function as the documented-order oracle up
    to rounding -- checked against the float64 model and against the default order -- and it must really be another
    order (different bits).  Widths that are not multiples of 32 or 4 exercise the pad handling of the ranges.
    On the GPU the HIP path equals this twin BIT FOR BIT (tests/test_gpu_mfma_order.py)."""
    ls, B = [3 * 37, 70, 45, 37], 16
    ws, bs = synth.make_weights(ls, seed=3)
    rng = np.random.default_rng(5)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(3 * B, 37, 3, seed=4)
    base = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
    assert base.train(inp, targ) == 3
    pyoracle.set_gemm_order("hip", s_out)
    try:
        twin = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
        assert twin.train(inp, targ) == 3
        cv_t = twin.cv_sqerr(inp, targ)
    finally:
        pyoracle.set_gemm_order("ref")
    r = Ref64(ls, *HP, 1.2, 1, ws, bs)
    for i in range(3):
        r.step(inp[i * B:(i + 1) * B], targ[i * B:(i + 1) * B])
    wt, bt = twin.get_weights()
    wb, bb = base.get_weights()
    for l in range(3):
        assert relmax(wt[l], r.W[l]) < 2e-6 and relmax(bt[l], r.b[l]) < 2e-5
        assert relmax(wt[l], wb[l]) < 2e-6
        assert relmax(twin.tensor("delta_w", l + 1), r.dW[l]) < 2e-5
    assert any(not np.array_equal(x, y) for x, y in zip(wt, wb))       # another order: other bits
    assert relmax(twin.tensor("scalefactor"), r.alpha) < 1e-5
    assert abs(cv_t - base.cv_sqerr(inp, targ)) <= 1e-5 * abs(cv_t)
    # exactly representable data: every order gives the same bits (small-integer inputs, weights in eighths)
    W = [(rng.integers(-4, 5, (ls[i], ls[i + 1])) * 0.125).astype(np.float32) for i in range(3)]
    W[0][:] = 0                                                        # hidden activations exactly 0.5
    W[1][:] = 0
    b0 = [np.zeros(n, np.float32) for n in ls[1:]]
    x = rng.integers(-3, 4, (B, ls[0])).astype(np.float32)
    t = rng.integers(-3, 4, (B, ls[3])).astype(np.float32)
    a = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, W, b0)
    a.train(x, t)
    pyoracle.set_gemm_order("hip", s_out)
    try:
        c = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, W, b0)
        c.train(x, t)
    finally:
        pyoracle.set_gemm_order("ref")
    assert np.array_equal(a.tensor("out", rows=B), c.tensor("out", rows=B))
    for o in (base, twin, a, c):
        o.close()


def test_mfma_order_twin_matches_its_goldens_bit_exact(pyoracle, synth):
    """tests/golden/mfma_order_twin.npz (oracle/make_golden.py twin()): the restated HIP summation order, with 4-wave
    layers, an output layer over 3 slabs, and a mixed per-layer kernel plan (64 x 64-tile layers = one chain)."""
    g = np.load(os.path.join(GOLD, "mfma_order_twin.npz"))
    ls, B = [15, 8, 8, 8, 5], 8
    for s_out, plan, key in ((1, None, "s1_w4"), (3, None, "s3_w4"), (1, [(1, 4), (1, 1), (4, 1), (4, 4)], "s1_plan")):
        ws, bs = synth.make_weights(ls, seed=3)
        inp, targ = synth.make_frames(3 * B, 5, 3, seed=4)
        pyoracle.set_gemm_order("hip", s_out, plan=plan)
        try:
            o = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
            o.train(inp, targ)
        finally:
            pyoracle.set_gemm_order("ref")
        w, b = o.get_weights()
        for l in range(4):
            assert np.array_equal(w[l], g["%s_W%d" % (key, l + 1)]), (key, l)
            assert np.array_equal(b[l], g["%s_b%d" % (key, l + 1)]), (key, l)
        assert np.array_equal(o.tensor("scalefactor"), g[key + "_alpha"])
        o.close()


@pytest.mark.parametrize("variant", ["strict", "fma"])
def test_blocked_gemms_equal_the_plain_loops_bit_for_bit(pyoracle, variant):
    """The oracle's GEMMs run register-blocked (oracle/mlggd_oracle.c, tile_4x16 / gemm_dx_blocked) for speed; the
    plain loops are the definition.  Same chain per output element => same bits, on ragged shapes (widths that are
    not multiples of 16 / 8 / 4, 1 and 7 frames), in the documented order, the split twins and the MFMA-order twin."""
    def run(ls, B, ml, beta, blocked, split=1, order="ref", s_out=1, plan=None):
        pyoracle.set_gemm_blocked(blocked, variant)
        pyoracle.set_gemm_split(split, variant)
        pyoracle.set_gemm_order(order, s_out, variant, plan)
        try:
            rng = np.random.default_rng(5)
            W = [(rng.standard_normal((ls[i], ls[i + 1])) * 0.1).astype(np.float32) for i in range(len(ls) - 1)]
            b = [(rng.standard_normal(ls[i + 1]) * 0.1).astype(np.float32) for i in range(len(ls) - 1)]
            net = pyoracle.OracleNet(ls, B, 0.05, 0.9, 1e-4, beta, ml, W, b, variant=variant)
            for _ in range(3):
                net.train_bunch(rng.standard_normal((B, ls[0])).astype(np.float32),
                                rng.standard_normal((B, ls[-1])).astype(np.float32))
            ws, bs = net.get_weights()
            res = ws + bs + [net.tensor(n, l) for l in range(1, len(ls)) for n in ("grad_w", "dedx", "y")]
            net.close()
            return res
        finally:
            pyoracle.set_gemm_blocked(True, variant)
            pyoracle.set_gemm_split(1, variant)
            pyoracle.set_gemm_order("ref", 1, variant)

    for ls, B in (([37, 50, 29, 19], 13), ([100, 17, 8, 257], 50), ([283, 96, 40, 31], 7), ([31, 1, 5, 3], 1),
                  ([33, 129, 65, 9], 128)):
        for ml, beta in ((0, 2.0), (1, 1.0), (1, 0.9)):
            for kw in ({}, {"split": 4}, {"split": 7}, {"order": "hip", "s_out": 7},
                       {"order": "hip", "s_out": 3, "plan": [(1, 1)] * (len(ls) - 1)}):
                a = run(ls, B, ml, beta, False, **kw)
                c = run(ls, B, ml, beta, True, **kw)
                for x, y in zip(a, c):
                    assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (ls, B, ml, beta, kw)


def test_exp_det_is_a_one_ulp_exponential_and_the_same_in_both_builds(pyoracle):
    """ora_exp_det: the sigmoid's exponential as the HIP kernels evaluate it (csrc/kernels.hip.h exp_det, the same
    statements; IEEE operations only, so the device returns the same bits: tests/test_gpu_loss_ulps.py).  Against
    float64 it is within 1 ulp wherever exp is a normal number, the sigmoid built on it is as close to the exact sigmoid as
    the one built on glibc's expf, the build with FMA contraction returns the same bits (the function opts out of it),
    and the edges behave: overflow to +inf, saturation at 7.4e-38 below -85.5, NaN in -> NaN out."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-85.5, 88.7, 2_000_000), rng.normal(0, 4, 1_000_000)]).astype(np.float32)
    e = pyoracle.exp_det(x)
    want = np.exp(x.astype(np.float64))
    ulp = np.ldexp(1.0, np.frexp(want)[1] - 1 - 23)
    err = np.abs(e.astype(np.float64) - want) / ulp
    assert err.max() < 1.0, err.max()                                   # measured 0.96
    assert np.array_equal(e.view(np.uint32), pyoracle.exp_det(x, variant="fma").view(np.uint32))
    s = pyoracle.exp_det(x, sigmoid=True)
    sw = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    su = np.ldexp(1.0, np.frexp(sw)[1] - 1 - 23)
    s_libm = (np.float32(1) / (np.float32(1) + np.exp(-x).astype(np.float32))).astype(np.float32)
    e_det, e_libm = np.abs(s - sw) / su, np.abs(s_libm - sw) / su
    assert e_det.max() < 2.6 and e_det.max() <= e_libm.max() + 0.5, (e_det.max(), e_libm.max())
    edge = pyoracle.exp_det(np.array([88.72283, 88.8, 1e9, -85.5, -87.4, -1e9, 0.0, np.nan], np.float32))
    assert np.isfinite(edge[0]) and edge[0] > 3.4e38 and np.isposinf(edge[1]) and np.isposinf(edge[2])
    assert edge[3] == edge[4] == edge[5] and 7.3e-38 < edge[3] < 7.5e-38
    assert edge[6] == 1.0 and np.isnan(edge[7])


def test_pow_det_is_the_correctly_rounded_power_but_for_a_few_in_a_million(pyoracle):
    """ora_pow_det: the loss chain's power as the HIP kernels evaluate it (csrc/kernels.hip.h pow_det, the same statements;
    IEEE double operations only, so the device returns the same bits: tests/test_gpu_loss_ulps.py).  Against the 80-bit
    long-double power it is never more than 1 ulp off and correctly rounded in all but ~5 of a million cases -- closer to
    the exact function than glibc's powf (the documented-order oracle's), let alone a 2-ulp CUDA powf; the build with FMA
    contraction returns the same bits; and the special cases follow pow()."""
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-7), np.log(30.0), 600000)), np.abs(rng.normal(0, 1, 400000))]).astype(np.float32)
    x = x[x > 0]
    tot = bad = 0
    for y in (0.9, 1.2, -0.1, 0.2, 1 / 0.9, 1 / 1.2, 2.0, 0.5, 3.0, -1.0):
        y32 = np.float32(y)
        got = pyoracle.pow_det(x, y32)
        want = np.power(x.astype(np.longdouble), np.longdouble(y32))
        cr = want.astype(np.float32)
        ulp = np.ldexp(np.longdouble(1), np.frexp(want)[1] - 1 - 23)
        assert float((np.abs(got.astype(np.longdouble) - want) / ulp).max()) <= 0.5001, y
        tot += x.size
        bad += int((got != cr).sum())
        assert np.array_equal(got.view(np.uint32), pyoracle.pow_det(x, y32, variant="fma").view(np.uint32))
    assert bad <= 2e-5 * tot, (bad, tot)                                 # measured 4.7e-6
    e = pyoracle.pow_det(np.array([0.0, 0.0, np.inf, np.inf, 2.0, 1e-45, 3e38, 1e-30], np.float32), 0.9)
    assert e[0] == 0 and np.isposinf(e[2]) and e[5] > 0 and np.isfinite(e[6])
    e = pyoracle.pow_det(np.array([0.0, np.inf, 2.0], np.float32), -0.1)
    assert np.isposinf(e[0]) and e[1] == 0 and 0.93 < e[2] < 0.94
    assert np.array_equal(pyoracle.pow_det(np.array([0.0, 5.0, np.inf], np.float32), 0.0), np.ones(3, np.float32))
    assert np.isposinf(pyoracle.pow_det(np.array([3e38], np.float32), 2.0)[0]) and pyoracle.pow_det(np.array([1e-30], np.float32), 2.0)[0] == 0
    assert np.isnan(pyoracle.pow_det(np.array([np.nan], np.float32), 0.9)[0])


@pytest.mark.parametrize("allreduce", [False, True])
def test_data_parallel_form_of_the_twin_is_the_same_step_in_another_order(pyoracle, synth, allreduce):
    """ora_set_dp_twin(world, allreduce): the MFMA-order twin as `world` ranks whose partial results meet in rank order (the
    ML statistic by k_colsum's wavefront reduction per rank; with the gradient all-reduce also the dW / bias chains).  It
    must be the same function as the plain twin up to rounding, must differ from it in bits (another order), and with one
    rank it must BE the plain twin.  (On the GPU the emulated worlds equal it bit for bit: tests/test_gpu_configs.py.)"""
    ls, B, world = [3 * 37, 70, 45, 37], 64, 4
    ws, bs = synth.make_weights(ls, seed=3)
    inp, targ = synth.make_frames(3 * B, 37, 3, seed=4)

    def run(dp_world, ar):
        pyoracle.set_gemm_order("hip", 3, dp_world=dp_world, dp_allreduce=ar)
        try:
            o = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
            assert o.train(inp, targ) == 3
            r = (o.get_weights(), o.tensor("scalefactor").copy())
            o.close()
            return r
        finally:
            pyoracle.set_gemm_order("ref")

    (w1, b1), a1 = run(1, False)
    (wd, bd), ad = run(world, allreduce)
    (w0, b0), a0 = run(1, allreduce)                                     # one rank: nothing to meet
    assert all(np.array_equal(x, y) for x, y in zip(w1 + b1, w0 + b0)) and np.array_equal(a1, a0)
    for x, y in zip(w1, wd):
        assert relmax(x, y) < 2e-6
    assert relmax(a1, ad) < 1e-5
    assert not np.array_equal(a1, ad)                                     # 4 x 16 frames by wavefront sums: other bits


def test_dropout_in_the_oracle(pyoracle, synth):
    """a25 (BP_GPU.cu:344-355, 484-501) in the oracle: omit-probabilities of 0 change nothing (bit for bit); otherwise the
    input / hidden activations are zeroed at the stated rates with NO rescale, another seed gives another mask, the same
    seed the same run; the CV pass equals a forward pass over weights scaled by the keep-probabilities, and its scale /
    unscale round trip leaves the weights within an ulp.  (The generator is the HIP engine's counter hash -- documented
    deviation from cuRAND -- and the GPU test requires the engine's bits: tests/test_gpu_mfma_order.py.)"""
    ls, B = [3 * 37, 96, 80, 37], 64
    ws, bs = synth.make_weights(ls, seed=3)
    inp, targ = synth.make_frames(2 * B, 37, 3, seed=4)
    inp = np.abs(inp) + 0.5

    def run(seed, vis, hid, steps=2):
        o = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, ws, bs, dropoutflag=1, visible_omit=vis, hid_omit=hid, random_seed=seed)
        assert o.train(inp[:steps * B], targ[:steps * B]) == steps
        return o

    plain = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, ws, bs)
    plain.train(inp, targ)
    z = run(7, 0.0, 0.0)
    assert all(np.array_equal(x, y) for x, y in zip(plain.get_weights()[0], z.get_weights()[0]))
    a, b, c = run(7, 0.2, 0.5), run(7, 0.2, 0.5), run(8, 0.2, 0.5)
    ya = a.tensor("y", 1, rows=B)
    assert 0.4 < float((ya == 0).mean()) < 0.6                          # hid_omit = 0.5; a sigmoid is never exactly 0
    kept = ya[ya != 0]
    assert kept.min() > 0 and kept.max() < 1                            # no rescale: still sigmoid values
    assert all(np.array_equal(x, y) for x, y in zip(a.get_weights()[0], b.get_weights()[0]))
    assert not np.array_equal(ya == 0, c.tensor("y", 1, rows=B) == 0)
    assert any(not np.array_equal(x, y) for x, y in zip(a.get_weights()[0], plain.get_weights()[0]))
    # CV: W * keep around each GEMM
    wa = [w.copy() for w in a.get_weights()[0]]
    out = a.cv_forward(inp[:B])
    scaled = pyoracle.OracleNet(ls, B, *HP, 2.0, 0, [w * np.float32(k) for w, k in zip(wa, (0.8, 0.5, 0.5))], a.get_weights()[1])
    assert np.array_equal(out, scaled.cv_forward(inp[:B]))
    for x, y in zip(a.get_weights()[0], wa):
        assert np.abs(x - y).max() <= 1.2e-7 * np.abs(y).max()
    for o in (plain, z, a, b, c, scaled):
        o.close()
