"""Where can the HIP loss chain differ from the oracle's AT ALL, given the same output-layer activations?

VERDICT r02 asked for the origin of the beta = 0.9 distance (epoch-horizon CV numbers 1e-4 from the oracle where
the oracle's own FMA twin sits 4e-5 away).  The loss chain (TC/BP_GPU.cu:413-423: kernerror, kernabsolutevalus,
kernindex2, kernSumcol, kernDivide, kernVecMulNum, kernindex2, kernfunc2, kernVecMulNum; TC/DevFunc.cu:219-227,
468-489) is IEEE-exact arithmetic in a fixed order on both sides EXCEPT its powf calls: the device's ocml powf and the
oracle's glibc powf are different implementations of a function that neither rounds correctly.  So:

  1. the device's powf / expf against the correctly rounded result and against glibc, in ulps;
  2. the whole chain on an output layer whose activations are EXACTLY representable (small-integer inputs and
     weights: every product and partial sum is exact in fp32, so `out` is bit-identical on both sides whatever the
     GEMM summation order): dEdX_L and alpha in ulps, for the one-launch kernel (k_loss_ml) and for the
     data-parallel pair (k_loss_err + k_loss_grad).

Measured (MI355X, ROCm 7.2, r03): see the bounds below and DESIGN.md section 2.  Parity of the powf calls themselves
is UNPINNED against the reference (CUDA's powf is a third implementation; no reference output exists)."""
import ctypes
import ctypes.util

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HP = (0.1, 0.9, 1e-5)


def ulp_dist(a, b):
    """distance in units in the last place between two float32 arrays (same sign assumed or zero)"""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def glibc_powf(x, y):
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    m.powf.restype = ctypes.c_float
    m.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    return np.array([m.powf(float(v), float(y)) for v in x], np.float32)


def tiny_engine(pkg, synth):
    ls = [32, 32]
    ws, bs = synth.make_weights(ls, seed=1)
    return pkg.BPGpu(1, 0, ls, 32, *HP, ws, bs, 2.0, 0)


def test_device_powf_and_expf_in_ulps(pkg, pyoracle, synth):
    eng = tiny_engine(pkg, synth)
    rng = np.random.default_rng(0)
    # |e| of a normalised LPS regression: mostly 1e-3 .. 10, tails down to 1e-7
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-7), np.log(30.0), 60000)), np.abs(rng.normal(0, 1, 60000))]).astype(np.float32)
    report = []
    for y in (0.9, 1.2, 0.9 - 1.0, 1.2 - 1.0, 1.0 / 0.9, 1.0 / 1.2, 2.0):
        y32 = np.float32(y)
        dev = eng.debug_math("powf", x, y32)
        exact = np.power(x.astype(np.float64), np.float64(y32)).astype(np.float32)
        host = glibc_powf(x[:20000], y32)
        d_dev, d_host, d_dh = ulp_dist(dev, exact), ulp_dist(host, exact[:20000]), ulp_dist(dev[:20000], host)
        report.append((y, int(d_dev.max()), float((d_dev > 0).mean()), int(d_host.max()), float((d_host > 0).mean()),
                       int(d_dh.max()), float((d_dh > 0).mean())))
        assert d_dev.max() <= 2, (y, d_dev.max())        # ocml powf: measured 1 ulp max, 11-23 % of the values 1 ulp off
        assert d_host.max() <= 1, (y, d_host.max())      # glibc powf: measured correctly rounded in 99.9 % of the cases
    # y = 1 and y = 0 (the shipped objective, MLflag = 1 shapefactor = 1, TC/finetune.pl:25-26, meets exactly these:
    # |e|^1, v2^(1/1), alpha^1, |e|^0): the correctly rounded results are x and 1.  glibc returns them; what ocml's
    # powf returns is REPORTED here, not relied on -- the kernels never call powf for these exponents (pow_or_self,
    # kernels.hip.h), so the beta = 1 chain is bit-identical to the oracle's (test below).
    for y, want in ((1.0, x), (0.0, np.ones_like(x))):
        dev = eng.debug_math("powf", x, np.float32(y))
        d = ulp_dist(dev, want)
        host = glibc_powf(x[:20000], np.float32(y))
        assert np.array_equal(host, want[:20000])        # glibc: exact
        print("powf(x, %.1f): device vs the exact result max %d ulp (%.1f %% differ); glibc exact"
              % (y, d.max(), 100 * (d > 0).mean()))
        assert d.max() <= 2
    # pow_det (kernels.hip.h): what the loss kernels call since r04 instead of ocml's powf -- IEEE double operations only,
    # restated statement for statement in the oracle (ora_pow_det): same bits on both sides for every argument, and the
    # correctly rounded power in all but a few of a million cases
    xs = np.concatenate([x, [0.0, 1e-45, 1e-38, 1.0, 3e38, np.inf]]).astype(np.float32)
    for y in (0.9, 1.2, 0.9 - 1.0, 1.2 - 1.0, 1.0 / 0.9, 1.0 / 1.2, 2.0, 0.5, -1.0, 0.0, 1.0):
        y32 = np.float32(y)
        dev = eng.debug_math("pow_det", xs, y32)
        assert np.array_equal(dev.view(np.uint32), pyoracle.pow_det(xs, y32).view(np.uint32)), y
        with np.errstate(divide="ignore", over="ignore"):
            exact = np.power(xs[:-5].astype(np.float64), np.float64(y32)).astype(np.float32)
        d = ulp_dist(dev[:-5], exact)
        assert d.max() <= 1, (y, d.max())
        if y in (0.9, 1.2, 2.0):
            print("pow_det(x, %.1f): device = CPU restatement in every bit; vs float64 pow rounded: max %d ulp (%.4f %% differ)"
                  % (y, d.max(), 100 * (d > 0).mean()))
    for r in report:
        print("powf(x, %+.4f): device vs exact max %d ulp (%.1f %% differ) | glibc vs exact max %d ulp (%.1f %%) | "
              "device vs glibc max %d ulp (%.1f %% differ)" % (r[0], r[1], 100 * r[2], r[3], 100 * r[4], r[5], 100 * r[6]))
    # the sigmoid of the hidden layers (kernSigmoid, DevFunc.cu:48) -- not part of the loss chain, same question.  Since
    # r04 the kernels evaluate its exponential in IEEE operations only (kernels.hip.h exp_det; ocml's expf is called
    # nowhere), restated statement for statement in the oracle (ora_exp_det): the two must agree in EVERY bit -- which is
    # what lets whole training runs equal the oracle's MFMA-order twin (tests/test_gpu_mfma_order.py) -- and how far that
    # exponential sits from the correctly rounded one is measured, not assumed
    v = np.concatenate([rng.normal(0, 4, 100000), rng.uniform(-100, 100, 20000),
                        [0.0, -0.0, 88.72283, 88.8, -85.5, -87.4, -200.0, 200.0, 1e-30, -1e-30]]).astype(np.float32)
    for fn, sig in (("exp_det", False), ("sigmoid", True), ("sigmoid4", True)):   # sigmoid4: the epilogues' packed two-at-a-time form
        assert np.array_equal(eng.debug_math(fn, v).view(np.uint32), pyoracle.exp_det(v, sigmoid=sig).view(np.uint32)), fn
    inr = np.abs(v) < 85.5
    d = ulp_dist(eng.debug_math("exp_det", v)[inr], np.exp(v[inr].astype(np.float64)).astype(np.float32))
    d_ocml = ulp_dist(eng.debug_math("expf", v)[inr], np.exp(v[inr].astype(np.float64)).astype(np.float32))
    print("exp_det: device vs correctly rounded max %d ulp (%.1f %% differ) | ocml expf max %d ulp (%.1f %% differ)"
          % (d.max(), 100 * (d > 0).mean(), d_ocml.max(), 100 * (d_ocml > 0).mean()))
    assert d.max() <= 1                                  # < 0.96 ulp of the exact value (tests/test_oracle.py) = at most 1 from its rounding
    dev = eng.debug_math("sigmoid", v)
    with np.errstate(over="ignore"):
        e64 = np.exp(-v.astype(np.float64)).astype(np.float32)  # expf correctly rounded, then the two IEEE operations
    exact = (np.float32(1) / (np.float32(1) + e64)).astype(np.float32)
    d = ulp_dist(dev[inr], exact[inr])
    print("sigmoid: device vs (correctly rounded expf, IEEE add / divide) max %d ulp (%.1f %% differ)" % (d.max(), 100 * (d > 0).mean()))
    assert d.max() <= 2                                  # one ulp of the exponential through an add and a divide
    # x / y is IEEE-exact on the device (hipcc's default correctly rounded fp32 division)
    q = eng.debug_math("div", x, 3.7)
    assert np.array_equal(q, (x / np.float32(3.7)).astype(np.float32))
    eng.close()


@pytest.mark.parametrize("beta", [0.9, 1.0, 1.2, 2.0])
@pytest.mark.parametrize("fused", [True, False])
def test_loss_chain_on_an_exactly_representable_output_layer(pkg, pyoracle, beta, fused, monkeypatch):
    """fused: k_loss_ml (single device, one launch); not fused: k_loss_err + k_loss_grad (the data-parallel pair)."""
    monkeypatch.setenv("MLGGD_LOSS_FUSE", "1" if fused else "0")
    K, D, B = 96, 257, 128
    ls = [K, D]
    rng = np.random.default_rng(7)
    W = (rng.integers(-4, 5, (K, D)) * 0.125).astype(np.float32)   # multiples of 1/8, |w| <= 1/2
    b = (rng.integers(-8, 9, D) * 0.25).astype(np.float32)
    x = rng.integers(-3, 4, (B, K)).astype(np.float32)             # small integers
    targ = rng.normal(0, 1.5, (B, D)).astype(np.float32)
    targ[3, 5] = np.float32((x[3] @ W[:, 5]) + b[5])               # one exact hit: e == 0 -> g = 0 (kernfunc2's middle branch)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, [W], [b], beta, 1)
    ora = pyoracle.OracleNet(ls, B, *HP, beta, 1, [W], [b])
    assert eng.train(x, targ) == 1 and ora.train(x, targ) == 1
    out_e, out_o = eng.debug_tensor("out"), ora.tensor("out", rows=B)
    assert np.array_equal(out_e, out_o)                            # exact activations: bit-identical whatever the order
    g_e, g_o = eng.debug_tensor("dedx", 1), ora.tensor("dedx", 1, rows=B)
    gt_e = eng.debug_tensor("dedxt", 1)
    assert np.array_equal(g_e, gt_e)                               # both layouts of the gradient hold the same bits
    a_e, a_o = eng.scalefactor(), ora.tensor("scalefactor")
    dg, da = ulp_dist(g_e, g_o), ulp_dist(a_e, a_o)
    assert g_e[3, 5] == 0.0 and g_o[3, 5] == 0.0
    assert np.array_equal(np.sign(g_e), np.sign(g_o))
    print("beta %.1f %s: dEdX_L max %d ulp (%.1f %% of elements differ, mean %.2f ulp) | alpha max %d ulp (%.1f %% differ)"
          % (beta, "k_loss_ml" if fused else "k_loss_err+k_loss_grad", dg.max(), 100 * (dg > 0).mean(), dg.mean(), da.max(),
             100 * (da > 0).mean()))
    # Measured r03 (MI355X, ROCm 7.2): dEdX_L max 5-6 ulp, mean 0.7-1.0 ulp, about half of the elements differ at all;
    # alpha max 1 (beta 2) .. 3 ulp.  Note powf(x, 2) is NOT x*x on the device (1 ulp off in 23 % of the cases), so
    # beta = 2 with MLflag = 1 is no exception; MLflag = 0 with beta = 2 takes pow(x, 1) = x and is bit-exact.
    if beta == 1.0:
        # the shipped objective: no libm call left in the chain (pow_or_self), kernSumcol's order kept -> same bits
        assert dg.max() == 0 and da.max() == 0
    else:
        # The bound, derived rather than chosen.  D = 1 ulp is the measured device-vs-glibc distance of ONE powf call
        # (first test of this file).  g = P * beta / q * inv_n with P = powf(|e|, beta-1), q = powf(alpha, beta):
        #   P: D;   q: alpha is da ulp off (measured in THIS run, below), powf scales a relative input error by its
        #   exponent, and a relative error is worth up to 2x as many ulps on the other side of a binade boundary
        #   -> 2 * beta * da + D;   the mul / div / mul that follow are IEEE on both sides but act on perturbed inputs:
        #   +1 ulp each for the two whose inputs differ.
        # alpha itself: v2 is a sum of 128 positive terms each within D -> within 2 D ulp; powf(v2, 1/beta) -> 2/beta * 2D + D.
        D = 1
        bound_a = int(np.ceil(4.0 * D / beta + D))
        bound_g = int(np.ceil(D + 2.0 * beta * int(da.max()) + D + 2))
        assert da.max() <= bound_a, (int(da.max()), bound_a)
        assert dg.max() <= bound_g, (int(dg.max()), bound_g)
        assert dg.mean() <= 1.5
    # against the oracle's MFMA-order twin (the same IEEE-only pow_det on both sides) the chain is the same bits at EVERY beta
    pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan())
    try:
        tw = pyoracle.OracleNet(ls, B, *HP, beta, 1, [W], [b])
        assert tw.train(x, targ) == 1
        assert np.array_equal(g_e, tw.tensor("dedx", 1, rows=B)) and np.array_equal(a_e, tw.tensor("scalefactor"))
        tw.close()
    finally:
        pyoracle.set_gemm_order("ref")
    eng.close()
    ora.close()
