"""GPU: the three GEMMs of the HIP path against the oracle's MFMA-order twin, BIT FOR BIT, on ordinary random data
("separate summation order from libm instead of arguing it").

`pyoracle.set_gemm_order("hip", s_out)` restates on the CPU the order in which the HIP kernels sum
(csrc/kernels.hip.h): v_mfma_f32_32x32x2_f32 = two chained fused multiply-adds per instruction (k, then k+1); the
forward reduction cut into the 4 waves' contiguous ranges of k-pairs (output layer: s_out slabs x 4 waves, slabs added in
order by the loss kernel, then the bias); dX over quads of the reduction index, {4j, 4j+2} then {4j+1, 4j+3}, 4 waves;
dW one chain per weight over the frames in order.  If the kernels equal that model bit for bit on data that is NOT
exactly representable, then the GEMMs of the HIP path are pinned to a CPU model, and everything that still separates
the HIP path from the oracle's documented order is (a) that re-association, which the twin makes measurable on the CPU
alone, and (b) libm -- expf in the sigmoid, powf in the loss chain (tests/test_gpu_loss_ulps.py).

Every case avoids libm on purpose: one-layer nets (no sigmoid), MMSE with beta = 2 (the gradient 2 e / n is IEEE), and for
dX a first layer with zero weights and biases, whose activations are exactly 0.5 on both sides (1 / (1 + expf(-0)))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HP = (0.1, 0.9, 1e-5)


def ulp_report(name, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    print("%s: %d of %d elements differ, max %d ulp" % (name, int((d > 0).sum()), d.size, int(d.max()) if d.size else 0))
    return int((d > 0).sum())


def twin_net(pyoracle, eng, *args):
    pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan())
    return pyoracle.OracleNet(*args)


@pytest.mark.parametrize("s_out_env", [None, "1", "3"])
@pytest.mark.parametrize("K,D,B", [(2048, 257, 128), (531, 257, 128), (2827, 2048, 64), (300, 40, 256)])
def test_forward_and_dw_of_a_one_layer_net_equal_the_twin_bitwise(pkg, pyoracle, monkeypatch, K, D, B, s_out_env):
    """`out` (forward, output-layer order: s_out slabs x 4 waves; MLGGD_S_OUT=1 is exactly the hidden layers' order: 4
    waves, no slabs -- same main loop, same split) and, after one MMSE step, dEdX, delta_w / delta_b and the updated W / b
    (dW chain over the frames + the fused update, all IEEE)."""
    if s_out_env is not None:
        monkeypatch.setenv("MLGGD_S_OUT", s_out_env)
    rng = np.random.default_rng(K + D + B)
    W = rng.normal(0, 0.05, (K, D)).astype(np.float32)
    b = rng.normal(0, 0.1, D).astype(np.float32)
    x = rng.normal(0, 1, (2 * B, K)).astype(np.float32)
    t = rng.normal(0, 1, (2 * B, D)).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K, D], B, *HP, [W], [b], 2.0, 0)
    if s_out_env is not None:
        assert eng.out_slabs() == int(s_out_env)
    try:
        ora = twin_net(pyoracle, eng, [K, D], B, *HP, 2.0, 0, [W], [b])
        assert eng.train(x[:B], t[:B]) == 1 and ora.train(x[:B], t[:B]) == 1
        bad = ulp_report("out (s_out %d)" % eng.out_slabs(), eng.debug_tensor("out"), ora.tensor("out", rows=B))
        bad += ulp_report("dEdX", eng.debug_tensor("dedx", 1), ora.tensor("dedx", 1, rows=B))
        bad += ulp_report("delta_w", eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1))
        bad += ulp_report("delta_b", eng.debug_tensor("delta_b", 1), ora.tensor("delta_b", 1))
        assert eng.train(x[B:], t[B:]) == 1 and ora.train(x[B:], t[B:]) == 1    # a second step: momentum, weight decay
        we, be = eng.returnWeights()
        wo, bo = ora.get_weights()
        bad += ulp_report("W after 2 steps", we[0], wo[0]) + ulp_report("b after 2 steps", be[0], bo[0])
        # and the default order really is another one (the twin is not vacuous)
        pyoracle.set_gemm_order("ref")
        ref = pyoracle.OracleNet([K, D], B, *HP, 2.0, 0, [W], [b])
        ref.train(x[:B], t[:B])
        differs = not np.array_equal(ref.tensor("out", rows=B), ora.tensor("out", rows=B))
        ref.close()
        ora.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()
    assert bad == 0
    assert differs


@pytest.mark.parametrize("H,D,B", [(96, 257, 128), (160, 2048, 128), (64, 1000, 64)])
def test_dx_equals_the_twin_bitwise(pkg, pyoracle, H, D, B):
    """dX with the reduction over D output units (quads, 4 waves, pad columns): a first layer with zero weights and biases
    has activations of exactly 0.5 on both sides, so dEdX_1 = 0.25 * (dEdX_2 . W_2^T) compares bit for bit."""
    K0 = 64
    rng = np.random.default_rng(H + D)
    W1, b1 = np.zeros((K0, H), np.float32), np.zeros(H, np.float32)
    W2 = rng.normal(0, 0.05, (H, D)).astype(np.float32)
    b2 = rng.normal(0, 0.1, D).astype(np.float32)
    x = rng.normal(0, 1, (B, K0)).astype(np.float32)
    t = rng.normal(0, 1, (B, D)).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K0, H, D], B, *HP, [W1, W2], [b1, b2], 2.0, 0)
    try:
        ora = twin_net(pyoracle, eng, [K0, H, D], B, *HP, 2.0, 0, [W1, W2], [b1, b2])
        assert eng.train(x, t) == 1 and ora.train(x, t) == 1
        assert np.all(eng.debug_tensor("y", 1) == 0.5) and np.all(ora.tensor("y", 1, rows=B) == 0.5)
        bad = ulp_report("out", eng.debug_tensor("out"), ora.tensor("out", rows=B))
        bad += ulp_report("dEdX_2", eng.debug_tensor("dedx", 2), ora.tensor("dedx", 2, rows=B))
        bad += ulp_report("dEdX_1 (dX over %d units)" % D, eng.debug_tensor("dedx", 1), ora.tensor("dedx", 1, rows=B))
        bad += ulp_report("delta_w_1", eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1))
        bad += ulp_report("delta_w_2", eng.debug_tensor("delta_w", 2), ora.tensor("delta_w", 2))
        ora.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()
    assert bad == 0


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.0), (1, 1.2), (1, 0.9), (1, 2.0), (0, 1.0)])
def test_a_whole_run_equals_the_twin_bit_for_bit(pkg, pyoracle, synth, ml, beta):
    """Every loss configuration -- MMSE, the shipped ML-GGD beta = 1 (TC/finetune.pl:25-26), BASELINE's beta = 1.2, the
    paper's 0.9, ...: the GEMMs equal the twin's (above), the sigmoid's exponential (exp_det) and the loss chain's power
    (pow_det) are IEEE operations only and restated statement for statement in the twin, the rest of a step is IEEE in a
    fixed order.  So 60 steps at 2827-2048^3-257 on ordinary data leave EXACTLY the twin's weights, biases, momentum,
    alpha and CV numbers -- every bit of 14.7 M weights -- and the activations of the last step.  Nothing but the
    summation order and the choice of a <= 1-ulp exponential / power -- which cuBLAS and CUDA's libm leave open --
    separates the HIP path from the documented-order oracle; any indexing slip, race or stale operand anywhere in a step
    would show here."""
    ls, B, steps = synth.baseline_layersizes(), 128, 60
    ws, bs = synth.make_weights(ls)
    inp, targ = synth.make_frames(steps * B, 257, 11)
    cin, ctarg = synth.make_frames(1000, 257, 11, seed=77)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    try:
        twin = twin_net(pyoracle, eng, ls, B, *HP, beta, ml, ws, bs)
        assert eng.train(inp, targ) == steps and twin.train(inp, targ) == steps
        for l in (1, 2, 3):                                                   # the last step's activations and output
            assert np.array_equal(eng.debug_tensor("y", l), twin.tensor("y", l, rows=B)), ("y", l)
        assert np.array_equal(eng.debug_tensor("out"), twin.tensor("out", rows=B))
        we, be = eng.returnWeights()
        wt, bt = twin.get_weights()
        for l in range(len(we)):
            assert np.array_equal(we[l], wt[l]), ("weights", l + 1)
            assert np.array_equal(be[l], bt[l]), ("bias", l + 1)
            assert np.array_equal(eng.debug_tensor("delta_w", l + 1), twin.tensor("delta_w", l + 1)), ("delta", l + 1)
        if ml:
            assert np.array_equal(eng.scalefactor(), twin.tensor("scalefactor"))
        sq, ab, ll = eng.cv_all(cin, ctarg)
        assert sq == twin.cv_sqerr(cin, ctarg) and ab == twin.cv_abserr(cin, ctarg)
        if ml:
            assert ll == twin.cv_loglik(cin, ctarg)
        assert any(not np.array_equal(a, b) for a, b in zip(we, ws))          # the net was trained
        pyoracle.set_gemm_order("ref")                                        # and the documented order gives other bits
        ref = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
        ref.train(inp[:B], targ[:B])
        assert not np.array_equal(ref.get_weights()[0][0], wt[0])
        ref.close()
        twin.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()


@pytest.mark.parametrize("K,D,B", [(2048, 2048, 64), (2827, 192, 128), (96, 64, 256)])
def test_the_64x64_tile_forward_equals_the_twin_bitwise(pkg, pyoracle, monkeypatch, K, D, B):
    """k_fwd64 (csrc/kernels64.hip.h): one chain per output element over k ascending.  Read back through an output layer
    wide enough for that tiling (MLGGD_TILE64=2 forces the form wherever the shape divides; one slab, so `out` is the
    kernel's sum + bias): bit-identical to the twin with waves = 1 -- on ordinary data, K not a multiple of 128 included
    (2827 -> 89 chunks: the remainder bodies of the 4-chunk ring)."""
    monkeypatch.setenv("MLGGD_TILE64", "2")
    monkeypatch.setenv("MLGGD_S_OUT", "1")
    rng = np.random.default_rng(K + D + B)
    W = rng.normal(0, 0.05, (K, D)).astype(np.float32)
    b = rng.normal(0, 0.1, D).astype(np.float32)
    x = rng.normal(0, 1, (B, K)).astype(np.float32)
    t = rng.normal(0, 1, (B, D)).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K, D], B, *HP, [W], [b], 2.0, 0)
    assert eng.gemm_plan() == [(1, 4)] and eng.out_slabs() == 1          # the 64 x 64 form is what runs
    try:
        ora = twin_net(pyoracle, eng, [K, D], B, *HP, 2.0, 0, [W], [b])
        assert eng.train(x, t) == 1 and ora.train(x, t) == 1
        bad = ulp_report("out (k_fwd64)", eng.debug_tensor("out"), ora.tensor("out", rows=B))
        bad += ulp_report("delta_w", eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1))
        pyoracle.set_gemm_order("hip", 1)                                # the 4-wave order is a different one
        four = pyoracle.OracleNet([K, D], B, *HP, 2.0, 0, [W], [b])
        four.train(x, t)
        differs = not np.array_equal(four.tensor("out", rows=B), ora.tensor("out", rows=B))
        four.close()
        ora.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()
    assert bad == 0 and differs


@pytest.mark.parametrize("H,D,B", [(128, 2048, 64), (64, 257, 128), (192, 1000, 64)])
def test_the_64x64_tile_dx_equals_the_twin_bitwise(pkg, pyoracle, monkeypatch, H, D, B):
    """k_dx64: one chain per output element over the reduction index in quads ({4j, 4j+2} then {4j+1, 4j+3}); pad columns
    (257 -> 288, 1000 -> 1024) and a chunk count that is not a multiple of 4 (288 / 32 = 9) included."""
    monkeypatch.setenv("MLGGD_TILE64", "2")
    K0 = 64
    rng = np.random.default_rng(H + D + 1)
    W1, b1 = np.zeros((K0, H), np.float32), np.zeros(H, np.float32)      # y_1 = 0.5 exactly on both sides
    W2 = rng.normal(0, 0.05, (H, D)).astype(np.float32)
    b2 = rng.normal(0, 0.1, D).astype(np.float32)
    x = rng.normal(0, 1, (B, K0)).astype(np.float32)
    t = rng.normal(0, 1, (B, D)).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K0, H, D], B, *HP, [W1, W2], [b1, b2], 2.0, 0)
    assert eng.gemm_plan()[1][1] == 1 and eng.gemm_plan()[0][0] == 1    # dX of layer 2 and the forward of layer 1: 64 x 64 form
    try:
        ora = twin_net(pyoracle, eng, [K0, H, D], B, *HP, 2.0, 0, [W1, W2], [b1, b2])
        assert eng.train(x, t) == 1 and ora.train(x, t) == 1
        assert np.all(eng.debug_tensor("y", 1) == 0.5) and np.all(ora.tensor("y", 1, rows=B) == 0.5)
        bad = ulp_report("out", eng.debug_tensor("out"), ora.tensor("out", rows=B))
        bad += ulp_report("dEdX_1 (k_dx64 over %d units)" % D, eng.debug_tensor("dedx", 1), ora.tensor("dedx", 1, rows=B))
        bad += ulp_report("delta_w_1", eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1))
        bad += ulp_report("delta_w_2", eng.debug_tensor("delta_w", 2), ora.tensor("delta_w", 2))
        ora.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()
    assert bad == 0


def test_hidden_layers_through_the_64x64_tile_kernels_match_the_oracle(pkg, pyoracle, synth, monkeypatch):
    """Three sigmoid layers, ML-GGD, every hidden forward and dX through k_fwd64 / k_dx64 (forced at a small shape), two
    steps against the documented-order oracle at the usual tolerances, both activation layouts checked, and against the
    twin: activations within the sigmoid's 2 ulp of it after the first layer."""
    monkeypatch.setenv("MLGGD_TILE64", "2")
    ls, B = [40 * 5, 192, 128, 64, 40], 128
    ws, bs = synth.make_weights(ls, seed=21)
    rng = np.random.default_rng(22)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(2 * B, 40, 5, seed=23)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 1.2, 1)
    assert eng.gemm_plan() == [(1, 4), (1, 1), (1, 1), (4, 1)]
    ora = pyoracle.OracleNet(ls, B, *HP, 1.2, 1, ws, bs)
    try:
        twin = twin_net(pyoracle, eng, ls, B, *HP, 1.2, 1, ws, bs)
        assert eng.train(inp[:B], targ[:B]) == 1 and twin.train(inp[:B], targ[:B]) == 1
        pyoracle.set_gemm_order("ref")
        assert ora.train(inp[:B], targ[:B]) == 1
        y1 = eng.debug_tensor("y", 1)
        d = np.abs(y1.view(np.int32).astype(np.int64) - twin.tensor("y", 1, rows=B).view(np.int32).astype(np.int64))
        assert d.max() <= 2, d.max()
        assert eng.train(inp[B:], targ[B:]) == 1 and ora.train(inp[B:], targ[B:]) == 1
        for l in (1, 2, 3):   # the two layouts every epilogue writes hold the same bits
            assert np.array_equal(eng.debug_tensor("y", l), eng.debug_tensor("yt", l)), l
            assert np.array_equal(eng.debug_tensor("dedx", l), eng.debug_tensor("dedxt", l)), l
        we, be = eng.returnWeights()
        wo, bo = ora.get_weights()
        for l in range(4):
            assert float(np.abs(we[l] - wo[l]).max() / np.abs(wo[l]).max()) < 2e-5, l
            assert float(np.abs(be[l] - bo[l]).max() / np.abs(bo[l]).max()) < 2e-5, l
            for name in ("dedx", "y"):
                if name == "y" and l == 3:
                    continue
                a, r = eng.debug_tensor(name, l + 1), ora.tensor(name, l + 1, rows=B)
                assert float(np.abs(a - r).max() / np.abs(r).max()) < 2e-4, (name, l)
        twin.close()
    finally:
        pyoracle.set_gemm_order("ref")
        ora.close()
        eng.close()


@pytest.mark.parametrize("K,D,B", [(300, 40, 256), (2827, 257, 512)])
def test_dw_over_256_and_512_frames_equals_the_twin_bitwise(pkg, pyoracle, K, D, B):
    """k_dwp<4> / k_dwp<8> (256 / 512 frames): one chain per weight over the frames in order + the fused update: delta_w /
    delta_b / W / b of a one-layer MMSE net after two steps against the MFMA-order twin, bit for bit."""
    rng = np.random.default_rng(K + D + B + 7)
    W = rng.normal(0, 0.05, (K, D)).astype(np.float32)
    b = rng.normal(0, 0.1, D).astype(np.float32)
    x = rng.normal(0, 1, (2 * B, K)).astype(np.float32)
    t = rng.normal(0, 1, (2 * B, D)).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K, D], B, *HP, [W], [b], 2.0, 0)
    try:
        ora = twin_net(pyoracle, eng, [K, D], B, *HP, 2.0, 0, [W], [b])
        assert eng.train(x, t) == 2 and ora.train(x, t) == 2
        bad = ulp_report("delta_w", eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1))
        bad += ulp_report("delta_b", eng.debug_tensor("delta_b", 1), ora.tensor("delta_b", 1))
        we, be = eng.returnWeights()
        wo, bo = ora.get_weights()
        bad += ulp_report("W after 2 steps", we[0], wo[0]) + ulp_report("b after 2 steps", be[0], bo[0])
        ora.close()
    finally:
        pyoracle.set_gemm_order("ref")
        eng.close()
    assert bad == 0


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.0)])
def test_dropout_runs_equal_the_twin_bit_for_bit(pkg, pyoracle, synth, ml, beta):
    """a25 (BP_GPU.cu:344-355, 484-501).  The reference's cuRAND stream cannot be matched (documented deviation); the engine
    draws its uniforms from a counter hash of (seed, step, layer, element), and the oracle restates THAT generator -- so a
    dropout run is comparable at all: the same masks on both sides, 6 training steps leave the twin's weights in every bit
    (and the documented-order oracle's to 2e-5), a CV forward pass (weights scaled by the keep-probability around each
    GEMM) returns the twin's outputs in every bit, and the scale / unscale round trip leaves the same perturbed weights."""
    ls, B, steps = [257 * 3, 256, 192, 257], 128, 6
    ws, bs = synth.make_weights(ls, seed=5)
    inp, targ = synth.make_frames(steps * B, 257, 3, seed=6)
    kw = dict(dropoutflag=1, visible_omit=0.2, hid_omit=0.5)
    eng = pkg.BPGpu(77, 0, ls, B, *HP, ws, bs, beta, ml, **kw)
    ref = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs, random_seed=77, **kw)
    try:
        pyoracle.set_gemm_order("hip", eng.out_slabs(), plan=eng.gemm_plan())
        twin = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs, random_seed=77, **kw)
        assert eng.train(inp, targ) == steps and twin.train(inp, targ) == steps
        for l in (1, 2):
            y = eng.debug_tensor("y", l)
            assert np.array_equal(y, twin.tensor("y", l, rows=B)), l
            assert 0.45 < float((y == 0).mean()) < 0.55                      # hid_omit = 0.5 took effect
        we, be = eng.returnWeights()
        wt, bt = twin.get_weights()
        for a, b in zip(list(we) + list(be), list(wt) + list(bt)):
            assert np.array_equal(a, b)
        out = eng.forward(inp[:300])                                        # CV: W * keep, GEMM, W * (1 / keep) -- per bunch,
        want = np.concatenate([twin.cv_forward(inp[i:min(i + B, 300)]) for i in range(0, 300, B)])   # as CrossValid loops them
        assert np.array_equal(out, want)
        for a, b in zip(eng.returnWeights()[0], twin.get_weights()[0]):      # the round trip's rounding, bit for bit
            assert np.array_equal(a, b)
        assert any(not np.array_equal(a, b) for a, b in zip(eng.returnWeights()[0], we))   # ... and it is not the identity
        pyoracle.set_gemm_order("ref")
        assert ref.train(inp, targ) == steps
        for a, b in zip(we, ref.get_weights()[0]):
            assert float(np.abs(a - b).max() / np.abs(b).max()) < 2e-5
        twin.close()
    finally:
        pyoracle.set_gemm_order("ref")
        ref.close()
        eng.close()
