"""GPU parity tests: the HIP path (through the C-ABI, libmlggd.so) against the CPU oracle on
the same seeded inputs.  Tolerances (fp32, different but fixed summation orders; the
reference's own cuBLAS order is unspecified):
  activations / gradients  rtol 2e-4 of the tensor's max magnitude
  weights after k steps    rtol 2e-5 of max |W| (updates are ~1e-3 of |W|)
  CV metrics               1e-4 relative (the north_star tolerance)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [(0, 2.0), (0, 1.0), (1, 2.0), (1, 1.2), (1, 1.0), (1, 0.9)]


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_pair(pkg, pyoracle, synth, ls, B, ml, beta, seed=3, lr=0.1, mom=0.9, wc=1e-5):
    ws, bs = synth.make_weights(ls, seed=seed)
    # non-zero biases so the bias path is exercised
    rng = np.random.default_rng(seed + 100)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    eng = pkg.BPGpu(1234, 0, ls, B, lr, mom, wc, ws, bs, beta, ml)
    ora = pyoracle.OracleNet(ls, B, lr, mom, wc, beta, ml, ws, bs)
    return eng, ora


@pytest.mark.parametrize("ml,beta", CASES)
def test_tiny_net_three_steps(pkg, pyoracle, synth, ml, beta):
    ls, B = [15, 8, 8, 8, 5], 8
    eng, ora = make_pair(pkg, pyoracle, synth, ls, B, ml, beta)
    inp, targ = synth.make_frames(3 * B + 3, 5, 3, seed=4)  # 3 full bunches + ragged tail (ignored)
    assert eng.train(inp, targ) == 3
    assert ora.train(inp, targ) == 3
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 2e-4
    assert relmax(eng.debug_tensor("out"), ora.tensor("out", rows=B)) < 2e-4
    assert relmax(eng.debug_tensor("dedx", 4), ora.tensor("dedx", 4, rows=B)) < 2e-4
    if ml == 1:
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    eng.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_intermediates_one_step_odd_shapes(pkg, pyoracle, synth, ml, beta):
    # sizes that are not multiples of 32 anywhere, bunch not a multiple of 32
    ls, B = [77, 45, 70, 33], 50
    eng, ora = make_pair(pkg, pyoracle, synth, ls, B, ml, beta)
    inp, targ = synth.make_frames(B, 11, 7, seed=9)
    targ = np.ascontiguousarray(np.tile(targ, (1, 3))[:, :33])
    eng.train(inp, targ)
    ora.train(inp, targ)
    for l in (1, 2):
        assert relmax(eng.debug_tensor("y", l), ora.tensor("y", l, rows=B)) < 2e-4
        assert relmax(eng.debug_tensor("yt", l), ora.tensor("y", l, rows=B)) < 2e-4
    for l in (1, 2, 3):
        assert relmax(eng.debug_tensor("dedx", l), ora.tensor("dedx", l, rows=B)) < 2e-4, l
        assert relmax(eng.debug_tensor("dedxt", l), ora.tensor("dedx", l, rows=B)) < 2e-4, l
        assert relmax(eng.debug_tensor("delta_w", l), ora.tensor("delta_w", l)) < 2e-4, l
        assert relmax(eng.debug_tensor("delta_b", l), ora.tensor("delta_b", l)) < 2e-4, l
    eng.close()


@pytest.mark.parametrize("B", [64, 96, 200, 256, 500, 512])
def test_every_bunch_size_variant_of_the_dw_kernel(pkg, pyoracle, synth, B):
    # Bp = 64, 128, 256, 512 take the persistent multi-layer dW kernel (H = 1, 2, 4, 8 units per tile),
    # Bp = 96 the per-layer fallback; 7 layers = 6 jobs in one launch, tile counts that are not
    # multiples of anything
    ls = [70, 130, 64, 33, 96, 40, 21]
    eng, ora = make_pair(pkg, pyoracle, synth, ls, B, 1, 1.2, seed=11)
    inp, targ = synth.make_frames(2 * B, 10, 7, seed=12)
    targ = np.ascontiguousarray(np.tile(targ, (1, 3))[:, :21])
    assert eng.train(inp, targ) == 2
    assert ora.train(inp, targ) == 2
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 2e-4, l
        assert relmax(eng.debug_tensor("delta_b", l + 1), ora.tensor("delta_b", l + 1)) < 2e-4, l
    eng.close()


def test_largest_advertised_bunch_with_the_ml_loss(pkg, pyoracle, synth):
    """mlggd_create accepts bunchsize up to 1152; the ML loss kernels then need 32*(Bp+1)+32 floats = 148 KB of
    dynamic LDS (the attribute has to be raised past 64 KB: ADVICE r01).  B = 1024 and B = 1152 with MLflag=1, two
    steps against the oracle (the dW kernel takes its non-persistent fallback at these sizes)."""
    for B in (1024, 1152):
        ls = [70, 130, 64, 33]
        eng, ora = make_pair(pkg, pyoracle, synth, ls, B, 1, 1.2, seed=21)
        inp, targ = synth.make_frames(2 * B + 7, 10, 7, seed=22)
        targ = np.ascontiguousarray(np.tile(targ, (1, 4))[:, :33])
        assert eng.train(inp, targ) == 2 and ora.train(inp, targ) == 2
        we, be = eng.returnWeights()
        wo, bo = ora.get_weights()
        for l in range(len(we)):
            assert relmax(we[l], wo[l]) < 2e-5, (B, l)
            assert relmax(be[l], bo[l]) < 2e-5, (B, l)
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
        sq, ab, ll = eng.cv_all(inp, targ)
        assert abs(ll - ora.cv_loglik(inp, targ)) <= 1e-4 * abs(ll)
        eng.close()
        ora.close()
    with pytest.raises(pkg.MlggdError, match="too large"):
        ws, bs = synth.make_weights([70, 130, 64, 33], seed=21)
        pkg.BPGpu(1, 0, [70, 130, 64, 33], 1153, 0.1, 0.9, 0.0, ws, bs, 1.2, 1)


def test_forward_and_dx_loop_variants(pkg, pyoracle, synth, monkeypatch):
    """The forward / dX main loops exist in several forms (software-pipelined inside the wave = default, forward with
    the operands by LDS-DMA or, MLGGD_FWD_PIPE=1 / MLGGD_DX_PIPE=1, through staging registers; the round-1 loops behind
    MLGGD_FWD_PIPE=0 / MLGGD_DX_PIPE=0) and for 4 (default) or 8 waves per workgroup.  The pipelined loop
    issues the same MFMAs in the same order: bit-identical weights for the same wave count.  Another wave count is
    another K split (summation order): equal to rounding, and each within the oracle bound.  Layer widths chosen so
    that a wave's K range is several full chunks plus a partial one, a single partial chunk, and nothing at all."""
    ls, B = [1210, 1060, 70, 262, 40], 96
    ws, bs = synth.make_weights(ls, seed=31)
    rng = np.random.default_rng(32)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    inp, targ = synth.make_frames(3 * B, 40, 1, seed=33)
    inp = np.ascontiguousarray(np.tile(inp, (1, 31))[:, :1210])

    def run(env):
        for k in ("MLGGD_FWD_PIPE", "MLGGD_DX_PIPE", "MLGGD_FWD_NW", "MLGGD_DX_NW"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 1.2, 1)
        assert eng.train(inp, targ) == 3
        w, b = eng.returnWeights()
        eng.close()
        return w + b

    ora = pyoracle.OracleNet(ls, B, 0.1, 0.9, 1e-5, 1.2, 1, ws, bs)
    assert ora.train(inp, targ) == 3
    wo, bo = ora.get_weights()
    old = {"MLGGD_FWD_PIPE": "0", "MLGGD_DX_PIPE": "0"}
    for nw in ("4", "8"):
        waves = {"MLGGD_FWD_NW": nw, "MLGGD_DX_NW": nw}
        new_loop, old_loop = run(waves), run(dict(waves, **old))
        staged = run(dict(waves, MLGGD_FWD_PIPE="1", MLGGD_DX_PIPE="1"))  # the pipelined loops through staging registers
        for x, y, z in zip(new_loop, old_loop, staged):
            assert np.array_equal(x, y) and np.array_equal(x, z), nw
        for x, y in zip(new_loop, wo + bo):
            assert relmax(x, y) < 2e-5, nw
    for x, y in zip(run({"MLGGD_FWD_NW": "16"}), wo + bo):  # 16 waves: the round-1 loop only (128-VGPR budget)
        assert relmax(x, y) < 2e-5
    ora.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2), (1, 1.0)])
def test_baseline_net_two_steps(pkg, pyoracle, synth, ml, beta):
    """BASELINE.json configs 2/3: 2827-2048x3-257, 128-frame minibatch."""
    ls, B = synth.baseline_layersizes(), 128
    eng, ora = make_pair(pkg, pyoracle, synth, ls, B, ml, beta, seed=27870775)
    inp, targ = synth.make_frames(2 * B, 257, 11)
    assert eng.train(inp, targ) == 2
    assert ora.train(inp, targ) == 2
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(4):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 5e-4, l
    assert relmax(eng.debug_tensor("out"), ora.tensor("out", rows=B)) < 2e-4
    # CV metrics incl. a ragged last bunch (300 frames = 2 bunches + 44)
    cin, ctarg = synth.make_frames(300, 257, 11, seed=77)
    sq, ab, ll = eng.cv_all(cin, ctarg)
    assert abs(sq - ora.cv_sqerr(cin, ctarg)) <= 1e-4 * abs(sq)
    assert abs(ab - ora.cv_abserr(cin, ctarg)) <= 1e-4 * abs(ab)
    assert abs(eng.CrossValid(cin, ctarg) - sq) <= 1e-6 * abs(sq)
    assert abs(eng.CrossValiddB(cin, ctarg) - ab) <= 1e-6 * abs(ab)
    if ml == 1:
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
        assert abs(ll - ora.cv_loglik(cin, ctarg)) <= 1e-4 * abs(ll)
        assert abs(eng.CrossValid2(cin, ctarg) - ll) <= 1e-6 * abs(ll)
    out = eng.forward(cin)
    assert relmax(out, ora.cv_forward(cin[:128])[:128] if False else np.vstack(
        [ora.cv_forward(cin[i:i + 128]) for i in range(0, 300, 128)])) < 2e-4
    eng.close()


def test_empty_and_short_chunks(pkg, synth):
    ls, B = [15, 8, 5], 8
    ws, bs = synth.make_weights(ls, seed=1)
    eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 0.0, ws, bs, 2.0, 0)
    inp, targ = synth.make_frames(5, 5, 3)
    assert eng.train(inp[:0], targ[:0]) == 0       # empty chunk
    assert eng.train(inp, targ) == 0               # shorter than one bunch: ignored (BP_GPU.cu:177-180)
    w2, _ = eng.returnWeights()
    for a, b in zip(w2, ws):
        assert np.array_equal(a, b)
    assert eng.forward(inp).shape == (5, 5)
    eng.close()


def test_errors_are_reported(pkg, synth):
    ls = [15, 8, 5]
    ws, bs = synth.make_weights(ls, seed=1)
    with pytest.raises(pkg.MlggdError, match="Not In Range"):
        pkg.BPGpu(1, 99, ls, 8, 0.1, 0.9, 0.0, ws, bs, 2.0, 0)
    eng = pkg.BPGpu(1, 0, ls, 8, 0.1, 0.9, 0.0, ws, bs, 2.0, 0)
    with pytest.raises(pkg.MlggdError):
        eng.train_resident(0, 8)  # nothing resident
    eng.close()


def test_refusals_leave_no_state_behind(pkg, synth, monkeypatch):
    """mlggd_alloc_pinned_on validates the device (the BP_GPU.cu:17-21 message, not an out-of-memory error
    from an IO thread) and leaves the caller's current device alone; an exchange mode the shape rules out is refused
    BEFORE a communicator exists, so no rank can be left inside ncclCommInitRank."""
    import ctypes as C
    L = pkg.load()
    p = C.c_void_p()
    assert L.mlggd_alloc_pinned_on(99, 4096, C.byref(p)) != 0 and b"Not In Range" in L.mlggd_last_error()
    assert L.mlggd_alloc_pinned_on(0, 4096, C.byref(p)) == 0 and p.value
    assert L.mlggd_free_pinned(p) == 0
    ls = [15, 8, 5]
    ws, bs = synth.make_weights(ls, seed=1)
    eng = pkg.BPGpu(1, 0, ls, 50, 0.1, 0.9, 0.0, ws, bs, 2.0, 0)   # 50 frames: not a multiple of 32
    monkeypatch.setenv("MLGGD_DP_MODE", "gather")
    with pytest.raises(pkg.MlggdError, match="MLGGD_DP_MODE=gather needs"):
        eng.comm_init(pkg.comm_unique_id(), 1, 0)
    assert eng.comm_info() == (0, -1) and eng.dp_mode() == 0          # nothing was created
    monkeypatch.setenv("MLGGD_DP_MODE", "allreduce")
    eng.comm_init(pkg.comm_unique_id(), 1, 0)                          # the same engine can still join
    assert eng.comm_info() == (1, 0) and eng.dp_mode() == 1
    eng.close()


def test_one_step_is_bit_exact_on_exactly_representable_data(pkg, pyoracle):
    """With small-integer inputs, weights that are multiples of 1/8 and targets that are multiples of 1/4 every product
    and partial sum of an MMSE step is exact in fp32, so GEMM summation order -- the one freedom cublasSgemm has
    (TC/BP_GPU.cu:361,432) -- cannot matter, and ONE step must give the oracle's output, gradient,
    momentum, weights and biases BIT FOR BIT: forward (k_fwd SLAB + k_loss_norm), dW + update (k_dwp with its fused
    kernUpdatedelta / kernAccSum epilogue, TC/DevFunc.cu:490-507,427-443) and the bias path (kernAccSumrow order).
    The shape -- 587 x 577 = 9 x 64 + 11 inputs, 9 x 64 + 1 outputs -- puts nearly empty edge tiles (11 valid rows,
    1 valid column, the corner) next to full ones.  (Round 3 also ran this test against a variant of k_dwp that forms
    those edge strips outside the tile walk: profiles/r03_dwp_edge_strips_ab.txt.)"""
    K, N, B = 587, 577, 128
    rng = np.random.default_rng(17)
    W = (rng.integers(-4, 5, (K, N)) * 0.125).astype(np.float32)
    b = (rng.integers(-8, 9, N) * 0.25).astype(np.float32)
    x = rng.integers(-3, 4, (B, K)).astype(np.float32)
    targ = (rng.integers(-8, 9, (B, N)) * 0.25).astype(np.float32)
    eng = pkg.BPGpu(1, 0, [K, N], B, 0.1, 0.9, 1e-5, [W], [b], 2.0, 0)
    ora = pyoracle.OracleNet([K, N], B, 0.1, 0.9, 1e-5, 2.0, 0, [W], [b])
    assert eng.train(x, targ) == 1 and ora.train(x, targ) == 1
    assert np.array_equal(eng.debug_tensor("out"), ora.tensor("out", rows=B))
    assert np.array_equal(eng.debug_tensor("dedx", 1), ora.tensor("dedx", 1, rows=B))
    dw_e, dw_o = eng.debug_tensor("delta_w", 1), ora.tensor("delta_w", 1)
    bad = np.argwhere(dw_e != dw_o)
    assert bad.size == 0, "first differing (k, n): %s of %d" % (bad[:5].tolist(), len(bad))
    assert np.array_equal(eng.debug_tensor("delta_b", 1), ora.tensor("delta_b", 1))
    (we,), (be,) = eng.returnWeights()
    (wo,), (bo,) = ora.get_weights()
    assert np.array_equal(we, wo) and np.array_equal(be, bo)
    assert np.abs(dw_e[576:, :]).max() > 0 and np.abs(dw_e[:, 576]).max() > 0   # the edge rows / column did move
    eng.close()
    ora.close()
