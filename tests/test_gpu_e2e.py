"""GPU: committed goldens, the exchange (data-parallel) code path on a 1-rank RCCL
communicator, and the BPtrain_Sigmoid executable end to end on synthetic pfiles."""
import os
import re
import subprocess

import numpy as np
import pytest

import hostlib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HP = (0.1, 0.9, 1e-5)


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (0, 1.0), (1, 2.0), (1, 1.2), (1, 1.0), (1, 0.9)])
def test_tiny_goldens(pkg, synth, ml, beta):
    g = np.load(os.path.join(GOLD, "tiny_net.npz"))
    ls, B = [int(x) for x in g["layersizes"]], int(g["bunch"])
    ws, bs = synth.make_weights(ls, seed=int(g["wseed"]))
    inp, targ = synth.make_frames(int(g["frames"]), 5, 3, seed=int(g["dseed"]))
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    assert eng.train(inp, targ) == 3
    key = "ml%d_b%s" % (ml, beta)
    w, b = eng.returnWeights()
    for l in range(4):
        assert relmax(w[l], g["%s_W%d" % (key, l + 1)]) < 2e-5
        assert relmax(b[l], g["%s_b%d" % (key, l + 1)]) < 2e-5
        assert relmax(eng.debug_tensor("delta_w", l + 1), g["%s_dW%d" % (key, l + 1)]) < 2e-4
    assert relmax(eng.debug_tensor("dedx", 4), g[key + "_dedx4"]) < 2e-4
    cv = g[key + "_cv"]
    sq, ab, ll = eng.cv_all(inp, targ)
    assert abs(sq - cv[0]) <= 1e-4 * abs(cv[0]) and abs(ab - cv[1]) <= 1e-4 * abs(cv[1])
    if ml:
        assert relmax(eng.scalefactor(), g[key + "_alpha"]) < 1e-5
        assert abs(ll - cv[2]) <= 1e-4 * abs(cv[2])
    eng.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2), (1, 1.0)])
def test_baseline_goldens(pkg, synth, ml, beta):
    g = np.load(os.path.join(GOLD, "baseline_net.npz"))
    ls, B = synth.baseline_layersizes(), int(g["bunch"])
    ws, bs = synth.make_weights(ls)
    inp, targ = synth.make_frames(2 * B, 257, 11)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    assert eng.train(inp, targ) == 2
    key = "ml%d_b%s" % (ml, beta)
    w, b = eng.returnWeights()
    for l in range(4):
        idx = g["%s_idx%d" % (key, l + 1)]
        assert relmax(w[l].ravel()[idx], g["%s_Wsamp%d" % (key, l + 1)]) < 2e-5
        dw = eng.debug_tensor("delta_w", l + 1)
        assert relmax(dw.ravel()[idx], g["%s_dWsamp%d" % (key, l + 1)]) < 1e-3
        s = g["%s_dWsum%d" % (key, l + 1)]
        assert abs(dw.astype(np.float64).sum() - s[0]) <= 1e-4 * s[1]     # checksum of the whole tensor
        assert relmax(b[l], g["%s_b%d" % (key, l + 1)]) < 2e-5
    cin, ctarg = synth.make_frames(300, 257, 11, seed=77)
    cv = g[key + "_cv"]
    sq, ab, ll = eng.cv_all(cin, ctarg)
    assert abs(sq - cv[0]) <= 1e-4 * abs(cv[0]) and abs(ab - cv[1]) <= 1e-4 * abs(cv[1])
    if ml:
        assert abs(ll - cv[2]) <= 1e-4 * abs(cv[2])
    eng.close()


@pytest.mark.parametrize("mode", ["allreduce", "gather", "shard", "shard_a2a"])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_exchange_path_on_one_rank_communicator(pkg, pyoracle, synth, monkeypatch, ml, beta, mode):
    """mlggd_comm_init(world=1): RCCL is dlopen'ed and the step takes a data-parallel exchange path --
    allreduce: unfused kernels, gradient buffers, all-reduce on the comm stream, k_apply_update;
    gather: all-gather of the gradient factors, then the fused kernel over the gathered minibatch;
    shard: the same with the update restricted to the rank's block of weight rows and W all-gathered in
    place; shard_a2a: the sharded update with the activations written owner-blocked and exchanged by ncclSend / ncclRecv
    pairs (an all-to-all; with one rank: to itself) -- and must equal the oracle."""
    monkeypatch.setenv("MLGGD_DP_MODE", mode)
    ls, B = [257 * 3, 256, 160, 257], 64
    ws, bs = synth.make_weights(ls, seed=5)
    inp, targ = synth.make_frames(3 * B, 257, 3, seed=6)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    assert eng.comm_info() == (0, -1)                      # no communicator yet
    eng.comm_init(pkg.comm_unique_id(), 1, 0)
    assert eng.dp_mode() == {"allreduce": 1, "gather": 2, "shard": 3, "shard_a2a": 4}[mode]
    assert eng.comm_info() == (1, 0)                       # ncclCommCount / ncclCommUserRank: bench.py's `rccl_ranks`
    ora = pyoracle.OracleNet(ls, B, *HP, beta, ml, ws, bs)
    assert eng.train(inp, targ) == 3 and ora.train(inp, targ) == 3
    assert eng.plan_count() <= 6                           # a handful of launch plans, however many steps
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(3):
        assert relmax(we[l], wo[l]) < 2e-5 and relmax(be[l], bo[l]) < 2e-5
        if mode == "allreduce":
            assert relmax(eng.debug_tensor("grad_w", l + 1), ora.tensor("grad_w", l + 1)) < 3e-4
    if ml:
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    eng.close()


@pytest.mark.parametrize("mode", ["gather", "shard", "allreduce", "shard_a2a"])
@pytest.mark.parametrize("world,B", [(2, 64), (2, 128), (4, 128), (8, 128), (4, 32), (3, 50)])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_emulated_world_equals_one_device_with_the_global_bunch(pkg, pyoracle, synth, ml, beta, world, B, mode):
    """SURVEY 8e parity definition: n ranks x B frames == one device with bunchsize n*B on the same frame
    order.  `world` ranks are emulated on one GPU, one after the other, each on ITS OWN rows
    [r*B,(r+1)*B) of every global minibatch (mlggd_debug_fake_world: device copies / adds in place of the
    RCCL calls); the oracle trains the same rows with bunchsize world*B.
    gather / shard: the rank-major row order of the gathered factors, the dW kernel over 2..16 units of 64
    gathered frames, the global 1/n and the ML statistic summed over ranks; shard: every emulated rank
    updates only its block of weight rows (uneven and empty blocks included), the bias update comes from
    the last rank's bias-only jobs.  allreduce: k_dwp<.,false> -> sum of the ranks' gradients ->
    k_apply_update / k_bias_apply with n_global != B (any shape, e.g. 3 x 50)."""
    if mode != "allreduce" and (B % 32 or world * B not in (64, 128, 256, 512, 1024)):
        pytest.skip("factor exchange needs world*B in {64,...,1024}")
    ls = [40 * 5, 160, 96, 40]
    ws, bs = synth.make_weights(ls, seed=8)
    rng = np.random.default_rng(9)
    bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
    steps = 2
    inp, targ = synth.make_frames(steps * world * B + 5, 40, 5, seed=10)   # + a ragged tail that is ignored
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    eng.fake_world(world, sharded=mode == "shard", allreduce=mode == "allreduce", a2a=mode == "shard_a2a")
    assert eng.dp_mode() == {"allreduce": 1, "gather": 2, "shard": 3, "shard_a2a": 4}[mode]
    ora = pyoracle.OracleNet(ls, world * B, *HP, beta, ml, ws, bs)
    assert eng.train(inp, targ) == steps and ora.train(inp, targ) == steps
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
        assert relmax(eng.debug_tensor("delta_w", l + 1), ora.tensor("delta_w", l + 1)) < 2e-4, l
        assert relmax(eng.debug_tensor("delta_b", l + 1), ora.tensor("delta_b", l + 1)) < 2e-4, l
        if mode == "shard_a2a" and l < len(we) - 1:   # the last emulated rank's owner-blocked activations, un-blocked
            assert relmax(eng.debug_tensor("y", l + 1), ora.tensor("y", l + 1, rows=world * B)[(world - 1) * B:]) < 2e-4, l
        if mode == "allreduce":
            assert relmax(eng.debug_tensor("grad_w", l + 1), ora.tensor("grad_w", l + 1)) < 3e-4, l
    if ml:
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    eng.close()


@pytest.mark.parametrize("harness", ["one_rank_communicator", "emulated_world_of_4"])
@pytest.mark.parametrize("mode", ["gather", "shard", "allreduce", "shard_a2a"])
def test_dp_launch_order_knobs_do_not_change_a_bit(pkg, synth, monkeypatch, mode, harness):
    """MLGGD_DP_MAINLINE (critical-path collectives on the main stream), MLGGD_DP_STOPEV (events riding on
    the producing kernel's dispatch packet) and, for the all-reduce arm, MLGGD_DP_AR_SHARD (reduce-scatter -> update
    of the rank's block -> all-gather of W, against all-reduce + full update) change WHERE and WHEN work is enqueued,
    never what is computed (and MLGGD_DP_STAT_COMM only which communicator carries the 257-float statistic): every arm
    must leave bit-identical weights, biases and alpha -- through a real 1-rank
    RCCL communicator (streams, events, RCCL calls) and in an emulated world of 4 ranks (different rows per rank,
    uneven blocks).  A premature send or a skipped wait shows up here as a different bit.  The defaults themselves
    rest on these one-GPU rehearsals: nothing has run between two GPUs."""
    monkeypatch.setenv("MLGGD_DP_MODE", mode)
    ls, B, steps, world = [40 * 5, 160, 96, 40], 64, 4, 4
    ws, bs = synth.make_weights(ls, seed=8)
    n = steps * (world if harness == "emulated_world_of_4" else 1) * B
    inp, targ = synth.make_frames(n, 40, 5, seed=10)

    def run(env):
        for k in ("MLGGD_DP_MAINLINE", "MLGGD_DP_STOPEV", "MLGGD_DP_AR_SHARD", "MLGGD_DP_STAT_COMM"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 1.2, 1)
        if harness == "emulated_world_of_4":
            eng.fake_world(world, sharded=mode == "shard", allreduce=mode == "allreduce", a2a=mode == "shard_a2a")
        else:
            eng.comm_init(pkg.comm_unique_id(), 1, 0)
        assert eng.train(inp, targ) == steps
        w, b = eng.returnWeights()
        out = [x.copy() for x in w + b] + [eng.scalefactor().copy()]
        eng.close()
        return out

    base = run({})
    arms = [{"MLGGD_DP_MAINLINE": "0"}, {"MLGGD_DP_STOPEV": "0"}, {"MLGGD_DP_MAINLINE": "0", "MLGGD_DP_STOPEV": "0"}]
    if mode == "allreduce":
        arms += [{"MLGGD_DP_AR_SHARD": "0"}, {"MLGGD_DP_AR_SHARD": "0", "MLGGD_DP_STOPEV": "0"}]
    if harness == "one_rank_communicator":
        # the ML statistic's all-reduce on a communicator of its own (ncclCommSplit): same sum, another queue
        arms += [{"MLGGD_DP_STAT_COMM": "1"}]
    for env in arms:
        got = run(env)
        for i, (x, y) in enumerate(zip(got, base)):
            assert np.array_equal(x, y), (env, i)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_emulated_world_with_the_fine_grained_factor_exchange(pkg, pyoracle, synth, monkeypatch, ml, beta, world):
    """MLGGD_DP_FINE=1 (no longer the default): every factor sent the moment it exists, the dW launch split in two
    (layers L-1..2, then layer 1) -- same contract as test_emulated_world_equals_one_device_with_the_global_bunch."""
    monkeypatch.setenv("MLGGD_DP_FINE", "1")
    B, steps = 128, 2
    ls = [40 * 5, 160, 96, 40]
    ws, bs = synth.make_weights(ls, seed=8)
    inp, targ = synth.make_frames(steps * world * B + 5, 40, 5, seed=10)
    eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    eng.fake_world(world)
    assert eng.dp_mode() == 2
    ora = pyoracle.OracleNet(ls, world * B, *HP, beta, ml, ws, bs)
    assert eng.train(inp, targ) == steps and ora.train(inp, targ) == steps
    we, be = eng.returnWeights()
    wo, bo = ora.get_weights()
    for l in range(len(we)):
        assert relmax(we[l], wo[l]) < 2e-5, l
        assert relmax(be[l], bo[l]) < 2e-5, l
    if ml:
        assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-5
    eng.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
@pytest.mark.parametrize("mode", ["gather", "shard", "allreduce", "shard_a2a"])
def test_frame_stream_chunks_on_the_exchange_path(pkg, synth, monkeypatch, mode, ml, beta):
    """What BPtrain_Sigmoid does on a data-parallel rank: frame-stream chunks (rows gathered on the device, the next
    minibatch staged ahead, chunks enqueued without waiting) WITH a communicator.  Through a 1-rank RCCL communicator
    the factor-exchange modes with the MMSE loss must reproduce the plain engine bit for bit (same kernels over the
    same rows); with the ML loss the statistic is summed in another order (wavefront reduction + all-reduce instead of
    kernSumcol's sequence) and the all-reduce mode applies its update in a separate pass: equal to rounding."""
    monkeypatch.setenv("MLGGD_DP_MODE", mode)
    dim, ctx, B, toff = 40, 5, 64, 2
    ls = [dim * ctx, 128, 96, dim]
    rng = np.random.default_rng(23)
    nfr = 700
    feat = rng.standard_normal((nfr, dim), dtype=np.float32)
    targ = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
    ws, bs = synth.make_weights(ls, seed=24)
    firsts = [rng.permutation(nfr - ctx + 1)[:4 * B + 7].astype(np.int32) for _ in range(3)]   # three chunks, ragged tails
    out = []
    for with_comm in (False, True):
        eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
        if with_comm:
            eng.comm_init(pkg.comm_unique_id(), 1, 0)
        for first in firsts:
            assert eng.train_frames(feat, targ, first, ctx, toff, wait=False) == 4
        eng.sync()
        w, b = eng.returnWeights()
        out.append((w + b, eng.scalefactor()))
        eng.close()
    for x, y in zip(out[0][0], out[1][0]):
        if mode == "allreduce" or ml:
            assert relmax(y, x) < 2e-5
        else:
            assert np.array_equal(x, y)
    if ml:
        assert relmax(out[1][1], out[0][1]) < 1e-5


@pytest.mark.parametrize("mode", ["gather", "shard", "allreduce", "shard_a2a"])
@pytest.mark.parametrize("ls", [[96, 70], [75, 64, 33]])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 0.9)])
def test_emulated_world_on_shallow_nets(pkg, pyoracle, synth, monkeypatch, ml, beta, ls, mode):
    """The data-parallel step on nets with ONE and TWO weight layers (no hidden activations to exchange; one dX launch or
    none; the last factor group is the only one), both factor-exchange granularities: 2 emulated ranks x 64 frames
    against the oracle with bunchsize 128 (TC/BP_GPU.cu:308-440 with numlayers 2 and 3)."""
    for fine in ("0", "1"):
        monkeypatch.setenv("MLGGD_DP_FINE", fine)
        world, B, steps = 2, 64, 3
        ws, bs = synth.make_weights(ls, seed=18)
        rng = np.random.default_rng(19)
        bs = [rng.uniform(-0.1, 0.1, b.shape).astype(np.float32) for b in bs]
        inp = rng.standard_normal((steps * world * B, ls[0]), dtype=np.float32)
        targ = rng.standard_normal((steps * world * B, ls[-1]), dtype=np.float32)
        eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
        eng.fake_world(world, sharded=mode == "shard", allreduce=mode == "allreduce", a2a=mode == "shard_a2a")
        ora = pyoracle.OracleNet(ls, world * B, *HP, beta, ml, ws, bs)
        assert eng.train(inp, targ) == steps and ora.train(inp, targ) == steps
        we, be = eng.returnWeights()
        wo, bo = ora.get_weights()
        for l in range(len(we)):
            assert relmax(we[l], wo[l]) < 5e-5, (fine, l)
            assert relmax(be[l], bo[l]) < 5e-5, (fine, l)
        if ml:
            assert relmax(eng.scalefactor(), ora.tensor("scalefactor")) < 1e-4
        eng.close()
        ora.close()
        if mode != "gather":
            break   # the granularity knob only exists for the replicated factor exchange


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_frame_stream_chunks_equal_expanded_chunks_bitwise(pkg, synth, ml, beta):
    """rows gathered on the device from the raw frame stream == the host-expanded
    matrix, bit for bit (weights, CV metrics, forward outputs), incl. a shuffled row order."""
    dim, ctx, B, toff = 257, 11, 128, 5
    ls = [dim * ctx, 512, 384, dim]
    rng = np.random.default_rng(3)
    nfr = 700
    feat = rng.standard_normal((nfr, dim), dtype=np.float32)
    targ = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
    first = rng.permutation(nfr - ctx + 1)[:3 * B + 17].astype(np.int32)   # shuffled windows + ragged tail
    idx = first[:, None] + np.arange(ctx)[None, :]
    inp = np.ascontiguousarray(feat[idx].reshape(len(first), ctx * dim))
    tg = np.ascontiguousarray(targ[first + toff])
    ws, bs = synth.make_weights(ls, seed=4)
    a = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    b = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
    assert a.train(inp, tg) == 3
    assert b.train_frames(feat, targ, first, ctx, toff) == 3
    for x, y in zip(a.returnWeights()[0] + a.returnWeights()[1], b.returnWeights()[0] + b.returnWeights()[1]):
        assert np.array_equal(x, y)
    assert a.cv_all(inp, tg) == b.cv_all_frames(feat, targ, first, ctx, toff)
    assert np.array_equal(a.forward(inp), b.forward_frames(feat, first, ctx))
    with pytest.raises(pkg.MlggdError, match="outside"):
        b.train_frames(feat, targ, np.array([nfr - 3], np.int32), ctx, toff)
    a.close()
    b.close()


@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
@pytest.mark.parametrize("frames_mode", [False, True])
def test_launch_plan_knobs_do_not_change_a_bit(pkg, synth, monkeypatch, frames_mode, ml, beta):
    """The launch-plan optimisations (next minibatch staged alongside the loss kernel, one dW launch
    for all layers, the loss in one launch -- k_loss_norm for MMSE, k_loss_ml for ML-GGD) reorder launches, not
    arithmetic: weights after 5 steps are bit-identical with each of them switched off, for expanded and for
    frame-stream chunks, and when the chunk is trained in two calls (the staged-ahead bunch is then dropped at the
    call boundary)."""
    dim, ctx, B, toff = 40, 5, 64, 2
    ls = [dim * ctx, 96, 70, dim]
    rng = np.random.default_rng(5)
    nfr = 600
    feat = rng.standard_normal((nfr, dim), dtype=np.float32)
    targ = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
    first = rng.permutation(nfr - ctx + 1)[:5 * B + 9].astype(np.int32)
    idx = first[:, None] + np.arange(ctx)[None, :]
    inp = np.ascontiguousarray(feat[idx].reshape(len(first), ctx * dim))
    tg = np.ascontiguousarray(targ[first + toff])
    ws, bs = synth.make_weights(ls, seed=6)

    def run(env, split=False):
        for k in ("MLGGD_STAGE_AHEAD", "MLGGD_DW_MERGE", "MLGGD_LOSS_FUSE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, beta, ml)
        if frames_mode:
            eng.load_frames(feat, targ, first, ctx, toff)
        else:
            eng.load_chunk(inp, tg)
        if split:
            assert eng.train_resident(0, 2 * B) == 2
            assert eng.train_resident(2 * B, 3 * B + 9) == 3
        else:
            assert eng.train_resident(0, len(first)) == 5
        w, b = eng.returnWeights()
        eng.close()
        return w + b

    ref = run({})
    for env, split in (({"MLGGD_STAGE_AHEAD": "0"}, False), ({"MLGGD_DW_MERGE": "0"}, False),
                       ({"MLGGD_LOSS_FUSE": "0"}, False), ({}, True)):
        for x, y in zip(ref, run(env, split)):
            assert np.array_equal(x, y), (env, split)


def test_chunks_enqueued_without_waiting_equal_chunks_trained_one_by_one(pkg, synth):
    """mlggd_train_frames_async: four chunks of different sizes back to back (the upload of chunk i+1 goes to
    the other device buffer set while chunk i's steps run, the host arrays are overwritten right after each
    call) give bit-identical weights to the synchronous calls."""
    dim, ctx, B, toff = 48, 5, 64, 2
    ls = [dim * ctx, 128, 96, dim]
    ws, bs = synth.make_weights(ls, seed=7)
    rng = np.random.default_rng(8)
    chunks = []
    for nfr in (900, 500, 1300, 700):
        feat = rng.standard_normal((nfr, dim), dtype=np.float32)
        targ = (0.5 * feat + 0.5 * rng.standard_normal((nfr, dim), dtype=np.float32)).astype(np.float32)
        first = rng.permutation(nfr - ctx + 1)[:(nfr // 70) * B // 2 + 5].astype(np.int32)
        chunks.append((feat, targ, first))
    a = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 2.0, 0)
    b = pkg.BPGpu(1, 0, ls, B, *HP, ws, bs, 2.0, 0)
    for feat, targ, first in chunks:
        a.train_frames(feat, targ, first, ctx, toff)
    scratch_f = np.empty((1300, dim), np.float32)
    for feat, targ, first in chunks:
        f, t, i = feat.copy(), targ.copy(), first.copy()
        b.train_frames(f, t, i, ctx, toff, wait=False)
        f[:] = 7.0   # the caller's buffers are free as soon as the call returns
        t[:] = -3.0
        i[:] = 0
    b.sync()
    for x, y in zip(a.returnWeights()[0] + a.returnWeights()[1], b.returnWeights()[0] + b.returnWeights()[1]):
        assert np.array_equal(x, y)
    a.close()
    b.close()


def test_linearity_of_forward_at_full_size(pkg, synth):
    """Size-independent property at BASELINE size: with one linear layer the network output is
    linear in the input; f(a x1 + b x2) = a f(x1) + b f(x2) - (a+b-1) bias."""
    ls = [2827, 257]
    ws, bs = synth.make_weights(ls, seed=3)
    bs = [np.linspace(-1, 1, 257).astype(np.float32)]
    eng = pkg.BPGpu(1, 0, ls, 128, *HP, ws, bs, 2.0, 0)
    x1, _ = synth.make_frames(300, 257, 11, seed=1)
    x2, _ = synth.make_frames(300, 257, 11, seed=2)
    f1, f2, f3 = eng.forward(x1), eng.forward(x2), eng.forward(0.5 * x1 - 2.0 * x2)
    want = 0.5 * f1 - 2.0 * f2 + 2.5 * bs[0]
    assert relmax(f3, want) < 1e-4
    eng.close()


def test_bptrain_sigmoid_executable(pkg, pyoracle, tmp_path):
    """finetune.pl-style invocation of the drop-in binary on synthetic pfiles vs the same
    pipeline driven from Python (real host IO code for the chunk/sample order + CPU oracle)."""
    exe = os.path.join(hostlib.HOST, "BPtrain_Sigmoid")
    subprocess.check_call(["make", "-C", hostlib.HOST, "-s"])
    rng = np.random.default_rng(11)
    dim, ctx, B = 20, 5, 16
    lens = [int(x) for x in rng.integers(30, 90, 24)] + [3]
    nfr = sum(lens)
    noisy = rng.normal(3, 2, (nfr, dim)).astype(np.float32)
    clean = (0.6 * noisy + rng.normal(0, 1, (nfr, dim))).astype(np.float32)
    hostlib.write_pfile(str(tmp_path / "n.pfile"), lens, noisy)
    hostlib.write_pfile(str(tmp_path / "c.pfile"), lens, clean)
    hostlib.write_norm(str(tmp_path / "n.norm"), noisy.mean(0), 1.0 / noisy.std(0))
    ls = [dim * ctx, 48, 40, dim]
    subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "4", *map(str, ls), str(tmp_path),
                           str(tmp_path / "init.wts"), "1", "2", "5"], stdout=subprocess.DEVNULL)
    kv = dict(gpu_used=0, numlayers=4, layersizes=",".join(map(str, ls)), bunchsize=B, MLflag=1, shapefactor=1.2,
              momentum=0.9, weightcost=1e-5, lrate=0.1, fea_dim=dim, fea_context=ctx, traincache=300,
              init_randem_seed=27870775, targ_offset=2, initwts_file=tmp_path / "init.wts", norm_file=tmp_path / "n.norm",
              fea_file=tmp_path / "n.pfile", targ_file=tmp_path / "c.pfile", outwts_file=tmp_path / "mlp.1.wts",
              log_file=tmp_path / "mlp.1.log", train_sent_range="0-19", cv_sent_range="20-24", dropoutflag=0,
              visible_omit=0.1, hid_omit=0.1)
    res = subprocess.run([exe] + ["%s=%s" % (k, v) for k, v in kv.items()], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all finish!" in res.stdout
    log = open(tmp_path / "mlp.1.log").read()

    # the same epoch from Python: real host IO code for order, CPU oracle for the math
    io = hostlib.HostIO(**kv)
    w0, b0 = hostlib.read_wts(str(tmp_path / "init.wts"), ls)
    ora = pyoracle.OracleNet(ls, B, 0.1, 0.9, 1e-5, 1.2, 1, w0, b0)
    starts, total = io.plan("0-19")
    order = io.shuffle(len(starts))
    assert len(starts) >= 3
    for n, ci in enumerate(order):
        inp, tg = io.read_chunk(ci, ls[0], dim, 300)
        assert "Starting chunk %d of %d containing %d samples." % (n + 1, len(starts), len(inp)) in log
        ora.train(inp, tg)
    cvs, cvtotal = io.plan("20-24", cv=True)
    sq = ab = ll = np.float32(0)
    for ci in range(len(cvs)):
        inp, tg = io.read_chunk(ci, ls[0], dim, 300, cv=True)
        sq += np.float32(ora.cv_sqerr(inp, tg))
        ab += np.float32(ora.cv_abserr(inp, tg))
        ll += np.float32(ora.cv_loglik(inp, tg))
    io.close()

    ws, bs = hostlib.read_wts(str(tmp_path / "mlp.1.wts"), ls)
    wo, bo = ora.get_weights()
    for l in range(3):
        assert relmax(ws[l], wo[l]) < 5e-5, l
        assert relmax(bs[l], bo[l]) < 5e-5, l
    got = [float(re.search(pat + r": (-?[\d.]+)", log).group(1)) for pat in
           ("CV over. squared error", "CV over. square root squared error", "CV2 over. CV log likelihood")]
    want = [float(sq) / cvtotal, float(ab) / cvtotal, float(ll) / cvtotal]
    for g_, w_ in zip(got, want):
        assert abs(g_ - w_) <= 1e-4 * abs(w_) + 1e-6, (got, want)
    for line in ("Get pfile info over: Training data has %d frames, %d sentences." % (nfr, len(lens)), "Total cost time:",
                 "Saving over.", "Starting CV."):
        assert line in log

    # the reference-style host expansion (MLGGD_EXPANDED=1) gives the same weights file, bit for bit
    kv2 = dict(kv, outwts_file=tmp_path / "mlp.exp.wts", log_file=tmp_path / "mlp.exp.log")
    res = subprocess.run([exe] + ["%s=%s" % (k, v) for k, v in kv2.items()], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, MLGGD_EXPANDED="1"))
    assert res.returncode == 0, res.stdout + res.stderr
    assert open(tmp_path / "mlp.exp.wts", "rb").read() == open(tmp_path / "mlp.1.wts", "rb").read()
    cvlines = lambda t: [l for l in t.splitlines() if l.startswith("CV")]
    assert cvlines(open(tmp_path / "mlp.exp.log").read()) == cvlines(log)

    # data parallel, a peer that never shows up: rank 0 of 2 sits in the id hand-off; the watchdog names rank and phase
    # and ends the process with status 3 instead of leaving it blocked (RCCL collectives have no timeout)
    kv3 = dict(kv, outwts_file=tmp_path / "mlp.dp.wts", log_file=tmp_path / "mlp.dp.log")
    res = subprocess.run([exe] + ["%s=%s" % (k, v) for k, v in kv3.items()], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MLGGD_ID_FILE=str(tmp_path / "id"),
                                  MLGGD_WATCHDOG_S="3", MLGGD_RENDEZVOUS_TIMEOUT="60"))
    assert res.returncode == 3, (res.returncode, res.stderr[-500:])
    assert "BPtrain watchdog: rank 0 of 2 made no progress for 3 s in phase 'engine + communicator'" in res.stderr

    # errors: message in the log, non-zero exit status
    kv["initwts_file"] = tmp_path / "missing.wts"
    kv["log_file"] = tmp_path / "err.log"
    res = subprocess.run([exe] + ["%s=%s" % (k, v) for k, v in kv.items()], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "can not open initial weights file" in open(tmp_path / "err.log").read()


def test_dropout_training_and_cv_scaling(pkg, pyoracle, synth):
    """a25 (BP_GPU.cu:344-355, 484-501): dropout zeroes inputs/hidden activations where u < p with
    NO rescale in training; CV scales W by the keep-probability around each GEMM.  The random
    stream differs from cuRAND by design, so training is checked statistically and CV exactly."""
    ls, B = [257 * 3, 256, 256, 257], 128
    ws, bs = synth.make_weights(ls, seed=5)
    inp, targ = synth.make_frames(B, 257, 3, seed=6)
    inp = np.abs(inp) + 0.5                                   # no exact zeros in the data
    eng = pkg.BPGpu(77, 0, ls, B, 0.0, 0.0, 0.0, ws, bs, 2.0, 0, dropoutflag=1, visible_omit=0.2, hid_omit=0.5)
    eng.train(inp, targ)                                      # lrate 0: weights stay, activations observable
    y1 = eng.debug_tensor("y", 1)
    frac_hidden = float((y1 == 0).mean())
    assert abs(frac_hidden - 0.5) < 0.02, frac_hidden         # sigmoid outputs are never exactly 0
    yt1 = eng.debug_tensor("yt", 1)
    assert np.array_equal(y1 == 0, yt1 == 0)                  # both layouts dropped identically
    # CV: forward with W scaled by keep-prob == oracle forward on pre-scaled weights
    out = eng.forward(inp)
    keep = [0.8, 0.5, 0.5]
    ora = pyoracle.OracleNet(ls, B, 0.0, 0.0, 0.0, 2.0, 0, [w * k for w, k in zip(ws, keep)], bs)
    want = ora.cv_forward(inp)
    assert np.abs(out - want).max() / np.abs(want).max() < 2e-4
    w_after, _ = eng.returnWeights()
    for a, b in zip(w_after, ws):
        assert np.abs(a - b).max() <= 2e-7 * np.abs(b).max()  # scale / unscale round trip
    # a different seed gives a different mask
    eng2 = pkg.BPGpu(78, 0, ls, B, 0.0, 0.0, 0.0, ws, bs, 2.0, 0, dropoutflag=1, visible_omit=0.2, hid_omit=0.5)
    eng2.train(inp, targ)
    assert not np.array_equal(eng2.debug_tensor("y", 1) == 0, y1 == 0)
    eng.close()
    eng2.close()


def test_enhance_lps_tool_matches_decode_m_math(pkg, pyoracle, tmp_path):
    """the inference tool vs a float64 restatement of Test_code/decode.m +
    frame_expand.m (edge-replicated 7-frame context, sigmoid MLP, de-normalisation, HTK out) -- and, bit for bit, vs the
    same pipeline in float32 with the oracle's MFMA-order twin as the MLP."""
    import struct
    subprocess.check_call(["make", "-C", hostlib.HOST, "-s"])
    rng = np.random.default_rng(21)
    dim, ctx, n = 257, 7, 168
    ls = [dim * ctx, 96, 80, dim]
    ws = [rng.normal(0, 0.05, (ls[i], ls[i + 1])).astype(np.float32) for i in range(3)]
    bs = [rng.normal(0, 0.1, ls[i + 1]).astype(np.float32) for i in range(3)]
    hostlib.write_wts(str(tmp_path / "mlp.wts"), ws, bs)
    mean = rng.normal(10, 2, dim).astype(np.float32)
    inv = (1.0 / rng.uniform(2, 4, dim)).astype(np.float32)
    hostlib.write_norm(str(tmp_path / "n.norm"), mean, inv)
    lps = rng.normal(10, 3, (n, dim)).astype(np.float32)
    with open(tmp_path / "noisy.lps", "wb") as f:                      # HTK big-endian, as Wav2LPS_be writes it
        f.write(struct.pack(">iihh", n, 160000, dim * 4, 9) + lps.astype(">f4").tobytes())
    res = subprocess.run([os.path.join(hostlib.HOST, "enhance_lps"), "wts=%s" % (tmp_path / "mlp.wts"),
                          "norm_file=%s" % (tmp_path / "n.norm"), "in=%s" % (tmp_path / "noisy.lps"),
                          "out=%s" % (tmp_path / "out.htk"), "fea_context=7"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(tmp_path / "out.htk", "rb").read()
    assert struct.unpack(">iihh", raw[:12]) == (n, 160000, dim * 4, 9)
    got = np.frombuffer(raw[12:], ">f4").reshape(n, dim).astype(np.float64)
    norm_mean, norm_inv = hostlib.HostNorm.read(str(tmp_path / "n.norm"), dim)   # what the tool parsed (text file)
    x = (lps - norm_mean) * norm_inv                                   # decode.m:31-33
    idx = np.clip(np.arange(n)[:, None] + np.arange(-3, 4)[None, :], 0, n - 1)   # frame_expand.m
    a = x[idx].reshape(n, ctx * dim).astype(np.float64)
    for i in range(3):
        a = a @ ws[i].astype(np.float64) + bs[i]
        if i < 2:
            a = 1.0 / (1.0 + np.exp(-a))
    want = a / norm_inv + norm_mean                                    # decode.m:59-61
    assert np.abs(got - want).max() < 2e-4 * np.abs(want).max()
    # the same in float32 around the oracle's MFMA-order twin (the tool's forward pass runs in bunches of 512 frames)
    probe = pkg.BPGpu(1, 0, ls, 512, 0.1, 0.9, 1e-5, ws, bs, 2.0, 0)
    pyoracle.set_gemm_order("hip", probe.out_slabs(), plan=probe.gemm_plan())
    probe.close()
    try:
        tw = pyoracle.OracleNet(ls, 512, 0.1, 0.9, 1e-5, 2.0, 0, ws, bs)
        y32 = tw.cv_forward(np.ascontiguousarray(x[idx].reshape(n, ctx * dim), np.float32))
        tw.close()
    finally:
        pyoracle.set_gemm_order("ref")
    want32 = (y32 / norm_inv.astype(np.float32) + norm_mean.astype(np.float32)).astype(np.float32)
    assert np.array_equal(np.frombuffer(raw[12:], ">f4").reshape(n, dim).astype(np.float32), want32)


def test_roctx_ranges_can_be_switched_on(pkg, synth):
    """MLGGD_ROCTX=1 (read once per process, at the first mlggd_create): the step's phases are bracketed with
    roctxRangePush / Pop for `rocprofv3 --marker-trace`; the library is dlopen'ed only then.  A child process trains two
    steps with the ranges on and must reproduce this process's weights bit for bit."""
    import subprocess
    import sys
    import zlib
    code = (
        "import importlib, sys, zlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "pkg = importlib.import_module(%r); synth = importlib.import_module(%r + '.synth')\n"
        "ls, B = [257 * 3, 128, 96, 257], 64\n"
        "ws, bs = synth.make_weights(ls, seed=5); inp, targ = synth.make_frames(2 * B, 257, 3, seed=6)\n"
        "eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 1.2, 1)\n"
        "assert eng.train(inp, targ) == 2\n"
        "w, b = eng.returnWeights(); eng.close()\n"
        "print('CRC', zlib.crc32(b''.join(x.tobytes() for x in w + b)))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pkg.__name__, pkg.__name__)
    env = dict(os.environ, MLGGD_ROCTX="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    assert "no roctx library" not in r.stderr
    ls, B = [257 * 3, 128, 96, 257], 64
    ws, bs = synth.make_weights(ls, seed=5)
    inp, targ = synth.make_frames(2 * B, 257, 3, seed=6)
    eng = pkg.BPGpu(1, 0, ls, B, 0.1, 0.9, 1e-5, ws, bs, 1.2, 1)
    assert eng.train(inp, targ) == 2
    w, b = eng.returnWeights()
    eng.close()
    assert "CRC %d" % zlib.crc32(b"".join(x.tobytes() for x in w + b)) in r.stdout


def test_bench_rehearses_the_data_parallel_path_on_one_gpu():
    """`bench.py --rehearse-dp`: the N-rank code of the benchmark -- RCCL communicator, the engine's exchange path, the
    ml_ggd leg, every exchange arm with its dp_breakdown, teardown -- on ONE GPU through a 1-rank communicator, i.e.
    everything of a multi-GPU run except the links.  (The launch / gloo side of it runs on the CPU:
    tests/test_bench_launcher.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-dp", "--steps", "5", "--warmup", "2",
                        "--windows", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "rehearsal" in d and d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and "incomplete" not in d
    assert d["config"]["dp_mode"] == "gather" and d["roofline"]["launches_timed"] == 64   # one dW launch per step (grouped exchange)
    assert list(d["dp_arms"]) == ["allreduce", "shard_a2a", "gather", "shard", "gather_other_granularity",
                                  "headline_mode_without_mainline", "allreduce_unsharded_update"]
    assert d["dp_arms"]["gather"]["same_as"] == "headline" and d["skipped"] == []        # everything fits the default budget
    for arm in ("allreduce", "shard", "shard_a2a", "gather_other_granularity", "headline_mode_without_mainline",
                "allreduce_unsharded_update"):
        a = d["dp_arms"][arm]
        assert a["value"] > 1e5 and a["dp_breakdown"]["compute_us_by_class"]["dw"] > 0, (arm, a)
    # the all-reduce arm updates the rank's block only (here the whole matrix: one rank) in one pass per layer
    assert d["dp_arms"]["allreduce"]["dp_breakdown"]["compute_us_by_class"]["update"] > 0   # k_apply_update ran
    assert d["ml_ggd"]["value"] > 1e5 and d["ml_ggd"]["dp_breakdown"]["compute_us_by_class"]["loss"] > 0
    assert d["ml_ggd"]["stat_comm"]["value"] > 1e5                                        # ncclCommSplit worked
    assert d["value"] > 1e5 and 0 < d["ms_per_step"] < 5
    # the parity leg of a multi-rank line (here world = 1): HIP through the communicator vs the oracle at bunchsize world x B
    for leg in (d["loss_vs_oracle"], d["ml_ggd"]["loss_vs_oracle"]):
        assert leg["replicas_identical"] is True and leg["steps"] == 5 and leg["oracle_bunchsize"] == 128
        assert leg["cv_sqerr_rel"] < 1e-4 and leg["cv_abserr_rel"] < 1e-4 and leg["weights_relmax"] < 2e-5
    assert d["ml_ggd"]["loss_vs_oracle"]["cv_loglik_rel"] < 1e-4
    assert d["loss_vs_oracle"]["hip_equals_mfma_order_twin_bitwise"] is True   # MMSE through the factor exchange: the twin's bits
