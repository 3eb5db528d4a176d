"""CPU, world_size 2, gloo: the data-parallel contract of SURVEY.md 8e.

Each rank runs the oracle's phases on ITS rows of every global minibatch (rows chosen by the
product's shard_rows), exchanges exactly what libmlggd.so exchanges over RCCL -- the
per-dimension sum |e|^beta (ML only) and either the weight/bias gradients (fp32 sum, "allreduce")
or their factors, the activations and dEdX of every rank ("gather") -- and applies the update with
the GLOBAL minibatch size.  The result must equal a single-process run with
bunchsize = world * B_local (up to summation order).

Round 4 adds the two sharded forms: "allreduce_rs" -- each rank receives the summed gradient of ITS block of weight rows
(the engine's reduce-scatter), updates W and delta for that block only and the ranks all-gather the W blocks; and
"shard_a2a" -- the sharded update from the factors, with the activations exchanged by ALL-TO-ALL: rank o receives from
every rank only the columns of Y_{l-1} that belong to its block (point-to-point sends of the owner-blocked pieces,
what the engine does with ncclSend / ncclRecv), dEdX_l all-gathered.  Blocks come from the product's own
`weight_row_block`."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
LS, BL, STEPS = [33, 24, 17, 11], 16, 3
LS_WIDE = [11 * 13, 150, 70, 11]   # wide enough for two 64-row tile rows per layer: both ranks own a block (the last one short)
HP = (0.1, 0.9, 1e-5)


def _worker(rank, world, ml, beta, initfile, outdir, mode="allreduce"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import pyoracle
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    LS = LS_WIDE if mode in ("allreduce_rs", "shard_a2a") else globals()["LS"]
    ctx = LS[0] // 11
    ws, bs = synth.make_weights(LS, seed=8)
    inp, targ = synth.make_frames(STEPS * world * BL, 11, ctx, seed=9)
    net = pyoracle.OracleNet(LS, BL, *HP, beta, ml, ws, bs)
    lr, mom, wc = (np.float32(v) for v in HP)
    gb = world * BL

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    for s in range(STEPS):
        lo, hi = pkg.shard_rows(gb, world, rank)
        x = inp[s * gb + lo:s * gb + hi]
        t = targ[s * gb + lo:s * gb + hi]
        net.forward(x)
        colsum = allreduce(net.loss_colsum(t)) if ml else np.zeros(LS[-1], np.float32)
        net.loss_grad(t, gb, colsum)
        net.backward(x)
        import ctypes as C

        def grad_view(name, l):
            cnt = C.c_long(0)
            p = pyoracle.lib().ora_tensor(net._h, name.encode(), l, C.byref(cnt))
            return np.ctypeslib.as_array(p, shape=(cnt.value,))

        def allgather_rows(a):
            t = torch.from_numpy(np.ascontiguousarray(a, np.float32))
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return np.concatenate([o.numpy() for o in out])  # rank-major rows = rows of the global minibatch

        def update_block_and_allgather(l, g_block, lo, hi):
            """kernUpdatedelta + kernAccSum (DevFunc.cu:490-507,427-443) on rows [lo, hi) of layer l only, then the W blocks
            of all ranks all-gathered in place -- what the engine's sharded updates do"""
            K, N = LS[l - 1], LS[l]
            W = grad_view("weights", l).reshape(K, N)
            D = grad_view("delta_w", l).reshape(K, N)
            if hi > lo:
                D[lo:hi] = mom * D[lo:hi] - lr * (g_block / np.float32(gb) + wc * W[lo:hi])
                W[lo:hi] = D[lo:hi] + np.float32(1.0) * W[lo:hi]
            for r in range(world):  # blocks differ in size: one broadcast per owner
                blo, bhi = pkg.weight_row_block(K, world, r)
                if bhi > blo:
                    t_ = torch.from_numpy(np.ascontiguousarray(W[blo:bhi]))
                    dist.broadcast(t_, src=r)
                    W[blo:bhi] = t_.numpy()

        def bias_allreduce_update(l):
            gbv = allreduce(grad_view("grad_b", l).copy())
            bv, db = grad_view("bias", l), grad_view("delta_b", l)
            db[:] = mom * db - lr * (gbv / np.float32(gb) + np.float32(0.0) * bv)
            bv[:] = db + np.float32(1.0) * bv

        for l in range(1, len(LS)):
            K, N = LS[l - 1], LS[l]
            lo_k, hi_k = pkg.weight_row_block(K, world, rank)
            if mode == "allreduce_rs":
                # reduce-scatter: this rank ends up with the sum of ITS rows of G_l (gloo has no reduce_scatter: all-reduce
                # and keep the block -- the same values); biases: all-reduce + the replicated bias update
                g = allreduce(grad_view("grad_w", l).copy()).reshape(K, N)
                update_block_and_allgather(l, g[lo_k:hi_k], lo_k, hi_k)
                bias_allreduce_update(l)
                continue
            if mode == "shard_a2a":
                # all-to-all of the activations: rank o receives, from every rank s, columns [block o) of s's Y_{l-1} rows
                y_loc = np.ascontiguousarray((x if l == 1 else net.tensor("y", l - 1, rows=BL)).astype(np.float32))
                pieces = [None] * world
                reqs = []
                for peer in range(world):
                    plo, phi = pkg.weight_row_block(K, world, peer)
                    if peer == rank:
                        pieces[rank] = y_loc[:, lo_k:hi_k].copy()
                        continue
                    if phi > plo:
                        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(y_loc[:, plo:phi])), dst=peer))
                    if hi_k > lo_k:
                        buf = torch.empty((BL, hi_k - lo_k), dtype=torch.float32)
                        reqs.append(dist.irecv(buf, src=peer))
                        pieces[peer] = buf
                for rq in reqs:
                    rq.wait()
                d_all = allgather_rows(net.tensor("dedx", l, rows=BL))
                if hi_k > lo_k:
                    y_blk = np.concatenate([np.asarray(p_) for p_ in pieces])          # [world x B][block], rank-major rows
                    g_blk = (y_blk.T.astype(np.float32) @ d_all.astype(np.float32))
                else:
                    g_blk = np.zeros((0, N), np.float32)
                update_block_and_allgather(l, g_blk, lo_k, hi_k)
                # biases: every rank forms the same sum from the gathered dEdX (the engine's bias-only tiles)
                bv, db = grad_view("bias", l), grad_view("delta_b", l)
                db[:] = mom * db - lr * (d_all.sum(axis=0, dtype=np.float32) / np.float32(gb) + np.float32(0.0) * bv)
                bv[:] = db + np.float32(1.0) * bv
                continue
            if mode == "allreduce":
                # the engine all-reduces G_l / gb_l in place; here: write the reduced values back
                for name in ("grad_w", "grad_b"):
                    view = grad_view(name, l)
                    view[:] = allreduce(view.copy())
            else:
                # the engine all-gathers the FACTORS and every rank forms the global gradient itself
                y_all = allgather_rows(x if l == 1 else net.tensor("y", l - 1, rows=BL))
                d_all = allgather_rows(net.tensor("dedx", l, rows=BL))
                grad_view("grad_w", l)[:] = (y_all.T.astype(np.float32) @ d_all.astype(np.float32)).ravel()
                grad_view("grad_b", l)[:] = d_all.sum(axis=0, dtype=np.float32)
        if mode in ("allreduce", "gather"):
            net.apply_update(gb)
    w, b = net.get_weights()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), *w, *b, alpha=net.tensor("scalefactor"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "gather", "allreduce_rs", "shard_a2a"])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_two_ranks_equal_single_process_with_doubled_bunch(tmp_path, pyoracle, synth, ml, beta, mode):
    """mode = which exchange of libmlggd.so is mirrored: all-reduce of the gradients, all-gather of their factors
    (Y_{l-1}, dEdX_l) with the global gradient formed on every rank, reduce-scatter + update of the rank's block of weight
    rows + all-gather of W, or the sharded update with the activations by all-to-all."""
    import torch.multiprocessing as mp
    world = 2
    initfile = str(tmp_path / "rendezvous")
    mp.spawn(_worker, args=(world, ml, beta, initfile, str(tmp_path), mode), nprocs=world, join=True)
    LS = LS_WIDE if mode in ("allreduce_rs", "shard_a2a") else globals()["LS"]
    ws, bs = synth.make_weights(LS, seed=8)
    inp, targ = synth.make_frames(STEPS * world * BL, 11, LS[0] // 11, seed=9)
    single = pyoracle.OracleNet(LS, world * BL, *HP, beta, ml, ws, bs)
    assert single.train(inp, targ) == STEPS
    w, b = single.get_weights()
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for i in range(3):
        assert np.array_equal(r0["arr_%d" % i], r1["arr_%d" % i])            # replicas stay identical
        d = np.abs(r0["arr_%d" % i] - w[i]).max() / np.abs(w[i]).max()
        assert d < 1e-5, (i, d)
        d = np.abs(r0["arr_%d" % (3 + i)] - b[i]).max() / max(np.abs(b[i]).max(), 1e-12)
        assert d < 1e-4, (i, d)
    if ml:
        assert np.allclose(r0["alpha"], single.tensor("scalefactor"), rtol=1e-6)
