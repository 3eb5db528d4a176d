"""CPU, world_size 2, gloo: the data-parallel contract of SURVEY.md 8e.

Each rank runs the oracle's phases on ITS rows of every global minibatch (rows chosen by the
product's shard_rows), exchanges exactly what libmlggd.so exchanges over RCCL -- the
per-dimension sum |e|^beta (ML only) and either the weight/bias gradients (fp32 sum, "allreduce")
or their factors, the activations and dEdX of every rank ("gather") -- and applies the update with
the GLOBAL minibatch size.  The result must equal a single-process run with
bunchsize = world * B_local (up to summation order)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"
LS, BL, STEPS = [33, 24, 17, 11], 16, 3
HP = (0.1, 0.9, 1e-5)


def _worker(rank, world, ml, beta, initfile, outdir, mode="allreduce"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import pyoracle
    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    ws, bs = synth.make_weights(LS, seed=8)
    inp, targ = synth.make_frames(STEPS * world * BL, 11, 3, seed=9)
    net = pyoracle.OracleNet(LS, BL, *HP, beta, ml, ws, bs)
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

        for l in range(1, len(LS)):
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
        net.apply_update(gb)
    w, b = net.get_weights()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), *w, *b, alpha=net.tensor("scalefactor"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "gather"])
@pytest.mark.parametrize("ml,beta", [(0, 2.0), (1, 1.2)])
def test_two_ranks_equal_single_process_with_doubled_bunch(tmp_path, pyoracle, synth, ml, beta, mode):
    """mode = which exchange of libmlggd.so is mirrored: all-reduce of the gradients, or all-gather of
    their factors (Y_{l-1}, dEdX_l) with the global gradient formed on every rank."""
    import torch.multiprocessing as mp
    world = 2
    initfile = str(tmp_path / "rendezvous")
    mp.spawn(_worker, args=(world, ml, beta, initfile, str(tmp_path), mode), nprocs=world, join=True)
    ws, bs = synth.make_weights(LS, seed=8)
    inp, targ = synth.make_frames(STEPS * world * BL, 11, 3, seed=9)
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
