"""Synthetic inputs for tests and bench.py: numpy only, all seeds fixed.

* features: an N(0,1) stream of [n + ctx - 1][fea_dim] frames context-expanded to
  [n][fea_dim*ctx] rows, so consecutive rows overlap like real pfile data
  (reference host code: Interface.cc:778-785);
* targets: 0.5 * centre frame + 0.5 * N(0,1)  ->  [n][fea_dim];
* weights: U(+-beta*sqrt(6)/sqrt(n_in+n_out)), biases 0 -- the rule of the reference's
  init tool (pretraining_weights/Gen_rand_net.cpp:84-103, flag=1, beta=2), drawn from
  numpy's PCG64 rather than rand().
"""
import numpy as np

DEFAULT_SEED = 27870775  # finetune.pl:31


def make_weights(layersizes, seed=DEFAULT_SEED, beta=2.0):
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for l in range(1, len(layersizes)):
        k, n = layersizes[l - 1], layersizes[l]
        r = beta * np.sqrt(6.0) / np.sqrt(k + n)
        ws.append(rng.uniform(-r, r, size=(k, n)).astype(np.float32))
        bs.append(np.zeros(n, np.float32))
    return ws, bs


def make_frames(n, fea_dim, ctx, seed=DEFAULT_SEED + 1):
    """Returns (in [n][fea_dim*ctx], targ [n][fea_dim]) float32, C-contiguous."""
    rng = np.random.default_rng(seed)
    stream = rng.standard_normal((n + ctx - 1, fea_dim), dtype=np.float32)
    idx = np.arange(n)[:, None] + np.arange(ctx)[None, :]
    inp = np.ascontiguousarray(stream[idx].reshape(n, ctx * fea_dim))
    centre = stream[np.arange(n) + (ctx - 1) // 2]
    noise = rng.standard_normal((n, fea_dim), dtype=np.float32)
    targ = np.ascontiguousarray((0.5 * centre + 0.5 * noise).astype(np.float32))
    return inp, targ


def baseline_layersizes(fea_dim=257, ctx=11, hidden=2048, nhid=3):
    return [fea_dim * ctx] + [hidden] * nhid + [fea_dim]
