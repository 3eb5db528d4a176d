#!/usr/bin/env python3
"""Where the CPU oracle's step goes (forward / loss / backward GEMMs / update) at 2827-2048^3-257, B = 128, with the
register-blocked and the plain GEMM loops -- the make-up of bench.py's cpu_baseline on this host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po
ls = [2827, 2048, 2048, 2048, 257]; B = 128
rng = np.random.default_rng(0)
W = [(rng.standard_normal((ls[i], ls[i + 1])) * 0.05).astype(np.float32) for i in range(4)]
b = [np.zeros(ls[i + 1], np.float32) for i in range(4)]
x = rng.standard_normal((B, ls[0])).astype(np.float32); t = rng.standard_normal((B, 257)).astype(np.float32)
print("threads", po.num_threads())
for name, blocked in (("blocked", True), ("plain", False)):
    po.set_gemm_blocked(blocked)
    net = po.OracleNet(ls, B, 0.01, 0.9, 1e-5, 1.0, 1, W, b)
    net.train_bunch(x, t)
    T = {}
    for nm, f in (("forward", lambda: net.forward(x)), ("loss", lambda: (net.loss_grad(t, B, net.loss_colsum(t)))),
                  ("backward", lambda: net.backward(x)), ("update", lambda: net.apply_update(B))):
        t0 = time.perf_counter()
        for _ in range(10): f()
        T[nm] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
    t0 = time.perf_counter()
    for _ in range(10): net.train_bunch(x, t)
    print(name, "ms:", T, "| train_bunch %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
    net.close()
po.set_gemm_blocked(True)
