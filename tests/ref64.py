"""Independent float64 NumPy model of one training step / CV metrics.

Written from the math in SURVEY.md 3.2 (not from the oracle's C code) to guard the
oracle's restatement: same formulas, float64 everywhere, library matmul.
"""
import math

import numpy as np


class Ref64:
    def __init__(self, layersizes, lrate, momentum, weightcost, shapefactor, MLflag, weights, bias):
        self.ls = list(layersizes)
        self.lr, self.mu, self.wc, self.beta, self.ml = lrate, momentum, weightcost, shapefactor, MLflag
        self.W = [np.asarray(w, np.float64).copy() for w in weights]
        self.b = [np.asarray(b, np.float64).copy() for b in bias]
        self.dW = [np.zeros_like(w) for w in self.W]
        self.db = [np.zeros_like(b) for b in self.b]
        self.alpha = None

    def forward(self, x):
        ys = [np.asarray(x, np.float64)]
        nl = len(self.W)
        for i in range(nl):
            z = ys[-1] @ self.W[i] + self.b[i]
            ys.append(1.0 / (1.0 + np.exp(-z)) if i < nl - 1 else z)
        return ys

    def loss_grad(self, out, targ, n_global=None):
        n = out.shape[0] if n_global is None else n_global
        e = out - np.asarray(targ, np.float64)
        beta = self.beta
        with np.errstate(divide="ignore", invalid="ignore"):
            p = np.where(e == 0, 0.0, np.sign(e) * np.abs(e) ** (beta - 1.0))
        if self.ml == 1:
            colsum = (np.abs(e) ** beta).sum(axis=0)
            self.alpha = (beta * colsum / n) ** (1.0 / beta)
            g = p * beta / self.alpha ** beta
        else:
            g = beta * p
        return g / n

    def step(self, x, targ):
        n = x.shape[0]
        ys = self.forward(x)
        d = self.loss_grad(ys[-1], targ)
        nl = len(self.W)
        grads = [None] * nl
        gbs = [None] * nl
        for i in range(nl - 1, -1, -1):
            grads[i] = ys[i].T @ d
            gbs[i] = d.sum(axis=0)
            if i > 0:
                dy = d @ self.W[i].T
                d = (1.0 - ys[i]) * ys[i] * dy
        for i in range(nl):
            self.dW[i] = self.mu * self.dW[i] - self.lr * (grads[i] / n + self.wc * self.W[i])
            self.db[i] = self.mu * self.db[i] - self.lr * (gbs[i] / n)
            self.W[i] += self.dW[i]
            self.b[i] += self.db[i]
        return grads, gbs

    def cv(self, x, targ):
        out = self.forward(x)[-1]
        e = out - np.asarray(targ, np.float64)
        D = e.shape[1]
        res = {"sqerr": float((e * e).sum()), "abserr": float(np.abs(e).sum() / D)}
        if self.alpha is not None:
            n = e.shape[0]
            beta = self.beta
            res["loglik"] = float(
                n * D * math.log(beta / (2.0 * math.gamma(1.0 / beta)))
                - n * np.log(self.alpha).sum()
                - ((np.abs(e) / self.alpha) ** beta).sum()
            )
        return res
