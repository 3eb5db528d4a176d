"""TEST INFRASTRUCTURE ONLY: a do-nothing stand-in for the product package, with just the surface bench.py touches, so
that tests/test_bench_launcher.py can run bench.py's multi-rank code (self-launch, gloo rendezvous, windows, exchange
arms, dp_breakdown, teardown) on a machine without GPUs.  Every "measurement" it returns is made up.  bench.py only
imports it under --stub-engine, which marks the printed line STUB_ENGINE."""
import os
import time

UNIQUE_ID_BYTES = 128


class MlggdError(RuntimeError):
    pass


def load():
    return None


def comm_unique_id():
    return bytes(range(128))


class BPGpu:
    def __init__(self, seed, gpu, layersizes, bunchsize, lrate, momentum, weightcost, weights, bias, shapefactor, MLflag):
        self.ls, self.B, self.world, self.rank, self.mode = list(layersizes), int(bunchsize), 1, 0, 0
        self.cls, self.n = None, 0
        self.closed = False
        self.w, self.b = [x.copy() for x in weights], [x.copy() for x in bias]

    def comm_init(self, uid, world, rank):
        assert uid == comm_unique_id() and 0 <= rank < world
        m = os.environ.get("MLGGD_DP_MODE")
        usable = self.B % 32 == 0 and world * self.B in (64, 128, 256, 512, 1024)
        if m in ("gather", "shard", "shard_a2a") and not usable:
            raise MlggdError("mlggd error 1: MLGGD_DP_MODE=%s needs bunchsize %% 32 == 0 ..." % m)
        self.mode = {"allreduce": 1, "gather": 2, "shard": 3, "shard_a2a": 4}.get(m, (3 if world >= 6 else 2) if usable else 1)
        self.world, self.rank = world, rank

    def load_chunk(self, inp, targ):
        assert inp.shape[0] == targ.shape[0] and inp.shape[1] == self.ls[0] and targ.shape[1] == self.ls[-1]
        self.frames = inp.shape[0]

    def train_resident(self, first, n):
        assert first >= 0 and first + n <= self.frames
        steps = n // self.B
        time.sleep(float(os.environ.get("MLGGD_BENCH_STUB_STEP_S", "20e-6")) * steps)
        if self.cls:
            self.n += steps
        return steps

    def sync(self):
        pass

    def dp_mode(self):
        return self.mode

    def comm_info(self):
        return (self.world, self.rank) if self.world > 1 else (0, -1)

    def profile_select(self, cls, layer=0, max_launches=4096, stride=1):
        self.cls, self.n = cls, 0

    def profile_read(self):
        n, self.n = self.n, 0
        return (10.0, n) if self.cls in ("fwd", "dx", "dw", "loss") else (0.0, 0)

    def dw_launches_per_step(self):
        return 1 if self.mode in (0, 3, 4) else 2 if self.mode == 2 else len(self.ls) - 1

    def kernel_work(self, cls, layer=0):
        return 3.7e9, 2.4e8

    # parity leg (made-up numbers: the stub only has to carry the control flow)
    def returnWeights(self):
        return self.w, self.b

    def cv_all(self, inp, targ):
        return 1.0, 1.0, 1.0

    def scalefactor(self):
        import numpy as np
        return np.ones(self.ls[-1], np.float32)

    def close(self):
        self.closed = True
