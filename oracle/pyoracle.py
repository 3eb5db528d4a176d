"""ctypes binding of the CPU oracle (oracle/libmlggd_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmlggd_oracle.so")
_LIB_FMA = os.path.join(_HERE, "libmlggd_oracle_fma.so")  # same source, FMA contraction on (ambiguity viii)
_libs = {}

_fp = C.POINTER(C.c_float)
_fpp = C.POINTER(_fp)


def build(force=False):
    src = os.path.getmtime(os.path.join(_HERE, "mlggd_oracle.c"))
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < src for p in (_LIB, _LIB_FMA)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib(variant="strict"):
    """variant "strict": no a*b+c is fused (the oracle every parity test uses); "fma": contraction on."""
    if variant not in _libs:
        build()
        L = C.CDLL({"strict": _LIB, "fma": _LIB_FMA}[variant])
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_int, _fpp, _fpp]
        L.ora_destroy.argtypes = [C.c_void_p]
        L.ora_set_dropout.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ora_train.restype = C.c_int
        L.ora_train.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        L.ora_train_bunch.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        L.ora_forward.argtypes = [C.c_void_p, C.c_int, _fp]
        L.ora_loss_colsum.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        L.ora_loss_grad.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp]
        L.ora_backward.argtypes = [C.c_void_p, C.c_int, _fp]
        L.ora_apply_update.argtypes = [C.c_void_p, C.c_int]
        L.ora_cv_bunch.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        for f in (L.ora_cv_sqerr, L.ora_cv_abserr, L.ora_cv_loglik):
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        L.ora_gamma.restype = C.c_float
        L.ora_gamma.argtypes = [C.c_float]
        L.ora_get_weights.argtypes = [C.c_void_p, _fpp, _fpp]
        L.ora_set_scalefactor.argtypes = [C.c_void_p, _fp]
        L.ora_tensor.restype = _fp
        L.ora_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_long)]
        L.ora_num_threads.restype = C.c_int
        L.ora_set_num_threads.argtypes = [C.c_int]
        L.ora_set_gemm_split.argtypes = [C.c_int]
        L.ora_set_gemm_blocked.argtypes = [C.c_int]
        L.ora_set_gemm_order.argtypes = [C.c_int, C.c_int]
        L.ora_set_gemm_plan.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ora_set_dp_twin.argtypes = [C.c_int, C.c_int]
        L.ora_exp_det_array.argtypes = [_fp, _fp, C.c_long, C.c_int]
        L.ora_pow_det_array.argtypes = [_fp, C.c_float, _fp, C.c_long]
        if "OMP_NUM_THREADS" not in os.environ:
            # a container often sees all host CPUs but may only use a share of them: more threads than
            # that share makes every OpenMP region slower, not faster
            L.ora_set_num_threads(usable_cpus())
        _libs[variant] = L
    return _libs[variant]


def usable_cpus():
    """CPUs this process may actually use: affinity mask and cgroup CPU quota (v2 and v1)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                quota, period = parse(txt)
                if quota != "max":
                    n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            else:
                quota = int(txt)
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                if quota > 0 and period > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def _ptr_array(arrs):
    """float*[numlayers] with index 0 unused (NULL), like BP_GPU's float** args."""
    n = len(arrs) + 1
    pa = (_fp * n)()
    for i, a in enumerate(arrs):
        pa[i + 1] = a.ctypes.data_as(_fp)
    return pa


class OracleNet:
    """Mirror of the reference's class BP_GPU (BP_GPU.h:45-70) on the CPU oracle."""

    def __init__(self, layersizes, bunchsize, lrate, momentum, weightcost, shapefactor, MLflag,
                 weights, bias, variant="strict", dropoutflag=0, visible_omit=0.0, hid_omit=0.0, random_seed=0):
        self._lib = lib(variant)
        self.layersizes = [int(x) for x in layersizes]
        self.L = len(self.layersizes)
        self.bunchsize = int(bunchsize)
        self.D = self.layersizes[-1]
        self._w = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self._b = [np.ascontiguousarray(b, dtype=np.float32) for b in bias]
        assert len(self._w) == self.L - 1
        for l in range(1, self.L):
            assert self._w[l - 1].shape == (self.layersizes[l - 1], self.layersizes[l])
        ls = (C.c_int * self.L)(*self.layersizes)
        self._h = self._lib.ora_create(self.L, ls, self.bunchsize, lrate, momentum, weightcost,
                                   shapefactor, int(MLflag), _ptr_array(self._w), _ptr_array(self._b))
        assert self._h
        if dropoutflag == 1:  # the HIP engine's counter-hash generator, restated (documented deviation from cuRAND)
            self._lib.ora_set_dropout(self._h, 1, visible_omit, hid_omit, int(random_seed))

    def close(self):
        if self._h:
            self._lib.ora_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def train(self, inp, targ):
        inp, pi = _f32(inp)
        targ, pt = _f32(targ)
        return self._lib.ora_train(self._h, inp.shape[0], pi, pt)

    def train_bunch(self, inp, targ):
        inp, pi = _f32(inp)
        targ, pt = _f32(targ)
        self._lib.ora_train_bunch(self._h, inp.shape[0], pi, pt)

    # phases (data-parallel contract)
    def forward(self, inp):
        inp, pi = _f32(inp)
        self._lib.ora_forward(self._h, inp.shape[0], pi)

    def loss_colsum(self, targ):
        targ, pt = _f32(targ)
        out = np.zeros(self.D, np.float32)
        self._lib.ora_loss_colsum(self._h, targ.shape[0], pt, out.ctypes.data_as(_fp))
        return out

    def loss_grad(self, targ, n_global, colsum):
        targ, pt = _f32(targ)
        cs, pc = _f32(colsum)
        self._lib.ora_loss_grad(self._h, targ.shape[0], int(n_global), pt, pc)

    def backward(self, inp):
        inp, pi = _f32(inp)
        self._lib.ora_backward(self._h, inp.shape[0], pi)

    def apply_update(self, n_global):
        self._lib.ora_apply_update(self._h, int(n_global))

    def cv_forward(self, inp):
        inp, pi = _f32(inp)
        out = np.zeros((inp.shape[0], self.D), np.float32)
        self._lib.ora_cv_bunch(self._h, inp.shape[0], pi, out.ctypes.data_as(_fp))
        return out

    def cv_sqerr(self, inp, targ):
        inp, pi = _f32(inp)
        targ, pt = _f32(targ)
        return float(self._lib.ora_cv_sqerr(self._h, inp.shape[0], pi, pt))

    def cv_abserr(self, inp, targ):
        inp, pi = _f32(inp)
        targ, pt = _f32(targ)
        return float(self._lib.ora_cv_abserr(self._h, inp.shape[0], pi, pt))

    def cv_loglik(self, inp, targ):
        inp, pi = _f32(inp)
        targ, pt = _f32(targ)
        return float(self._lib.ora_cv_loglik(self._h, inp.shape[0], pi, pt))

    def get_weights(self):
        ws = [np.zeros((self.layersizes[l - 1], self.layersizes[l]), np.float32) for l in range(1, self.L)]
        bs = [np.zeros(self.layersizes[l], np.float32) for l in range(1, self.L)]
        self._lib.ora_get_weights(self._h, _ptr_array(ws), _ptr_array(bs))
        return ws, bs

    def set_scalefactor(self, alpha):
        a, pa = _f32(alpha)
        assert a.shape == (self.D,)
        self._lib.ora_set_scalefactor(self._h, pa)

    def tensor(self, name, layer=0, rows=None):
        cnt = C.c_long(0)
        p = self._lib.ora_tensor(self._h, name.encode(), int(layer), C.byref(cnt))
        if not p:
            raise KeyError(name)
        a = np.ctypeslib.as_array(p, shape=(cnt.value,)).copy()
        if name in ("grad_w", "delta_w", "weights"):
            return a.reshape(self.layersizes[layer - 1], self.layersizes[layer])
        if name in ("out",):
            a = a.reshape(-1, self.D)
            return a if rows is None else a[:rows]
        if name in ("x", "y", "dedx", "dedy"):
            a = a.reshape(-1, self.layersizes[layer])
            return a if rows is None else a[:rows]
        return a


def set_gemm_split(s, variant="strict"):
    """GEMM summation-order twin of the oracle (process-wide for that library): 1 = the documented orders;
    S > 1 = forward / dX reductions as S contiguous partial sums (what a split-K GEMM does).  Tests use the distance
    between the two to say what a mere change of summation order -- which cuBLAS leaves open -- does to a run."""
    lib(variant).ora_set_gemm_split(int(s))


def set_gemm_blocked(on, variant="strict"):
    """True (default): the register-blocked forms of the GEMM loops; False: the plain loops that define the oracle.  The
    two run the same chain per output element and give the same bits (tests/test_oracle.py)."""
    lib(variant).ora_set_gemm_blocked(1 if on else 0)


def set_gemm_order(order, s_out=1, variant="strict", plan=None, dp_world=1, dp_allreduce=False):
    """MFMA-order twin (process-wide for that library): order "hip" / 1 = the HIP kernels' own summation order with
    fused multiply-adds -- forward / dX reductions over the 4 waves' contiguous ranges, the output layer over `s_out`
    slabs x 4 waves (the engine's choice: BPGpu.out_slabs()), dW over the frames in order; "ref" / 0 = the documented
    orders every parity test compares against.  With order "hip" the GEMMs of the HIP path equal this CPU model bit for
    bit (tests/test_gpu_mfma_order.py), which leaves libm (expf, powf) as the only difference between the two.
    plan: [(fwd_waves, dx_waves)] per layer 1..L-1 (BPGpu.gemm_plan()): 4 = reduction over the 4 waves of a 32 x 32-tile
    workgroup (default), 1 = the 64 x 64-tile kernels' single chain per output element.
    dp_world > 1: the data-parallel form -- `dp_world` ranks of bunchsize / dp_world frames whose ML statistic (k_colsum's
    wavefront reduction per rank) and, with dp_allreduce, weight / bias gradients meet in rank order, as the one-GPU
    emulation of a world (BPGpu.fake_world) does."""
    lib(variant).ora_set_gemm_order(1 if order in (1, "hip") else 0, int(s_out))
    for l, (fw, dx) in enumerate(plan or [], start=1):  # BPGpu.gemm_plan(): which GEMM kernel each layer takes
        lib(variant).ora_set_gemm_plan(l, int(fw), int(dx))
    lib(variant).ora_set_dp_twin(int(dp_world), 1 if dp_allreduce else 0)


def exp_det(x, sigmoid=False, variant="strict"):
    """The HIP kernels' exponential (csrc/kernels.hip.h exp_det, restated: same statements, IEEE operations only), or
    the sigmoid 1 / (1 + exp_det(-x)) built on it, elementwise."""
    x, px = _f32(np.ravel(x))
    out = np.empty_like(x)
    lib(variant).ora_exp_det_array(px, out.ctypes.data_as(_fp), x.size, 1 if sigmoid else 0)
    return out


def pow_det(x, y, variant="strict"):
    """x ** y (x >= 0) as the HIP loss kernels evaluate it (csrc/kernels.hip.h pow_det, restated), elementwise."""
    x, px = _f32(np.ravel(x))
    out = np.empty_like(x)
    lib(variant).ora_pow_det_array(px, float(np.float32(y)), out.ctypes.data_as(_fp), x.size)
    return out


def gamma(x):
    return float(lib().ora_gamma(float(x)))


def num_threads():
    return int(lib().ora_num_threads())
