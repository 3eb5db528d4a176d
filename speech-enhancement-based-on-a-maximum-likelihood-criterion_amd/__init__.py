"""MI355X-native ML-GGD DNN trainer: Python host-side binding of the C-ABI (include/mlggd.h).

`BPGpu` mirrors the reference's device-engine class `BP_GPU`
(Train_code_ML_GGD/BP_GPU.h:45-70: train / CrossValid / CrossValiddB / CrossValid2 /
returnWeights) over libmlggd.so.  There is NO CPU fallback: if the HIP library is missing
or no GPU is present, construction raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_CSRC, "libmlggd.so")
MAXLAYER = 10
UNIQUE_ID_BYTES = 128

_fp = C.POINTER(C.c_float)
_fpp = C.POINTER(_fp)


class MlggdError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("random_seed", C.c_int32), ("device", C.c_int32),
        ("numlayers", C.c_int32), ("layersizes", C.c_int32 * MAXLAYER), ("bunchsize", C.c_int32),
        ("lrate", C.c_float), ("momentum", C.c_float), ("weightcost", C.c_float),
        ("shapefactor", C.c_float), ("MLflag", C.c_int32), ("dropoutflag", C.c_int32),
        ("visible_omit", C.c_float), ("hid_omit", C.c_float), ("max_cache_frames", C.c_int32),
        ("reserved", C.c_int32 * 7),
    ]


# every symbol include/mlggd.h declares (tests check the library exports all of them)
EXPORTS = [
    "mlggd_create", "mlggd_destroy", "mlggd_last_error", "mlggd_device_count", "mlggd_train_chunk",
    "mlggd_load_chunk", "mlggd_train_resident", "mlggd_sync", "mlggd_cv_sqerr", "mlggd_cv_abserr",
    "mlggd_cv_loglik", "mlggd_cv_all", "mlggd_forward", "mlggd_get_weights", "mlggd_set_weights",
    "mlggd_get_scalefactor", "mlggd_set_scalefactor", "mlggd_set_lrate", "mlggd_gamma",
    "mlggd_debug_tensor", "mlggd_comm_unique_id", "mlggd_comm_init", "mlggd_last_train_ms",
    "mlggd_profile_select", "mlggd_profile_stride", "mlggd_profile_read", "mlggd_profile_overhead",
    "mlggd_kernel_work", "mlggd_dw_launches_per_step", "mlggd_dp_mode", "mlggd_debug_fake_world",
    "mlggd_debug_stamp_select", "mlggd_debug_stamp_read",
    "mlggd_load_frames", "mlggd_train_frames", "mlggd_train_frames_async", "mlggd_cv_all_frames", "mlggd_forward_frames",
    "mlggd_alloc_pinned", "mlggd_alloc_pinned_on", "mlggd_free_pinned", "mlggd_set_cv_device_reduce",
    "mlggd_comm_info", "mlggd_debug_plan_count", "mlggd_debug_math", "mlggd_debug_out_slabs", "mlggd_debug_gemm_plan",
]

_lib = None


def build(force=False):
    """Compile libmlggd.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("engine.hip", "kernels.hip.h", "kernels64.hip.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "mlggd.h"))
    stale = not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return LIB_PATH


def load():
    """Load libmlggd.so (never builds implicitly; raises if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MlggdError("HIP extension %s is missing: run __graft_entry__.build() "
                         "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.mlggd_last_error.restype = C.c_char_p
    L.mlggd_gamma.restype = C.c_float
    L.mlggd_gamma.argtypes = [C.c_float]
    L.mlggd_create.argtypes = [C.POINTER(_Config), _fpp, _fpp, C.POINTER(C.c_void_p)]
    L.mlggd_destroy.argtypes = [C.c_void_p]
    L.mlggd_device_count.argtypes = [C.POINTER(C.c_int)]
    L.mlggd_train_chunk.argtypes = [C.c_void_p, C.c_int, _fp, _fp, C.POINTER(C.c_int)]
    L.mlggd_load_chunk.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
    L.mlggd_train_resident.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mlggd_sync.argtypes = [C.c_void_p]
    for f in (L.mlggd_cv_sqerr, L.mlggd_cv_abserr, L.mlggd_cv_loglik):
        f.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp]
    L.mlggd_cv_all.argtypes = [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp]
    L.mlggd_forward.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
    L.mlggd_get_weights.argtypes = [C.c_void_p, _fpp, _fpp]
    L.mlggd_set_weights.argtypes = [C.c_void_p, _fpp, _fpp]
    L.mlggd_get_scalefactor.argtypes = [C.c_void_p, _fp]
    L.mlggd_set_scalefactor.argtypes = [C.c_void_p, _fp]
    L.mlggd_set_lrate.argtypes = [C.c_void_p, C.c_float]
    L.mlggd_set_cv_device_reduce.argtypes = [C.c_void_p, C.c_int]
    L.mlggd_alloc_pinned_on.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mlggd_debug_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _fp, C.c_size_t]
    L.mlggd_comm_unique_id.argtypes = [C.c_void_p]
    L.mlggd_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.mlggd_last_train_ms.argtypes = [C.c_void_p, _fp, C.POINTER(C.c_int)]
    L.mlggd_profile_select.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    L.mlggd_profile_stride.argtypes = [C.c_void_p, C.c_int]
    L.mlggd_profile_read.argtypes = [C.c_void_p, _fp, C.POINTER(C.c_int)]
    L.mlggd_profile_overhead.argtypes = [C.c_void_p, _fp]
    L.mlggd_kernel_work.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]
    L.mlggd_dw_launches_per_step.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mlggd_dp_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mlggd_debug_fake_world.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mlggd_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mlggd_debug_plan_count.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mlggd_debug_out_slabs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.mlggd_debug_gemm_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mlggd_debug_math.argtypes = [C.c_void_p, C.c_char_p, _fp, C.c_float, _fp, C.c_size_t]
    L.mlggd_debug_stamp_select.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.mlggd_debug_stamp_read.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_int)]
    _ip = C.POINTER(C.c_int32)
    L.mlggd_load_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp, C.c_int, _ip, C.c_int]
    L.mlggd_train_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp, C.c_int, _ip, C.c_int, C.POINTER(C.c_int)]
    L.mlggd_train_frames_async.argtypes = L.mlggd_train_frames.argtypes
    L.mlggd_cv_all_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, _fp, C.c_int, _ip, C.c_int, _fp, _fp, _fp]
    L.mlggd_forward_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp, C.c_int, _ip, _fp]
    L.mlggd_alloc_pinned.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.mlggd_free_pinned.argtypes = [C.c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MlggdError("mlggd error %d: %s" % (rc, load().mlggd_last_error().decode()))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


def _p(a):
    return a.ctypes.data_as(_fp)


def _ptr_array(arrs):
    """float*[numlayers] with slot 0 unused, like BP_GPU's float** arguments."""
    pa = (_fp * (len(arrs) + 1))()
    for i, a in enumerate(arrs):
        pa[i + 1] = _p(a)
    return pa


def device_count():
    n = C.c_int(0)
    _check(load().mlggd_device_count(C.byref(n)))
    return n.value


def gamma(x):
    return float(load().mlggd_gamma(float(x)))


def comm_unique_id():
    buf = (C.c_char * UNIQUE_ID_BYTES)()
    _check(load().mlggd_comm_unique_id(buf))
    return bytes(buf)


class BPGpu:
    """Same constructor arguments and methods as the reference's BP_GPU (BP_GPU.h:48-59)."""

    def __init__(self, random_seed, gpu, layersizes, bunchsize, lrate, momentum, weightcost, weights, bias,
                 shapefactor, MLflag, dropoutflag=0, visible_omit=0.0, hid_omit=0.0, max_cache_frames=0):
        self._h = None
        self.layersizes = [int(x) for x in layersizes]
        self.numlayers = len(self.layersizes)
        if not 2 <= self.numlayers <= MAXLAYER:
            raise ValueError("numlayers must be 2..%d" % MAXLAYER)
        self.bunchsize = int(bunchsize)
        self.D = self.layersizes[-1]
        self.K0 = self.layersizes[0]
        ws = [_f32(w, (self.layersizes[l], self.layersizes[l + 1])) for l, w in enumerate(weights)]
        bs = [_f32(b, (self.layersizes[l + 1],)) for l, b in enumerate(bias)]
        if len(ws) != self.numlayers - 1 or len(bs) != self.numlayers - 1:
            raise ValueError("need numlayers-1 weight matrices and bias vectors")
        cfg = _Config()
        cfg.struct_size = C.sizeof(_Config)
        cfg.random_seed = int(random_seed)
        cfg.device = int(gpu)
        cfg.numlayers = self.numlayers
        for i, v in enumerate(self.layersizes):
            cfg.layersizes[i] = v
        cfg.bunchsize = self.bunchsize
        cfg.lrate, cfg.momentum, cfg.weightcost = lrate, momentum, weightcost
        cfg.shapefactor, cfg.MLflag = shapefactor, int(MLflag)
        cfg.dropoutflag, cfg.visible_omit, cfg.hid_omit = int(dropoutflag), visible_omit, hid_omit
        cfg.max_cache_frames = int(max_cache_frames)
        h = C.c_void_p()
        rc = load().mlggd_create(C.byref(cfg), _ptr_array(ws), _ptr_array(bs), C.byref(h))
        if rc != 0:
            msg = load().mlggd_last_error().decode()
            if h:
                load().mlggd_destroy(h)
            raise MlggdError("mlggd_create failed (%d): %s" % (rc, msg))
        self._h = h

    # -- lifetime
    def close(self):
        if self._h:
            load().mlggd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- BP_GPU::train
    def train(self, inp, targ):
        inp = _f32(inp)
        targ = _f32(targ)
        n = inp.shape[0]
        if inp.shape != (n, self.K0) or targ.shape != (n, self.D):
            raise ValueError("in must be [n][%d] and targ [n][%d]" % (self.K0, self.D))
        trained = C.c_int(0)
        _check(load().mlggd_train_chunk(self._h, n, _p(inp), _p(targ), C.byref(trained)))
        return trained.value

    def load_chunk(self, inp, targ):
        inp = _f32(inp)
        targ = _f32(targ)
        n = inp.shape[0]
        if inp.shape != (n, self.K0) or targ.shape != (n, self.D):
            raise ValueError("in must be [n][%d] and targ [n][%d]" % (self.K0, self.D))
        _check(load().mlggd_load_chunk(self._h, n, _p(inp), _p(targ)))

    def train_resident(self, first_frame, n_frames):
        trained = C.c_int(0)
        _check(load().mlggd_train_resident(self._h, int(first_frame), int(n_frames), C.byref(trained)))
        return trained.value

    # -- frame-stream chunks (input pipeline on the device)
    def _frames_args(self, feat, targ, first_frame, fea_context):
        feat = _f32(feat)
        first = np.ascontiguousarray(first_frame, dtype=np.int32)
        if feat.ndim != 2 or feat.shape[1] * fea_context != self.K0:
            raise ValueError("feat must be [n_frames][%d/fea_context]" % self.K0)
        if targ is not None:
            targ = _f32(targ, (feat.shape[0], self.D))
        return feat, targ, first

    def load_frames(self, feat, targ, first_frame, fea_context, targ_offset):
        feat, targ, first = self._frames_args(feat, targ, first_frame, fea_context)
        _check(load().mlggd_load_frames(self._h, feat.shape[0], int(fea_context), _p(feat),
                                        _p(targ) if targ is not None else None, first.size,
                                        first.ctypes.data_as(C.POINTER(C.c_int32)), int(targ_offset)))

    def train_frames(self, feat, targ, first_frame, fea_context, targ_offset, wait=True):
        """wait=False: mlggd_train_frames_async -- returns once the chunk is on the device and its steps are
        enqueued (the arrays may be reused at once); sync() waits."""
        feat, targ, first = self._frames_args(feat, targ, first_frame, fea_context)
        trained = C.c_int(0)
        fn = load().mlggd_train_frames if wait else load().mlggd_train_frames_async
        _check(fn(self._h, feat.shape[0], int(fea_context), _p(feat), _p(targ), first.size,
                  first.ctypes.data_as(C.POINTER(C.c_int32)), int(targ_offset), C.byref(trained)))
        return trained.value

    def cv_all_frames(self, feat, targ, first_frame, fea_context, targ_offset):
        feat, targ, first = self._frames_args(feat, targ, first_frame, fea_context)
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        _check(load().mlggd_cv_all_frames(self._h, feat.shape[0], int(fea_context), _p(feat), _p(targ), first.size,
                                          first.ctypes.data_as(C.POINTER(C.c_int32)), int(targ_offset), C.byref(a),
                                          C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def forward_frames(self, feat, first_frame, fea_context):
        feat, _, first = self._frames_args(feat, None, first_frame, fea_context)
        out = np.empty((first.size, self.D), np.float32)
        _check(load().mlggd_forward_frames(self._h, feat.shape[0], int(fea_context), _p(feat), first.size,
                                           first.ctypes.data_as(C.POINTER(C.c_int32)), _p(out)))
        return out

    def sync(self):
        _check(load().mlggd_sync(self._h))

    def last_train_ms(self):
        ms, steps = C.c_float(0), C.c_int(0)
        _check(load().mlggd_last_train_ms(self._h, C.byref(ms), C.byref(steps)))
        return ms.value, steps.value

    # -- BP_GPU::CrossValid / CrossValiddB / CrossValid2 / cv_bunch_single
    def _cv(self, fn, inp, targ):
        inp = _f32(inp)
        targ = _f32(targ)
        out = C.c_float(0)
        _check(fn(self._h, inp.shape[0], _p(inp), _p(targ), C.byref(out)))
        return out.value

    def CrossValid(self, inp, targ):
        return self._cv(load().mlggd_cv_sqerr, inp, targ)

    def CrossValiddB(self, inp, targ):
        return self._cv(load().mlggd_cv_abserr, inp, targ)

    def CrossValid2(self, inp, targ):
        return self._cv(load().mlggd_cv_loglik, inp, targ)

    def cv_all(self, inp, targ):
        inp = _f32(inp)
        targ = _f32(targ)
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        _check(load().mlggd_cv_all(self._h, inp.shape[0], _p(inp), _p(targ), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def forward(self, inp):
        inp = _f32(inp)
        out = np.empty((inp.shape[0], self.D), np.float32)
        _check(load().mlggd_forward(self._h, inp.shape[0], _p(inp), _p(out)))
        return out

    # -- BP_GPU::returnWeights and friends
    def returnWeights(self):
        ws = [np.empty((self.layersizes[l], self.layersizes[l + 1]), np.float32) for l in range(self.numlayers - 1)]
        bs = [np.empty(self.layersizes[l + 1], np.float32) for l in range(self.numlayers - 1)]
        _check(load().mlggd_get_weights(self._h, _ptr_array(ws), _ptr_array(bs)))
        return ws, bs

    def set_weights(self, weights, bias):
        ws = [_f32(w, (self.layersizes[l], self.layersizes[l + 1])) for l, w in enumerate(weights)]
        bs = [_f32(b, (self.layersizes[l + 1],)) for l, b in enumerate(bias)]
        _check(load().mlggd_set_weights(self._h, _ptr_array(ws), _ptr_array(bs)))

    def scalefactor(self):
        a = np.empty(self.D, np.float32)
        _check(load().mlggd_get_scalefactor(self._h, _p(a)))
        return a

    def set_scalefactor(self, alpha):
        a = _f32(alpha, (self.D,))
        _check(load().mlggd_set_scalefactor(self._h, _p(a)))

    def set_lrate(self, lrate):
        _check(load().mlggd_set_lrate(self._h, float(lrate)))

    def debug_tensor(self, name, layer=0):
        if name == "scalefactor":
            shape = (self.D,)
        elif name == "out":
            shape = (self.bunchsize, self.D)
        elif name in ("y", "dedx", "yt", "dedxt"):
            shape = (self.bunchsize, self.layersizes[layer])
        elif name in ("weights", "delta_w", "grad_w"):
            shape = (self.layersizes[layer - 1], self.layersizes[layer])
        elif name in ("bias", "delta_b"):
            shape = (self.layersizes[layer],)
        else:
            raise KeyError(name)
        a = np.empty(shape, np.float32)
        _check(load().mlggd_debug_tensor(self._h, name.encode(), int(layer), _p(a), a.size))
        return a

    # -- data parallel
    def comm_init(self, unique_id, world_size, rank):
        buf = (C.c_char * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _check(load().mlggd_comm_init(self._h, buf, int(world_size), int(rank)))

    # -- kernel-class timing
    def profile_select(self, kernel_class, layer=0, max_launches=4096, stride=1):
        kc = kernel_class.encode() if kernel_class else None
        _check(load().mlggd_profile_select(self._h, kc, int(layer), int(max_launches)))
        _check(load().mlggd_profile_stride(self._h, int(stride)))

    def profile_read(self):
        us, n = C.c_float(0), C.c_int(0)
        _check(load().mlggd_profile_read(self._h, C.byref(us), C.byref(n)))
        return us.value, n.value

    def profile_overhead(self):
        us = C.c_float(0)
        _check(load().mlggd_profile_overhead(self._h, C.byref(us)))
        return us.value

    def kernel_work(self, kernel_class, layer=0):
        f, b = C.c_double(0), C.c_double(0)
        _check(load().mlggd_kernel_work(self._h, kernel_class.encode(), int(layer), C.byref(f), C.byref(b)))
        return f.value, b.value

    def dw_launches_per_step(self):
        n = C.c_int(0)
        _check(load().mlggd_dw_launches_per_step(self._h, C.byref(n)))
        return n.value

    def dp_mode(self):
        """0 single device, 1 all-reduce of gradients, 2 all-gather of the gradient factors, 3 = 2 + sharded update,
        4 = 3 with the activations exchanged by all-to-all (each rank receives only its block's units)"""
        n = C.c_int(0)
        _check(load().mlggd_dp_mode(self._h, C.byref(n)))
        return n.value

    def out_slabs(self):
        """split-K slabs of the output-layer forward GEMM (the oracle's MFMA-order twin restates the same split)"""
        n = C.c_int(0)
        _check(load().mlggd_debug_out_slabs(self._h, C.byref(n)))
        return n.value

    def gemm_plan(self):
        """[(fwd_waves, dx_waves)] per layer 1..L-1: 4 = the 32 x 32-tile kernels (reduction over 4 waves), 1 = the
        64 x 64-tile kernels (one chain per output element)"""
        out = []
        for l in range(1, len(self.layersizes)):
            f, d = C.c_int(0), C.c_int(0)
            _check(load().mlggd_debug_gemm_plan(self._h, l, C.byref(f), C.byref(d)))
            out.append((f.value, d.value))
        return out

    def comm_info(self):
        """(ranks, rank) of the engine's RCCL communicator as RCCL reports them; (0, -1) without one."""
        n, r = C.c_int(0), C.c_int(-1)
        _check(load().mlggd_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def debug_math(self, fn, x, y=0.0):
        """fn(x, y) elementwise on the device with the kernels' own libm ("powf" "expf" "sigmoid" "div")."""
        x = _f32(x).ravel()
        out = np.empty_like(x)
        _check(load().mlggd_debug_math(self._h, fn.encode(), _p(x), float(y), _p(out), x.size))
        return out

    def plan_count(self):
        n = C.c_int(0)
        _check(load().mlggd_debug_plan_count(self._h, C.byref(n)))
        return n.value

    def set_cv_device_reduce(self, on=True):
        """CV sums formed on the device (no n x D copy back) instead of the reference-order host loop."""
        _check(load().mlggd_set_cv_device_reduce(self._h, 1 if on else 0))

    def fake_world(self, world_size, sharded=False, allreduce=False, a2a=False):
        """Emulate world_size ranks on this GPU: every step consumes world_size*bunchsize rows, rank r owns
        rows [r*bunchsize,(r+1)*bunchsize) of them (test hook, mlggd_debug_fake_world)."""
        _check(load().mlggd_debug_fake_world(self._h, int(world_size), 3 if a2a else 2 if allreduce else 1 if sharded else 0))

    def stamp_select(self, kernel_class, layer):
        _check(load().mlggd_debug_stamp_select(self._h, kernel_class.encode(), int(layer)))

    def stamp_read(self, cap_blocks=8192):
        buf = np.zeros((cap_blocks, 8), np.int64)
        n = C.c_int(0)
        _check(load().mlggd_debug_stamp_read(self._h, buf.ctypes.data_as(C.POINTER(C.c_longlong)), cap_blocks,
                                             C.byref(n)))
        return buf[:n.value]


def shard_rows(global_frames, world_size, rank):
    """Row range [lo, hi) of a global minibatch that `rank` trains (SURVEY.md 8e partition)."""
    if global_frames % world_size:
        raise ValueError("global minibatch %d not divisible by world size %d" % (global_frames, world_size))
    per = global_frames // world_size
    return rank * per, (rank + 1) * per


def weight_row_block(in_units, world_size, rank):
    """Rows [lo, hi) of a layer's weight matrix W [in][out] that `rank` owns in the sharded updates (the `shard`,
    `shard_a2a` and `allreduce` exchanges; engine.hip shard_alloc): the ceil32-padded input width is cut into 64-row tile
    rows, ceil(tile rows / world) of them per rank; hi is clipped to the true width, so late ranks may own fewer rows or
    none.  Also the block of UNITS of the layer below that rank receives in the all-to-all form."""
    kp = (int(in_units) + 31) // 32 * 32
    tile_rows = (kp + 63) // 64
    per = (tile_rows + world_size - 1) // world_size
    lo = min(rank * per * 64, int(in_units))
    hi = min((rank + 1) * per * 64, int(in_units))
    return lo, hi
