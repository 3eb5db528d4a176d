"""ctypes access to the real C++ host code (tests/host_api.cc -> tests/libmlggd_host.so, built by host/Makefile) + independent NumPy
restatements of the reference's host algorithms for cross-checking (Interface.cc)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd", "host")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", HOST, "-s", "../../tests/libmlggd_host.so", "gen_rand_net"])
        L = C.CDLL(os.path.join(ROOT, "tests", "libmlggd_host.so"))
        L.mlggd_host_open.restype = C.c_void_p
        L.mlggd_host_open.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.mlggd_host_close.argtypes = [C.c_void_p]
        L.mlggd_host_last_error.restype = C.c_char_p
        L.mlggd_host_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_int), C.c_int]
        L.mlggd_host_norm.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.mlggd_host_plan.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint)]
        L.mlggd_host_read_chunk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mlggd_host_read_chunk_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                                   C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mlggd_host_shuffle.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.mlggd_host_para.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
        L.mlggd_host_weights.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mlggd_host_write_pfile.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.mlggd_host_rank_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.mlggd_host_rendezvous.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_double]
        L.mlggd_host_rendezvous_cleanup.argtypes = [C.c_char_p, C.c_int]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


class HostIO:
    """The trainer's `Interface` opened with a finetune.pl-style argument list."""

    def __init__(self, **kv):
        args = [b"BPtrain_Sigmoid"] + [("%s=%s" % (k, v)).encode() for k, v in kv.items()]
        self._open(args)
        self.kv = kv

    @classmethod
    def raw(cls, args):
        self = cls.__new__(cls)
        self._open([b"BPtrain_Sigmoid"] + [a.encode() for a in args])
        return self

    def _open(self, args):
        arr = (C.c_char_p * len(args))(*args)
        self.h = lib().mlggd_host_open(len(args), arr)
        if not self.h:
            raise HostError(lib().mlggd_host_last_error().decode())

    def close(self):
        if self.h:
            lib().mlggd_host_close(self.h)
            self.h = None

    def info(self):
        s, f = C.c_uint(0), C.c_uint(0)
        table = (C.c_int * 100000)()
        nl = lib().mlggd_host_info(self.h, C.byref(s), C.byref(f), table, 100000)
        return s.value, f.value, list(table[:s.value]), nl

    def norm(self, dim):
        m = np.zeros(dim, np.float32)
        v = np.zeros(dim, np.float32)
        lib().mlggd_host_norm(self.h, m.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_float)), dim)
        return m, v

    def plan(self, rng, cv=False):
        starts = (C.c_int * 200000)()
        tot = C.c_uint(0)
        n = lib().mlggd_host_plan(self.h, rng.encode(), int(cv), starts, 200000, C.byref(tot))
        if n < 0:
            raise HostError(lib().mlggd_host_last_error().decode())
        return list(starts[:n]), tot.value

    def read_chunk(self, index, K0, D, cap, cv=False):
        inp = np.zeros((cap, K0), np.float32)
        targ = np.zeros((cap, D), np.float32)
        n = lib().mlggd_host_read_chunk(self.h, index, int(cv), inp.ctypes.data_as(C.POINTER(C.c_float)),
                                        targ.ctypes.data_as(C.POINTER(C.c_float)))
        if n < 0:
            raise HostError(lib().mlggd_host_last_error().decode())
        return inp[:n], targ[:n]

    def read_chunk_frames(self, index, dim, D, cap_frames, cap_samples, cv=False):
        feat = np.zeros((cap_frames, dim), np.float32)
        targ = np.zeros((cap_frames, D), np.float32)
        first = np.zeros(cap_samples, np.int32)
        nfr = C.c_int(0)
        n = lib().mlggd_host_read_chunk_frames(self.h, index, int(cv), feat.ctypes.data_as(C.POINTER(C.c_float)),
                                               targ.ctypes.data_as(C.POINTER(C.c_float)),
                                               first.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nfr))
        if n < 0:
            raise HostError(lib().mlggd_host_last_error().decode())
        return feat[:nfr.value], targ[:nfr.value], first[:n]

    def para(self, key):
        """one parsed WorkPara field as text (ints %d, floats %.9g, strings as stored)"""
        buf = C.create_string_buffer(4096)
        if lib().mlggd_host_para(self.h, key.encode(), buf, len(buf)) < 0:
            raise KeyError(key)
        return buf.value.decode()

    def shuffle(self, n):
        v = (C.c_int * n)(*range(n))
        lib().mlggd_host_shuffle(self.h, v, n)
        return list(v)

    def weights(self, layer, K, N):
        w = np.zeros((K, N), np.float32)
        b = np.zeros(N, np.float32)
        assert lib().mlggd_host_weights(self.h, layer, w.ctypes.data_as(C.POINTER(C.c_float)),
                                        b.ctypes.data_as(C.POINTER(C.c_float))) == 0
        return w, b


def write_pfile(path, sent_lengths, feats):
    feats = np.ascontiguousarray(feats, np.float32)
    sl = (C.c_int * len(sent_lengths))(*sent_lengths)
    rc = lib().mlggd_host_write_pfile(path.encode(), sl, len(sent_lengths), feats.shape[1],
                                      feats.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0, lib().mlggd_host_last_error()


def write_norm(path, mean, inv_std):
    with open(path, "w") as f:
        f.write("vec %d\n" % len(mean))
        f.writelines("%.9g\n" % x for x in mean)
        f.write("vec %d\n" % len(inv_std))
        f.writelines("%.9g\n" % x for x in inv_std)


def write_wts(path, weights, bias):
    """MATLAB level-4 container as Interface::Writeweights lays it out (Interface.cc:484-516)."""
    with open(path, "wb") as f:
        for i, (w, b) in enumerate(zip(weights, bias), start=1):
            for name, mrows, ncols, data in (("weights%d%d" % (i, i + 1), w.shape[1], w.shape[0], w),
                                             ("bias%d" % (i + 1), 1, b.shape[0], b)):
                f.write(struct.pack("<5i", 10, mrows, ncols, 0, len(name) + 1))
                f.write(name.encode() + b"\0")
                f.write(np.ascontiguousarray(data, np.float32).tobytes())


def read_wts(path, layersizes):
    ws, bs = [], []
    with open(path, "rb") as f:
        for i in range(1, len(layersizes)):
            for kind in ("w", "b"):
                t, mrows, ncols, imagf, namelen = struct.unpack("<5i", f.read(20))
                name = f.read(namelen)
                data = np.frombuffer(f.read(4 * mrows * ncols), np.float32)
                if kind == "w":
                    assert (t, mrows, ncols, imagf) == (10, layersizes[i], layersizes[i - 1], 0), name
                    ws.append(data.reshape(ncols, mrows).copy())
                else:
                    assert (t, mrows, ncols) == (10, 1, layersizes[i])
                    bs.append(data.copy())
        assert f.read() == b""
    return ws, bs


# ---------------- independent restatements (NumPy / pure Python)
class Rand48:
    """srand48/lrand48 (POSIX: X' = (0x5DEECE66D X + 0xB) mod 2^48; lrand48 = X' >> 17)."""

    def __init__(self, seed):
        self.x = ((seed & 0xFFFFFFFF) << 16) | 0x330E

    def lrand48(self):
        self.x = (0x5DEECE66D * self.x + 0xB) & ((1 << 48) - 1)
        return self.x >> 17

    def read_chunk_frames(self, index, dim, D, cap_frames, cap_samples, cv=False):
        feat = np.zeros((cap_frames, dim), np.float32)
        targ = np.zeros((cap_frames, D), np.float32)
        first = np.zeros(cap_samples, np.int32)
        nfr = C.c_int(0)
        n = lib().mlggd_host_read_chunk_frames(self.h, index, int(cv), feat.ctypes.data_as(C.POINTER(C.c_float)),
                                               targ.ctypes.data_as(C.POINTER(C.c_float)),
                                               first.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nfr))
        if n < 0:
            raise HostError(lib().mlggd_host_last_error().decode())
        return feat[:nfr.value], targ[:nfr.value], first[:n]

    def para(self, key):
        """one parsed WorkPara field as text (ints %d, floats %.9g, strings as stored)"""
        buf = C.create_string_buffer(4096)
        if lib().mlggd_host_para(self.h, key.encode(), buf, len(buf)) < 0:
            raise KeyError(key)
        return buf.value.decode()

    def shuffle(self, n):  # Interface::GetRandIndex, Interface.cc:975-986
        v = list(range(n))
        for i in range(n - 1):
            idx = self.lrand48() % (n - i)
            v[idx], v[n - 1 - i] = v[n - 1 - i], v[idx]
        return v


def plan_chunks(ends, sent_st, sent_en, ctx, cache):
    """Interface::get_chunk_info restated (Interface.cc:588-650): ends[i] = end frame of sentence i."""
    frame = 0 if sent_st == 0 else ends[sent_st - 1]
    starts, count = [frame], 0
    for s in range(sent_st, sent_en + 1):
        length = ends[s] - frame
        frame = ends[s]
        count += length - (ctx - 1) if length >= ctx else 0
        while count >= cache:
            nxt = frame - (count - cache)
            starts.append(nxt)
            count = frame - nxt - ctx + 1 if frame - nxt > ctx - 1 else 0
    return starts, (len(starts) - 1) * cache + count


def read_chunk(feat, targ, ends, starts, total_samples, sent_en, index, ctx, toff, cache, mean, inv_std, order):
    """Interface::Readchunk restated (Interface.cc:719-838) on in-memory [frames][dim] arrays."""
    last = index == len(starts) - 1
    st = starts[index]
    en = ends[sent_en] if last else starts[index + 1]
    samples = total_samples - cache * index if last else cache
    f = ((feat[st:en] - mean) * inv_std).astype(np.float32)
    t = ((targ[st:en] - mean) * inv_std).astype(np.float32)
    dim = feat.shape[1]
    inp = np.zeros((samples, ctx * dim), np.float32)
    tg = np.zeros((samples, dim), np.float32)
    sent = int(np.searchsorted(np.asarray(ends), st, side="right"))
    s = 0
    pos = st
    while pos < en:
        send = min(ends[sent], en)
        for j in range(pos, send - ctx + 1):
            if s < samples:
                inp[order[s]] = f[j - st:j - st + ctx].reshape(-1)
                tg[order[s]] = t[j - st + toff]
            s += 1
        pos = send
        sent += 1
    return inp, tg


class HostNorm:
    @staticmethod
    def read(path, dim):
        """norm file as the trainer/decoder parse it: 'vec N', N means, 'vec N', N inverse std-devs."""
        lines = open(path).read().split("\n")
        mean = np.array([float(x) for x in lines[1:1 + dim]], np.float32)
        inv = np.array([float(x) for x in lines[2 + dim:2 + 2 * dim]], np.float32)
        return mean, inv


def rank_rows(n_samples, bunch, world, rank):
    """dp_launch.h rank_sample_rows: the sample rows of a chunk that `rank` trains, in training order."""
    cap = max(n_samples, 1)
    buf = (C.c_int * cap)()
    n = lib().mlggd_host_rank_rows(n_samples, bunch, world, rank, buf, cap)
    return np.array(buf[:n], np.int64)


def rendezvous(path, world, rank, ident=None, timeout=20.0):
    """dp_launch.h rendezvous: rank 0 passes the 128-byte id in, the others get it back."""
    buf = (C.c_ubyte * 128)(*(ident if ident is not None else bytes(128)))
    if lib().mlggd_host_rendezvous(str(path).encode(), world, rank, buf, float(timeout)) != 0:
        raise HostError(lib().mlggd_host_last_error().decode())
    return bytes(buf)
