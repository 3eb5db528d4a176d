"""CPU: the C-ABI library loads without a GPU, exports every symbol include/mlggd.h
declares, fails loudly (no CPU fallback) and its host-only entry points work."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mlggd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mlggd_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    pkg.build()
    lib = ctypes.CDLL(pkg.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "libmlggd.so does not export %s" % s
    assert sorted(pkg.EXPORTS) == syms  # the Python binding covers the whole header


def test_no_cpu_fallback_without_gpu(pkg, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ls = [15, 8, 5]
    ws, bs = synth.make_weights(ls, seed=1)
    with pytest.raises(pkg.MlggdError, match="no HIP device|failed"):
        pkg.BPGpu(1, 0, ls, 8, 0.1, 0.9, 0.0, ws, bs, 2.0, 0)


def test_config_struct_matches_header(pkg):
    # struct_size guard: 4*(4+10+1) + 4*4 + 4*2 + 4*2 + 4 + 28 bytes
    assert ctypes.sizeof(pkg._Config) == 4 * (4 + 10 + 1 + 4 + 2 + 2 + 1 + 7)


def test_missing_library_raises(pkg, monkeypatch):
    monkeypatch.setattr(pkg, "_lib", None)
    monkeypatch.setattr(pkg, "LIB_PATH", "/nonexistent/libmlggd.so")
    with pytest.raises(pkg.MlggdError, match="missing"):
        pkg.load()


def test_host_gamma_equals_oracle(pkg, pyoracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "gamma.npz"))
    for x, want in zip(g["x"], g["gamma"]):
        assert np.float32(pkg.gamma(float(x))) == want == np.float32(pyoracle.gamma(float(x)))


def test_shard_rows(pkg):
    assert [pkg.shard_rows(1024, 8, r) for r in (0, 3, 7)] == [(0, 128), (384, 512), (896, 1024)]
    with pytest.raises(ValueError):
        pkg.shard_rows(100, 8, 0)


def test_create_rejects_bad_arguments_before_touching_a_device(pkg):
    """Argument checks of mlggd_create that need no GPU: a bunch larger than the loss kernels' LDS tile, a layer
    whose padded weight matrix a kernel could not address with 32-bit byte offsets."""
    L = pkg.load()
    fp = ctypes.POINTER(ctypes.c_float)
    arr = (fp * pkg.MAXLAYER)()  # never dereferenced: the checks come first

    def create(layersizes, bunch):
        cfg = pkg._Config()
        cfg.struct_size = ctypes.sizeof(pkg._Config)
        cfg.numlayers = len(layersizes)
        for i, v in enumerate(layersizes):
            cfg.layersizes[i] = v
        cfg.bunchsize = bunch
        h = ctypes.c_void_p()
        rc = L.mlggd_create(ctypes.byref(cfg), arr, arr, ctypes.byref(h))
        return rc, L.mlggd_last_error().decode()

    rc, msg = create([30000, 30000, 257], 128)
    assert rc != 0 and "2 GiB" in msg
    rc, msg = create([100, 50, 10], 1153)
    assert rc != 0 and "too large" in msg
    rc, msg = create([100, 0, 10], 8)
    assert rc != 0 and "layersizes[1]" in msg
