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


def test_weight_row_blocks(pkg):
    """The block table of the sharded updates (engine.hip shard_alloc): 64-row tile rows of the
    ceil32-padded width, ceil(rows / world) per rank, clipped to the true width."""
    assert [pkg.weight_row_block(2827, 8, r) for r in (0, 1, 6, 7)] == [(0, 384), (384, 768), (2304, 2688), (2688, 2827)]  # 45 = 8 x 6 - 3
    assert [pkg.weight_row_block(2048, 8, r) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    assert [pkg.weight_row_block(96, 4, r) for r in range(4)] == [(0, 64), (64, 96), (96, 96), (96, 96)]            # late ranks own nothing
    assert pkg.weight_row_block(300, 1, 0) == (0, 300)


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


REF_TC = "/root/reference/Train_code_ML_GGD"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_TC, "BPtrain.cc")), reason="reference tree not present")
def test_reference_main_builds_and_links_against_the_bp_gpu_shim(pkg, tmp_path):
    """INTEGRATION.md section 2 as a test: the reference's own host half -- TC/BPtrain.cc (main, threadFetch) and
    TC/Interface.cc, unmodified -- compiles and links against `class BP_GPU` of host/bp_gpu.h + libmlggd.so, with the
    one change a maintainer makes: BP_GPU.h (which hard-wires /usr/local/cuda-9.0 headers, TC/BP_GPU.h:3-5) becomes a
    one-line forward to host/bp_gpu.h.  Call sites covered by the link: the constructor (BPtrain.cc:77-78), train
    (:98), returnWeights (:108), CrossValid / CrossValiddB / CrossValid2 (:124-128), the destructor (:143).
    In-container only; everything lives in tmp_path (the sources are symlinked there so that `#include "BP_GPU.h"`
    resolves to the forwarding header, not to the file next to them); nothing of the reference is committed.  The
    binary is then RUN on the reference's sample pfiles with the reference's shipped argument list (finetune.pl's
    epoch-1 form): without a GPU it must get as far as the engine's constructor and stop with the engine's own
    message; with one it trains the epoch."""
    import json
    import subprocess
    import hostlib
    pkg.build()
    hostlib.lib()                                   # builds gen_rand_net as well
    host = os.path.join(os.path.dirname(pkg.LIB_PATH), "..", "host")
    csrc = os.path.dirname(pkg.LIB_PATH)
    for f in ("BPtrain.cc", "Interface.cc", "Interface.h"):
        os.symlink(os.path.join(REF_TC, f), tmp_path / f)
    (tmp_path / "BP_GPU.h").write_text('#include "%s"\n' % os.path.abspath(os.path.join(host, "bp_gpu.h")))
    exe = tmp_path / "BPtrain_ref"
    r = subprocess.run(["g++", "-O1", "-w", "-o", str(exe), str(tmp_path / "BPtrain.cc"), str(tmp_path / "Interface.cc"),
                        os.path.join(host, "bp_gpu.cc"), "-L" + csrc, "-lmlggd", "-lpthread", "-Wl,-rpath," + csrc,
                        "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the public members of the reference class the shim must carry (TC/BP_GPU.h:48-59), train_bunch_single included
    hdr = open(os.path.join(host, "bp_gpu.h")).read()
    for member in ("train(", "train_bunch_single(", "CrossValid(", "CrossValiddB(", "CrossValid2(", "cv_bunch_single(",
                   "returnWeights(", "Gamma("):
        assert member in hdr, member
    ls = [1799, 2048, 2048, 2048, 257]
    subprocess.check_call([os.path.join(host, "gen_rand_net"), "5", *map(str, ls), str(tmp_path), str(tmp_path / "init.wts"),
                           "1", "2", "5"], stdout=subprocess.DEVNULL)
    argv = json.load(open(os.path.join(ROOT, "tests", "golden", "finetune_argv.json")))["argv"][0]
    sub = {"initwts_file": tmp_path / "init.wts", "outwts_file": tmp_path / "mlp.1.wts", "log_file": tmp_path / "mlp.1.log",
           "norm_file": "/root/reference/tools_pfile/train_noisy.norm", "fea_file": "/root/reference/tools_pfile/train_noisy.pfile",
           "targ_file": "/root/reference/tools_pfile/train_clean.pfile"}
    argv = ["%s=%s" % (a.split("=", 1)[0], sub.get(a.split("=", 1)[0], a.split("=", 1)[1])) for a in argv]
    r = subprocess.run([str(exe)] + argv, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        assert "all finish" in r.stdout and os.path.getsize(tmp_path / "mlp.1.wts") > 4 * 1799 * 2048
    else:
        assert "mlggd_device_count failed" in r.stdout + r.stderr     # reached BP_GPU::BP_GPU through the shim
        assert "Get pfile info" not in r.stdout                        # ... and stopped there (BPtrain.cc:77 comes first)
