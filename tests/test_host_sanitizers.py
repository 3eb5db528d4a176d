"""CPU: the trainer's host side under sanitizers (the reference has no race detection and carries
a latent one in its condition-variable hand-off, Interface.cc:25-30).  tests/host_sanitize.cc walks an epoch -- chunk
plan, shuffles, reader thread (host/prefetch.h) ahead of a consumer that touches every byte, both chunk forms, CV
chunks, weight file -- built once with AddressSanitizer + UBSan and once with ThreadSanitizer; any report fails."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd", "host")
SRC = [os.path.join(ROOT, "tests", "host_sanitize.cc"), os.path.join(HOST, "trainer_io.cc"), os.path.join(HOST, "dp_launch.cc")]


@pytest.mark.parametrize("name,flags", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp"]),
    ("tsan", ["-fsanitize=thread"]),  # without OpenMP (libgomp is not TSan-instrumented); the reader thread is what it checks
])
def test_host_side_is_clean_under(name, flags, tmp_path):
    exe = tmp_path / ("host_sanitize_" + name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-Wno-unknown-pragmas",
                           "-I", HOST, *flags, *SRC, "-o", str(exe)])
    scratch = tmp_path / "corpus"
    scratch.mkdir()
    env = dict(os.environ, MLGGD_IO_THREADS="2", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
               TSAN_OPTIONS="halt_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe), str(scratch)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_sanitize OK" in r.stdout
    for bad in ("AddressSanitizer", "ThreadSanitizer", "LeakSanitizer", "runtime error:"):
        assert bad not in r.stderr, r.stderr[-4000:]
    assert os.path.getsize(scratch / "out.wts") > 0
