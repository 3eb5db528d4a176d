"""CPU tests of the measurement helpers (no GPU): the PMC post-processing that feeds bench.py's
`roofline.traffic`, and the oracle's thread-count rule."""
import csv
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_pass(d, counter, rows):
    os.makedirs(os.path.join(d, "host"), exist_ok=True)
    with open(os.path.join(d, "host", "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Correlation_Id", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (kernel, value) in enumerate(rows):
            w.writerow([i, i, kernel, counter, value])


def test_pmc_traffic_applies_the_gfx950_fetch_correction(tmp_path, monkeypatch):
    pt = _load("pmc_traffic")
    fetch, write, out = str(tmp_path / "f"), str(tmp_path / "w"), str(tmp_path / "o.json")
    dw = "void k_dwp<2, true>(DwpJobs, long long*)"
    _write_pass(fetch, "FETCH_SIZE", [(dw, 100.0), (dw, 300.0), ("void k_fwd<1, 4>(FwdArgs, long long*)", 10.0),
                                      ("__amd_rocclr_copyBuffer", 999.0)])
    _write_pass(write, "WRITE_SIZE", [(dw, 50.0), (dw, 150.0), ("void k_fwd<1, 4>(FwdArgs, long long*)", 4.0)])
    monkeypatch.setattr("sys.argv", ["pmc_traffic.py", fetch, write, out])
    pt.main()
    res = json.load(open(out))
    # mean FETCH 200 KB counted at half -> 400 KB, + mean WRITE 100 KB
    assert res["k_dw"]["hbm_bytes_per_launch"] == (2 * 200 + 100) * 1024
    assert res["k_dw"]["launches"] == 2
    assert res["k_fwd_slab"]["hbm_bytes_per_launch"] == (2 * 10 + 4) * 1024
    assert not any("rocclr" in k for k in res)


def test_oracle_thread_count_follows_the_usable_cpu_share(pyoracle):
    n = pyoracle.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if "OMP_NUM_THREADS" not in os.environ:
        assert pyoracle.num_threads() == n
