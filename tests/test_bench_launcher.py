"""bench.py's own multi-rank launch path, on the CPU (no GPU is touched: --dry-launch stops after the gloo rendezvous).

The driver runs `python bench.py --gpus N --steps K --warmup W` for N > 1 as well as the torchrun form; both must
reach the ranks' rendezvous.  (Round 2: the first form exited with status 1 before starting anything.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def _json_line(stdout):
    """stdout carries exactly ONE line, the JSON object -- no library chatter (gloo prints to stdout from C++)"""
    lines = stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_self_launch_reaches_the_rendezvous(world):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps", "20", "--warmup", "5", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["dry_launch"] and d["world"] == world and d["backend"] == "gloo"
    assert d["ranks"] == list(range(world)) and d["local_ranks"] == list(range(world))
    assert d["distinct_pids"] == world and d["launched_by"] == "bench.py" and d["gpu_touched"] is False


def test_torchrun_form_reaches_the_rendezvous():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "20",
                        "--warmup", "5", "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["world"] == 2 and d["ranks"] == [0, 1] and d["launched_by"] == "external launcher"


def test_single_rank_dry_launch_stays_in_process():
    r = subprocess.run([sys.executable, BENCH, "--dry-launch"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr
    d = _json_line(r.stdout)
    assert d["world"] == 1 and d["launched_by"] == "direct"


def test_without_a_gpu_the_failure_is_the_missing_gpu_not_the_launcher():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the real run is covered by the gpu tests")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True,
                       text=True, timeout=300, env=_env())
    assert r.returncode == 1
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr  # both ranks were started and said so
    assert "torch.distributed.run" not in r.stderr


def test_a_rank_that_dies_takes_the_launch_down_with_its_status():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300,
                       env=_env(MLGGD_BENCH_TEST_FAIL_RANK="1", MLGGD_BENCH_GRACE_S="2"))
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert "rank 1 exited with status 7" in r.stderr


def test_watchdog_names_rank_and_phase():
    # rank 1 never arrives; its watchdog (budget 60 s x 0.02) says where it sat and exits 3, the launcher stops rank 0
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=300,
                       env=_env(MLGGD_BENCH_TEST_HANG_RANK="1", MLGGD_BENCH_WATCHDOG_SCALE="0.02", MLGGD_BENCH_GRACE_S="2"))
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "bench watchdog: rank 1 of 2 stuck in phase 'test hang'" in r.stderr


@pytest.mark.parametrize("form", ["self-launch", "torchrun"])
def test_the_rank_code_runs_end_to_end_on_a_stub_engine(form):
    """The world > 1 branches of bench.py (communicator hand-off over gloo, windows with max over ranks, ml_ggd leg, the
    three exchange arms, dp_breakdown, teardown) first run for real on the driver's multi-GPU node; here they run over
    tests/bench_stub.py so that a slip in that control flow shows on the CPU.  The numbers mean nothing."""
    cmd = [BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "3", "--stub-engine"]
    if form == "torchrun":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + cmd
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    # the driver's contract for the line: every key it reads is there, with the fixed values where they are fixed
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and d["warmup"] == 2
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert "STUB_ENGINE" in d and d["n_gpus"] == 2 and d["steps"] == 4 and d["rccl_ranks"] == 2
    assert d["rendezvous"] == "gloo" and d["config"]["dp_mode"] == "gather" and d["config"]["global_minibatch"] == 256
    assert d["timing"]["windows"] == 3 and len(d["timing"]["window_ms"]) == 3
    assert d["timing"]["window_ms_min"] <= d["ms_per_step"] * 4 <= d["timing"]["window_ms_max"]
    assert abs(d["value"] - 2 * 128 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-3 * d["value"]
    assert list(d["dp_arms"]) == ["allreduce", "shard_a2a", "gather", "shard", "gather_other_granularity",
                                  "headline_mode_without_mainline", "allreduce_unsharded_update"]  # north_star's exchange first, the launch-order switches last
    assert d["dp_arms"]["gather"]["same_as"] == "headline" and d["dp_arms"]["gather_other_granularity"]["MLGGD_DP_FINE"] == 1
    for arm in ("allreduce", "shard", "shard_a2a", "gather_other_granularity", "headline_mode_without_mainline",
                "allreduce_unsharded_update"):
        assert d["dp_arms"][arm]["value"] > 0 and "dp_breakdown" in d["dp_arms"][arm]
    assert d["dp_arms"]["headline_mode_without_mainline"]["env"] == {"MLGGD_DP_MAINLINE": "0"}
    assert d["dp_arms"]["allreduce_unsharded_update"]["env"] == {"MLGGD_DP_AR_SHARD": "0"}
    assert d["dp_breakdown"]["compute_us_by_class"]["fwd"] == 10.0 and "ml_ggd" in d and "dp_breakdown" in d["ml_ggd"]
    assert d["ml_ggd"]["stat_comm"]["env"] == {"MLGGD_DP_STAT_COMM": "1"}
    assert "incomplete" not in d and d["skipped"] == [] and d["budget_s"] == 300.0
    assert "bench.py headline after" in r.stderr     # the early copy of the headline


def test_stub_engine_arm_the_shape_rules_out_is_reported_not_fatal():
    # 2 x 96 frames: the factor exchange is unusable, the default is the all-reduce, the other two arms say why not
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "2", "--bunch", "96",
                        "--no-ml", "--stub-engine"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["config"]["dp_mode"] == "allreduce" and d["dp_arms"]["allreduce"]["same_as"] == "headline"
    assert "unavailable" in d["dp_arms"]["gather"] and "unavailable" in d["dp_arms"]["shard"] and "unavailable" in d["dp_arms"]["shard_a2a"]
    assert "unavailable" in d["dp_arms"]["gather_other_granularity"]


def test_stub_engine_two_ranks_take_the_parity_leg():
    """N > 1 `loss_vs_oracle` (BASELINE.json's "loss-vs-ref delta" in a multi-GPU line): K global minibatches on a fresh
    engine, CRCs of every rank's weights gathered over gloo, rank 0 runs the CPU oracle at bunchsize world x B on the
    rank-major rows while the other ranks wait in a barrier.  The stub's numbers mean nothing; the control flow (and the
    REAL oracle at bunchsize 256 on the rows of both ranks) is what runs here."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "2", "--no-dp-arms",
                        "--hidden", "64", "--nhid", "1", "--stub-engine"], capture_output=True, text=True, timeout=600,
                       env=_env(MLGGD_BENCH_STUB_PARITY="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    for leg in (d["loss_vs_oracle"], d["ml_ggd"]["loss_vs_oracle"]):
        assert leg["replicas_identical"] is True and leg["oracle_bunchsize"] == 256 and leg["steps"] == 3
        assert leg["dp_mode"] == "gather" and "cv_sqerr_rel" in leg and "weights_relmax" in leg
    assert "cv_loglik_rel" in d["ml_ggd"]["loss_vs_oracle"] and "alpha_relmax" in d["ml_ggd"]["loss_vs_oracle"]
    assert "cv_loglik_rel" not in d["loss_vs_oracle"]


def test_budget_skips_optional_legs_and_says_so():
    """--budget-s: the headline always runs; a leg whose estimated cost no longer fits is not started and is listed
    under `skipped` (rank 0 decides, the decision is broadcast so that no rank enters a collective alone)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "3", "--budget-s", "1",
                        "--stub-engine"], capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["value"] > 0 and d["roofline"] is not None and "incomplete" not in d
    legs = [x["leg"] for x in d["skipped"]]
    assert "ml_ggd" in legs and "dp_arms.allreduce" in legs and "dp_arms.shard" in legs and "dp_breakdown" in legs
    assert "ml_ggd" not in d and d["dp_arms"] == {"gather": {"value": d["value"], "ms_per_step": d["ms_per_step"], "same_as": "headline"}}
    assert all(x["budget_s"] == 1.0 and x["elapsed_s"] > 0 for x in d["skipped"])


def test_sigterm_after_the_headline_still_delivers_the_line():
    """The caller's time limit arrives as SIGTERM at the launcher: it is passed on to the ranks, rank 0 prints what it
    has -- the headline, marked `incomplete` -- to stdout and the launch ends with status 0.  (The ranks wait for the
    signal in a thread of their own: the main thread may sit inside a HIP or gloo call.)"""
    import signal
    import time
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "3", "--stub-engine"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=_env(MLGGD_BENCH_TEST_HANG_AFTER_HEADLINE="0"))   # rank 0 parks after its headline
    err = []
    deadline = time.time() + 240
    while time.time() < deadline:                      # wait for the early copy of the headline on stderr
        line = p.stderr.readline()
        if not line:
            break
        err.append(line)
        if "bench.py headline after" in line:
            break
    assert any("bench.py headline after" in x for x in err), "".join(err)[-2000:]
    time.sleep(0.5)
    p.send_signal(signal.SIGTERM)
    out, rest = p.communicate(timeout=120)
    assert p.returncode == 0, (p.returncode, rest[-2000:])
    d = _json_line(out)
    assert d["value"] > 0 and d["incomplete"].startswith("SIGTERM in phase 'test hang after the headline'")


def test_watchdog_after_the_headline_delivers_the_line_but_fails_the_run():
    """a rank stuck in a secondary leg must not end the launch with status 0.  Rank 0 still prints the
    headline (marked `incomplete`), every rank leaves with 4, and so does the launcher."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--windows", "3", "--stub-engine"],
                       capture_output=True, text=True, timeout=600,
                       env=_env(MLGGD_BENCH_TEST_HANG_AFTER_HEADLINE="0", MLGGD_BENCH_WATCHDOG_SCALE="0.05", MLGGD_BENCH_GRACE_S="10"))
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    d = _json_line(r.stdout)
    assert d["value"] > 0 and d["incomplete"].startswith("watchdog: stuck in phase 'test hang after the headline'")
