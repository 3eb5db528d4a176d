#!/usr/bin/env python3
"""bench.py -- training frames/s of the ML-GGD DNN trainer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 needs no launcher: when WORLD_SIZE is not set, this process starts N child ranks itself -- BEFORE it imports
torch or touches a GPU -- (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), relays rank
0's JSON line and exits with the children's status.  The torchrun form works too
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...):
with WORLD_SIZE set this process is simply one rank.

A "step" is one pass of the hot path (BP_GPU::train_bunch_single, BP_GPU.cu:308-440) over one 128-frame minibatch
per GPU of synthetic 257x11 -> 2048x3 -> 257 data already resident in HBM.  Prints ONE JSON line (rank 0).
N > 1 is data parallel (weak scaling: 128 frames per GPU; the exchange runs over RCCL inside libmlggd.so -- by default
an all-gather of the gradient's factors with the update replicated up to 5 ranks and sharded from 6, DESIGN.md
section 6; `--dp-mode allreduce` / MLGGD_DP_MODE=allreduce selects BASELINE.json's all-reduce of the weight
gradients, which is also measured as `dp_arms.allreduce` in the same invocation).  torch.distributed is used over
GLOO only -- rendezvous, barriers, max over ranks; the only RCCL communicator is the engine's own.

Timing: `ms_per_step` is the MEDIAN over `--windows` (default 9) back-to-back windows of exactly --steps steps each;
a window is [barrier + device sync] t0 [K steps] [device sync] t1 on every rank, its time the MAX over ranks of
t1 - t0 (no collective and no event inside a window).  `window_ms_min/max` give the spread.

`loss_vs_oracle` (and `ml_ggd.loss_vs_oracle`, `ml_ggd_beta0.9.loss_vs_oracle`) is BASELINE.json's "loss-vs-ref delta":
relative difference of the CV numbers the reference logs (BPtrain.cc:131-138) between this engine and the CPU oracle
after the same steps.  The `roofline` object comes from an UNTIMED post-pass of 64 further steps in which every
launch of the dominant kernel carries a HIP start/stop event pair on the engine's stream (hipExtLaunchKernelGGL: the
dispatch's own begin/end timestamps, comparable with rocprofv3's average); `ml_ggd` is BASELINE.json configs[2]
(MLflag=1, beta=1.2) measured the same way in the same invocation.  `dp_breakdown` (untimed post-passes, one per
kernel class): device time of the rank's own kernels per step against the step's wall time -- the difference is the
exposed part of the exchange.

Time budget (`--budget-s`, default 300 s from process start): the headline always runs; every further leg (dp_breakdown,
N > 1 `loss_vs_oracle`, `dp_arms.*` -- all-reduce first, it is the exchange BASELINE.json names --, `ml_ggd`, the 1-rank
rehearsal, the CPU baseline and its parity legs) is started only if its estimated cost still fits, else it is listed
under `skipped` (rank 0 decides, the decision is broadcast).  The headline is copied to stderr the moment it exists;
on SIGTERM rank 0 prints what it has to stdout (marked `incomplete`) and exits 0, so a run that outlives the caller's
limit still delivers its line.  A watchdog that fires AFTER the headline also prints the partial line but exits 4.

N > 1 parity (`loss_vs_oracle`, BASELINE.json's "loss-vs-ref delta"): a fresh engine on the same communicator layout
trains K = 20 global minibatches, rank 0 scores a held-out chunk and compares with the CPU oracle run at bunchsize
world x B on the same rows (rank-major: rank r owns rows [rB, (r+1)B) of every global minibatch);
`replicas_identical` = the CRC of every rank's weights and biases agrees (gloo all-gather).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
import zlib

T0 = time.time()  # the budget of --budget-s counts from here (interpreter start-up and imports included)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix peak, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBPS = 8000.0        # HBM3E peak (spec), same table

DP_EXCHANGE = {0: None, 1: "all-reduce of weight gradients (RCCL)",
               2: "all-gather of the gradient factors Y, dEdX (RCCL); every rank forms the global gradient",
               3: "all-gather of the gradient factors Y, dEdX; each rank updates its block of weight rows; all-gather of W",
               4: "all-to-all of the activations Y (each rank receives only the units of its block of weight rows), all-gather of dEdX; "
                  "each rank updates its block; all-gather of W"}
DP_MODE_NAME = {0: None, 1: "allreduce", 2: "gather", 3: "shard", 4: "shard_a2a"}


def flop_per_frame(ls):
    P = sum(ls[i] * ls[i + 1] for i in range(len(ls) - 1))
    return 2 * P + 2 * P + 2 * (P - ls[0] * ls[1])  # fwd + dW + dX (no dX for layer 1)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--windows", type=int, default=9, help="timed windows of --steps steps each; the median is reported")
    ap.add_argument("--loss", choices=["mmse", "ml"], default="mmse",
                    help="mmse = BASELINE.json configs[1] (MLflag=0, beta=2); ml = configs[2] (MLflag=1, beta=1.2)")
    ap.add_argument("--bunch", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--nhid", type=int, default=3)
    ap.add_argument("--dp-mode", choices=["auto", "allreduce", "gather", "shard", "shard_a2a"], default="auto",
                    help="gradient exchange of the headline number for N > 1 (auto: engine default, DESIGN.md section 6)")
    ap.add_argument("--no-dp-arms", action="store_true", help="N > 1: do not also measure the other exchange modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-ml", action="store_true", help="skip the ml_ggd (configs[2]) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp", type=int, default=512,
                    help="untimed clock-ramp steps before the headline's warm-up (profiler passes over slow shapes lower it)")
    ap.add_argument("--budget-s", type=float, default=300.0,
                    help="seconds from process start within which optional legs may still be STARTED (module docstring)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, rendezvous over gloo, report rank/world and exit before touching a GPU")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launcher: seconds before the ranks are killed")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="one GPU: run the data-parallel code of this script -- RCCL communicator (of ONE rank), exchange "
                         "path of the engine, ml_ggd leg, all exchange arms, dp_breakdown -- exactly as an N-rank run would; "
                         "the line is marked `rehearsal` and its numbers are those of a 1-rank exchange, not a scaling result")
    ap.add_argument("--no-dp-rehearsal", action="store_true",
                    help="N = 1: skip the short 1-rank-communicator leg (`dp_rehearsal_1rank`)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TESTS ONLY (tests/test_bench_launcher.py): run the rank code over tests/bench_stub.py, a do-nothing "
                         "stand-in for the package, so the multi-rank control flow is exercised on a machine without GPUs; "
                         "the line it prints is marked and means nothing")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# self-launcher: N child ranks, started before anything in this process touches a GPU (it never does)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    n = args.gpus
    env0 = dict(os.environ)
    env0.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "WORLD_SIZE": str(n),
                 "LOCAL_WORLD_SIZE": str(n), "MLGGD_BENCH_LAUNCHED": "1"})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    env0.setdefault("OMP_NUM_THREADS", "4")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the JSON line (relayed below); the other ranks' stdout joins stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    deadline = time.time() + args.launch_timeout
    first_bad, grace = None, None
    term = {"at": None}

    def on_term(signum, frame):  # the caller's time limit: pass it on; rank 0 prints what it has and the ranks leave
        if term["at"] is None:
            term["at"] = time.time()
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()

    signal.signal(signal.SIGTERM, on_term)
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and first_bad is None and term["at"] is None:
            # the peers' own watchdogs get a chance to say where they were
            first_bad, grace = bad[0], time.time() + float(os.environ.get("MLGGD_BENCH_GRACE_S", "20"))
        now = time.time()
        if now > deadline and term["at"] is None:
            sys.stderr.write("bench launcher: no result after %.0f s; asking the ranks to finish (SIGTERM)\n" % args.launch_timeout)
            on_term(None, None)
        if (grace is not None and now > grace) or (term["at"] is not None and now > term["at"] + 15):
            why = ("rank %d exited with status %d" % first_bad) if first_bad else "the ranks did not leave within 15 s of SIGTERM"
            sys.stderr.write("bench launcher: %s; stopping the remaining ranks\n" % why)
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.1)
    for p in procs:
        p.wait()
    t.join(timeout=5)
    codes = [p.returncode for p in procs]
    if any(codes):
        sys.stderr.write("bench launcher: rank exit codes %s\n" % codes)
    if term["at"] is not None:  # asked to finish: the run counts as delivered if rank 0 printed its (partial) line and left with 0
        return codes[0] if codes[0] is not None else 1
    return next((c for c in codes if c), 0) if first_bad is None else first_bad[1]


# ----------------------------------------------------------------------------------------------------------------
class Watchdog:
    """A rank that sits in one phase longer than that phase's budget says where it was and ends the process (status 3)
    instead of leaving its peers -- and the driver -- blocked in a collective until an outer time limit."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.phase, self.detail, self.deadline, self.t_enter = "start", "", None, time.time()
        self.on_fire = None  # rank 0: print what has already been measured
        self.scale = float(os.environ.get("MLGGD_BENCH_WATCHDOG_SCALE", "1"))
        threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase, seconds, detail=""):
        self.phase, self.detail, self.t_enter = phase, detail, time.time()
        self.deadline = self.t_enter + seconds * self.scale if seconds else None

    def _run(self):
        while True:
            time.sleep(0.25)
            d = self.deadline
            if d is not None and time.time() > d:
                sys.stderr.write("bench watchdog: rank %d of %d stuck in phase '%s' for %.0f s (%s); giving up\n"
                                 % (self.rank, self.world, self.phase, time.time() - self.t_enter,
                                    self.detail or "no collective in flight"))
                sys.stderr.flush()
                rc = 3
                if self.on_fire is not None:
                    try:
                        rc = self.on_fire(self.phase)
                    except Exception as ex:  # noqa: BLE001 -- last words only
                        sys.stderr.write("bench watchdog: %r\n" % (ex,))
                sys.stdout.flush()
                os._exit(rc)


def claim_stdout():
    """stdout is for ONE JSON line.  Libraries chat on it (gloo prints "[Gloo] Rank 0 is connected to ..." from C++), so
    the rank keeps a private handle on the real stdout for that line and points fd 1 at stderr for everything else."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def dry_launch(rank, world, local_rank):
    """Rendezvous only: proves the launch path (children, env, gloo) without a GPU."""
    import datetime
    import torch
    import torch.distributed as dist
    real_stdout = claim_stdout()
    wd = Watchdog(rank, world)
    # test hooks for tests/test_bench_launcher.py: a rank that dies early / a rank that never arrives
    if os.environ.get("MLGGD_BENCH_TEST_FAIL_RANK") == str(rank):
        sys.stderr.write("rank %d: failing on request\n" % rank)
        return 7
    if os.environ.get("MLGGD_BENCH_TEST_HANG_RANK") == str(rank):
        wd.enter("test hang", 60, "nothing: this rank never joins the rendezvous")
        time.sleep(3600)
    wd.enter("gloo rendezvous", 120, "init_process_group(gloo)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        t = torch.tensor([float(rank)], dtype=torch.float64)
        wd.enter("gloo all_reduce", 60, "all_reduce(sum of ranks)")
        dist.all_reduce(t)
        got = [None] * world
        dist.all_gather_object(got, {"rank": rank, "local_rank": local_rank, "pid": os.getpid()})
        assert t.item() == world * (world - 1) / 2, t
        dist.barrier()
        dist.destroy_process_group()
    else:
        got = [{"rank": 0, "local_rank": local_rank, "pid": os.getpid()}]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "world": world, "backend": "gloo" if world > 1 else None,
                          "ranks": sorted(g["rank"] for g in got), "local_ranks": sorted(g["local_rank"] for g in got),
                          "distinct_pids": len({g["pid"] for g in got}),
                          "launched_by": "bench.py" if os.environ.get("MLGGD_BENCH_LAUNCHED") else
                                         ("external launcher" if world > 1 else "direct"),
                          "gpu_touched": False}), file=real_stdout, flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------
def rank_main(args):
    # SIGTERM is blocked in this thread BEFORE anything that starts threads is imported (numpy's BLAS pool, torch, gloo:
    # they inherit the mask), so the signal stays pending until term_waiter's sigwait takes it -- a process-directed
    # signal goes to ANY thread that does not block it, and the default action there would end the rank without its line
    if not args.dry_launch:
        signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
    import datetime
    import importlib

    import numpy as np

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.dry_launch:
        return dry_launch(rank, world, local_rank)
    real_stdout = claim_stdout()

    import torch
    import torch.distributed as dist

    stub = args.stub_engine
    if stub:  # tests only: no GPU, no package -- the control flow of the ranks and nothing else
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        pkg = importlib.import_module("bench_stub")
        args.no_cpu_baseline = True

        def device_sync():
            pass
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        pkg = importlib.import_module(PKG)
        device_sync = torch.cuda.synchronize
    wd = Watchdog(rank, world)
    synth = importlib.import_module(PKG + ".synth")
    pkg.load()

    last_coll = {"what": ""}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        wd.enter("gloo rendezvous", 300, "init_process_group(gloo)")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))

    def barrier(what="barrier"):
        device_sync()
        if world > 1:
            last_coll["what"] = "gloo " + what
            dist.barrier()
        device_sync()

    def max_over_ranks(vals):
        if world == 1:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64)
        last_coll["what"] = "gloo all_reduce(max)"
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    dp = world > 1 or args.rehearse_dp  # the engine joins an RCCL communicator (of `world` ranks; 1 in a rehearsal)
    B = args.bunch
    ls = synth.baseline_layersizes(hidden=args.hidden, nhid=args.nhid)
    ml, beta = (1, 1.2) if args.loss == "ml" else (0, 2.0)
    ws, bs = synth.make_weights(ls)  # same weights on every rank
    nb = min(max(args.steps, args.warmup, 1), 64)  # resident bunches, cycled
    inp, targ = synth.make_frames(nb * B, 257, 11, seed=synth.DEFAULT_SEED + 1 + rank)
    fpf = flop_per_frame(ls)
    frames = args.steps * B * world

    def make_engine(ml_, beta_, dp_mode, comm=None):
        """An engine on this rank's GPU with the chunk resident; N > 1: joined to a fresh RCCL communicator."""
        comm = dp if comm is None else comm
        if dp_mode and dp_mode != "auto":
            os.environ["MLGGD_DP_MODE"] = dp_mode
        elif "MLGGD_DP_MODE_USER" in os.environ:
            os.environ["MLGGD_DP_MODE"] = os.environ["MLGGD_DP_MODE_USER"]
        else:
            os.environ.pop("MLGGD_DP_MODE", None)
        wd.enter("engine create", 300)
        eng = pkg.BPGpu(synth.DEFAULT_SEED, local_rank, ls, B, 0.1, 0.9, 1e-5, ws, bs, beta_, ml_)
        if comm:
            wd.enter("RCCL communicator init (%s)" % (dp_mode or "auto"), 600, "gloo broadcast of the id, then ncclCommInitRank")
            uid = [pkg.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            eng.comm_init(uid[0], world, rank)
        eng.load_chunk(inp, targ)
        return eng

    def run_steps(eng, k):
        done = 0
        while done < k:
            m = min(k - done, nb)
            got = eng.train_resident(0, m * B)
            assert got == m
            done += m

    def measure(eng, label, ramp):
        """Median over args.windows windows of exactly args.steps steps (module docstring)."""
        # clock / power-state ramp: the GPU needs tens of milliseconds under load to reach its steady clocks, a short
        # --warmup does not get there (100 timed steps after 10 warm-up steps read 8 % low).  Untimed.  A fixed step
        # count, not a time: data-parallel ranks must all run the same number of steps.
        wd.enter("%s: ramp + warm-up" % label, 300, "RCCL collectives of the training steps")
        run_steps(eng, ramp)
        eng.sync()
        run_steps(eng, args.warmup)
        eng.sync()
        wins = []
        for w in range(max(1, args.windows)):
            wd.enter("%s: timed window %d" % (label, w), 180, "RCCL collectives of the training steps / gloo barrier")
            barrier("barrier before window %d" % w)
            t0 = time.perf_counter()
            run_steps(eng, args.steps)
            eng.sync()
            wins.append(time.perf_counter() - t0)
        wd.enter("%s: max over ranks" % label, 120, "gloo all_reduce(max)")
        wins = max_over_ranks(wins)
        srt = sorted(wins)
        med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        return {"dt": med, "value": frames / med, "ms_per_step": med / args.steps * 1e3,
                "window_ms": [round(x * 1e3, 4) for x in wins]}

    def class_times(eng, label, steps=32):
        """Device time of this rank's own kernels per step, by kernel class (untimed post-passes)."""
        out = {}
        for cls in ("transpose", "fwd", "loss", "dx", "dw", "update"):
            wd.enter("%s: kernel-class pass '%s'" % (label, cls), 180, "RCCL collectives of the training steps")
            eng.profile_select(cls, 0, steps * 2 * (len(ls) + 1), stride=1)
            run_steps(eng, steps)
            eng.sync()
            us, n = eng.profile_read()
            eng.profile_select(None)
            out[cls] = us * n / steps
        return out

    out = {}
    state = {"headline_done": False, "phase": "start"}
    skipped = []

    def deliver(reason):
        """rank 0: the line as far as it has got, marked incomplete (watchdog, SIGTERM)"""
        out["incomplete"] = reason
        if skipped:
            out["skipped"] = skipped
        print(json.dumps(out), file=real_stdout, flush=True)

    def last_words(phase):
        # watchdog fired.  If the headline was already measured (a secondary leg got stuck) rank 0 still delivers it,
        # with the reason -- but the process status says that something hung: 4 (3 = nothing to deliver)
        if rank == 0 and state["headline_done"]:
            deliver("watchdog: stuck in phase '%s' (%s)" % (phase, last_coll["what"]))
        return 4 if state["headline_done"] else 3

    wd.on_fire = last_words

    def term_waiter():
        # SIGTERM = the caller's time limit.  Waited for in a thread of its own (the signal is blocked in every thread,
        # so it stays pending until sigwait takes it): the main thread may sit inside a HIP / gloo call for minutes,
        # where a Python-level handler would never get to run.
        signal.sigwait({signal.SIGTERM})
        sys.stderr.write("bench.py rank %d: SIGTERM in phase '%s'\n" % (rank, wd.phase))
        if rank == 0 and state["headline_done"]:
            deliver("SIGTERM in phase '%s': the legs after it were not run" % wd.phase)
        sys.stderr.flush()
        os._exit(0 if state["headline_done"] else 143)

    threading.Thread(target=term_waiter, daemon=True).start()

    cost = {"measure": 20.0}  # seconds one engine + ramp + windows leg took (the headline sets it)

    def fits(leg, estimate_s):
        """May an optional leg still be started?  Rank 0 decides (elapsed time differs per rank), everyone follows."""
        ok = [time.time() - T0 + estimate_s <= args.budget_s]
        if world > 1:
            last_coll["what"] = "gloo broadcast (budget decision for '%s')" % leg
            dist.broadcast_object_list(ok, src=0)
        if not ok[0]:
            skipped.append({"leg": leg, "estimate_s": round(estimate_s, 1), "elapsed_s": round(time.time() - T0, 1),
                            "budget_s": args.budget_s})
        return ok[0]

    t_leg = time.time()
    eng = make_engine(ml, beta, args.dp_mode)
    mode = eng.dp_mode()
    head = measure(eng, "headline", args.ramp)
    cost["measure"] = time.time() - t_leg

    roofline = None
    if not args.no_kernel_timing:
        # untimed post-pass: a HIP event pair for EVERY launch of the dominant kernel over 64 steps.  The pair is
        # handed to the launch itself (hipExtLaunchKernelGGL start / stop events on the engine's stream), so it reads
        # the dispatch's own begin / end timestamps -- the quantity rocprofv3 --kernel-trace reports -- and no
        # bracket cost has to be calibrated away
        post = 64
        wd.enter("roofline post-pass", 180, "RCCL collectives of the training steps")
        eng.profile_select("dw", 0, post * (len(ls) - 1), stride=1)
        run_steps(eng, post)
        eng.sync()
        us_raw, nlaunch = eng.profile_read()
        eng.profile_select(None)
        us = max(us_raw, 1e-3)
        # the dominant kernel (largest share of the step in profiles/): k_dwp, the weight-gradient GEMM
        # with the fused momentum / weight-decay / bias update; mean over its launches (all layers)
        nl = eng.dw_launches_per_step()  # 1: all layers share one persistent launch (single GPU)
        fl = eng.kernel_work("dw", 0)[0] / nl
        by = eng.kernel_work("dw", 0)[1] / nl
        if nlaunch > 0 and us > 0:
            ach = fl / (us * 1e-6) / 1e12
            traffic, traffic_source = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            # HBM bytes per launch from rocprofv3 --pmc runs (profiles/README.md); they were collected on
            # the single-GPU plan (one launch for all layers, default shape) and only describe that one
            default_shape = (B == 128 and args.hidden == 2048 and args.nhid == 3)
            if os.path.exists(pmc) and world == 1 and not dp and nl == 1 and default_shape:
                try:
                    traffic = json.load(open(pmc)).get("k_dw", {}).get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_traffic.json (committed; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of this workload, not measured in this run)")
                except Exception:
                    traffic = None
            # which roof binds this kernel: arithmetic intensity against the machine balance
            # 157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B (at B = 128 the kernel moves 16 B per 2*B flops -> HBM)
            gbps = by / (us * 1e-6) / 1e9
            units = max(1, ((B + 31) // 32 * 32) * (world if mode >= 2 else 1) // 64)
            n_glob = B * world
            name = "k_dwp<%d,%s,%s> (persistent dW GEMM + fused momentum/weight-decay/bias update, %s%s)" % (
                units, "true" if mode != 1 else "false", "true" if n_glob & (n_glob - 1) == 0 else "false",
                "all layers in one launch" if nl == 1 else "%d launches per step" % nl,
                {0: "", 1: "", 2: ", over the %d gathered frames of all ranks" % (B * world),
                 3: ", this rank's block of weight rows over the %d gathered frames of all ranks" % (B * world),
                 4: ", this rank's block of weight rows over the %d frames of all ranks (activations by all-to-all)" % (B * world)}[mode])
            if fl / by < MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9):
                roofline = {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic}
            else:
                roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic}
            roofline.update({"traffic_source": traffic_source, "mean_launch_us": round(us, 2), "launches_timed": nlaunch,
                             "timing": "hipExtLaunchKernelGGL start/stop events per launch, untimed post-pass of %d steps" % post,
                             "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                             "algorithmic_TFLOPs": round(ach, 2), "algorithmic_GBps": round(gbps, 1)})

    def breakdown(eng_, step_ms, label):
        ct = class_times(eng_, label)
        comp = sum(ct.values())
        return {"step_us": round(step_ms * 1e3, 2), "compute_us": round(comp, 2),
                "compute_us_by_class": {k: round(v, 2) for k, v in ct.items()},
                "exposed_exchange_us": round(step_ms * 1e3 - comp, 2),
                "note": "compute = device time of this rank's own kernels per step (GEMM classes: the dispatches' own "
                        "begin/end timestamps; transpose / loss / update: event brackets, which add ~2 us each and, "
                        "data parallel, include the wait for the 257-float all-reduce inside the loss bracket); "
                        "exposed = step wall time - compute; rank 0's view"}

    if stub:
        out["STUB_ENGINE"] = "tests only: no GPU was used, every number in this line is meaningless"
    if args.rehearse_dp and world == 1:
        out["rehearsal"] = ("--rehearse-dp: the data-parallel path of bench.py and of the engine on ONE GPU through a 1-rank RCCL "
                            "communicator; not a scaling measurement")
    out.update({
        "metric": "training frames/sec (%d-frame minibatch)" % B,
        "value": round(head["value"], 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s-%s sigmoid DNN, %s, %d-frame minibatch per GPU, synthetic pfile-shaped frames"
                   % (ls[0], "x".join(str(x) for x in ls[1:]),
                      "ML-GGD loss (MLflag=1, beta=1.2)" if ml else "MMSE loss (MLflag=0, beta=2)", B),
                   "layersizes": ls, "bunchsize_per_gpu": B, "global_minibatch": B * world,
                   "parallelism": "dp%d" % world, "flop_per_frame": fpf,
                   "dp_exchange": DP_EXCHANGE[mode], "dp_mode": DP_MODE_NAME[mode]},
        "timing": {"windows": len(head["window_ms"]), "statistic": "median of the windows' max-over-ranks times",
                   "window_ms": head["window_ms"], "window_ms_min": min(head["window_ms"]),
                   "window_ms_max": max(head["window_ms"]),
                   "window": "[gloo barrier + device sync] t0 [--steps steps] [device sync] t1; nothing else inside"},
        "step_roofline_frac": round(head["value"] * fpf / (world * MFMA_F32_PEAK_TFLOPS * 1e12), 4),
        "roofline": roofline,
        "rccl_ranks": eng.comm_info()[0] if dp else 0,
        "rendezvous": "gloo" if world > 1 else None,
        "launched_by": "bench.py (child ranks)" if os.environ.get("MLGGD_BENCH_LAUNCHED") else
                       ("external launcher" if world > 1 else "direct"),
        "budget_s": args.budget_s,
    })
    state["headline_done"] = True
    if rank == 0:  # an early copy for whoever watches stderr: a run cut short after this point is not an empty run
        sys.stderr.write("bench.py headline after %.0f s (early copy; the line on stdout follows at the end): %s\n"
                         % (time.time() - T0, json.dumps(out)))
        sys.stderr.flush()
    if os.environ.get("MLGGD_BENCH_TEST_HANG_AFTER_HEADLINE") == str(rank):  # tests/test_bench_launcher.py
        wd.enter("test hang after the headline", 60, "nothing: test hook")
        time.sleep(3600)
    if not args.no_kernel_timing and fits("dp_breakdown", 5):
        out["dp_breakdown"] = breakdown(eng, head["ms_per_step"], "headline")
    eng.close()
    eng = None

    # ---- N-rank parity (also in the 1-rank rehearsal): K global minibatches, then the numbers the reference logs
    def dp_parity(ml_, beta_, dp_mode, label, env=None):
        kpar = min(20, nb)
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            e2 = make_engine(ml_, beta_, dp_mode)
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)
        wd.enter("%s: %d parity steps" % (label, kpar), 300, "RCCL collectives of the training steps")
        assert e2.train_resident(0, kpar * B) == kpar
        e2.sync()
        w_, b_ = e2.returnWeights()
        crc = zlib.crc32(b"".join(np.ascontiguousarray(x).tobytes() for x in list(w_) + list(b_)))
        crcs = [crc]
        if world > 1:
            wd.enter("%s: replica check" % label, 120, "gloo all_gather_object")
            crcs = [None] * world
            dist.all_gather_object(crcs, crc)
        res = {"steps": kpar, "oracle_bunchsize": world * B, "replicas_identical": len(set(crcs)) == 1,
               "dp_mode": DP_MODE_NAME[e2.dp_mode()]}
        if rank == 0:
            wd.enter("%s: oracle at bunchsize %d" % (label, world * B), 900)
            from oracle import pyoracle  # the checker; nothing here is timed
            pyoracle.lib().ora_set_num_threads(pyoracle.usable_cpus())  # the other ranks are parked in a barrier
            cin, ctarg = synth.make_frames(1000, 257, 11, seed=77)
            sq, ab, ll = e2.cv_all(cin, ctarg)
            rows = [(inp, targ)] + [synth.make_frames(nb * B, 257, 11, seed=synth.DEFAULT_SEED + 1 + r) for r in range(1, world)]
            ora = pyoracle.OracleNet(ls, world * B, 0.1, 0.9, 1e-5, beta_, ml_, ws, bs)
            for t in range(kpar):  # global minibatch t: rank r's bunch t in rows [rB, (r+1)B)
                ora.train_bunch(np.concatenate([x[t * B:(t + 1) * B] for x, _ in rows]),
                                np.concatenate([y[t * B:(t + 1) * B] for _, y in rows]))
            osq, oab = ora.cv_sqerr(cin, ctarg), ora.cv_abserr(cin, ctarg)
            res.update({"cv_sqerr_rel": abs(sq - osq) / abs(osq), "cv_abserr_rel": abs(ab - oab) / abs(oab)})
            if ml_:
                oll = ora.cv_loglik(cin, ctarg)
                alpha, oalpha = e2.scalefactor(), ora.tensor("scalefactor")
                res["cv_loglik_rel"] = abs(ll - oll) / abs(oll)
                res["alpha_relmax"] = float(np.abs(alpha - oalpha).max() / np.abs(oalpha).max())
            wo, _ = ora.get_weights()
            res["weights_relmax"] = max(float(np.abs(a_ - o_).max() / np.abs(o_).max()) for a_, o_ in zip(w_, wo))
            ora.close()
            if not ml_ and e2.dp_mode() >= 2 and hasattr(e2, "out_slabs"):
                # factor exchanges + MMSE: the weight gradient is one chain per weight over the frames of the global
                # minibatch in their order and no statistic crosses the ranks, so the N replicas must hold exactly the bits
                # of the oracle's MFMA-order twin at bunchsize world x B (tests/test_gpu_configs.py, 8 emulated ranks)
                pyoracle.set_gemm_order("hip", e2.out_slabs(), plan=e2.gemm_plan())
                try:
                    tw = pyoracle.OracleNet(ls, world * B, 0.1, 0.9, 1e-5, beta_, ml_, ws, bs)
                    for t in range(kpar):
                        tw.train_bunch(np.concatenate([x[t * B:(t + 1) * B] for x, _ in rows]),
                                       np.concatenate([y[t * B:(t + 1) * B] for _, y in rows]))
                    wt, bt = tw.get_weights()
                    tw.close()
                finally:
                    pyoracle.set_gemm_order("ref")
                res["hip_equals_mfma_order_twin_bitwise"] = all(np.array_equal(a_, t_) for a_, t_ in zip(list(w_) + list(b_), list(wt) + list(bt)))
            res = {k: (float("%.2e" % v) if isinstance(v, float) else v) for k, v in res.items()}
        e2.close()
        wd.enter("%s: barrier after the oracle" % label, 900, "gloo barrier (rank 0 is running the CPU oracle)")
        barrier("barrier after the parity leg")
        return res

    est_parity = cost["measure"] + 10 + 0.25 * world * min(20, nb)  # the oracle: ~0.2 s per 1,024-frame step
    # (the stub engine of the tests has no oracle-grade numbers to offer; MLGGD_BENCH_STUB_PARITY=1 runs the leg's control flow anyway)
    want_parity = dp and (not args.no_cpu_baseline or os.environ.get("MLGGD_BENCH_STUB_PARITY") == "1")
    if want_parity and fits("loss_vs_oracle", est_parity):
        out["loss_vs_oracle"] = dp_parity(ml, beta, args.dp_mode, "loss_vs_oracle")

    def measure_arm(label, ml_, beta_, dp_mode, env=None, with_breakdown=True):
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            e2 = make_engine(ml_, beta_, dp_mode)
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)
        m2 = measure(e2, label, 256)
        r = {"value": round(m2["value"], 1), "unit": "frames/s", "ms_per_step": round(m2["ms_per_step"], 5),
             "window_ms_min": min(m2["window_ms"]), "window_ms_max": max(m2["window_ms"]),
             "step_roofline_frac": round(m2["value"] * fpf / (world * MFMA_F32_PEAK_TFLOPS * 1e12), 4)}
        if dp:
            r["dp_exchange"] = DP_EXCHANGE[e2.dp_mode()]
        if env:
            r["env"] = dict(env)
        if with_breakdown and not args.no_kernel_timing:
            r["dp_breakdown"] = breakdown(e2, m2["ms_per_step"], label)
        e2.close()
        return r

    est_arm = 1.3 * cost["measure"] + 5
    arms = {}
    arm_specs = []
    if dp and not args.no_dp_arms:
        # The other gradient exchanges on the same ranks, same data, same windows, ordered by what they decide: first
        # BASELINE.json's own exchange (the all-reduce of the weight gradients: reduce-scatter -> update of the rank's
        # block -> all-gather of W), then the all-to-all form of the sharded update (the one exchange whose link
        # arithmetic fits >= 6x at 8 ranks), then the factor all-gather in both forms, then the launch-order switches
        # whose defaults rest on one-GPU rehearsals (MLGGD_DP_FINE, MLGGD_DP_MAINLINE, MLGGD_DP_AR_SHARD).
        fine_default = os.environ.get("MLGGD_DP_FINE")
        arm_specs = [("allreduce", "allreduce", None), ("shard_a2a", "shard_a2a", None), ("gather", "gather", None),
                     ("shard", "shard", None),
                     ("gather_other_granularity", "gather", {"MLGGD_DP_FINE": "0" if fine_default == "1" else "1"}),
                     ("headline_mode_without_mainline", DP_MODE_NAME[mode], {"MLGGD_DP_MAINLINE": "0"}),
                     ("allreduce_unsharded_update", "allreduce", {"MLGGD_DP_AR_SHARD": "0"})]

    def run_arm(name, dp_mode, env):
        if name == DP_MODE_NAME[mode]:
            arms[name] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "same_as": "headline"}
            return
        if not fits("dp_arms." + name, est_arm):
            return
        try:
            arms[name] = measure_arm("dp arm " + name, ml, beta, dp_mode, env)
        except pkg.MlggdError as ex:  # the shape rules the factor exchange out (every rank takes this branch)
            arms[name] = {"unavailable": str(ex)[:160]}
        if name == "gather_other_granularity" and "value" in arms[name]:
            arms[name]["MLGGD_DP_FINE"] = int(env["MLGGD_DP_FINE"])

    if arm_specs:
        out["dp_arms"] = arms
        run_arm(*arm_specs[0])  # north_star's exchange before anything else optional

    if args.loss == "mmse" and not args.no_ml and fits("ml_ggd", est_arm):
        # BASELINE.json configs[2] in the same invocation: ML-GGD loss (MLflag=1, beta=1.2), same data, same steps
        out["ml_ggd"] = dict({"workload": "the same net and data with the ML-GGD loss (MLflag=1, beta=1.2): BASELINE.json configs[2]"},
                             **measure_arm("ml_ggd", 1, 1.2, args.dp_mode, with_breakdown=dp))
        if want_parity and fits("ml_ggd.loss_vs_oracle", est_parity):
            out["ml_ggd"]["loss_vs_oracle"] = dp_parity(1, 1.2, args.dp_mode, "ml_ggd.loss_vs_oracle")
        if dp and fits("ml_ggd.stat_comm", est_arm):
            # the 257-float statistic on a communicator of its own (ncclCommSplit): does it stop queueing behind the
            # factor / weight collectives?  (first thing to try once two GPUs are available)
            try:
                out["ml_ggd"]["stat_comm"] = measure_arm("ml_ggd with MLGGD_DP_STAT_COMM=1", 1, 1.2, args.dp_mode,
                                                         {"MLGGD_DP_STAT_COMM": "1"}, with_breakdown=False)
            except pkg.MlggdError as ex:
                out["ml_ggd"]["stat_comm"] = {"unavailable": str(ex)[:160]}

    for spec in arm_specs[1:]:
        run_arm(*spec)

    def cpu_and_parity_legs():
        wd.enter("cpu baseline + loss_vs_oracle", 900)
        from oracle import pyoracle  # CPU oracle = the checker, timed here only as the reported CPU baseline
        ora = pyoracle.OracleNet(ls, B, 0.1, 0.9, 1e-5, beta, ml, ws, bs)
        ora.train_bunch(inp[:B], targ[:B])
        t1 = time.perf_counter()
        ora.train_bunch(inp[:B], targ[:B])
        one = time.perf_counter() - t1
        n = int(min(max(args.cpu_seconds / max(one, 1e-4), 4), 20000))  # ~cpu_seconds of work, cycling the resident minibatches
        t1 = time.perf_counter()
        for i in range(n):
            ora.train_bunch(inp[(i % nb) * B:(i % nb + 1) * B], targ[(i % nb) * B:(i % nb + 1) * B])
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(n * B / cdt, 1), "unit": "frames/s", "cores": pyoracle.num_threads(),
                               "kind": "port", "sample": "%d steps over the same %d-frame minibatches (oracle, OpenMP, threads = usable CPU share)"
                               % (n, B), "gpu_over_cpu": round(head["value"] / (n * B / cdt), 1)}
        step_s = cdt / n

        # BASELINE.json's "loss-vs-ref delta": a fresh engine trains exactly the steps the oracle has just been timed
        # on (the oracle here is the CHECKER; nothing below is timed), then both score a held-out synthetic chunk with
        # the numbers the reference logs after an epoch (BPtrain.cc:131-138).  The ML-GGD legs get a shorter run.
        cin, ctarg = synth.make_frames(1000, 257, 11, seed=77)

        def scores(o_or_eng, ml_, is_engine):
            import zlib
            if is_engine:
                sq, ab, ll = o_or_eng.cv_all(cin, ctarg)
                al = o_or_eng.scalefactor() if ml_ else None
                w, b_ = o_or_eng.returnWeights()
            else:
                sq, ab = o_or_eng.cv_sqerr(cin, ctarg), o_or_eng.cv_abserr(cin, ctarg)
                ll = o_or_eng.cv_loglik(cin, ctarg) if ml_ else 0.0
                al = o_or_eng.tensor("scalefactor").copy() if ml_ else None
                w, b_ = o_or_eng.get_weights()
            crc = 0
            for a in list(w) + list(b_):  # every bit of the trained net
                crc = zlib.crc32(np.ascontiguousarray(a, np.float32).tobytes(), crc)
            return sq, ab, ll, al, crc

        def delta(a, r, ml_):
            d = {"cv_sqerr_rel": abs(a[0] - r[0]) / abs(r[0]), "cv_abserr_rel": abs(a[1] - r[1]) / abs(r[1])}
            if ml_:
                d["cv_loglik_rel"] = abs(a[2] - r[2]) / abs(r[2])
                d["alpha_relmax"] = float(np.abs(a[3] - r[3]).max() / np.abs(r[3]).max())
            return {k: float("%.2e" % v) for k, v in d.items()}

        def engine_after(ml_, beta_, n_):
            chk = pkg.BPGpu(synth.DEFAULT_SEED, local_rank, ls, B, 0.1, 0.9, 1e-5, ws, bs, beta_, ml_)
            chk.train(inp[:B], targ[:B])
            chk.train(inp[:B], targ[:B])
            chk.load_chunk(inp, targ)
            done = 0
            while done < n_:
                m = min(n_ - done, nb)
                chk.train_resident(0, m * B)
                done += m
            r = scores(chk, ml_, True)
            s_out = (chk.out_slabs(), chk.gemm_plan()) if hasattr(chk, "out_slabs") else (1, None)
            chk.close()
            return r, s_out

        def oracle_after(ml_, beta_, n_, split=1, order="ref", s_out=(1, None), variant="strict"):
            pyoracle.set_gemm_split(split, variant)
            pyoracle.set_gemm_order(order, s_out[0], variant, plan=s_out[1])  # the engine's slabs and per-layer GEMM kernels
            try:
                o = pyoracle.OracleNet(ls, B, 0.1, 0.9, 1e-5, beta_, ml_, ws, bs, variant=variant)
                o.train_bunch(inp[:B], targ[:B])
                o.train_bunch(inp[:B], targ[:B])
                for i in range(n_):
                    o.train_bunch(inp[(i % nb) * B:(i % nb + 1) * B], targ[(i % nb) * B:(i % nb + 1) * B])
                r = scores(o, ml_, False)
                o.close()
                return r
            finally:
                pyoracle.set_gemm_split(1, variant)
                pyoracle.set_gemm_order("ref", 1, variant)

        def parity_leg(ml_, beta_, n_, ref=None):
            """HIP vs oracle after the same steps, next to the YARDSTICKS: the same oracle in other, equally valid
            arithmetic -- forward / dX reductions as 4 resp. 7 contiguous partial sums (what any split-K GEMM does; cuBLAS
            leaves the order open), the build with FMA contraction on (nvcc's default), and the MFMA-order twin (the HIP
            kernels' exact summation order, exponential and power on the CPU: the HIP path equals it in every bit)."""
            hip, s_out = engine_after(ml_, beta_, n_)
            ref = ref if ref is not None else oracle_after(ml_, beta_, n_)
            leg = {"steps": n_ + 2, "loss_vs_oracle": delta(hip, ref, ml_)}
            yard = {}
            for name, kw in (("split4", {"split": 4}), ("split7", {"split": 7}), ("fma_build", {"variant": "fma"}),
                             ("mfma_order", {"order": "hip", "s_out": s_out})):
                if not ml_ and name in ("split7", "fma_build"):
                    continue  # MMSE is smooth (every twin sits at 1e-7): two yardsticks say it, four cost 50 s of CPU
                if time.time() - T0 + (n_ + 2) * step_s * 1.2 > args.budget_s:
                    skipped.append({"leg": "yardstick %s (ml %d beta %.1f)" % (name, ml_, beta_), "elapsed_s": round(time.time() - T0, 1),
                                    "budget_s": args.budget_s})
                    continue
                tw = oracle_after(ml_, beta_, n_, **kw)
                yard[name] = delta(tw, ref, ml_)
                if name == "mfma_order":
                    # the HIP kernels' summation order, exponential and power restated on the CPU: the HIP path must
                    # equal THIS twin in every bit (tests/test_gpu_mfma_order.py) -- a CRC over all weights and biases
                    # says so here, and the distances are 0
                    leg["hip_vs_mfma_order_twin"] = delta(hip, tw, ml_)
                    leg["hip_equals_mfma_order_twin_bitwise"] = bool(hip[4] == tw[4])
            leg["oracle_twins_vs_oracle"] = yard
            if yard:
                worst = {k: max(y[k] for y in yard.values()) for k in leg["loss_vs_oracle"]}
                leg["hip_over_largest_twin"] = {k: (round(leg["loss_vs_oracle"][k] / worst[k], 2) if worst[k] > 0 else None)
                                                for k in worst}
            return leg

        ref_head = scores(ora, ml, False)
        ora.close()
        head_leg = parity_leg(ml, beta, n, ref=ref_head)
        out["loss_vs_oracle"] = dict(head_leg["loss_vs_oracle"], steps=head_leg["steps"])
        out["loss_vs_oracle_twins"] = {"oracle_twins_vs_oracle": head_leg["oracle_twins_vs_oracle"],
                                       "hip_over_largest_twin": head_leg.get("hip_over_largest_twin"),
                                       "hip_vs_mfma_order_twin": head_leg.get("hip_vs_mfma_order_twin"),
                                       "hip_equals_mfma_order_twin_bitwise": head_leg.get("hip_equals_mfma_order_twin_bitwise")}
        if "ml_ggd" in out:
            n_ml = min(n, 150)
            for key, b_, what in (("ml_ggd", 1.2, None),
                                  ("ml_ggd_beta1.0", 1.0, "the same net and data, MLflag=1, beta=1 -- the objective the reference's finetune.pl ships (TC/finetune.pl:25-26); parity only, not timed"),
                                  ("ml_ggd_beta0.9", 0.9, "the same net and data, MLflag=1, beta=0.9 (the paper's best shape factor); parity only, not timed")):
                if time.time() - T0 + 2.5 * (n_ml + 2) * step_s > args.budget_s:
                    skipped.append({"leg": key + ".loss_vs_oracle", "elapsed_s": round(time.time() - T0, 1), "budget_s": args.budget_s})
                    continue
                leg = parity_leg(1, b_, n_ml)
                tgt = out.setdefault(key, {"workload": what})
                tgt["loss_vs_oracle"] = dict(leg["loss_vs_oracle"], steps=leg["steps"])
                tgt["oracle_twins_vs_oracle"] = leg["oracle_twins_vs_oracle"]
                tgt["hip_over_largest_twin"] = leg.get("hip_over_largest_twin")
                tgt["hip_vs_mfma_order_twin"] = leg.get("hip_vs_mfma_order_twin")
                tgt["hip_equals_mfma_order_twin_bitwise"] = leg.get("hip_equals_mfma_order_twin_bitwise")
    # cpu_baseline is part of the line's contract at N = 1: it runs whatever the budget says (bounded: ~2 x --cpu-seconds);
    # its parity legs check the budget themselves
    if rank == 0 and world == 1 and not dp and not args.no_cpu_baseline:
        cpu_and_parity_legs()

    if world == 1 and not dp and not args.no_dp_rehearsal and not stub:
        # What the data-parallel exchange path costs BEFORE any link time, measured in the driver's own single-GPU run:
        # the same engine through a 1-rank RCCL communicator (which serves its collectives as device copies), per
        # exchange mode.  Not a scaling result -- the floor under every multi-GPU step.
        reh = {"what": "this workload through a 1-rank RCCL communicator: the exchange path's fixed cost without links; "
                       "single-GPU step for comparison: ms_per_step above", "ms_per_step": {}}
        try:
            for arm, env in (("allreduce", None), ("gather", None), ("shard", None), ("shard_a2a", None),
                             ("allreduce_unsharded_update", {"MLGGD_DP_AR_SHARD": "0"})):
                if not fits("dp_rehearsal_1rank." + arm, est_arm):
                    continue
                for k, v in (env or {}).items():
                    os.environ[k] = v
                try:
                    e2 = make_engine(ml, beta, "allreduce" if arm.startswith("allreduce") else arm, comm=True)
                finally:
                    for k in (env or {}):
                        os.environ.pop(k, None)
                m2 = measure(e2, "1-rank rehearsal " + arm, 256)
                reh["ms_per_step"][arm] = round(m2["ms_per_step"], 5)
                reh.setdefault("window_ms", {})[arm] = [min(m2["window_ms"]), max(m2["window_ms"])]
                e2.close()
        except Exception as ex:  # noqa: BLE001 -- an optional leg must not cost the line
            reh["error"] = str(ex)[:200]
        out["dp_rehearsal_1rank"] = reh

    out["skipped"] = skipped
    out["elapsed_s"] = round(time.time() - T0, 1)
    wd.enter("teardown", 120, "ncclCommDestroy")
    if eng is not None:
        eng.close()
    if world > 1:
        last_coll["what"] = "gloo barrier (teardown)"
        dist.barrier()
        dist.destroy_process_group()
    wd.enter("done", 0)
    if rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "MLGGD_DP_MODE" in os.environ and "MLGGD_DP_MODE_USER" not in os.environ:
        os.environ["MLGGD_DP_MODE_USER"] = os.environ["MLGGD_DP_MODE"]  # --dp-mode auto keeps a mode the caller exported
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torch import, no HIP call above this line: the children are the first processes to touch a GPU
        sys.exit(launch_ranks(args, argv))
    sys.exit(rank_main(args))


if __name__ == "__main__":
    main()
