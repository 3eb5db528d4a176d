#!/usr/bin/env python3
"""bench.py -- training frames/s of the ML-GGD DNN trainer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (BP_GPU::train_bunch_single, BP_GPU.cu:308-440) over one
128-frame minibatch per GPU of synthetic 257x11 -> 2048x3 -> 257 data already resident in HBM.
Prints ONE JSON line (rank 0).  N>1 is data parallel (weak scaling: 128 frames per GPU; the exchange runs
over RCCL inside libmlggd.so -- by default an all-gather of the gradient's factors, with the update replicated
up to 5 ranks and sharded from 6; MLGGD_DP_MODE=allreduce selects the all-reduce of
the weight gradients); torch.distributed is only used for the rendezvous, the barriers and the max-over-ranks
of the wall time.

`loss_vs_oracle` (and `ml_ggd.loss_vs_oracle`) is BASELINE.json's "loss-vs-ref delta": relative difference of the CV
numbers the reference logs (BPtrain.cc:131-138) between this engine and the CPU oracle after the same steps.

The timed region contains nothing but the K steps.  The `roofline` object comes from an UNTIMED post-pass of
64 further steps in which every launch of the dominant kernel carries a HIP start/stop event pair on the engine's
stream (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps, comparable with rocprofv3's average); `ml_ggd` is BASELINE.json configs[2] (MLflag=1, beta=1.2) measured the same way in the same invocation.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "speech-enhancement-based-on-a-maximum-likelihood-criterion_amd"

MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X fp32 matrix peak, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBPS = 8000.0        # HBM3E peak (spec), same table


def flop_per_frame(ls):
    P = sum(ls[i] * ls[i + 1] for i in range(len(ls) - 1))
    return 2 * P + 2 * P + 2 * (P - ls[0] * ls[1])  # fwd + dW + dX (no dX for layer 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--loss", choices=["mmse", "ml"], default="mmse",
                    help="mmse = BASELINE.json configs[1] (MLflag=0, beta=2); ml = configs[2] (MLflag=1, beta=1.2)")
    ap.add_argument("--bunch", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--nhid", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-ml", action="store_true", help="skip the ml_ggd (configs[2]) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    pkg.load()

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B = args.bunch
    ls = synth.baseline_layersizes(hidden=args.hidden, nhid=args.nhid)
    ml, beta = (1, 1.2) if args.loss == "ml" else (0, 2.0)
    ws, bs = synth.make_weights(ls)  # same weights on every rank
    nb = min(max(args.steps, args.warmup, 1), 64)  # resident bunches, cycled
    inp, targ = synth.make_frames(nb * B, 257, 11, seed=synth.DEFAULT_SEED + 1 + rank)

    eng = pkg.BPGpu(synth.DEFAULT_SEED, local_rank, ls, B, 0.1, 0.9, 1e-5, ws, bs, beta, ml)
    if world > 1:
        uid = [pkg.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], world, rank)
    eng.load_chunk(inp, targ)

    def run_steps(k):
        done = 0
        while done < k:
            m = min(k - done, nb)
            got = eng.train_resident(0, m * B)
            assert got == m
            done += m

    # clock / power-state ramp: the GPU needs tens of milliseconds under load to reach its steady clocks, a
    # short --warmup does not get there (100 timed steps after 10 warm-up steps read 8 % low).  Untimed.
    # A fixed step count, not a time: data-parallel ranks must all run the same number of steps.
    run_steps(512)
    eng.sync()
    run_steps(args.warmup)
    eng.sync()
    def timed(engine_steps, k):
        barrier()
        t0 = time.perf_counter()
        engine_steps(k)
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    def steps_and_sync(k):
        run_steps(k)
        eng.sync()

    dt = timed(steps_and_sync, args.steps)

    roofline = None
    if not args.no_kernel_timing:
        # untimed post-pass: a HIP event pair for EVERY launch of the dominant kernel over 64 steps.  The pair is
        # handed to the launch itself (hipExtLaunchKernelGGL start / stop events on the engine's stream), so it reads
        # the dispatch's own begin / end timestamps -- the quantity rocprofv3 --kernel-trace reports -- and no
        # bracket cost has to be calibrated away
        post = 64
        eng.profile_select("dw", 0, post * (len(ls) - 1), stride=1)
        run_steps(post)
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
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            # HBM bytes per launch from rocprofv3 --pmc runs (profiles/README.md); they were collected on
            # the single-GPU plan (one launch for all layers, default shape) and only describe that one
            default_shape = (B == 128 and args.hidden == 2048 and args.nhid == 3)
            if os.path.exists(pmc) and world == 1 and nl == 1 and default_shape:
                try:
                    traffic = json.load(open(pmc)).get("k_dw", {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            # which roof binds this kernel: arithmetic intensity against the machine balance
            # 157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B (at B = 128 the kernel moves 16 B per 2*B flops -> HBM)
            gbps = by / (us * 1e-6) / 1e9
            mode = eng.dp_mode() if world > 1 else 0  # 2, 3: the fused kernel runs over the gathered minibatch
            units = max(1, ((B + 31) // 32 * 32) * (world if mode >= 2 else 1) // 64)
            n_glob = B * world
            name = "k_dwp<%d,%s,%s> (persistent dW GEMM + fused momentum/weight-decay/bias update, %s%s)" % (
                units, "true" if mode != 1 else "false", "true" if n_glob & (n_glob - 1) == 0 else "false",
                "all layers in one launch" if nl == 1 else "one launch per layer",
                {0: "", 1: "", 2: ", over the %d gathered frames of all ranks" % (B * world),
                 3: ", this rank's block of weight rows over the %d gathered frames of all ranks" % (B * world)}[mode])
            if fl / by < MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9):
                roofline = {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic}
            else:
                roofline = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic}
            roofline.update({"mean_launch_us": round(us, 2), "launches_timed": nlaunch,
                             "timing": "hipExtLaunchKernelGGL start/stop events per launch, untimed post-pass of %d steps" % post,
                             "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                             "algorithmic_TFLOPs": round(ach, 2), "algorithmic_GBps": round(gbps, 1)})

    frames = args.steps * B * world
    value = frames / dt
    fpf = flop_per_frame(ls)
    out = {
        "metric": "training frames/sec (%d-frame minibatch)" % B,
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s-%s sigmoid DNN, %s, %d-frame minibatch per GPU, synthetic pfile-shaped frames"
                   % (ls[0], "x".join(str(x) for x in ls[1:]),
                      "ML-GGD loss (MLflag=1, beta=1.2)" if ml else "MMSE loss (MLflag=0, beta=2)", B),
                   "layersizes": ls, "bunchsize_per_gpu": B, "global_minibatch": B * world,
                   "parallelism": "dp%d" % world, "flop_per_frame": fpf,
                   "dp_exchange": {0: None, 1: "all-reduce of weight gradients (RCCL)",
                                   2: "all-gather of the gradient factors Y, dEdX (RCCL); every rank forms the global gradient",
                                   3: "all-gather of the gradient factors Y, dEdX; each rank updates its block of weight rows; all-gather of W"}[eng.dp_mode()]},
        "step_roofline_frac": round(value * fpf / (world * MFMA_F32_PEAK_TFLOPS * 1e12), 4),
        "roofline": roofline,
    }

    if args.loss == "mmse" and not args.no_ml:
        # BASELINE.json configs[2] in the same invocation: ML-GGD loss (MLflag=1, beta=1.2), same data, same steps
        eng.close()
        eng = pkg.BPGpu(synth.DEFAULT_SEED, local_rank, ls, B, 0.1, 0.9, 1e-5, ws, bs, 1.2, 1)
        if world > 1:
            uid = [pkg.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            eng.comm_init(uid[0], world, rank)
        eng.load_chunk(inp, targ)
        run_steps(256)
        run_steps(args.warmup)
        eng.sync()
        dt_ml = timed(steps_and_sync, args.steps)
        out["ml_ggd"] = {"workload": "the same net and data with the ML-GGD loss (MLflag=1, beta=1.2): BASELINE.json configs[2]",
                         "value": round(frames / dt_ml, 1), "unit": "frames/s", "ms_per_step": round(dt_ml / args.steps * 1e3, 5),
                         "step_roofline_frac": round(frames / dt_ml * fpf / (world * MFMA_F32_PEAK_TFLOPS * 1e12), 4)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
                               % (n, B), "gpu_over_cpu": round(value / (n * B / cdt), 1)}

        # BASELINE.json's "loss-vs-ref delta": a fresh engine trains exactly the steps the oracle has just been timed
        # on (the oracle here is the CHECKER; nothing below is timed), then both score a held-out synthetic chunk with
        # the numbers the reference logs after an epoch (BPtrain.cc:131-138).  configs[2] (ML-GGD) gets a shorter run.
        def loss_delta(ml_, beta_, ora_, n_):
            chk = pkg.BPGpu(synth.DEFAULT_SEED, local_rank, ls, B, 0.1, 0.9, 1e-5, ws, bs, beta_, ml_)
            chk.train(inp[:B], targ[:B])
            chk.train(inp[:B], targ[:B])
            chk.load_chunk(inp, targ)
            done = 0
            while done < n_:
                m = min(n_ - done, nb)
                chk.train_resident(0, m * B)
                done += m
            cin, ctarg = synth.make_frames(1000, 257, 11, seed=77)
            sq, ab, ll = chk.cv_all(cin, ctarg)
            osq, oab = ora_.cv_sqerr(cin, ctarg), ora_.cv_abserr(cin, ctarg)
            d = {"steps": n_ + 2, "cv_sqerr_rel": abs(sq - osq) / abs(osq), "cv_abserr_rel": abs(ab - oab) / abs(oab)}
            if ml_:
                oll = ora_.cv_loglik(cin, ctarg)
                alpha, oalpha = chk.scalefactor(), ora_.tensor("scalefactor")
                d["cv_loglik_rel"] = abs(ll - oll) / abs(oll)
                d["alpha_relmax"] = float(np.abs(alpha - oalpha).max() / np.abs(oalpha).max())
            chk.close()
            return {k: (v if k == "steps" else float("%.2e" % v)) for k, v in d.items()}

        out["loss_vs_oracle"] = loss_delta(ml, beta, ora, n)
        ora.close()
        if "ml_ggd" in out:
            n_ml = min(n, 150)
            ora = pyoracle.OracleNet(ls, B, 0.1, 0.9, 1e-5, 1.2, 1, ws, bs)
            ora.train_bunch(inp[:B], targ[:B])
            ora.train_bunch(inp[:B], targ[:B])
            for i in range(n_ml):
                ora.train_bunch(inp[(i % nb) * B:(i % nb + 1) * B], targ[(i % nb) * B:(i % nb + 1) * B])
            out["ml_ggd"]["loss_vs_oracle"] = loss_delta(1, 1.2, ora, n_ml)
            ora.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
