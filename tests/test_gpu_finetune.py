"""GPU: BPtrain_Sigmoid driven the way the reference's epoch driver drives it, at the reference's SHIPPED
configuration (SURVEY 2 #5 "script must run unmodified against the new binary except $exe").

`perl finetune.pl` (TC/finetune.pl:8-32) trains 1799-2048-2048-2048-257 (fea_context=7, targ_offset=3) with MLflag=1
shapefactor=1, one process per epoch, in three forms: epoch 1 from the random init (:50-76), epochs 2-10 resuming from
the previous epoch's weights file with the seed advanced by 345 (:80-115), epochs 11-50 the same with the learning
rate x0.9 per epoch (:118-153).  The argument lists are the script's own output, recorded in-container
(tests/finetune_recorder.py -> tests/golden/finetune_argv.json; nothing of the script is committed), used here
VERBATIM -- relative paths included, so the test lays out the directory tree the script assumes
(`Train_code_ML_GGD/{pretraining_weights,MLGGD1}`, `tools_pfile/`), with a synthetic pfile pair + norm file where the
reference keeps its sample data (the GPU box has no reference tree) and a gen_rand_net init file where finetune.pl:47
names one the reference does not ship.

Epochs 1, 2 and 11 run back to back (epoch 11 resumes from `mlp.10.wts`, here a copy of `mlp.2.wts`: epochs 3-10 are
the epoch-2 form again).  Checked against the same three epochs driven from Python -- the host IO code for chunk / sample
order, the CPU oracle for the math (TC/BP_GPU.cu:408-423, TC/BPtrain.cc:94-139), an INDEPENDENT chain: the oracle's
epoch 2 resumes from the oracle's own epoch-1 file -- weights files to 5e-5 of max|W|, the three CV log lines to 1e-4."""
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import finetune_recorder
import hostlib

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_bptrain_sigmoid_as_finetune_pl_drives_it_epochs_1_2_11(pkg, pyoracle, tmp_path):
    argvs = json.load(open(finetune_recorder.FIXTURE))["argv"]
    assert len(argvs) == 50
    ls, dim, ctx, B = finetune_recorder.LAYERS, 257, 7, 128
    kv1 = dict(a.split("=", 1) for a in argvs[0])
    assert (kv1["MLflag"], kv1["shapefactor"], kv1["layersizes"], kv1["fea_context"], kv1["targ_offset"]) == \
        ("1", "1", "1799,2048,2048,2048,257", "7", "3")          # the shipped objective and topology

    exe = os.path.join(hostlib.HOST, "BPtrain_Sigmoid")
    subprocess.check_call(["make", "-C", hostlib.HOST, "-s"])
    tc = tmp_path / "Train_code_ML_GGD"
    (tc / "pretraining_weights").mkdir(parents=True)
    (tc / "MLGGD1").mkdir()                                        # finetune.pl:42-43
    (tc / "ORA").mkdir()                                           # the oracle chain's own weights files
    tools = tmp_path / "tools_pfile"
    tools.mkdir()
    # synthetic stand-ins for tools_pfile/train_{noisy,clean}.pfile + train_noisy.norm: 10 sentences (the script's
    # train_sent_range=0-7 / cv_sent_range=8-9), long enough for ~40 steps of 128 frames per epoch
    rng = np.random.default_rng(71)
    lens = [int(x) for x in rng.integers(600, 700, 10)]
    nfr = sum(lens)
    noisy = rng.normal(8, 3, (nfr, dim)).astype(np.float32)
    clean = (0.7 * noisy + rng.normal(0, 1.5, (nfr, dim))).astype(np.float32)
    hostlib.write_pfile(str(tools / "train_noisy.pfile"), lens, noisy)
    hostlib.write_pfile(str(tools / "train_clean.pfile"), lens, clean)
    hostlib.write_norm(str(tools / "train_noisy.norm"), noisy.mean(0), 1.0 / noisy.std(0))
    init = tc / "pretraining_weights" / finetune_recorder.INIT_WTS
    subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "5", *map(str, ls), str(init.parent), str(init),
                           "1", "2", "5"], stdout=subprocess.DEVNULL)

    cwd = os.getcwd()
    os.chdir(tc)                                                   # the script's paths are relative to its directory
    try:
        prev_ora = None
        for epoch in (1, 2, 11):
            argv = argvs[epoch - 1]
            kv = dict(a.split("=", 1) for a in argv)
            assert kv["outwts_file"] == "./MLGGD1/mlp.%d.wts" % epoch
            if epoch == 11:                                        # epochs 3..10 are not run: stand in for mlp.10.wts
                shutil.copy("./MLGGD1/mlp.2.wts", "./MLGGD1/mlp.10.wts")
                shutil.copy("./ORA/mlp.2.wts", "./ORA/mlp.10.wts")
            assert not os.path.exists(kv["outwts_file"])           # finetune.pl:49,88,126 would skip the epoch otherwise
            res = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=600)
            assert res.returncode == 0, res.stdout + res.stderr
            assert "all finish!" in res.stdout
            log = open(kv["log_file"]).read()

            # the same epoch from Python; the oracle resumes from ITS OWN previous file
            okv = dict(kv, outwts_file="./ORA/unused.wts", log_file="./ORA/unused.log")
            if epoch > 1:
                okv["initwts_file"] = kv["initwts_file"].replace("./MLGGD1/", "./ORA/")
            io = hostlib.HostIO(**okv)
            lrate = float(io.para("lrate"))                        # atof -> float, as Interface.cc parses it
            assert abs(lrate - 0.1 * 0.9 ** max(0, epoch - 10)) < 1e-7
            assert int(io.para("init_randem_seed")) == 27870775 + 345 * (epoch - 1)   # finetune.pl:31,86
            w0, b0 = hostlib.read_wts(okv["initwts_file"], ls)
            ora = pyoracle.OracleNet(ls, B, lrate, float(io.para("momentum")), float(io.para("weightcost")),
                                     float(io.para("shapefactor")), int(io.para("MLflag")), w0, b0)
            starts, total = io.plan(kv["train_sent_range"])
            order = io.shuffle(len(starts))
            steps = 0
            for n, ci in enumerate(order):
                inp, tg = io.read_chunk(ci, ls[0], dim, int(kv["traincache"]))
                assert "Starting chunk %d of %d containing %d samples." % (n + 1, len(starts), len(inp)) in log
                steps += ora.train(inp, tg)
            assert steps >= 35
            cvs, cvtotal = io.plan(kv["cv_sent_range"], cv=True)
            sq = ab = ll = np.float32(0)
            for ci in range(len(cvs)):
                inp, tg = io.read_chunk(ci, ls[0], dim, int(kv["traincache"]), cv=True)
                sq += np.float32(ora.cv_sqerr(inp, tg))
                ab += np.float32(ora.cv_abserr(inp, tg))
                ll += np.float32(ora.cv_loglik(inp, tg))
            io.close()
            wo, bo = ora.get_weights()
            hostlib.write_wts("./ORA/mlp.%d.wts" % epoch, wo, bo)
            ws, bs = hostlib.read_wts(kv["outwts_file"], ls)
            dw = max(relmax(ws[l], wo[l]) for l in range(4))
            db = max(relmax(bs[l], bo[l]) for l in range(4))
            got = [float(re.search(pat + r": (-?[\d.]+)", log).group(1)) for pat in
                   ("CV over. squared error", "CV over. square root squared error", "CV2 over. CV log likelihood")]
            want = [float(sq) / cvtotal, float(ab) / cvtotal, float(ll) / cvtotal]
            print("finetune.pl epoch %2d (lrate %.6g, %d steps): weights %.1e biases %.1e of max | CV lines %s vs oracle %s"
                  % (epoch, lrate, steps, dw, db, got, ["%.6f" % w for w in want]))
            assert dw < 5e-5 and db < 5e-5, (epoch, dw, db)
            for g_, w_ in zip(got, want):
                assert abs(g_ - w_) <= 1e-4 * abs(w_) + 1e-6, (epoch, got, want)
            if prev_ora is not None:                               # training moved the weights: not a copy of the input
                assert any(not np.array_equal(a, b) for a, b in zip(wo, prev_ora))
            prev_ora = wo
            ora.close()
    finally:
        os.chdir(cwd)
