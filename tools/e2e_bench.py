#!/usr/bin/env python3
"""End-to-end epoch throughput of the BPtrain_Sigmoid executable on a synthetic corpus:
frame-stream chunks (default) vs the reference-style host-side context expansion."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostlib
nsent, slen, dim, ctx = int(os.environ.get("NSENT", "240")), 1000, 257, 11
d = tempfile.mkdtemp(prefix="e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
rng = np.random.default_rng(1)
lens = [slen] * nsent
noisy = rng.standard_normal((nsent * slen, dim), dtype=np.float32) * 3 + 10
clean = (noisy * 0.7 + rng.standard_normal((nsent * slen, dim), dtype=np.float32)).astype(np.float32)
hostlib.write_pfile(d + "/n.pfile", lens, noisy); hostlib.write_pfile(d + "/c.pfile", lens, clean)
hostlib.write_norm(d + "/n.norm", noisy.mean(0), 1.0 / noisy.std(0))
ls = [dim * ctx, 2048, 2048, 2048, dim]
subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "5", *map(str, ls), d, d + "/init.wts", "1", "2", "5"],
                      stdout=subprocess.DEVNULL)
exe = os.path.join(hostlib.HOST, "BPtrain_Sigmoid")
ntrain = nsent - 8
samples = ntrain * (slen - ctx + 1)
for mode, env in (("frame-stream (device gather)", {}), ("frame-stream, CV sums on the device", {"MLGGD_CV_DEVICE": "1"}),
                  ("host expansion (reference style)", {"MLGGD_EXPANDED": "1"})):
    kv = dict(gpu_used=0, numlayers=5, layersizes=",".join(map(str, ls)), bunchsize=128, MLflag=1, shapefactor=1.2,
              momentum=0.9, weightcost=1e-5, lrate=0.1, fea_dim=dim, fea_context=ctx, traincache=102400,
              init_randem_seed=27870775, targ_offset=5, initwts_file=d + "/init.wts", norm_file=d + "/n.norm",
              fea_file=d + "/n.pfile", targ_file=d + "/c.pfile", outwts_file=d + "/out.wts", log_file=d + "/log.txt",
              train_sent_range="0-%d" % (ntrain - 1), cv_sent_range="%d-%d" % (ntrain, nsent - 1), dropoutflag=0,
              visible_omit=0.1, hid_omit=0.1)
    t0 = time.time()
    r = subprocess.run([exe] + ["%s=%s" % kv_ for kv_ in kv.items()], env=dict(os.environ, **env), capture_output=True, text=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    log = open(d + "/log.txt").read()
    cv = [l for l in log.splitlines() if l.startswith("CV")]
    print("%-34s %7.2f s wall for %d training samples + CV  ->  %8.0f frames/s end to end   | %s" %
          (mode, dt, samples, samples / dt, " ; ".join(cv)), flush=True)
    for l in r.stderr.splitlines():
        if l.startswith("[timing]"):
            print("      " + l, flush=True)
