"""Runs the reference's epoch driver (Train_code_ML_GGD/finetune.pl) under perl against a recorder and returns the
argument lists its three system() call sites emit (finetune.pl:50-76 epoch 1, :89-115 epochs 2-10, :127-153 epochs
11-50).  In-container only: the script is read from /root/reference, a TEMPORARY copy gets its `$exe` line (the one
line a user edits) pointed at the recorder; nothing of the script is committed.  What IS committed is its
OUTPUT -- the 50 argument lists -- as tests/golden/finetune_argv.json (`python tests/finetune_recorder.py` writes it), so
that the GPU box, which has no reference tree, can drive BPtrain_Sigmoid exactly as the script does
(tests/test_gpu_finetune.py)."""
import json
import os
import re
import shutil
import subprocess
import sys

FINETUNE = "/root/reference/Train_code_ML_GGD/finetune.pl"
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "finetune_argv.json")
INIT_WTS = "Rand_1799_3hid2048_257_beta2.wts"      # finetune.pl:47 names it; the reference does not ship it
LAYERS = [1799, 2048, 2048, 2048, 257]


def available():
    return os.path.exists(FINETUNE) and shutil.which("perl") is not None


def record(workdir, make_init):
    """workdir: an empty directory; make_init(path) writes the initial weights file finetune.pl expects.
    Returns (argvs, tc) -- the recorded argument lists and the directory the script ran in."""
    top = os.path.join(str(workdir), "ref")
    tc = os.path.join(top, "Train_code_ML_GGD")
    os.makedirs(os.path.join(tc, "pretraining_weights"))
    os.symlink("/root/reference/tools_pfile", os.path.join(top, "tools_pfile"))  # $ROOT_DIR/tools_pfile: the sample data
    script = open(FINETUNE).read()
    rec, log = os.path.join(tc, "recorder.py"), os.path.join(tc, "argv.jsonl")
    with open(rec, "w") as f:
        f.write("#!%s\nimport json, os, sys\n"
                "open(%r, 'a').write(json.dumps(sys.argv[1:]) + '\\n')\n"
                "kv = dict(a.split('=', 1) for a in sys.argv[1:])\n"
                "# the next epoch starts from this epoch's output: stand in for it with the initial weights\n"
                "os.symlink(os.path.realpath(kv['initwts_file']), kv['outwts_file'])\n" % (sys.executable, log))
    os.chmod(rec, 0o755)
    patched, n = re.subn(r'(my \$exe\s*=\s*)"[^"]*";', lambda m: m.group(1) + '"./recorder.py";', script)
    assert n == 1  # the ONE line a user edits
    with open(os.path.join(tc, "finetune.pl"), "w") as f:
        f.write(patched)
    make_init(os.path.join(tc, "pretraining_weights", INIT_WTS))
    r = subprocess.run(["perl", "finetune.pl"], cwd=tc, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return [json.loads(x) for x in open(log)], tc


if __name__ == "__main__":
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        argvs, _ = record(d, lambda p: open(p, "wb").close())   # the recorder never reads the file
    with open(FIXTURE, "w") as f:
        json.dump({"source": "argument lists emitted by Train_code_ML_GGD/finetune.pl (run by perl against a recorder)",
                   "argv": argvs}, f, indent=0)
    print("wrote %s: %d epochs" % (FIXTURE, len(argvs)))
