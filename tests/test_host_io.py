"""CPU: the C++ host code (CLI, norm, .wts, pfile reader, chunk planner, chunk reader,
lrand48 shuffles) against independent restatements, synthetic pfiles, and -- in the build
container only -- the reference's own sample pfiles/norm file (SURVEY.md 4 / 8c KATs)."""
import json
import os
import re
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

import finetune_recorder
import hostlib

REF = "/root/reference/tools_pfile"
have_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "train_noisy.pfile")),
                              reason="reference sample files only exist in the build container")


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    """12 sentences incl. one shorter than the context window; distinct clean/noisy streams."""
    d = tmp_path_factory.mktemp("corpus")
    rng = np.random.default_rng(5)
    lens = [40, 33, 7, 61, 3, 50, 29, 45, 38, 11, 52, 31]
    dim = 6
    noisy = rng.normal(5, 2, (sum(lens), dim)).astype(np.float32)
    clean = rng.normal(4, 2, (sum(lens), dim)).astype(np.float32)
    mean = noisy.mean(0).astype(np.float32)
    inv = (1.0 / noisy.std(0)).astype(np.float32)
    hostlib.write_pfile(str(d / "noisy.pfile"), lens, noisy)
    hostlib.write_pfile(str(d / "clean.pfile"), lens, clean)
    hostlib.write_norm(str(d / "noisy.norm"), mean, inv)
    ls = [dim * 5, 16, dim]
    rngw = np.random.default_rng(6)
    ws = [rngw.normal(0, 0.1, (ls[i], ls[i + 1])).astype(np.float32) for i in range(2)]
    bs = [rngw.normal(0, 0.1, ls[i + 1]).astype(np.float32) for i in range(2)]
    hostlib.write_wts(str(d / "init.wts"), ws, bs)
    return dict(dir=d, lens=lens, dim=dim, noisy=noisy, clean=clean, ls=ls, ws=ws, bs=bs,
                ends=list(np.cumsum(lens)))


def open_io(c, ctx=5, cache=64, seed=1234, toff=2, **extra):
    init = c["dir"] / ("init_ctx%d.wts" % ctx)
    if ctx == 5:
        init = c["dir"] / "init.wts"
    elif not init.exists():
        lsz = [c["dim"] * ctx] + c["ls"][1:]
        hostlib.write_wts(str(init), [np.zeros((lsz[i], lsz[i + 1]), np.float32) for i in range(2)],
                          [np.zeros(lsz[i + 1], np.float32) for i in range(2)])
    kv = dict(gpu_used=0, numlayers=3, layersizes=",".join(map(str, [c["dim"] * ctx] + c["ls"][1:])), bunchsize=8,
              MLflag=1, shapefactor=1.2, momentum=0.9, weightcost=1e-5, lrate=0.1, fea_dim=c["dim"], fea_context=ctx,
              traincache=cache, init_randem_seed=seed, targ_offset=toff, initwts_file=init,
              norm_file=c["dir"] / "noisy.norm", fea_file=c["dir"] / "noisy.pfile", targ_file=c["dir"] / "clean.pfile",
              outwts_file=c["dir"] / "out.wts", log_file=c["dir"] / "log.txt", train_sent_range="0-9",
              cv_sent_range="10-11", dropoutflag=0, visible_omit=0.1, hid_omit=0.1)
    kv.update(extra)
    return hostlib.HostIO(**kv)


def test_pfile_info_norm_and_weights(corpus):
    io = open_io(corpus)
    sents, frames, table, nl = io.info()
    assert (sents, frames, nl) == (12, sum(corpus["lens"]), 3)
    assert table == corpus["ends"]
    mean, inv = io.norm(corpus["dim"])
    assert np.allclose(mean, corpus["noisy"].mean(0), rtol=1e-6)
    w1, b1 = io.weights(1, 30, 16)
    assert np.array_equal(w1, corpus["ws"][0]) and np.array_equal(b1, corpus["bs"][0])
    io.close()


@pytest.mark.parametrize("ctx,cache", [(5, 64), (5, 1000), (7, 50), (3, 33), (9, 97)])
def test_chunk_planner_matches_restatement(corpus, ctx, cache):
    io = open_io(corpus, ctx=ctx, cache=cache, toff=(ctx - 1) // 2)
    starts, total = io.plan("0-9")
    want_starts, want_total = hostlib.plan_chunks(corpus["ends"], 0, 9, ctx, cache)
    assert (starts, total) == (want_starts, want_total)
    cvs, cvt = io.plan("10-11", cv=True)
    assert (cvs, cvt) == hostlib.plan_chunks(corpus["ends"], 10, 11, ctx, cache)
    # samples = frames minus (ctx-1) per usable sentence minus the frames lost at chunk cuts
    usable = sum(max(0, l - ctx + 1) for l in corpus["lens"][:10])
    assert total <= usable
    io.close()


@pytest.mark.parametrize("ctx,cache,seed", [(5, 64, 1234), (7, 50, 99), (5, 1000, 27870775)])
def test_readchunk_matches_restatement_with_lrand48_order(corpus, ctx, cache, seed):
    toff = (ctx - 1) // 2
    io = open_io(corpus, ctx=ctx, cache=cache, seed=seed, toff=toff)
    starts, total = io.plan("0-9")
    mean, inv = io.norm(corpus["dim"])
    r48 = hostlib.Rand48(seed)
    order_chunks = r48.shuffle(len(starts))          # BPtrain.cc:87 draws first
    assert io.shuffle(len(starts)) == order_chunks
    K0, D = ctx * corpus["dim"], corpus["dim"]
    for ci in order_chunks:                           # then one shuffle per Readchunk, in read order
        last = ci == len(starts) - 1
        n = total - cache * ci if last else cache
        order = r48.shuffle(n)
        inp, tg = io.read_chunk(ci, K0, D, cache)
        winp, wtg = hostlib.read_chunk(corpus["noisy"], corpus["clean"], corpus["ends"], starts, total, 9, ci, ctx,
                                       toff, cache, mean, inv, order)
        assert inp.shape[0] == n
        assert np.array_equal(inp, winp)
        assert np.array_equal(tg, wtg)
    # CV chunks are read in order, unshuffled (Interface.cc:875-878)
    cvs, cvt = io.plan("10-11", cv=True)
    inp, tg = io.read_chunk(0, K0, D, cache, cv=True)
    winp, wtg = hostlib.read_chunk(corpus["noisy"], corpus["clean"], corpus["ends"], cvs, cvt, 11, 0, ctx, toff, cache,
                                   mean, inv, list(range(cache)))
    assert np.array_equal(inp, winp) and np.array_equal(tg, wtg)
    io.close()


@pytest.mark.parametrize("ctx,cache", [(5, 64), (7, 50)])
def test_frame_stream_reader_equals_expanding_reader(corpus, ctx, cache):
    """Readchunk_frames (raw frames + first frame per row) describes exactly the matrix Readchunk
    expands on the host: row r == frames[first[r] : first[r]+ctx] flattened, target == frame
    first[r]+targ_offset -- with the same lrand48 draws."""
    toff = (ctx - 1) // 2
    dim = corpus["dim"]
    # lrand48 is process-global state: run the two readers one after the other (each open re-seeds)
    a = open_io(corpus, ctx=ctx, cache=cache, toff=toff)
    starts, total = a.plan("0-9")
    order = a.shuffle(len(starts))
    expanded = [a.read_chunk(ci, ctx * dim, dim, cache) for ci in order]
    a.plan("10-11", cv=True)
    expanded_cv = a.read_chunk(0, ctx * dim, dim, cache, cv=True)
    a.close()
    b = open_io(corpus, ctx=ctx, cache=cache, toff=toff)
    b.plan("0-9")
    assert b.shuffle(len(starts)) == order
    for ci, (inp, tg) in zip(order, expanded):
        feat, ftarg, first = b.read_chunk_frames(ci, dim, dim, 4000, cache)
        assert len(first) == len(inp)
        idx = first[:, None] + np.arange(ctx)[None, :]
        assert np.array_equal(feat[idx].reshape(len(first), ctx * dim), inp)
        assert np.array_equal(ftarg[first + toff], tg)
    b.plan("10-11", cv=True)
    feat, ftarg, first = b.read_chunk_frames(0, dim, dim, 4000, cache, cv=True)
    idx = first[:, None] + np.arange(ctx)[None, :]
    assert np.array_equal(feat[idx].reshape(len(first), ctx * dim), expanded_cv[0])
    assert np.array_equal(ftarg[first + toff], expanded_cv[1])
    b.close()


def test_cli_and_file_errors(corpus):
    with pytest.raises(hostlib.HostError, match="Format Error"):
        hostlib.HostIO.raw(["fea_dim"])
    with pytest.raises(hostlib.HostError, match="can not open feature file"):
        open_io(corpus, fea_file="/nonexistent.pfile")
    with pytest.raises(hostlib.HostError, match="feadim times context"):
        open_io(corpus, fea_dim=5)
    with pytest.raises(hostlib.HostError, match="node nums do not match"):
        open_io(corpus, layersizes="30,17,6")
    with pytest.raises(hostlib.HostError, match="please set initial weights file"):
        open_io(corpus, initwts_file="")
    io = open_io(corpus)
    with pytest.raises(hostlib.HostError, match="number error"):
        io.plan("5-40")
    with pytest.raises(hostlib.HostError, match="format error"):
        io.plan("5")
    io.close()
    # unknown keys are ignored, as the reference ignores numlayers= (Interface.cc:150-315)
    open_io(corpus, some_future_key=1).close()


def test_wts_container_kat(corpus, tmp_path):
    """MAT-v4 header bytes: 0a000000 | mrows | ncols | 0 | namelen | "weights12\\0" (SURVEY 8c-6)."""
    raw = open(corpus["dir"] / "init.wts", "rb").read()
    assert raw[:20] == struct.pack("<5i", 10, 16, 30, 0, 10) and raw[20:30] == b"weights12\0"
    # gen_rand_net writes the same container and the Gen_rand_net rule: |w| <= beta*sqrt(6)/sqrt(n_i+n_j)
    out = tmp_path / "rand.wts"
    subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "3", "30", "16", "6", str(tmp_path), str(out),
                           "1", "2", "7"], stdout=subprocess.DEVNULL)
    ws, bs = hostlib.read_wts(str(out), [30, 16, 6])
    assert abs(np.abs(ws[0]).max() - 2 * np.sqrt(6) / np.sqrt(46)) < 1e-3 and np.all(bs[0] == 0)
    assert ws[0].std() > 0.3 * np.abs(ws[0]).max()


@have_ref
def test_reference_sample_pfiles(tmp_path):
    """The reference's own 10-sentence sample (tools_pfile/): reader + planner KATs, SURVEY.md 4."""
    ls = [257 * 7, 8, 257]
    rng = np.random.default_rng(0)
    hostlib.write_wts(str(tmp_path / "i.wts"), [rng.normal(size=(ls[i], ls[i + 1])).astype(np.float32) for i in range(2)],
                      [np.zeros(ls[i + 1], np.float32) for i in range(2)])

    def io_for(ctx, cache):
        lsz = [257 * ctx, 8, 257]
        hostlib.write_wts(str(tmp_path / "i.wts"),
                          [np.zeros((lsz[i], lsz[i + 1]), np.float32) for i in range(2)],
                          [np.zeros(lsz[i + 1], np.float32) for i in range(2)])
        return hostlib.HostIO(layersizes=",".join(map(str, lsz)), bunchsize=128, fea_dim=257, fea_context=ctx,
                              traincache=cache, init_randem_seed=27870775, targ_offset=(ctx - 1) // 2,
                              initwts_file=tmp_path / "i.wts", norm_file=REF + "/train_noisy.norm",
                              fea_file=REF + "/train_noisy.pfile", targ_file=REF + "/train_clean.pfile",
                              outwts_file=tmp_path / "o.wts", log_file=tmp_path / "l.txt")

    io = io_for(7, 102400)
    sents, frames, table, _ = io.info()
    assert (sents, frames) == (10, 1885)
    assert table == [146, 289, 536, 763, 931, 1108, 1300, 1491, 1681, 1885]
    mean, inv = io.norm(257)
    assert abs(mean[0] - 14.2505) < 1e-4 and abs(inv[0] - 0.312086) < 1e-6
    assert io.plan("0-7") == ([0], 1443)          # 11 full 128-frame bunches, 35 frames dropped
    assert io.plan("8-9", cv=True) == ([1491], 382)
    # first CV sample = frames 1491..1497 of the noisy pfile, z-normalised; first raw noisy
    # feature of the file is 13.5513
    inp, tg = io.read_chunk(0, 257 * 7, 257, 102400, cv=True)
    assert inp.shape == (382, 1799)
    with open(REF + "/train_noisy.pfile", "rb") as f:
        f.seek(32768 + 8)
        first = struct.unpack(">3f", f.read(12))
        assert [round(x, 4) for x in first] == [13.5513, 15.4622, 15.6788]
        f.seek(32768 + 1491 * 259 * 4 + 8)
        row = np.frombuffer(f.read(257 * 4), ">f4").astype(np.float32)
    assert np.array_equal(inp[0, :257], ((row - mean) * inv).astype(np.float32))
    with open(REF + "/train_clean.pfile", "rb") as f:
        f.seek(32768 + 8)
        assert [round(x, 4) for x in struct.unpack(">2f", f.read(8))] == [10.8714, 9.1289]
        f.seek(32768 + (1491 + 3) * 259 * 4 + 8)
        crow = np.frombuffer(f.read(257 * 4), ">f4").astype(np.float32)
    assert np.array_equal(tg[0], ((crow - mean) * inv).astype(np.float32))  # targets use the NOISY stats
    io.close()
    io = io_for(11, 102400)
    assert io.plan("0-7")[1] == 1411 and io.plan("8-9", cv=True)[1] == 374
    io.close()
    io = io_for(7, 512)
    assert io.plan("0-7") == ([0, 530, 1066], 1431)
    io.close()


def test_rank_rows_are_the_global_minibatch_partition():
    """bptrain_main's per-rank slicing (dp_launch.h): rank r trains rows [r*B,(r+1)*B) of every COMPLETE global
    minibatch of world*B samples -- the partition the emulated-world GPU tests and SURVEY 8e use -- and the
    ranks together cover each global minibatch exactly once, in order; the ragged tail belongs to nobody."""
    for ns, B, world in ((1000, 16, 4), (1024, 128, 8), (130, 128, 2), (5, 8, 2), (96, 32, 3)):
        per_rank = [hostlib.rank_rows(ns, B, world, r) for r in range(world)]
        nglob = ns // (B * world)
        for r, rows in enumerate(per_rank):
            want = (np.arange(nglob)[:, None] * B * world + r * B + np.arange(B)[None, :]).ravel()
            assert np.array_equal(rows, want)
        if nglob:
            # rank-major concatenation of step g == rows [g*W*B, (g+1)*W*B): what the gathered factor buffers hold
            for g in range(nglob):
                cat = np.concatenate([rows[g * B:(g + 1) * B] for rows in per_rank])
                assert np.array_equal(cat, np.arange(g * B * world, (g + 1) * B * world))
    assert len(hostlib.rank_rows(100, 16, 4, 4)) == 0     # rank out of range: nothing


def test_rendezvous_never_accepts_a_stale_id_file(tmp_path):
    """finetune.pl starts one BPtrain_Sigmoid launch per epoch, so MLGGD_ID_FILE from the previous
    epoch (or a crashed launch: id, go and ack files all present) is already there when the ranks of the next
    launch start.  Ranks 1..3 start BEFORE rank 0, see the complete stale set, and must still come out with
    the NEW id; afterwards rank 0's cleanup leaves nothing behind."""
    import struct
    import threading
    import time
    path = tmp_path / "rccl.id"
    world = 4
    stale_id, new_id = bytes([7] * 128), bytes(range(128))
    # a complete, well-formed set of files from an "earlier launch" with nonce 0x1111 and M_r = 0x2220 + r
    open(path, "wb").write(b"MLGGDID2" + struct.pack("<Q", 0x1111) + stale_id)
    open(str(path) + ".go", "wb").write(struct.pack("<4Q", 0x1111, 0x2221, 0x2222, 0x2223))
    for r in range(1, world):
        open("%s.ack.%d" % (path, r), "wb").write(struct.pack("<2Q", 0x1111, 0x2220 + r))
    got, errs = {}, []

    def run(rank, ident):
        try:
            got[rank] = hostlib.rendezvous(path, world, rank, ident)
        except Exception as e:  # noqa: BLE001
            errs.append((rank, e))

    others = [threading.Thread(target=run, args=(r, None)) for r in range(1, world)]
    for t in others:
        t.start()
    time.sleep(0.3)                       # the other ranks are polling the stale files now
    assert not got                        # ... and none of them has accepted the stale id
    t0 = threading.Thread(target=run, args=(0, new_id))
    t0.start()
    for t in others + [t0]:
        t.join(30)
    assert not errs, errs
    assert all(got[r] == new_id for r in range(world))
    hostlib.lib().mlggd_host_rendezvous_cleanup(str(path).encode(), world)
    assert [f for f in os.listdir(tmp_path)] == []
    # a rank whose peers never show up gives up with a message instead of hanging
    with pytest.raises(hostlib.HostError, match="timed out"):
        hostlib.rendezvous(path, 2, 1, None, timeout=0.2)


@pytest.mark.skipif(not finetune_recorder.available(), reason="reference tree / perl not present")
def test_reference_finetune_pl_argv_through_the_parser(tmp_path):
    """the reference's epoch driver `finetune.pl` must run unmodified against the new binary
    except `$exe`.  The script itself is run here by perl (in-container only; nothing of it is committed or shipped:
    a temporary copy gets its `$exe` line pointed at a recorder, tests/finetune_recorder.py), and every argv its three
    system() call sites emit (finetune.pl:50-76 epoch 1, :89-115 epochs 2-10, :127-153 epochs 11-50) goes through
    Interface::Initial; each key must land where the reference's parser puts it (Interface.cc:150-315): strings
    verbatim, ints by atoi, floats by atof -> float; `numlayers=` is not a key of that parser (the count comes from
    layersizes=).  The recorded lists are also what tests/golden/finetune_argv.json holds (the GPU test
    tests/test_gpu_finetune.py drives BPtrain_Sigmoid with them): the fixture must equal today's recording."""
    ls = finetune_recorder.LAYERS
    hostlib.lib()

    def make_init(path):   # the init file finetune.pl:47 names is not shipped by the reference: gen_rand_net makes it
        subprocess.check_call([os.path.join(hostlib.HOST, "gen_rand_net"), "5", *map(str, ls), os.path.dirname(path), path,
                               "1", "2", "5"], stdout=subprocess.DEVNULL)

    argvs, tc = finetune_recorder.record(tmp_path, make_init)
    assert len(argvs) == 50  # one process per epoch
    assert all(len(a) == 25 and a[1] == "numlayers=5" for a in argvs)
    assert json.load(open(finetune_recorder.FIXTURE))["argv"] == argvs

    seed, lrate = 27870775, 0.1
    for epoch in (1, 2, 10, 11, 12, 50):
        argv = argvs[epoch - 1]
        kv = dict(a.split("=", 1) for a in argv)
        cwd = os.getcwd()
        os.chdir(tc)  # the script's paths are relative to its directory
        try:
            io = hostlib.HostIO.raw(argv)
        finally:
            os.chdir(cwd)
        for key in ("fea_file", "norm_file", "targ_file", "outwts_file", "log_file", "initwts_file", "train_sent_range",
                    "cv_sent_range"):
            assert io.para(key) == kv[key], key
        for key in ("fea_dim", "fea_context", "targ_offset", "dropoutflag", "MLflag", "traincache", "bunchsize", "gpu_used",
                    "init_randem_seed"):
            assert int(io.para(key)) == int(kv[key]), key
        for key in ("momentum", "shapefactor", "weightcost", "lrate", "visible_omit", "hid_omit"):
            assert np.float32(float(io.para(key))) == np.float32(float(kv[key])), key  # atof -> float
        assert io.para("layersizes") == "1799,2048,2048,2048,257" and int(io.para("numlayers")) == 5
        # keys the script never passes keep the reference's defaults (Interface.cc:140-143)
        assert [float(io.para("init_randem_%s" % k)) for k in ("weight_min", "weight_max", "bias_min", "bias_max")] == \
            [np.float32(-0.1), np.float32(0.1), np.float32(-0.1), np.float32(0.1)]
        # the schedule finetune.pl implements (:27,31,80,86,118-123): seed += 345 per epoch, lr x0.9 from epoch 11 on
        assert int(io.para("init_randem_seed")) == seed + 345 * (epoch - 1)
        want_lr = lrate * 0.9 ** max(0, epoch - 10)
        assert abs(float(io.para("lrate")) - want_lr) <= 1e-6 * want_lr
        assert kv["outwts_file"] == "./MLGGD1/mlp.%d.wts" % epoch and kv["log_file"] == "./MLGGD1/mlp.%d.log" % epoch
        assert kv["initwts_file"] == ("./pretraining_weights/Rand_1799_3hid2048_257_beta2.wts" if epoch == 1
                                      else "./MLGGD1/mlp.%d.wts" % (epoch - 1))
        # the sample pfiles the script points at were really opened: 10 sentences, 1885 frames
        sents, frames, _, nl = io.info()
        assert (sents, frames, nl) == (10, 1885, 5)
        io.close()

    # numlayers= is ignored even when it contradicts layersizes= (the reference never parses it)
    argv = [a if not a.startswith("numlayers=") else "numlayers=9" for a in argvs[0]]
    cwd = os.getcwd()
    os.chdir(tc)
    try:
        io = hostlib.HostIO.raw(argv)
    finally:
        os.chdir(cwd)
    assert int(io.para("numlayers")) == 5
    io.close()
