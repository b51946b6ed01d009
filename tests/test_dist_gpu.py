"""Two ranks sharing ONE MI355X (gloo for the exchange, HipOps for the device work): the sharded path on
real device memory equals the unsharded index.  (RCCL itself needs one GPU per rank; the 8-GPU run is the
driver's.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd import synth
        from cbird_amd.dist import HipOps, ShardedDctHashIndex

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        imgs = torch.from_numpy(synth.make_images(n, seed=99)).to(dev)
        sh = ShardedDctHashIndex(HipOps(0), record_capacity=1 << 12)
        a, b = sh.shard_range(n, rank, world)
        h_local = sh.ops.hash_images(imgs[a:b])
        allh = sh.gather_hashes(h_local, n)
        ids = torch.arange(a + 1, b + 1, dtype=torch.int32, device=dev)
        sh.load_shard(h_local, ids)
        out = {}
        for dht in (1, 2, 7):
            i, s, c = sh.similar(allh, dht, 4)
            out[dht] = (i.cpu().numpy().copy(), s.cpu().numpy().copy(), c.cpu().numpy().copy())
        # the pipelined sweep (scan of threshold i+1 overlapping the exchange/sort/cut of threshold i) must give
        # the same answers -- including thresholds whose records overflow the tiny buffer and force a rescan
        sweep = sh.similar_sweep(allh, (1, 2, 7, 9, 3), 4)
        torch.cuda.synchronize()
        for dht in (1, 2, 7):
            i, s, c = sweep[dht]
            assert (i.cpu().numpy() == out[dht][0]).all() and (s.cpu().numpy() == out[dht][1]).all()
            assert (c.cpu().numpy() == out[dht][2]).all()
        for dht in (9, 3):
            i, s, c = sh.similar(allh, dht, 4)
            assert (sweep[dht][0] == i).all() and (sweep[dht][1] == s).all() and (sweep[dht][2] == c).all()
        q_out.put((rank, allh.cpu().numpy().copy(), out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_index(gpu, orc):
    from cbird_amd import synth

    n = 601
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    imgs = synth.make_images(n, seed=99)
    h = orc.dcthash64_batch(imgs)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    for rank, allh, out in res:
        assert (allh.view(np.uint64) == h).all()
        for dht in (1, 2, 7):
            wi, ws, wc = orc.find64_batch(h, ids, h, dht, 4)
            gi, gs, gc = out[dht]
            assert (gc == wc.astype(np.int32)).all() and (gi.view(np.uint32) == wi).all() and (gs == ws).all()


def _video_worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd import synth_video
        from cbird_amd.dist import NeedleParallel
        from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

        clips = synth_video.make_clips(200, 150, seed=21, subclip_frac=0.1, max_gap=8)

        class M:
            pass

        media = []
        for i, (f, h) in enumerate(clips):
            m = M()
            m.id, m.path, m.videoIndex = i + 1, f"c{i}", VideoIndex(f.tolist(), [int(x) for x in h])
            media.append(m)
        idx = DctVideoIndex()
        idx.add(media)
        p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=10, minFramesNear=30)
        key = lambda r: [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
        res = NeedleParallel().run(media, lambda chunk: [key(r) for r in idx.find_videos_batch(chunk, p)])
        q_out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_video_needle_parallel_two_ranks(gpu):
    """BASELINE configs[4] shape (video search spread over ranks), 2 ranks on one GPU: needle-parallel result ==
    single-process result"""
    from cbird_amd import synth_video
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_video_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=300) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    clips = synth_video.make_clips(200, 150, seed=21, subclip_frac=0.1, max_gap=8)

    class M:
        pass

    media = []
    for i, (f, h) in enumerate(clips):
        m = M()
        m.id, m.path, m.videoIndex = i + 1, f"c{i}", VideoIndex(f.tolist(), [int(x) for x in h])
        media.append(m)
    idx = DctVideoIndex()
    idx.add(media)
    p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=10, minFramesNear=30)
    want = [[(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
            for r in idx.find_videos_batch(media, p)]
    assert sum(len(r) for r in want) > 5
    for rank, res in got:
        assert res == want


def test_sweep_single_rank_equals_per_threshold(gpu, orc):
    """similar_sweep on one rank (no process group): pipelined result == oracle for every threshold, with a record
    buffer small enough that some thresholds overflow and are redone"""
    from cbird_amd import synth
    from cbird_amd.dist import HipOps, ShardedDctHashIndex

    n = 5000
    h, ids = synth.make_hashes(n, seed=3, planted_frac=0.4)
    dev = torch.device("cuda", 0)
    sh = ShardedDctHashIndex(HipOps(0), record_capacity=1 << 13)
    dh = torch.from_numpy(h.view(np.int64)).to(dev)
    sh.load_shard(dh, torch.from_numpy(ids.view(np.int32)).to(dev))
    ths = (1, 4, 5, 8, 12, 2)
    res = sh.similar_sweep(dh, ths, 6)
    torch.cuda.synchronize()
    for t in ths:
        wi, ws, wc = orc.find64_batch(h, ids, h, t, 6)
        gi, gs, gc = res[t]
        assert (gc.cpu().numpy() == wc.astype(np.int32)).all(), t
        assert (gi.cpu().numpy().view(np.uint32) == wi).all() and (gs.cpu().numpy() == ws).all(), t


def _rccl_worker(port, n, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CBH_DIST_FORCE_COLLECTIVES="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from cbird_amd import synth
        from cbird_amd.dist import HipOps, ShardedDctHashIndex

        imgs = torch.from_numpy(synth.make_images(n, seed=5)).to(dev)
        sh = ShardedDctHashIndex(HipOps(0), record_capacity=1 << 12)
        assert sh.collective and dist.get_backend() == "nccl"
        h_local = sh.ops.hash_images(imgs)
        allh = sh.gather_hashes(h_local, n)  # all_gather_into_tensor over RCCL
        sh.load_shard(h_local, torch.arange(1, n + 1, dtype=torch.int32, device=dev))
        work = sh.ops.work_stream()
        with sh.ops.stream_ctx(work):
            sweep = sh.similar_sweep(allh, (1, 2, 7, 9), 4)  # one all-gather of {count, records} per threshold
        torch.cuda.synchronize()
        out = {d: tuple(t.cpu().numpy().copy() for t in sweep[d]) for d in (1, 2, 7, 9)}
        # the other two sharded indexes: their one all-gather each over RCCL (device tensors), == the plain index
        from cbird_amd import synth_video
        from cbird_amd.cvfeatures import CvFeaturesIndex
        from cbird_amd.dist import ShardedCvFeaturesIndex, ShardedDctVideoIndex
        from cbird_amd.index import SearchParams
        from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

        class M:
            pass

        media = []
        for i, (f, hh) in enumerate(synth_video.make_clips(60, 120, seed=3, subclip_frac=0.2, max_gap=8)):
            m = M()
            m.id, m.path, m.videoIndex = i + 1, f"c{i}", VideoIndex(f.tolist(), [int(x) for x in hh])
            media.append(m)
        vp = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=10, minFramesNear=30)
        sv = ShardedDctVideoIndex(lambda: DctVideoIndex(0), device=dev)
        sv.add(media)
        plain = DctVideoIndex(0)
        plain.add(media)
        key = lambda r: [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
        assert [key(r) for r in sv.find_videos_batch(media[:20], vp)] == [key(r) for r in plain.find_videos_batch(media[:20], vp)]
        rng = np.random.default_rng(4)
        om = []
        for i in range(40):
            m = M()
            m.id, m.path = i + 1, f"o{i}"
            m.keyPointDescriptors = rng.integers(0, 256, (50, 32), dtype=np.uint8)
            om.append(m)
        so = ShardedCvFeaturesIndex(lambda: CvFeaturesIndex(0), device=dev)
        so.add(om)
        po = CvFeaturesIndex(0)
        po.add(om)
        sp = SearchParams(cvThresh=120)
        got = so.find_batch(om[:8], sp)
        want = [po.find(m, sp) for m in om[:8]]
        assert [[(x.mediaId, x.score) for x in r] for r in got] == [[(x.mediaId, x.score) for x in r] for r in want]
        q_out.put((allh.cpu().numpy().copy(), out))
    finally:
        dist.destroy_process_group()


def test_rccl_transport_world1_equals_oracle(gpu, orc):
    """The RCCL backend itself ("nccl"), forced through the collectives at world size 1: the hash all-gather and the
    per-threshold block all-gather of the sharded sweep run on the device, on side streams, exactly as at R > 1 (one
    GPU per box: R > 1 over RCCL is the driver's run)."""
    from cbird_amd import synth

    n = 777
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_rccl_worker, args=(port, n, q))
    p.start()
    allh, out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    h = orc.dcthash64_batch(synth.make_images(n, seed=5))
    ids = np.arange(1, n + 1, dtype=np.uint32)
    assert (allh.view(np.uint64) == h).all()
    for dht in (1, 2, 7, 9):
        wi, ws, wc = orc.find64_batch(h, ids, h, dht, 4)
        gi, gs, gc = out[dht]
        assert (gc == wc.astype(np.int32)).all() and (gi.view(np.uint32) == wi).all() and (gs == ws).all()


def test_pipelined_sweep_inside_the_bench_step_is_deterministic(gpu):
    """Regression (round 2): hash -> reload -> pipelined threshold sweep, repeated as bench.py's step does at a SMALL
    index, where the scans are so short that the scan of threshold i+1 (needle-tile scratch, main stream) always
    overlaps the cut of threshold i (counting-select scratch, side stream).  With both scratches in the device's
    default stream-ordered pool the match counts came out wrong in about one bench run in three; every stream now has
    its own pool (cbh_internal.h: malloc_async).  The counts of every step must equal the un-pipelined ones."""
    import bench
    from cbird_amd.dist import HipOps, ShardedDctHashIndex

    dev = torch.device("cuda", 0)
    n = 40000
    ops = HipOps(0)
    sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)
    imgs = bench.gen_images(torch, dev, 0, n, n, 1234)
    ids = torch.arange(1, n + 1, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()  # generated on torch's default stream; the work stream below does not wait for it
    dhts = [1, 2, 3, 4, 5, 6, 7, 8]
    with ops.stream_ctx(ops.work_stream()):
        h = ops.hash_images(imgs)
        sh.load_shard(h, ids)
        ref = {d: sh.similar(h, d, 8) for d in dhts}
        torch.cuda.synchronize()
        want = {d: tuple(t.cpu().numpy().copy() for t in ref[d]) for d in dhts}
        assert n < want[2][2].sum() < 2 * n  # self matches + the planted near-duplicates
        sh.similar_sweep(h, dhts, 8)
        torch.cuda.synchronize()
        sh.fit_capacity()
        for it in range(120):
            h2 = ops.hash_images(imgs)
            sh.load_shard(h2, ids)
            r = sh.similar_sweep(h2, dhts, 8)
            torch.cuda.synchronize()
            for d in dhts:
                cnt = r[d][2].cpu().numpy()
                assert (cnt == want[d][2]).all(), (it, d)
                live = np.arange(8)[None, :] < np.minimum(cnt, 8)[:, None]  # places beyond a needle's count are not written
                assert (r[d][0].cpu().numpy()[live] == want[d][0][live]).all(), (it, d)
                assert (r[d][1].cpu().numpy()[live] == want[d][1][live]).all(), (it, d)
