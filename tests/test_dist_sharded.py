"""SURVEY.md 8(e) for the two indexes whose reduction is per media: DctVideoIndex sharded BY VIDEO and
CvFeaturesIndex sharded BY IMAGE (cbird_amd/dist.py).  CPU: world 2 over gloo with oracle-backed stand-ins for
the shard-local index (the sharding, the fixed-size exchange and the merge are the product code under test);
GPU: two ranks sharing the one MI355X with the real indexes.  Expected results: the unsharded oracle."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


class _M:
    pass


def _video_media(clips, first_id=100):
    from cbird_amd.video import VideoIndex

    out = []
    for i, (f, h) in enumerate(clips):
        m = _M()
        m.id, m.path, m.videoIndex, m.dctHash = first_id + i, f"v{i}", VideoIndex(f.tolist(), [int(x) for x in h]), 0
        out.append(m)
    return out


def _orb_media(rows, ids, per_media):
    out = []
    for i, mid in enumerate(ids.tolist()):
        m = _M()
        m.id, m.path, m.keyPointDescriptors = int(mid), f"i{i}", rows[i * per_media:(i + 1) * per_media]
        out.append(m)
    return out


class FakeVideoIndex:
    """shard-local DctVideoIndex stand-in on the oracle (test infrastructure)"""

    def __init__(self):
        from oracle import VideoOracle

        self.vorc, self.media = VideoOracle(), []

    def add(self, media):
        self.media += list(media)

    def find_videos_batch(self, needles, p):
        from cbird_amd.index import Match, MatchRange

        videos = [(m.id, np.asarray(m.videoIndex.frames, np.int32), np.asarray(m.videoIndex.hashes, np.uint64))
                  for m in self.media]
        entries = self.vorc.build_entries(videos, p.skipFrames)
        out = []
        for m in needles:
            r = self.vorc.find_video(entries, m.videoIndex.frames, m.videoIndex.hashes, m.id, p.dctThresh,
                                     p.skipFrames, p.minFramesMatched, p.minFramesNear) if videos else []
            out.append([Match(a, b, MatchRange(c, d, e)) for a, b, c, d, e in r])
        return out


class FakeCvIndex:
    """shard-local CvFeaturesIndex stand-in on the oracle"""

    def __init__(self):
        from oracle import CvOracle

        self.cvo, self.rows, self.first, self.ids = CvOracle(), np.zeros((0, 32), np.uint8), [], []

    def add(self, media):
        for m in media:
            self.first.append(len(self.rows))
            self.ids.append(m.id)
            self.rows = np.concatenate([self.rows, np.asarray(m.keyPointDescriptors, np.uint8)])

    def knn_media(self, needles, k, thresh):
        nq = len(needles)
        if len(self.rows) == 0:
            z = np.zeros((nq, k), np.uint32)
            return z, np.zeros((nq, k), np.uint16), z.copy(), np.zeros(nq, np.uint32)
        r, d, c = self.cvo.knn(self.rows, needles, k, thresh)
        media = np.asarray(self.ids, np.uint32)[np.searchsorted(np.asarray(self.first), r, side="right") - 1]
        media[np.arange(k)[None, :] >= np.minimum(c, k)[:, None]] = 0
        return r, d.astype(np.uint16), media, c


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _video_case():
    from cbird_amd import synth_video

    return synth_video.make_clips(41, 120, seed=5, subclip_frac=0.25, max_gap=6)


def _orb_case():
    from test_cvfeatures import make_descriptors

    return make_descriptors(23, 40, 9) + (40,)


def _worker(rank, world, port, real, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd.dist import ShardedCvFeaturesIndex, ShardedDctVideoIndex
        from cbird_amd.index import SearchParams
        from cbird_amd.video import VideoSearchParams

        if real:
            import torch

            from cbird_amd.cvfeatures import CvFeaturesIndex
            from cbird_amd.video import DctVideoIndex

            torch.cuda.set_device(0)
            mk_v, mk_c = DctVideoIndex, CvFeaturesIndex
        else:
            mk_v, mk_c = FakeVideoIndex, FakeCvIndex
        media = _video_media(_video_case())
        sv = ShardedDctVideoIndex(mk_v)
        sv.add(media)
        p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=5, minFramesNear=20)
        vres = [[(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
                for r in sv.find_videos_batch(media, p)]
        rows, first, ids, per = _orb_case()
        om = _orb_media(rows, ids, per)
        sc = ShardedCvFeaturesIndex(mk_c)
        sc.add(om)
        ores = [[(x.mediaId, x.score) for x in r] for r in sc.find_batch(om[::3], SearchParams(cvThresh=25))]
        q_out.put((rank, vres, ores, sc.row_offset))
    finally:
        dist.destroy_process_group()


def _run(real):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, real, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted(q.get(timeout=240) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return got


def _check(got):
    from oracle import CvOracle, VideoOracle

    vorc, cvo = VideoOracle(), CvOracle()
    clips = _video_case()
    media = _video_media(clips)
    entries = vorc.build_entries([(m.id, f, h) for m, (f, h) in zip(media, clips)], 0)
    want_v = [vorc.find_video(entries, m.videoIndex.frames, m.videoIndex.hashes, m.id, 5, 0, 5, 20) for m in media]
    assert sum(len(w) for w in want_v) > 5
    rows, first, ids, per = _orb_case()
    om = _orb_media(rows, ids, per)
    want_o = []
    for m in om[::3]:
        wi, ws = cvo.find(rows, first, ids, m.keyPointDescriptors, 10, 25)
        want_o.append(list(zip(wi.tolist(), ws.tolist())))
    assert got[1][3] == 11 * per  # rank 1's rows start after the 11 images of rank 0
    for rank, vres, ores, _ in got:
        assert vres == want_v, rank
        assert ores == want_o, rank


def test_sharded_video_and_orb_world2_gloo_equal_unsharded_oracle():
    _check(_run(real=False))


@pytest.mark.gpu
def test_sharded_video_and_orb_two_ranks_one_gpu(gpu):
    _check(_run(real=True))


def test_gather_rows_grows_on_overflow():
    """single process: the helper is a no-op; the overflow path is covered by world 2 below"""
    from cbird_amd.dist import _gather_rows

    r = np.arange(12, dtype=np.int32).reshape(4, 3)
    assert (_gather_rows(r) == r).all()


def _gr_worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd.dist import _gather_rows

        n = 3 if rank == 0 else 700  # rank 1 overflows the initial capacity of 64 rows
        rows = (np.arange(n * 2, dtype=np.int32).reshape(n, 2) + 100000 * rank)
        q_out.put((rank, _gather_rows(rows, cap0=64)))
    finally:
        dist.destroy_process_group()


def test_gather_rows_world2_ragged_overflow():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gr_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    want = np.concatenate([np.arange(6, dtype=np.int32).reshape(3, 2),
                           np.arange(1400, dtype=np.int32).reshape(700, 2) + 100000])
    assert (got[0] == want).all() and (got[1] == want).all()
