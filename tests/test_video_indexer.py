"""Media::makeVideoIndex (src/media.cpp:925-1037) as a streaming indexer: cbh_vindexer_* / cbird_amd.video.VideoIndexer
against the oracle's restatement (autocrop + dctHash64 per frame, then the near-frame filter over all hashes)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def vo():
    from oracle import VideoOracle

    return VideoOracle()


@pytest.fixture(scope="module")
def po():
    from oracle import PrestageOracle

    return PrestageOracle()


def scene(rng, h, w):
    """smooth content with a few strong shapes: stable hashes under small noise"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w), np.float32)
    for _ in range(5):
        fx, fy = rng.uniform(0.5, 4, 2)
        img += rng.uniform(20, 60) * np.sin(xx * fx * 6.283 / w + rng.uniform(0, 6)) * np.cos(yy * fy * 6.283 / h)
    for _ in range(4):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        img[y0:y0 + int(rng.integers(8, h // 2)), x0:x0 + int(rng.integers(8, w // 2))] += rng.uniform(-90, 90)
    return img + 128


def clip(seed, n, h, w, bars=(0, 0, 0, 0), cut_every=17, noise=3.0):
    """n grey frames: scenes that drift a little per frame, hard cuts every `cut_every` frames, optional black bars
    (top, bottom, left, right) with a little noise in them (a decoded letterbox is not exactly black)"""
    rng = np.random.default_rng(seed)
    top, bottom, left, right = bars
    out = np.zeros((n, h, w), np.uint8)
    cur = None
    for i in range(n):
        if i % cut_every == 0:
            cur = scene(rng, h - top - bottom, w - left - right)
        f = cur + rng.normal(0, noise, cur.shape)
        frame = np.full((h, w), 16.0) + rng.integers(0, 3, (h, w))
        frame[top:h - bottom, left:w - right] = np.clip(f, 40, 255)
        out[i] = frame.astype(np.uint8)
    return out


def oracle_index(po, vo, frames, threshold, autocrop=20, resume=None):
    hashes = np.array([po.process_image(f, autocrop=autocrop)[0] for f in frames], np.uint64)
    return vo.make_video_index(hashes, threshold, resume=resume), hashes


def test_oracle_make_video_index_rules(vo):
    """the loop's rules on a hand case (threshold 8): first frame stored and NOT in the window, so the second frame
    is never stored; a far frame is stored and restarts the window; the last frame is always appended"""
    A, B = 0x00FF00FF00FF00FF, 0xFF00FF00FF00FF00
    f, h = vo.make_video_index([A, A, A, B, B], 8)
    assert f.tolist() == [0, 3, 4] and h.tolist() == [A, B, B]
    f, h = vo.make_video_index([A, B], 8)           # second frame: empty window -> not stored, but it is the last one
    assert f.tolist() == [0, 1] and h.tolist() == [A, B]
    f, h = vo.make_video_index([A, B, A], 8)        # third frame far from window [B] -> stored
    assert f.tolist() == [0, 2]
    f, h = vo.make_video_index([A, B, A], 0)        # threshold <= 0 stores everything
    assert f.tolist() == [0, 1, 2]
    f, h = vo.make_video_index([], 8)
    assert len(f) == 0
    f, h = vo.make_video_index([A], 8)
    assert f.tolist() == [0]
    # resume (:929-936): numbering continues, the first new frame is stored unconditionally
    f, h = vo.make_video_index([B, B, B], 8, resume=(np.array([0, 3], np.int32), np.array([A, B], np.uint64)))
    assert f.tolist() == [0, 3, 4, 6] and h.tolist() == [A, B, B, B]
    f, h = vo.make_video_index([], 8, resume=(np.array([0, 3], np.int32), np.array([A, B], np.uint64)))
    assert f.tolist() == [0, 3]
    # MAX_FRAMES_PER_VIDEO (:1013-1016), scaled down: decoding stops when frameNumber reaches the limit
    f, h = vo.make_video_index([A, B, A, B, A, B], 0, max_frames=4)
    assert f.tolist() == [0, 1, 2, 3]


def test_oracle_index_equals_dedup_of_all_frames(vo):
    rng = np.random.default_rng(3)
    for thr in (0, 4, 8, 20):
        for n in (1, 2, 3, 64, 700):
            h = rng.integers(1, 2 ** 63, n, dtype=np.uint64)
            for i in range(1, n):
                if rng.random() < 0.75:
                    h[i] = h[i - 1] ^ np.uint64(1 << int(rng.integers(0, 64)))
            f, hh = vo.make_video_index(h, thr)
            keep = vo.dedup(h, thr)
            assert (f == np.nonzero(keep)[0]).all() and (hh == h[keep]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("geom,bars", [((240, 320), (30, 30, 0, 0)), ((180, 320), (0, 0, 0, 0)),
                                        ((256, 256), (0, 0, 0, 0)), ((200, 360), (0, 0, 40, 40))])
def test_indexer_equals_oracle_for_any_chunking(vo, po, geom, bars):
    from cbird_amd.video import VideoIndexer

    h, w = geom
    frames = clip(11, 75, h, w, bars)
    for thr in (8, 0):
        (wf, wh), hashes = oracle_index(po, vo, frames, thr)
        if thr == 8:
            assert 3 < len(wf) < len(frames)  # the filter drops near frames and keeps the cuts
        for chunk in (1, 7, 75):
            ix = VideoIndexer(threshold=thr)
            for i in range(0, len(frames), chunk):
                ix.push(frames[i:i + chunk])
            got = ix.finish()
            assert ix.frames_seen == len(frames)
            assert got.frames == wf.tolist() and got.hashes == [int(x) for x in wh], (thr, chunk)
            assert ix.finish().frames == got.frames  # finish() does not change the state


@pytest.mark.gpu
def test_indexer_letterbox_is_cropped_before_hashing(vo, po):
    """the hashes in the index are those of the autocropped frames (media.cpp:961, :989)"""
    from cbird_amd.video import VideoIndexer

    frames = clip(5, 20, 240, 320, (30, 30, 0, 0))
    assert po.autocrop(frames[0]).tolist() == [0, 30, 320, 210]
    ix = VideoIndexer(threshold=8)
    ix.push(frames)
    got = ix.finish()
    ix0 = VideoIndexer(threshold=8, autocrop_range=-1)
    ix0.push(frames)
    assert got.hashes[0] == po.process_image(frames[0], autocrop=20)[0]
    assert ix0.finish().hashes[0] == po.process_image(frames[0], autocrop=None)[0] != got.hashes[0]


@pytest.mark.gpu
def test_indexer_device_frames_strides_and_geometry_change(vo, po):
    import torch

    from cbird_amd.video import VideoIndexer

    a = clip(21, 40, 240, 320, (30, 30, 0, 0))
    b = clip(22, 30, 180, 240)
    (wf, wh), _ = oracle_index(po, vo, list(a) + list(b), 8)
    # host, padded rows and padded frames
    pad = np.zeros((40, 250, 352), np.uint8)
    pad[:, :240, :320] = a
    ix = VideoIndexer(threshold=8)
    ix.push(pad[:, :240, :320])
    ix.push(b)
    got = ix.finish()
    assert got.frames == wf.tolist() and got.hashes == [int(x) for x in wh]
    # device tensors (a hardware decoder's output), also strided, single frames as (h, w)
    dev = torch.device("cuda", 0)
    ta = torch.from_numpy(pad).to(dev)[:, :240, :320]
    tb = torch.from_numpy(b).to(dev)
    ix = VideoIndexer(threshold=8)
    ix.push(ta[:13])
    ix.push(ta[13])
    ix.push(ta[14:])
    ix.push(tb)
    got = ix.finish()
    assert got.frames == wf.tolist() and got.hashes == [int(x) for x in wh]


@pytest.mark.gpu
def test_indexer_resume_and_edges(vo, po):
    from cbird_amd._lib import CbhError
    from cbird_amd.video import VideoIndex, VideoIndexer

    frames = clip(31, 60, 200, 200)
    # nothing pushed
    assert VideoIndexer().finish().isEmpty()
    # one frame
    ix = VideoIndexer()
    ix.push(frames[0])
    one = ix.finish()
    assert one.frames == [0] and one.hashes == [po.process_image(frames[0])[0]]
    # stop after 25 frames, resume from the written index with the rest
    ix = VideoIndexer(threshold=8)
    ix.push(frames[:25])
    part = ix.finish()
    assert part.frames[-1] == 24  # the last frame is always included: decoding resumes at 25
    res = VideoIndexer(threshold=8, resume=part)
    assert res.frames_seen == 25
    res.push(frames[25:])
    got = res.finish()
    (wf, wh), _ = oracle_index(po, vo, frames[25:], 8,
                               resume=(np.array(part.frames, np.int32), np.array(part.hashes, np.uint64)))
    assert got.frames == wf.tolist() and got.hashes == [int(x) for x in wh]
    assert got.frames[:len(part.frames)] == part.frames and got.frames[-1] == 59
    # a resumed indexer with no new frames returns the old index
    assert VideoIndexer(resume=part).finish().frames == part.frames
    # resume() is refused once frames went in; bad resume lists are refused
    import ctypes as C

    from cbird_amd import _lib
    L = _lib.lib()
    f = np.array([0, 5], np.int32)
    hh = np.array([1, 2], np.uint64)
    assert L.cbh_vindexer_resume(ix._h, f.ctypes.data, hh.ctypes.data, 2) == _lib.CBH_E_INVAL
    with pytest.raises(CbhError):
        VideoIndexer(resume=VideoIndex([3, 3], [1, 2]))
    # bad geometry
    assert L.cbh_vindexer_push(ix._h, frames.ctypes.data, 2, 200, 200, 100, 40000) == _lib.CBH_E_INVAL
    assert L.cbh_vindexer_push(None, frames.ctypes.data, 2, 200, 200, 200, 40000) == _lib.CBH_E_INVAL
    assert L.cbh_vindexer_finish(None, None, None, 0) == _lib.CBH_E_INVAL
    del C


@pytest.mark.gpu
def test_indexer_output_round_trips_through_vdx_and_findvideo(vo, po, tmp_path):
    """the produced index is what DctVideoIndex consumes: save as .vdx, load, find the clip it came from"""
    from cbird_amd.index import Media
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoIndexer, VideoSearchParams

    vids = []
    for k in range(3):
        ix = VideoIndexer(threshold=8)
        ix.push(clip(100 + k, 90, 180, 240, cut_every=9))
        vi = ix.finish()
        vi.save(str(tmp_path / f"{k + 1}.vdx"))
        assert VideoIndex.isValid(str(tmp_path / f"{k + 1}.vdx"))
        vids.append(vi)
    idx = DctVideoIndex(0, str(tmp_path))
    idx.load([1, 2, 3])
    assert idx.count() == 3
    p = VideoSearchParams(dctThresh=1, skipFrames=0, minFramesMatched=1, minFramesNear=1, filterSelf=False)
    needle = Media(id=2)
    needle.videoIndex = VideoIndex.load(str(tmp_path / "2.vdx"))
    assert needle.videoIndex.frames == vids[1].frames
    m = idx.find(needle, p)
    assert [x.mediaId for x in m][:1] == [2]


@pytest.mark.gpu
def test_indexers_on_worker_threads_are_independent(vo, po):
    """Scanner::processVideo runs one video per worker thread: four indexers fed concurrently (different geometries,
    chunk sizes and letterboxes) give what each gives alone"""
    from concurrent.futures import ThreadPoolExecutor

    from cbird_amd.video import VideoIndexer

    jobs = [(41, 60, 240, 320, (30, 30, 0, 0), 7), (42, 50, 180, 320, (0, 0, 0, 0), 16),
            (43, 70, 200, 360, (0, 0, 40, 40), 5), (44, 40, 256, 256, (0, 0, 0, 0), 40)]
    clips = [clip(s, n, h, w, bars) for (s, n, h, w, bars, _) in jobs]
    want = [oracle_index(po, vo, c, 8)[0] for c in clips]

    def run(k):
        ix = VideoIndexer(threshold=8)
        ch = jobs[k][5]
        for rep in range(3):  # several rounds so that the threads really overlap
            ix = VideoIndexer(threshold=8)
            for i in range(0, len(clips[k]), ch):
                ix.push(clips[k][i:i + ch])
        return ix.finish()

    with ThreadPoolExecutor(4) as ex:
        got = list(ex.map(run, range(4)))
    for k in range(4):
        assert got[k].frames == want[k][0].tolist() and got[k].hashes == [int(x) for x in want[k][1]], k


@pytest.mark.gpu
def test_indexer_when_the_kept_region_changes_inside_a_clip(vo, po):
    """the hash launch of a chunk speculates on the region the previous chunk ended on (the first chunk: on the region of
    one of its middle frames) and hashes again only the frames whose rectangle came out different (vindexer.hip,
    hash_chunk): a clip that fades in from black (nothing to crop), is letterboxed, switches to pillarbox with a logo in a
    bar and ends without bars must give the oracle's index for any chunking"""
    from cbird_amd.video import VideoIndexer

    h, w = 216, 384
    parts = [clip(31, 5, h, w, (0, 0, 0, 0)), clip(32, 23, h, w, (28, 28, 0, 0)), clip(33, 19, h, w, (0, 0, 48, 48)),
             clip(34, 9, h, w, (28, 28, 0, 0)), clip(35, 11, h, w, (0, 0, 0, 0))]
    parts[0][:] = 16  # black lead-in: every pixel is border
    parts[2][:, h // 2, 10] = 200  # a logo pixel inside the left bar
    frames = np.concatenate(parts)
    rects = {tuple(po.autocrop(f).tolist()) for f in frames}
    assert len(rects) >= 3  # whole frame, letterbox, pillarbox
    (wf, wh), _ = oracle_index(po, vo, frames, 8)
    for chunk in (1, 9, 16, 40, len(frames)):
        ix = VideoIndexer(threshold=8)
        for i in range(0, len(frames), chunk):
            ix.push(frames[i:i + chunk])
        got = ix.finish()
        assert got.frames == wf.tolist() and got.hashes == [int(x) for x in wh], chunk
