"""DctVideoIndex / VideoIndex (src/dctvideoindex.cpp, src/videoindex.cpp): .vdx codec known answers from
the reference's own test vectors (unit/testvideoindex.cpp:174-258), oracle vs the real RadixMap, product
host code vs oracle on CPU, and the GPU find path vs oracle."""
import numpy as np
import pytest

# the in-source vectors of unit/testvideoindex.cpp:174-258 (frames, hashes)
KAT = [([0, 1, 2, 3], [4, 3, 2, 1]), ([0, 1, 2000, 2001], [4, 3, 2, 1]), ([0, 1, 2, 2000], [4, 3, 2, 1]),
       ([0, 1000, 1001, 1002], [4, 3, 2, 1]), ([0, 1000, 2000, 3000], [4, 3, 2, 1]),
       ([0, 1000, 1001, 2000, 2001, 3000, 3001, 4000], [4, 3, 2, 1, 1, 2, 3, 4])]


@pytest.fixture(scope="module")
def vorc():
    from oracle import VideoOracle

    return VideoOracle()


def test_vdx_roundtrip_reference_vectors(vorc):
    from cbird_amd.video import VideoIndex

    for frames, hashes in KAT + [([], [])]:
        data = vorc.vdx_encode(frames, hashes)
        f, h = vorc.vdx_decode(data)
        assert f.tolist() == frames and h.tolist() == hashes
        # the product codec (host code of libcbird_hip.so) writes the same bytes and reads them back
        vi = VideoIndex(frames, hashes)
        assert vi.to_bytes() == data
        back = VideoIndex.from_bytes(data)
        assert back.frames == frames and back.hashes == hashes


def test_vdx_byte_layout(vorc):
    """hand-derived from save_v2 (videoindex.cpp:271-347) for frames {0,1,2000,2001}"""
    data = vorc.vdx_encode([0, 1, 2000, 2001], [4, 3, 2, 1])
    header = b"cbird video index:0.8.1:2:1:1:8:4:\n"
    assert data.startswith(header)
    # frame 0 -> 0x00; +1 -> 0x01; +1999 = 0b1111_1001111 -> 0xCF (0x4F|0x80), 0x0F; +1 -> 0x01
    packed = bytes([0x00, 0x01, 0xCF, 0x0F, 0x01])
    off = len(header)
    assert data[off:off + 4] == (5).to_bytes(4, "little")
    assert data[off + 4:off + 9] == packed
    pad = (8 - (off + 4 + 5) % 8) % 8
    hs = off + 9 + pad
    assert data[off + 9:hs] == b"\0" * pad
    assert data[hs:hs + 32] == b"".join(int(x).to_bytes(8, "little") for x in (4, 3, 2, 1))
    assert data[hs + 32:] == b"cbir"


def test_vdx_rejects_corruption(vorc, tmp_path):
    from cbird_amd.video import VideoIndex

    good = vorc.vdx_encode([0, 5, 9], [1, 2, 3])
    for bad in (good[:-5], good[:30], b"not a cbird video index:\n", good.replace(b":2:1:1:8:", b":3:1:1:8:")):
        with pytest.raises(ValueError):
            vorc.vdx_decode(bad)
        with pytest.raises(ValueError):
            VideoIndex.from_bytes(bad)
        assert not vorc.vdx_verify(bad)
    with pytest.raises(ValueError):
        vorc.vdx_encode([1, 2], [1, 2])  # first frame must be 0
    with pytest.raises(ValueError):
        vorc.vdx_encode([0, 2, 2], [1, 2, 3])  # non-sequential


def _v1(frames, hashes, count=None):
    """a version-1 file by the format (videoindex.cpp:448-476): u16 count, u16 frame numbers, u64 hashes"""
    n = len(frames) if count is None else count
    return (n.to_bytes(2, "little") + b"".join(int(f).to_bytes(2, "little") for f in frames) +
            b"".join(int(h).to_bytes(8, "little") for h in hashes))


def test_vdx_version1_files_load_like_the_reference(vorc, tmp_path):
    """unit/testvideoindex.cpp testV1Load / testV1Save / testLoad: VideoIndex::load and isValid pick the version by the
    "cbird" magic (videoindex.cpp:41-68); load_v1's repairs (:503-535); product host code == oracle == the format"""
    from cbird_amd import _lib
    from cbird_amd.video import VideoIndex

    L = _lib.lib()

    def version(data):
        b = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        return L.cbh_vdx_version(b.ctypes.data, len(data))

    def valid(data):
        b = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        return bool(L.cbh_vdx_verify(b.ctypes.data, len(data)))

    def encode_v1(frames, hashes):
        f, h = np.ascontiguousarray(frames, np.int32), np.ascontiguousarray(hashes, np.uint64)
        n = L.cbh_vdx_encode_v1(f.ctypes.data, h.ctypes.data, len(f), None, 0)
        out = np.zeros(n, np.uint8)
        L.cbh_vdx_encode_v1(f.ctypes.data, h.ctypes.data, len(f), out.ctypes.data, n)
        return out.tobytes()

    # testV1Save (:71-92): save_v1 then load_v1 round trip
    frames, hashes = [0, 10, 30], [40404040, 10101010, 30303030]
    data = encode_v1(frames, hashes)
    assert data == _v1(frames, hashes) == vorc.vdx_encode_v1(frames, hashes)
    assert version(data) == 1 and valid(data) and vorc.vdx_any_verify(data)
    vi = VideoIndex.from_bytes(data)
    assert vi.frames == frames and vi.hashes == hashes
    # testLoad (:94-118): a 201-frame version-1 index (frames 0..1999) and its version-2 conversion load identically
    f201 = list(range(0, 2000, 10)) + [1999]
    h201 = [int(x) for x in np.random.default_rng(1).integers(1, 2 ** 63, 201, dtype=np.uint64)]
    v1 = VideoIndex.from_bytes(_v1(f201, h201))
    assert len(v1.frames) == 201 and v1.frames[0] == 0 and v1.frames[-1] == 1999
    v2data = v1.to_bytes()
    assert version(v2data) == 2 and valid(v2data)
    v2 = VideoIndex.from_bytes(v2data)
    assert v2.frames == v1.frames and v2.hashes == v1.hashes == h201
    # version1-truncated / empty / version1-empty (:44-68)
    cut = _v1(f201, h201)[:-5]
    assert not valid(cut) and not vorc.vdx_any_verify(cut) and vorc.vdx_any_decode(cut) is None
    with pytest.raises(ValueError):
        VideoIndex.from_bytes(cut)
    assert not valid(b"") and not vorc.vdx_any_verify(b"") and vorc.vdx_any_decode(b"") is None
    with pytest.raises(ValueError):
        VideoIndex.from_bytes(b"")
    none = _v1([], [])
    assert valid(none) and vorc.vdx_any_verify(none) and VideoIndex.from_bytes(none).isEmpty()
    # load_v1's repairs: frame numbers that wrapped past 65535 cut the index (:503-516) ...
    wrap = [0, 30000, 65400, 65500, 100, 200]
    got = VideoIndex.from_bytes(_v1(wrap, [1, 2, 3, 4, 5, 6]))
    assert got.frames == [0, 30000, 65400, 65500, 65535] and got.hashes == [1, 2, 3, 4, 5]
    got = VideoIndex.from_bytes(_v1([0, 7, 65535, 3], [1, 2, 3, 4]))
    assert got.frames == [0, 7, 65535] and got.hashes == [1, 2, 3]
    with pytest.raises(ValueError):  # out of order without the wrap signature: corrupt
        VideoIndex.from_bytes(_v1([0, 500, 100], [1, 2, 3]))
    # ... and an index that does not start at frame 0 gets one in front (:530-535)
    got = VideoIndex.from_bytes(_v1([5, 10], [11, 12]))
    assert got.frames == [0, 5, 10] and got.hashes == [0, 11, 12]
    # save_v1 drops what the format cannot hold (:452-468)
    assert VideoIndex.from_bytes(encode_v1([0, 100, 70000, 70001], [1, 2, 3, 4])).frames == [0, 100]
    # product host code == oracle on random version-1 files, good and damaged
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(0, 40))
        fr = np.sort(rng.choice(65536, n, replace=False)) if rng.random() < 0.7 else rng.integers(0, 65536, n)
        if n and rng.random() < 0.5:
            fr[0] = 0
        data = _v1(fr.tolist(), rng.integers(0, 2 ** 63, n).tolist())
        if rng.random() < 0.2:
            data = data[: int(rng.integers(0, len(data) + 1))]
        want = vorc.vdx_any_decode(data)
        assert valid(data) == vorc.vdx_any_verify(data)
        if want is None:
            with pytest.raises(ValueError):
                VideoIndex.from_bytes(data)
        else:
            got = VideoIndex.from_bytes(data)
            assert got.frames == want[0].tolist() and got.hashes == [int(x) for x in want[1]]


def test_vdx_load_does_not_need_the_trailer_but_isvalid_does(vorc, tmp_path):
    """load_v2 (src/videoindex.cpp:350-429) never looks at "cbir"; verify_v2 (:248-269, what isValid runs and what
    Engine::update uses to re-queue videos) does.  A file cut inside the trailer therefore LOADS and is INVALID."""
    from cbird_amd.video import VideoIndex

    good = vorc.vdx_encode([0, 5, 9, 400], [1, 2, 3, 4])
    assert vorc.vdx_verify(good)
    for cut in (good[:-4], good[:-1]):
        f, h = vorc.vdx_decode(cut)
        assert f.tolist() == [0, 5, 9, 400] and h.tolist() == [1, 2, 3, 4]
        vi = VideoIndex.from_bytes(cut)
        assert vi.frames == [0, 5, 9, 400] and vi.hashes == [1, 2, 3, 4]
        assert not vorc.vdx_verify(cut)
        p = tmp_path / "cut.vdx"
        p.write_bytes(cut)
        assert not VideoIndex.isValid(str(p))
    p = tmp_path / "good.vdx"
    p.write_bytes(good)
    assert VideoIndex.isValid(str(p)) and not VideoIndex.isValid(str(tmp_path / "missing.vdx"))
    empty = vorc.vdx_encode([], [])
    assert vorc.vdx_verify(empty)  # "no frames stored": valid without a trailer (:256-259)
    p.write_bytes(empty)
    assert VideoIndex.isValid(str(p))


def test_dedup_rule(vorc):
    """src/media.cpp:958-1024 on a hand-made sequence (threshold 8)"""
    from cbird_amd.video import make_video_index

    a = 0x0F0F0F0F0F0F0F00
    far = a ^ 0xFFFF  # 16 bits away
    seq = [a, far, a ^ 1, a ^ 3, far, far ^ 1, a]
    keep = vorc.dedup(seq, 8)
    # frame0 stored; frame1 never (empty window); frame2: window={far}: far vs a^1 is 15 away -> stored;
    # frame3: window={a^1}: 1 away -> dropped; frame4: window={a^1,a^3}: far -> stored; frame5: window={far}
    # near -> dropped; frame6 (last): window={far,far^1}: far -> stored
    assert keep.tolist() == [True, False, True, False, True, False, True]
    vi = make_video_index(seq, 8)
    assert vi.frames == [0, 2, 4, 6] and vi.hashes == [seq[0], seq[2], seq[4], seq[6]]
    assert vorc.dedup(seq, 0).all()
    rng = np.random.default_rng(1)
    walk = np.cumsum(rng.integers(0, 2, 500)).astype(np.uint64) * np.uint64(0x0101010101010101)
    k = vorc.dedup(walk, 8)
    assert k[0] and k[-1]
    assert make_video_index(walk, 8).frames == np.nonzero(k)[0].tolist()
    # static scenes and slow drift: a few hashes repeating (the library keeps the window as a set), a hash that wanders
    # a bit at a time (near its neighbours, eventually far from the window's oldest entry), both with cuts in between
    for seed in range(6):
        rng = np.random.default_rng(100 + seed)
        cur, out = int(rng.integers(0, 1 << 62)), []
        for i in range(3000):
            r = rng.random()
            if r < 0.004:
                cur = int(rng.integers(0, 1 << 62))  # cut
            elif r < (0.02, 0.3, 0.6)[seed % 3]:
                cur ^= 1 << int(rng.integers(0, 64))  # drift
            out.append(cur ^ ((1 << int(rng.integers(0, 64))) if rng.random() < 0.3 else 0))  # flicker
        out = np.array(out, np.uint64)
        for thr in (1, 4, 8, 20):
            assert make_video_index(out, thr).frames == np.nonzero(vorc.dedup(out, thr))[0].tolist(), (seed, thr)


def test_oracle_candidates_vs_real_radixmap(vorc):
    import oracle

    if not oracle.ref_qt_available():
        pytest.skip("oracle/_ref/libcbird_ref_qt.so not built")
    from cbird_amd import synth_video

    clips = synth_video.make_clips(40, 120, seed=5, subclip_frac=0.2)
    videos = [(100 + i, f, h) for i, (f, h) in enumerate(clips)]
    entries = vorc.build_entries(videos, skip=0)
    ev, ef, eh, mids = entries
    for radix in (0, 10):
        rm = oracle.RefRadixMap(radix)
        rm.insert(ev, ef.astype(np.uint32), eh)
        for q in eh[::37].tolist():
            rv, rf, rh, rd = rm.search(q, 6)
            # reference findFrame reduction (dctvideoindex.cpp:346-356) over the real map's matches
            nearest = {}
            for v, f, d in zip(rv.tolist(), rf.tolist(), rd.tolist()):
                if v not in nearest or d < nearest[v][0]:
                    nearest[v] = (d, f)
            want = [(int(mids[v]), d, 0, f, 1) for v, (d, f) in sorted(nearest.items())]
            assert vorc.find_frame(entries, q, 6, -1, radix=radix) == want
    # radix 10 returns a subset of radix 0
    full = vorc.find_frame(entries, int(eh[5]), 9, -1, radix=0)
    part = vorc.find_frame(entries, int(eh[5]), 9, -1, radix=10)
    assert {x[0] for x in part} <= {x[0] for x in full}


def test_insert_filter_rules(vorc):
    frames = np.arange(0, 1000, 10, dtype=np.int32)  # lastFrame 990
    hashes = np.full(100, 0x00FF00FF00FF00FF, np.uint64)
    hashes[3] = 0xF  # 4 ones -> dropped
    hashes[4] = ~np.uint64(0xF)  # 4 zeros -> dropped
    keep = np.zeros(100, np.uint8)
    vorc.L.orc_video_insert_filter(frames, hashes, 100, 300, keep)  # 990/2 > 300 -> trim both ends
    want = (frames >= 300) & (frames <= 690)
    assert keep.astype(bool).tolist() == want.tolist()
    vorc.L.orc_video_insert_filter(frames, hashes, 100, 600, keep)  # 990/2 < 600 -> no trimming at all
    want = np.ones(100, bool)
    want[[3, 4]] = False
    assert keep.astype(bool).tolist() == want.tolist()


def _mk_index(gpu, clips, first_id=100):
    from cbird_amd.video import DctVideoIndex, VideoIndex

    idx = DctVideoIndex()

    class M:
        pass

    media = []
    for i, (f, h) in enumerate(clips):
        m = M()
        m.id, m.path, m.videoIndex, m.dctHash = first_id + i, f"v{i}", VideoIndex(f.tolist(), [int(x) for x in h]), 0
        media.append(m)
    idx.add(media)
    return idx, media


@pytest.mark.gpu
def test_gpu_find_video_and_frame_vs_oracle(gpu, vorc, reduce_path):
    from cbird_amd import synth_video
    from cbird_amd.video import VideoSearchParams

    clips = synth_video.make_clips(300, 300, seed=11, subclip_frac=0.1, max_gap=8)
    idx, media = _mk_index(gpu, clips)
    assert idx.count() == 300 and idx.isLoaded()
    assert idx.memoryUsage() == 0  # `_tree ? _tree->stats().memory : 0` (dctvideoindex.cpp:57-59): nothing built yet
    videos = [(m.id, f, h) for m, (f, h) in zip(media, clips)]
    for skip, thr, vfm, vfn in ((0, 5, 30, 60), (0, 2, 5, 10), (40, 7, 10, 30)):
        idx2, _ = _mk_index(gpu, clips)  # the tree is built once per index with the first query's vtrim
        p = VideoSearchParams(dctThresh=thr, skipFrames=skip, minFramesMatched=vfm, minFramesNear=vfn)
        entries = vorc.build_entries(videos, skip)
        assert idx2.entries(skip) == len(entries[0])
        assert idx2.memoryUsage() == 14 * len(entries[0])  # hash_t + packed VideoTreeIndex per entry
        n_hits = 0
        for m in media[::7] + media[-30:]:
            got = [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in idx2.findVideo(m, p)]
            want = vorc.find_video(entries, m.videoIndex.frames, m.videoIndex.hashes, m.id, thr, skip, vfm, vfn)
            assert got == want, (m.id, skip, thr)
            n_hits += len(got)
        assert n_hits > 10
        # image needle
        for m in media[::29]:
            m.dctHash = m.videoIndex.hashes[len(m.videoIndex.hashes) // 2]
            got = [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in idx2.findFrame(m, p)]
            assert got == vorc.find_frame(entries, m.dctHash, thr, -1)
    # video -> itself is the single result when filterSelf is off (unit/testdctvideoindex.cpp:73-76 shape)
    p = VideoSearchParams(dctThresh=1, skipFrames=0, minFramesMatched=1, minFramesNear=1, filterSelf=False)
    r = idx.findVideo(media[0], p)
    assert [x.mediaId for x in r][:1] == [media[0].id] and r[0].score <= 10 and r[0].range.srcIn == 0


@pytest.mark.gpu
def test_gpu_video_batch_remove_add(gpu, vorc, reduce_path):
    from cbird_amd import synth_video
    from cbird_amd.video import VideoSearchParams

    clips = synth_video.make_clips(120, 200, seed=3, subclip_frac=0.15, max_gap=8)
    idx, media = _mk_index(gpu, clips)
    p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=10, minFramesNear=30)
    singles = [idx.findVideo(m, p) for m in media]
    batch = idx.find_videos_batch(media, p)
    key = lambda r: [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
    assert [key(a) for a in singles] == [key(b) for b in batch]
    victims = [m.id for m, r in zip(media, singles) if r][:2]
    idx.remove(victims)
    assert idx.count() == 118
    for r in idx.find_videos_batch(media, p):
        assert all(x.mediaId not in victims for x in r)
    idx.add([m for m in media if m.id in victims])
    assert idx.count() == 120


@pytest.mark.gpu
def test_gpu_video_search_index_batch_equals_per_needle(gpu):
    """Database::searchIndex over DctVideoIndex for a needle batch (cbh_vidx_search_index_batch: one batched findVideo
    per dctThresh level) == one searchIndex per needle; findVideo itself drops the needle's own video when filterSelf
    (src/dctvideoindex.cpp:494), so the count that decides about another level excludes it"""
    import warnings

    from cbird_amd import synth_video
    from cbird_amd.database import search_index, search_index_batch
    from cbird_amd.video import VideoSearchParams

    clips = synth_video.make_clips(150, 200, seed=5, subclip_frac=0.2, max_gap=8)
    idx, media = _mk_index(gpu, clips)
    for m in media:
        m.isValid = lambda: True
        m.score = -1
    id_map = {m.id: m for m in media if m.id % 11 != 0}
    needles = media[::2]
    for kw in (dict(dctThresh=5, filterSelf=False), dict(dctThresh=1, maxThresh=7, minMatches=1, filterSelf=True),
               dict(dctThresh=2, maxThresh=4, minMatches=2, maxMatches=2, filterSelf=False)):
        p = VideoSearchParams(skipFrames=0, minFramesMatched=10, minFramesNear=30, **kw)
        p.algo = p.AlgoVideo
        got = search_index_batch(idx, needles, p, id_map)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = [search_index(idx, m, p, id_map) for m in needles]
        key = lambda g: [(x.id, x.score, x.matchRange.srcIn, x.matchRange.dstIn, x.matchRange.len) for x in g]
        assert [key(g) for g in got] == [key(g) for g in want], kw
        assert any(len(g) for g in got)


@pytest.mark.gpu
def test_gpu_radix_compatible_mode_equals_bucket_search(gpu, vorc, reduce_path):
    """`-p.vradix N` of the reference: a needle frame only sees its RadixMap bucket.  The oracle's bucket rule is
    pinned to the real RadixMap (test_oracle_candidates_vs_real_radixmap); DctVideoIndex(radix_compat=True) must
    equal it, and differ from the exact search where the exact search finds more."""
    from cbird_amd import synth_video
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

    clips = synth_video.make_clips(200, 200, seed=21, subclip_frac=0.15, max_gap=6)

    class M:
        pass

    media = []
    for i, (f, h) in enumerate(clips):
        m = M()
        m.id, m.path, m.videoIndex, m.dctHash = 500 + i, f"v{i}", VideoIndex(f.tolist(), [int(x) for x in h]), 0
        media.append(m)
    videos = [(m.id, f, h) for m, (f, h) in zip(media, clips)]
    entries = vorc.build_entries(videos, 0)
    exact_more = 0
    for radix in (10, 3, 24):
        idx = DctVideoIndex(radix_compat=True)
        idx.add(media)
        p = VideoSearchParams(dctThresh=7, skipFrames=0, minFramesMatched=5, minFramesNear=20, videoRadix=radix)
        for m in media[::5]:
            got = [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in idx.findVideo(m, p)]
            want = vorc.find_video(entries, m.videoIndex.frames, m.videoIndex.hashes, m.id, 7, 0, 5, 20, radix=radix)
            assert got == want, (m.id, radix)
            full = vorc.find_video(entries, m.videoIndex.frames, m.videoIndex.hashes, m.id, 7, 0, 5, 20, radix=0)
            exact_more += got != full
        for m in media[::17]:
            m.dctHash = int(m.videoIndex.hashes[3]) ^ 0x40000000  # a near frame outside the bucket for small radix
            got = [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in idx.findFrame(m, p)]
            assert got == vorc.find_frame(entries, m.dctHash, 7, -1, radix=radix), (m.id, radix)
    assert exact_more > 0


# ---- fixture pin of a7: the real RadixMap's answers, committed (tests/golden/gen_golden_radix.py) --------------------
def _radix_golden():
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "radixmap_r0_r10.npz"))
    offs = g["clip_offs"]
    clips = [(g["frames"][a:b], g["hashes"][a:b]) for a, b in zip(offs[:-1], offs[1:])]
    return g, clips


def _golden_rows(g, key, qi):
    a, b = g[key + "_offs"][qi], g[key + "_offs"][qi + 1]
    return [tuple(int(x) for x in r) for r in g[key][a:b]]


def test_oracle_equals_radixmap_golden(vorc):
    """the oracle's bucket rule + findFrame reduction against the real RadixMap's committed answers (radix 0 and 10) --
    the same pin as test_oracle_candidates_vs_real_radixmap, but from a fixture, so it also holds where
    oracle/_ref/libcbird_ref_qt.so cannot be loaded"""
    g, clips = _radix_golden()
    videos = [(int(m), f, h) for m, (f, h) in zip(g["media_ids"], clips)]
    entries = vorc.build_entries(videos, skip=0)
    ev, ef, eh, _ = entries
    assert len(eh) == len(g["hashes"])  # no hash is dropped by the insert filter: the golden's entry list is every frame
    n_rows = 0
    for radix in g["radixes"].tolist():
        for thr in g["thresholds"].tolist():
            key = f"r{radix}_t{thr}"
            for qi, q in enumerate(g["needles"].tolist()):
                want = [(m, d, 0, f, 1) for m, d, f in _golden_rows(g, key + "_frame", qi)]
                assert vorc.find_frame(entries, q, thr, -1, radix=radix) == want, (key, qi)
                n_rows += len(want)
                # the raw matches: RadixMap::indexOf (radix.h:135-141) + hamm64 < threshold, restated in numpy
                mask = np.uint64((1 << radix) - 1)
                same = ((eh >> np.uint64(1)) & mask) == ((np.uint64(q) >> np.uint64(1)) & mask)
                d = np.array([bin(int(x) ^ q).count("1") for x in eh[same]], np.int64)
                hit = d < thr
                got = sorted(zip(ev[same][hit].tolist(), ef[same][hit].tolist(), d[hit].tolist()))
                assert got == _golden_rows(g, key + "_raw", qi), (key, qi)
    assert n_rows > 500
    # radix 10 must actually hide matches that radix 0 finds, or the fixture pins nothing about buckets
    assert len(g["r10_t6_raw"]) < len(g["r0_t6_raw"])


@pytest.mark.gpu
def test_gpu_find_frame_equals_radixmap_golden(gpu, reduce_path):
    """a7 on the GPU box without any checker library: DctVideoIndex(radix_compat) findFrame == the real RadixMap's
    committed answers, radix 0 (exact: also the default index) and radix 10"""
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

    g, clips = _radix_golden()

    class M:
        pass

    media = []
    for mid, (f, h) in zip(g["media_ids"].tolist(), clips):
        m = M()
        m.id, m.path, m.videoIndex, m.dctHash = mid, f"v{mid}", VideoIndex(f.tolist(), [int(x) for x in h]), 0
        media.append(m)
    needle = M()
    needle.id, needle.path, needle.videoIndex = 1, "needle", VideoIndex([], [])
    n_rows = 0
    for radix in g["radixes"].tolist():
        for compat in ((True, False) if radix == 0 else (True,)):
            idx = DctVideoIndex(radix_compat=compat)
            idx.add(media)
            for thr in g["thresholds"].tolist():
                p = VideoSearchParams(dctThresh=thr, skipFrames=0, minFramesMatched=1, minFramesNear=1, videoRadix=radix)
                key = f"r{radix}_t{thr}_frame"
                for qi, q in enumerate(g["needles"].tolist()):
                    needle.dctHash = q
                    got = [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in idx.findFrame(needle, p)]
                    want = [(m, d, 0, f, 1) for m, d, f in _golden_rows(g, key, qi)]
                    assert got == want, (radix, compat, thr, qi)
                    n_rows += len(want)
    assert n_rows > 700
