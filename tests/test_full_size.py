"""BASELINE.json's configurations at their FULL sizes on the MI355X, checked through size-independent properties
(the oracle finishes only small cases in seconds; those are the other test modules):

  configs[1]  1M-entry DctHashIndex, all-pairs find: self-match, monotone in the threshold, symmetry of the match
              relation, equality with the oracle on a needle sample, cut == prefix of the full list
  configs[3]  100k images x 500 descriptors (5e7 rows, 1.6 GB) CvFeaturesIndex: self rows at distance 0, k = 4 is a
              prefix of k = 10, planted neighbours found at their exact distance, oracle equality on a row window
  configs[4]  10k clips x 300 frame hashes DctVideoIndex: every planted sub-clip finds its source, batch == single,
              sharded-by-video wrapper (world 1) == plain index
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_one_million_hashes_properties(gpu, orc):
    from cbird_amd import synth

    n = 1_000_000
    h, ids = synth.make_hashes(n, seed=1234, planted_frac=0.02, max_dist=8)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    assert idx.count() == n and idx.memoryUsage() == 12 * n
    rng = np.random.default_rng(5)
    prev_counts = None
    for dht in (1, 3, 5, 8):
        gi, gs, gc = idx.find_batch(h, dht, 8)
        assert (gc >= 1).all()                                   # every needle finds itself ...
        assert (gs[:, 0] == 0).all()                             # ... at distance 0
        first_is_self_or_dup = (gi[:, 0] <= ids)                 # ties at distance 0 are ordered by id: self or an
        assert first_is_self_or_dup.all()                        # earlier exact duplicate comes first
        assert ((gs[:, 1:] >= gs[:, :-1]) | (gs[:, 1:] == 0) & (gi[:, 1:] == 0)).all()  # ascending scores (0 pads)
        if prev_counts is not None:
            assert (gc >= prev_counts).all()                     # monotone in the threshold
        prev_counts = gc
        # symmetry on the needles that have a second match: j in matches(i)  =>  i in matches(j)
        multi = np.nonzero(gc >= 2)[0]
        for i in rng.choice(multi, min(200, len(multi)), replace=False).tolist():
            for t in range(min(int(gc[i]), 8)):
                j = int(gi[i, t]) - 1
                if j == i:
                    continue
                back = [x.mediaId for x in idx.find(gpu.Media(id=0, dctHash=int(h[j])), gpu.SearchParams(dctThresh=dht))]
                assert int(ids[i]) in back
        # oracle equality on a sample of needles (full haystack)
        sample = rng.choice(n, 48, replace=False)
        wi, ws, wc = orc.find64_batch(h, ids, h[sample], dht, 8)
        assert (gc[sample] == wc).all() and (gi[sample] == wi).all() and (gs[sample] == ws).all()


def test_config1_every_needle_equals_the_real_vptree(gpu):
    """north_star's acceptance line at full size: ALL 10^6 needles against the 10^6-entry index, every needle's complete
    (mediaId, distance) list equal to the reference's own VP-tree (src/tree/vptree.h compiled in place, oracle/_ref) in
    canonical (distance, mediaId) order -- at dht 2 (BASELINE configs[0]/[1]) for ALL needles, at 5 and 8 for every fourth."""
    import os

    import oracle
    from cbird_amd import synth

    if not oracle.ref_available():
        pytest.skip("oracle/_ref/libcbird_ref.so absent (built by __graft_entry__.build() where /root/reference exists)")
    n = 1_000_000
    h, ids = synth.make_hashes(n, seed=1234, planted_frac=0.02, max_dist=8)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    tree = oracle.RefTree(h, ids)
    threads = len(os.sched_getaffinity(0))
    # dht 2 (the BASELINE threshold): every needle.  dht 5 and 8: the GPU still answers all 10^6 needles, the VP-tree --
    # whose searches at those thresholds are what this test's minute is spent on -- every fourth of them (bench.py's
    # full_identity repeats the dht 2 comparison on every run of the driver)
    for dht, step in ((2, 1), (5, 4), (8, 4)):
        gi, gs, gc = idx.find_batch(h, dht, 8)
        kmax = int(gc.max())
        if kmax > 8:
            gi, gs, gc = idx.find_batch(h, dht, kmax)
        gi, gs, gc = gi[::step], gs[::step], gc[::step]
        keep = np.arange(gi.shape[1])[None, :] < gc[:, None]
        off, ci, cd = tree.search_lists(np.ascontiguousarray(h[::step]), dht, threads=threads)
        assert (np.diff(off.astype(np.int64)) == gc).all()
        assert off[-1] > len(gc)                                 # (the planted neighbours are there)
        assert (gi[keep].astype(np.uint32) == ci).all() and (gs[keep].astype(np.int32) == cd).all()


def test_config3_orb_50m_rows_properties(gpu):
    from cbird_amd import _lib
    from cbird_amd.cvfeatures import CvFeaturesIndex

    L = _lib.lib()
    n_img, per = 100_000, 500
    rng = np.random.default_rng(1234)
    idx = CvFeaturesIndex()
    keep = {}
    chunk = 2000
    for c0 in range(0, n_img, chunk):
        rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
        if c0 == 0:
            # plant neighbours of image 77's descriptors inside image 1500 (same chunk), flipping d bits of descriptor d
            for d in range(0, 24):
                r = rows[76 * per + d].copy()
                for b in range(d):
                    r[b >> 3] ^= np.uint8(1 << (b & 7))
                rows[1499 * per + 100 + d] = r
            keep = {77: rows[76 * per: 77 * per].copy(), 1500: rows[1499 * per: 1500 * per].copy()}
        for i in range(chunk):
            _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
    assert idx.count() == n_img * per and idx.memoryUsage() == 2 * 32 * n_img * per
    needle = keep[77]
    r10, d10, c10 = idx.knn(needle, 10, 25)
    r4, d4, c4 = idx.knn(needle, 4, 25)
    first_row = 76 * per
    assert (c10 >= 1).all() and (d10[:, 0] == 0).all()
    assert (r10[:, 0] == first_row + np.arange(per)).all()        # each descriptor's own row, distance 0
    valid4 = np.arange(4)[None, :] < np.minimum(c4, 4)[:, None]   # (places beyond a row's count are unspecified)
    assert (c4 == c10).all() and (r4[valid4] == r10[:, :4][valid4]).all() and (d4[valid4] == d10[:, :4][valid4]).all()
    # ^ k = 4 is a prefix of k = 10
    for d in range(1, 24):                                         # the planted neighbour at exactly d bits
        assert c10[d] >= 2 and d10[d, 1] == d and r10[d, 1] == 1499 * per + 100 + d
    assert (c10[24:] == 1).all()                                   # random 256-bit rows: nothing else under 25 bits
    assert np.array_equal(idx.descriptorsForMediaId(1500), keep[1500])
    # find(): image 77 votes for itself (500 zero distances -> median 0) and for image 1500 (24 votes, d = 0..23)
    class M:
        pass
    m = M()
    m.id, m.path, m.keyPointDescriptors = 77, "", needle
    res = {x.mediaId: x.score for x in idx.find(m, gpu.SearchParams(cvThresh=25))}
    assert res[77] == 0 and res[1500] == ((11 + 12) // 2) * 1000 // 24 and set(res) == {77, 1500}  # 24 votes, d = 0..23


def test_config4_video_10k_clips_properties(gpu):
    from cbird_amd import synth_video
    from cbird_amd.dist import ShardedDctVideoIndex
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

    n_clips = 10_000
    clips = synth_video.make_clips_fast(n_clips, 300, seed=1234, subclip_frac=0.01, noise_bits=0, max_gap=8)

    class M:
        pass

    media = []
    for i, (f, h) in enumerate(clips):
        m = M()
        m.id, m.path, m.videoIndex, m.dctHash = i + 1, "", VideoIndex(f, h), 0
        media.append(m)
    idx = DctVideoIndex()
    idx.add(media)
    assert idx.count() == n_clips
    p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=30, minFramesNear=60)
    subs = media[-100:]
    batch = idx.find_videos_batch(subs + media[:100], p)
    key = lambda r: [(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
    found_source = 0
    for m, r in zip(subs, batch[:100]):
        # an exact sub-clip (>= 150 frames of its source): the source matches with nearly every frame adjacent (the
        # adjacency walk starts from frame 0, dctvideoindex.cpp:603-613, and repeated hashes tie to the earlier frame:
        # a few percent are not) -> score = 100 - percentNear <= 10, and the range starts at the needle's first frame
        best = [x for x in r if x.score <= 10 and x.range.srcIn == 0 and x.range.len >= 100]
        found_source += bool(best)
        assert all(x.mediaId != m.id for x in r)                  # filterSelf
    assert found_source == 100
    for m, r in list(zip(subs, batch[:100]))[::10]:               # batch == single
        assert key(idx.findVideo(m, p)) == key(r)
    sv = ShardedDctVideoIndex(DctVideoIndex)                      # world 1: the sharded wrapper is the plain index
    sv.add(media)
    assert [key(r) for r in sv.find_videos_batch(subs[:20], p)] == [key(r) for r in batch[:20]]
