"""The reference's own index test, end to end on the device: unit/testindexbase.cpp scans a data set of 40 images in
5 sizes with the Scanner, adds the results to the index under test and checks `similar` / `similarTo` / remove / re-add
(baseTestLoad :112-146, baseTestAddRemove :148-218; one subclass per index: testdcthashindex.cpp, testdctfeaturesindex.cpp,
testcvfeaturesindex.cpp, testcolordescindex.cpp, testdctvideoindex.cpp).

Here the "scanner" is cbh_index_images (gray -> autocrop -> dctHash64, sizeLongestSide + ORB + keypoint hashes,
ColorDescriptor::create) and cbh_vindexer (video), the indexes are the five device indexes, `similar` is
cbird_amd.database.  The data set is synthetic (40 scenes x 5 sizes; cbird's test images are not redistributable): this
is a behavioural test of the whole path, the bit-level parity lives in the per-stage tests."""
import copy

import numpy as np
import pytest

SIZES = (1.0, 0.875, 0.75, 0.625, 0.5)
N_SCENES = 40


def scene(seed, h=384, w=512):
    """a photo-like colour image: smooth illumination, a few dozen textured shapes, a scene-specific palette"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    palette = rng.integers(30, 256, (6, 3)).astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    base = palette[0] * 0.6
    img += base
    img += (40 * np.sin(xx / rng.uniform(60, 200) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(60, 200)))[..., None]
    for _ in range(28):
        x0, y0 = int(rng.integers(0, w - 24)), int(rng.integers(0, h - 24))
        ww, hh = int(rng.integers(16, w // 3)), int(rng.integers(16, h // 3))
        col = palette[int(rng.integers(1, 6))] * rng.uniform(0.6, 1.0)
        if rng.random() < 0.5:
            img[y0:y0 + hh, x0:x0 + ww] = col
        else:  # a disc
            cy, cx, r = y0 + hh / 2, x0 + ww / 2, min(ww, hh) / 2
            img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = col
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def resized(img, f):
    """a smaller copy (bilinear on a 2x box-prefiltered image: what an image editor's "resize" roughly does)"""
    from scipy import ndimage

    if f == 1.0:
        return img
    h, w = img.shape[:2]
    nh, nw = int(round(h * f)), int(round(w * f))
    out = ndimage.zoom(ndimage.uniform_filter(img.astype(np.float32), size=(2, 2, 1)), (nh / h, nw / w, 1), order=1)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


class M:
    """the slice of cbird's Media the indexes read"""

    def __init__(self, id_, path):
        self.id, self.path, self.score = id_, path, -1
        self.dctHash = 0
        self.keyPointHashes = []
        self.keyPointDescriptors = np.zeros((0, 32), np.uint8)
        self.colorDescriptor = None
        self.videoIndex = None

    def isValid(self):
        return self.id != 0


@pytest.fixture(scope="module")
def corpus(gpu):
    """Scanner::scanDirectory over 40 images x 5 sizes: one cbh_index_images call per size (one geometry each)"""
    from cbird_amd import orb
    from cbird_amd.scanner import IndexParams, process_images

    orb.set_pattern(orb.synthetic_pattern())
    scenes = [scene(1000 + s) for s in range(N_SCENES)]
    media = []
    for si, f in enumerate(SIZES):
        batch = np.stack([resized(s, f) for s in scenes])
        res = process_images(batch, IndexParams(algos=15))
        for s, r in enumerate(res):
            m = M(1 + s * len(SIZES) + si, f"/data/40x5/{s:02d}_{int(f * 1000):04d}.png")
            m.dctHash = int(r.dctHash)
            m.keyPointHashes = [int(x) for x in r.keyPointHashes]
            m.keyPointDescriptors = np.ascontiguousarray(r.keyPointDescriptors)
            m.colorDescriptor = r.colorDescriptor
            media.append(m)
    media.sort(key=lambda m: m.id)
    return media


def scene_of(m):
    return (m.id - 1) // len(SIZES)


def make_index(algo):
    from cbird_amd import DctFeaturesIndex, DctHashIndex, SearchParams
    from cbird_amd.colordesc import ColorDescIndex
    from cbird_amd.cvfeatures import CvFeaturesIndex

    return {SearchParams.AlgoDCT: DctHashIndex, SearchParams.AlgoDCTFeatures: DctFeaturesIndex,
            SearchParams.AlgoCVFeatures: CvFeaturesIndex, SearchParams.AlgoColor: ColorDescIndex}[algo]()


def params_for(algo):
    from cbird_amd import SearchParams

    # the reference's defaults (dctThresh 5, cvThresh 25, minMatches 1, maxMatches 5) and, as in every
    # unit/test*index.cpp, filterSelf = false: an indexed image is expected to find at least itself
    return SearchParams(algo=algo, filterSelf=False)


def populate(index, algo, media):
    """Database::similar's first step: Index::load from the database columns (add() on an index that was never loaded
    is a no-op in the reference: the database is the store)"""
    if algo == 0:
        index.load([m.dctHash for m in media], [m.id for m in media])
    elif algo == 1:
        index.load([(m.id, m.keyPointHashes) for m in media])
    else:
        index.add(media)


def check_defaults(index):
    """baseTestDefaults (:57-62)"""
    assert not index.isLoaded() and index.count() == 0 and index.memoryUsage() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_load_similar_and_similar_to(corpus, algo):
    """baseTestLoad (:112-146): `similar` finds at least the 40 groups, every group has more than its needle, and every
    image -- processed again as the scanner would -- finds more than itself"""
    from cbird_amd.database import search_index, similar

    index = make_index(algo)
    check_defaults(index)
    populate(index, algo, corpus)
    assert index.isLoaded() and index.count() > 0 and index.memoryUsage() > 0
    p = params_for(algo)
    groups = similar(index, corpus, p)
    assert len(groups) >= N_SCENES
    assert all(len(g) > 1 for g in groups)
    id_map = {m.id: m for m in corpus}
    others = same = wrong = 0
    for m in corpus:
        g = search_index(index, copy.copy(m), p, id_map)
        assert len(g) >= 1 and m.id in [x.id for x in g], (algo, m.path)  # "each image should at least match itself"
        rest = [x for x in g if x.id != m.id]
        others += bool(rest)
        same += sum(1 for x in rest if scene_of(x) == scene_of(m))
        wrong += sum(1 for x in rest if scene_of(x) != scene_of(m))
    print(f"algo {algo}: {others} of {len(corpus)} images found another image; same-scene matches {same}, other-scene {wrong}")
    # not asserted by the reference's test, but what the data set is for: the other sizes of a picture are found
    assert others >= 0.9 * len(corpus) and same > 4 * wrong


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_add_remove(corpus, algo):
    """baseTestAddRemove (:148-218): remove one member of three groups, they vanish from the results of the others;
    re-add them (processed again), the groups are back as they were"""
    from cbird_amd.database import search_index, similar

    index = make_index(algo)
    populate(index, algo, corpus)
    p = params_for(algo)
    id_map = {m.id: m for m in corpus}
    before = similar(index, corpus, p)
    assert len(before) >= N_SCENES
    queried = [before[0][0], before[1][0], before[2][0]]
    b = [search_index(index, copy.copy(q), p, id_map) for q in queried]
    for q, g in zip(queried, b):
        assert q.id in [x.id for x in g]  # present in the results as expected (:171-176)
    index.remove([q.id for q in queried])
    left = {k: v for k, v in id_map.items() if k not in {q.id for q in queried}}
    for q in queried:  # the removed item, processed again, no longer finds itself (:196-201)
        assert q.id not in [x.id for x in search_index(index, copy.copy(q), p, left)]
    index.add(queried)
    after = [search_index(index, copy.copy(q), p, id_map) for q in queried]
    for x, y in zip(b, after):  # Media::groupCompareByContents (media.cpp:276-292): same count, same paths
        assert sorted(m.path for m in x) == sorted(m.path for m in y)


@pytest.mark.gpu
def test_video_index_end_to_end(gpu, tmp_path):
    """testdctvideoindex.cpp in the same spirit: clips indexed by the streaming indexer (Scanner::processVideo ->
    Media::makeVideoIndex), saved as .vdx, loaded into DctVideoIndex; a re-encoded copy (noise, another size) of a clip
    finds its original, a frame grabbed from a clip finds the clip (image -> video search), removal and re-adding work"""
    from test_video_indexer import clip

    from cbird_amd.hashing import dct_hash64_batch
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoIndexer, VideoSearchParams

    def index_clip(frames):
        ix = VideoIndexer(threshold=8)
        for i in range(0, len(frames), 32):
            ix.push(frames[i:i + 32])
        return ix.finish()

    clips = [clip(500 + k, 120, 180, 240, (20, 20, 0, 0) if k % 2 else (0, 0, 0, 0), cut_every=6) for k in range(8)]
    media = []
    for k, c in enumerate(clips):
        m = M(k + 1, f"/data/video/{k}.mp4")
        m.videoIndex = index_clip(c)
        assert 10 < len(m.videoIndex.frames) < 120 and m.videoIndex.frames[0] == 0 and m.videoIndex.frames[-1] == 119
        m.videoIndex.save(str(tmp_path / f"{m.id}.vdx"))
        media.append(m)
    idx = DctVideoIndex(0, str(tmp_path))
    assert not idx.isLoaded() and idx.count() == 0
    idx.load([m.id for m in media])
    assert idx.isLoaded() and idx.count() == 8
    p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=5, minFramesNear=10)
    rng = np.random.default_rng(9)
    for k in (0, 3, 5):
        # a "re-encode": the same frames with noise; the middle 80 frames only (a cut of the original)
        noisy = np.clip(clips[k][20:100].astype(np.int16) + rng.integers(-3, 4, clips[k][20:100].shape), 0, 255)
        needle = M(0, "/data/needle.mp4")
        needle.videoIndex = index_clip(noisy.astype(np.uint8))
        found = idx.find(needle, p)
        assert found and found[0].mediaId == k + 1, k
        assert abs(found[0].range.dstIn - 20) <= 8  # where in the original the needle starts
        # image -> video: a single frame of the clip
        frame = M(0, "/data/frame.png")
        frame.dctHash = int(dct_hash64_batch(clips[k][40:41, 20 if k % 2 else 0:160 if k % 2 else 180])[0])
        hit = idx.find(frame, p)
        assert hit and hit[0].mediaId == k + 1
    idx.remove([4])
    needle = M(0, "/data/needle.mp4")
    needle.videoIndex = index_clip(clips[3][10:110])
    assert all(x.mediaId != 4 for x in idx.find(needle, p))
    idx.add([media[3]])
    assert idx.find(needle, p)[0].mediaId == 4


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [0, 1, 2, 3])
def test_search_index_batch_equals_one_search_index_per_needle(corpus, algo):
    """Database::searchIndex for the whole needle list behind the C-ABI (cbh_search_index_batch and the four
    cbh_*_search_index_batch of searchbatch.hip: one batched find per threshold level) against the reference's shape,
    one searchIndex per needle (cbird_amd.database.search_index over the same index): maxThresh escalation (+1 for the
    dct-based algorithms, +5 for ORB, none for colour), the order, filterSelf, the maxMatches cut, and ids the idMap
    does not hold"""
    import warnings

    from cbird_amd import SearchParams
    from cbird_amd.database import search_index, search_index_batch

    index = make_index(algo)
    populate(index, algo, corpus)
    id_map = {m.id: m for m in corpus if m.id % 9 != 0}  # every 9th media is not in the caller's idMap
    needles = corpus[::3] + corpus[1:40:7]
    variants = [dict(filterSelf=False), dict(filterSelf=True, maxMatches=3), dict(filterSelf=True, minMatches=3, maxThresh=9),
                dict(dctThresh=1, cvThresh=5, minMatches=2, maxThresh=30, maxMatches=7),
                dict(dctThresh=2, cvThresh=10, minMatches=1, maxThresh=4)]
    for kw in variants:
        p = SearchParams(algo=algo, **kw)
        got = search_index_batch(index, needles, p, id_map)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "no media with id"
            want = [search_index(index, m, p, id_map) for m in needles]
        assert [[(x.id, x.score) for x in g] for g in got] == [[(x.id, x.score) for x in g] for g in want], (algo, kw)
        assert any(len(g) for g in got)
