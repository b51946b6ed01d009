"""CvFeaturesIndex (src/cvfeaturesindex.cpp:438-604): exact 256-bit knn + median scoring.  The reference's
candidate generator is OpenCV FLANN LSH (approximate, third-party, absent here): parity is against the exact
brute-force statement in the oracle; the scoring rules are checked on hand-made cases."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def cvo():
    from oracle import CvOracle

    return CvOracle()


def make_descriptors(n_media, per_media, seed, planted=0.3, max_flip=30):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, 256, (n_media * per_media, 32), dtype=np.uint8)
    n = len(rows)
    for i in rng.choice(np.arange(1, n), int(n * planted), replace=False).tolist():
        src = int(rng.integers(0, i))
        rows[i] = rows[src]
        for b in rng.choice(256, int(rng.integers(0, max_flip)), replace=False).tolist():
            rows[i, b >> 3] ^= np.uint8(1 << (b & 7))
    first = np.arange(0, n, per_media, dtype=np.uint32)
    ids = np.arange(1, n_media + 1, dtype=np.uint32)
    return rows, first, ids


def test_score_rules(cvo):
    """cvfeaturesindex.cpp:564-596: median of the vote distances * 1000 / votes (integer arithmetic)"""
    z = np.zeros((1, 32), np.uint8)

    def flip(k):
        r = z.copy()
        for b in range(k):
            r[0, b >> 3] |= 1 << (b & 7)
        return r

    rows = np.concatenate([flip(3), flip(5), flip(10), flip(2), flip(200)])  # media 1: 3 rows, media 2: 2 rows
    first, ids = np.array([0, 3], np.uint32), np.array([7, 9], np.uint32)
    i, s = cvo.find(rows, first, ids, z, 10, 25)
    # media 7: distances [3,5,10] -> median 5 -> 5*1000/3 = 1666 ; media 9: [2] -> 2*1000/1
    assert i.tolist() == [7, 9] and s.tolist() == [1666, 2000]
    i, s = cvo.find(rows, first, ids, z, 10, 6)  # only 3,5 and 2 under the threshold: even count -> (3+5)/2
    assert i.tolist() == [7, 9] and s.tolist() == [4 * 1000 // 2, 2000]
    i, s = cvo.find(rows, first, np.array([0, 9], np.uint32), z, 10, 25)  # media 7 removed -> id 0 skipped
    assert i.tolist() == [9]
    # knn cut happens BEFORE the removed-media skip: with k=1 the nearest row (media 9) wins alone
    i, s = cvo.find(rows, first, ids, z, 1, 25)
    assert i.tolist() == [9] and s.tolist() == [2000]


def test_knn_oracle_against_numpy(cvo):
    rows, _, _ = make_descriptors(20, 30, 1)
    needles = rows[::41].copy()
    needles[0, 0] ^= 1
    r, d, c = cvo.knn(rows, needles, 4, 40)
    bits = np.unpackbits(rows, axis=1).astype(np.int32)
    nb = np.unpackbits(needles, axis=1).astype(np.int32)
    for q in range(len(needles)):
        dist = (bits != nb[q]).sum(1)
        order = np.lexsort((np.arange(len(rows)), dist))
        order = order[dist[order] < 40]
        assert c[q] == len(order)
        m = min(4, len(order))
        assert r[q, :m].tolist() == order[:m].tolist() and d[q, :m].tolist() == dist[order[:m]].tolist()


class _M:
    def __init__(self, id_, desc, path=""):
        self.id, self.keyPointDescriptors, self.path = id_, desc, path


@pytest.mark.gpu
@pytest.mark.parametrize("n_media,per_media", [(7, 5), (300, 100), (41, 500)])
def test_gpu_knn_and_find_vs_oracle(gpu, cvo, scan256_path, n_media, per_media):
    from cbird_amd.cvfeatures import CvFeaturesIndex

    rows, first, ids = make_descriptors(n_media, per_media, n_media + per_media)
    idx = CvFeaturesIndex()
    media = [_M(int(ids[i]), rows[i * per_media:(i + 1) * per_media]) for i in range(n_media)]
    idx.add(media[: n_media // 2])
    idx.add(media[n_media // 2:])
    assert idx.count() == len(rows) and idx.isLoaded() and idx.memoryUsage() == 2 * 32 * len(rows)
    for k, thr in ((10, 25), (4, 25), (10, 60), (3, 1), (10, 130), (2, 257)):
        needles = rows[:: max(1, len(rows) // 97)]
        gr, gd, gc = idx.knn(needles, k, thr)
        wr, wd, wc = cvo.knn(rows, needles, k, thr)
        assert (gc == wc).all(), (k, thr)
        for q in range(len(needles)):
            m = min(k, int(wc[q]))
            assert gr[q, :m].tolist() == wr[q, :m].tolist() and gd[q, :m].tolist() == wd[q, :m].tolist()
    p = gpu.SearchParams(cvThresh=25)
    for m in media[:: max(1, n_media // 9)]:
        got = idx.find(m, p)
        wi, ws = cvo.find(rows, first, ids, m.keyPointDescriptors, 10, 25)
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
    # removal keeps the rows (they still take knn places) but the media no longer votes
    victims = [int(ids[1]), int(ids[n_media - 1])]
    idx.remove(victims)
    assert idx.count() == len(rows)
    ids2 = ids.copy()
    ids2[np.isin(ids, victims)] = 0
    for m in media[:: max(1, n_media // 5)]:
        got = idx.find(m, p)
        wi, ws = cvo.find(rows, first, ids2, m.keyPointDescriptors, 10, 25)
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
    # needle without descriptors -> taken from the index by id (cvfeaturesindex.cpp:443)
    got = idx.find(_M(int(ids[0]), None), p)
    wi, ws = cvo.find(rows, first, ids2, rows[:per_media], 10, 25)
    assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
    res = idx.find_batch(media[:6], p)
    for m, r in zip(media[:6], res):
        assert [(x.mediaId, x.score) for x in r] == [(x.mediaId, x.score) for x in idx.find(m, p)]


@pytest.mark.gpu
@pytest.mark.parametrize("n_train", [300, 6000])
def test_gpu_radius_match_vs_oracle(gpu, cvo, scan256_path, n_train):
    """cv::BFMatcher(NORM_HAMMING).radiusMatch as TemplateMatcher uses it (src/templatematcher.cpp:134,217): every
    train row within max_dist (inclusive), per query in ascending (distance, trainIdx) order.  Expected lists: the
    oracle's exact knn with k = all rows and thresh = max_dist + 1."""
    from cbird_amd.cvfeatures import CvFeaturesIndex

    rows, _, _ = make_descriptors(n_train // 100, 100, 7)
    rng = np.random.default_rng(n_train)
    queries = rows[rng.choice(len(rows), 150, replace=False)].copy()
    flips = rng.integers(0, 40, len(queries))
    for q, f in zip(queries, flips):  # planted neighbours at distances 0..39
        for b in rng.choice(256, int(f), replace=False):
            q[b >> 3] ^= 1 << (b & 7)
    idx = CvFeaturesIndex()
    idx.add([_M(1, rows)])
    for max_dist in (0, 25, 60, 110, 256):
        got, first = idx.radius_match(queries, max_dist)
        k = len(rows) if max_dist >= 110 else 64
        wr, wd, wc = cvo.knn(rows, queries, k, max_dist + 1)
        assert first[-1] == len(got) == int(wc.sum()), max_dist
        for q in range(len(queries)):
            g = got[first[q]: first[q + 1]]
            assert len(g) == wc[q] <= k
            assert (g[:, 0] == q).all()
            assert g[:, 1].tolist() == wr[q, : wc[q]].tolist() and g[:, 2].tolist() == wd[q, : wc[q]].tolist()
        if max_dist == 256:
            assert len(got) == len(queries) * len(rows)
    e, f = CvFeaturesIndex().radius_match(queries, 25)  # empty train set
    assert len(e) == 0 and (f == 0).all()
