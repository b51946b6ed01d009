"""ColorDescIndex / ColorDescriptor::distance (src/cvutil.cpp:682-749, src/colordescindex.cpp:250-278): the
oracle against an independent numpy float32 statement and hand-made cases; the GPU bit-exact vs the oracle
(scores are int(distance); raw float distances are compared at 1e-5 as the north star states)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def co():
    from oracle import ColorOracle

    return ColorOracle()


def synth_descriptors(n, seed, dup_frac=0.3):
    from cbird_amd.colordesc import COLOR_DTYPE

    rng = np.random.default_rng(seed)
    d = np.zeros(n, COLOR_DTYPE)
    num = rng.integers(0, 33, n)
    num[rng.random(n) < 0.05] = 0  # grayscale images: stored with no colours (colordescindex.cpp:73-75)
    for i in range(n):
        if i and rng.random() < dup_frac:  # near-duplicate palette of an earlier entry
            src = int(rng.integers(0, i))
            d[i] = d[src]
            k = int(d[i]["numColors"])
            if k:
                jit = rng.integers(-600, 601, (k, 4))
                d[i]["colors"][:k] = np.clip(d[i]["colors"][:k].astype(np.int64) + jit, 0, 65535)
                drop = int(rng.integers(0, 3))
                d[i]["numColors"] = max(0, k - drop)
        else:
            k = int(num[i])
            d[i]["colors"][:k] = rng.integers(0, 65536, (k, 4))
            d[i]["numColors"] = k
    ids = np.arange(1, n + 1, dtype=np.uint32)
    return d, ids


def np_distance(a, b):
    f = np.float32
    na, nb = int(a["numColors"]), int(b["numColors"])
    if na == 0 or nb == 0 or abs(na - nb) > 2:
        return np.finfo(f).max
    if na < nb:
        a, b = b, a
        na, nb = nb, na

    def dec(x, k):
        c = x["colors"][:k].astype(f)
        return (c[:, 0] * f(100.0) / f(65535), c[:, 1] * f(354.0) / f(65535) - f(134.0),
                c[:, 2] * f(262.0) / f(65535) - f(140.0))

    l1, u1, v1 = dec(a, na)
    l2, u2, v2 = dec(b, nb)
    dl = l1[:, None] - l2[None, :]
    du = u1[:, None] - u2[None, :]
    dv = v1[:, None] - v2[None, :]
    dist = np.sqrt((dl * dl + du * du) + dv * dv, dtype=f)
    score = f(1)
    for x in dist.min(axis=1):
        score = f(score + x)
    return score


def test_oracle_against_numpy_float32(co):
    d, _ = synth_descriptors(300, 1)
    rng = np.random.default_rng(2)
    n_finite = 0
    for _ in range(1500):
        i, j = rng.integers(0, 300, 2)
        want = np_distance(d[i], d[j])
        got = co.distance(d[i], d[j])
        assert got == want, (i, j, got, want)
        n_finite += want < 1e30
    assert n_finite > 100


def test_hand_cases(co):
    from cbird_amd.colordesc import make_descriptor

    a = make_descriptor([[0, 0, 0, 1], [65535, 0, 0, 1]])  # L = 0 and 100, u = -134, v = -140
    b = make_descriptor([[0, 0, 0, 9]])
    assert co.distance(a, b) == pytest.approx(1 + 0 + 100.0, abs=1e-4)  # a has more colours: 2 terms
    assert co.distance(b, a) == co.distance(a, b)  # sides swap, same value
    assert co.distance(a, make_descriptor([])) > 1e38  # no colours -> FLT_MAX
    c5 = make_descriptor(np.zeros((5, 4)))
    assert co.distance(a, c5) > 1e38  # counts differ by 3
    assert co.distance(a, a) == 1.0
    # numColors is what counts, not the stored colours
    assert co.distance(make_descriptor([[0, 0, 0, 0]] * 3, num_colors=1), b) == 1.0


class _M:
    def __init__(self, id_, desc=None):
        self.id, self.colorDescriptor, self.path = id_, desc, f"m{id_}"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 255, 4097])
def test_gpu_find_bit_exact(gpu, co, n):
    from cbird_amd.colordesc import ColorDescIndex

    d, ids = synth_descriptors(n, n)
    idx = ColorDescIndex()
    assert not idx.isLoaded() and idx.memoryUsage() == 0
    media = [_M(int(i), x) for i, x in zip(ids, d)]
    idx.add(media[: n // 2])
    idx.add(media[n // 2:])
    assert idx.count() == n and idx.memoryUsage() == (258 + 4) * n  # unit/testcolordescindex.cpp:25-29
    total = 0
    for m in media[:: max(1, n // 60)]:
        got = idx.find(m)
        wi, ws = co.find(d, ids, m.colorDescriptor)
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist(), m.id
        total += len(got)
    if n > 100:
        assert total > 50
    victims = [int(ids[0]), int(ids[n // 2])]
    idx.remove(victims)
    d2, ids2 = d.copy(), ids.copy()
    for v in victims:
        d2[v - 1] = np.zeros((), d.dtype)
        ids2[v - 1] = 0
    for m in media[1:: max(1, n // 20)]:
        got = idx.find(m)
        wi, ws = co.find(d2, ids2, m.colorDescriptor)
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
    # needle without descriptor -> findIndexData (colordescindex.cpp:256-262)
    if n > 10:
        m = _M(int(ids[5]))
        got = idx.find(m)
        wi, ws = co.find(d2, ids2, d2[5])
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
        with pytest.warns(UserWarning):
            assert idx.find(_M(999999)) == []
    # batch: sorted by (score, id), cut at k
    qs = d[: min(n, 16)]
    gi, gs, gc = idx.find_batch(qs, 5)
    for q in range(len(qs)):
        wi, ws = co.find(d2, ids2, qs[q])
        order = np.lexsort((wi, ws))
        assert gc[q] == len(wi)
        m_ = min(5, len(wi))
        assert gi[q, :m_].tolist() == wi[order][:m_].tolist() and gs[q, :m_].tolist() == ws[order][:m_].tolist()


@pytest.mark.gpu
def test_gpu_scores_of_single_finds(gpu, co):
    """k_color_dist2 (packed f32, what find() runs); k_color_dist3 -- the kernel that also returns the raw floats -- is
    covered by test_gpu_float_distances_are_the_references_bit_for_bit"""
    from cbird_amd.colordesc import ColorDescIndex

    d, ids = synth_descriptors(3000, 31)
    idx = ColorDescIndex()
    idx.add([_M(int(i), x) for i, x in zip(ids, d)])
    for q in (0, 7, 1500, 2999):
        got = idx.find(_M(0, d[q]))
        wi, ws = co.find(d, ids, d[q])
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()


@pytest.mark.gpu
def test_gpu_float_distances_are_the_references_bit_for_bit(gpu, co):
    """the raw float distances (north_star: float colour distances within 1e-5): the VALU kernel keeps the reference's
    operation order, so the comparison is BITWISE against the oracle's ColorDescriptor::distance
    (src/cvutil.cpp:682-749); (int) of them are the scores"""
    from cbird_amd.colordesc import ColorDescIndex

    n = 700
    d, ids = synth_descriptors(n, 5)
    d["numColors"][3] = 0       # no colours -> FLT_MAX
    d["numColors"][4] = 32      # counts differing by more than 2 from most -> FLT_MAX
    idx = ColorDescIndex()
    idx.add([_M(int(i), x) for i, x in zip(ids, d)])
    needles = d[[0, 1, 3, 4, 250, 699]]
    got = idx.distances(needles)
    want = np.array([[co.distance(q, x) for x in d] for q in needles], np.float32)
    assert got.shape == want.shape == (6, n)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    finite = want < 1e38
    assert finite.sum() > 300 and (~finite).sum() > n  # both branches present
    rel = np.abs(got[finite] - want[finite]) / want[finite]
    assert rel.max() <= 1e-5  # (trivially: they are identical)
    # and the int scores are exactly (int) of these floats
    for qi, q in enumerate(needles):
        sc = {x.mediaId: x.score for x in idx.find(_M(0, q))}
        for i in np.nonzero(finite[qi])[0]:
            assert sc[int(ids[i])] == int(got[qi, i])
