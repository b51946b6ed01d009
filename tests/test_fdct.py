"""DctFeaturesIndex::find (src/dctfeaturesindex.cpp:260-358): oracle pinned to golden vectors from the
real HammingTree; GPU path bit-exact vs oracle and golden."""
import numpy as np
import pytest

from conftest import load_golden


def _cases(g):
    no, ro = g["needle_offs"], g["res_offs"]
    for c in range(len(g["needle_ids"])):
        yield (g["needle_hashes"][no[c]:no[c + 1]], int(g["needle_ids"][c]), int(g["thresh"][c]),
               g["res_ids"][ro[c]:ro[c + 1]], g["res_scores"][ro[c]:ro[c + 1]])


def test_oracle_matches_real_hammingtree_golden(orc):
    g = load_golden("fdct_single_leaf.npz")
    n = 0
    for nh, nid, thr, wi, ws in _cases(g):
        gi, gs = orc.fdct_find(g["hashes"], g["ids"], nh, nid, thr)
        assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist(), (nid, thr)
        n += 1
    assert n > 50


def test_oracle_vs_real_tree_live(orc):
    import oracle

    if not oracle.ref_qt_available():
        pytest.skip("oracle/_ref/libcbird_ref_qt.so not built")
    from cbird_amd import synth

    m, k = 40, 150
    h, _ = synth.make_hashes(m * k, seed=77, planted_frac=0.3, max_dist=6)
    ids = np.repeat(np.arange(1, m + 1, dtype=np.uint32), k)
    t = oracle.RefHammingTree()
    t.insert(ids, h)
    assert t.size() == m * k
    # raw candidates: same multiset as the brute-force predicate, ascending distance
    for x in h[::397].tolist():
        ri, rh, rd = t.search(x, 6)
        assert (np.diff(rd) >= 0).all()
        oi, od = orc.scan64(h, ids, x, 6)
        assert sorted(zip(rd.tolist(), ri.tolist())) == sorted(zip(od.tolist(), oi.tolist()))
    for needle in (1, 7, 40):
        nh = h[ids == needle][:50]
        a = t.fdct_find(nh, needle, 5)
        b = orc.fdct_find(h, ids, nh, needle, 5)
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_score_rules(orc):
    """dctfeaturesindex.cpp:334-355 on a hand-made index"""
    A, B, C = 0xF0F0F0F0F0F0F0F0, 0x0F0F0F0F0F0F0F0E, 0x123456789ABCDEF0
    hashes = np.array([A, A ^ 2, B, B ^ 4, C], np.uint64)
    ids = np.array([1, 2, 1, 2, 3], np.uint32)
    # needle = media 1 with hashes A,B: media 2 matches both at distance 1 -> maxMatches=2
    i, s = orc.fdct_find(hashes, ids, np.array([A, B], np.uint64), 1, 3)
    assert i.tolist() == [1, 2] and s.tolist() == [-1, 0]  # self -> -1; 2 votes -> 2-2 = 0
    # only one vote for the other media -> 10 * average distance
    i, s = orc.fdct_find(hashes, ids, np.array([A], np.uint64), 1, 3)
    assert i.tolist() == [1, 2] and s.tolist() == [-1, 10]
    # unknown needle id (0): nothing is "self"
    i, s = orc.fdct_find(hashes, ids, np.array([A, B], np.uint64), 0, 3)
    assert i.tolist() == [1, 2] and s.tolist() == [0, 0]
    # removed entries (id 0) take part in the 10-cut but never vote
    ids2 = ids.copy()
    ids2[1] = 0
    i, s = orc.fdct_find(hashes, ids2, np.array([A], np.uint64), 1, 3)
    assert i.tolist() == [1] and s.tolist() == [-1]


@pytest.mark.gpu
def test_gpu_matches_golden_and_oracle(gpu, orc, scan_path, reduce_path):
    g = load_golden("fdct_single_leaf.npz")
    idx = gpu.DctFeaturesIndex()
    h, ids = g["hashes"], g["ids"]
    # build the way cbird does: load, then remove (ids of removed media are already 0 in the fixture,
    # so load the original ids through add + remove)
    idx.load([(int(i), [int(x)]) for i, x in zip(ids.tolist(), h.tolist())])
    assert idx.count() == len(h)
    p = gpu.SearchParams()
    for nh, nid, thr, wi, ws in _cases(g):
        p.dctThresh = thr
        got = idx.find(gpu.Media(id=nid, keyPointHashes=nh.tolist()), p)
        assert [m.mediaId for m in got] == wi.tolist() and [m.score for m in got] == ws.tolist(), (nid, thr)


@pytest.mark.gpu
def test_gpu_add_remove_findindex_batch(gpu, orc, scan_path, reduce_path):
    from cbird_amd import synth

    m, k = 300, 120  # 36k entries: beyond a single reference leaf, exact vs oracle
    h, _ = synth.make_hashes(m * k, seed=5, planted_frac=0.4, max_dist=7)
    ids = np.repeat(np.arange(1, m + 1, dtype=np.uint32), k)
    media = [gpu.Media(id=i, keyPointHashes=h[ids == i].tolist()) for i in range(1, m + 1)]
    idx = gpu.DctFeaturesIndex()
    idx.load([])
    idx.add(media[:200])
    idx.add(media[200:])
    assert idx.count() == m * k and idx.isLoaded()
    idx.remove([3, 9])
    assert idx.count() == m * k  # size() keeps removed values
    ids_after = ids.copy()
    ids_after[np.isin(ids, [3, 9])] = 0
    p = gpu.SearchParams(dctThresh=6)
    for nd in media[:40:3]:
        got = idx.find(nd, p)
        wi, ws = orc.fdct_find(h, ids_after, np.array(nd.keyPointHashes, np.uint64), nd.id, 6)
        assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist(), nd.id
    # needle without hashes: taken from the index by id (dctfeaturesindex.cpp:270-276)
    got = idx.find(gpu.Media(id=11), p)
    wi, ws = orc.fdct_find(h, ids_after, h[ids == 11], 11, 6)
    assert [x.mediaId for x in got] == wi.tolist() and [x.score for x in got] == ws.tolist()
    with pytest.warns(UserWarning):
        assert idx.find(gpu.Media(id=0), p) == []
    # batch == singles
    res = idx.find_batch(media[50:80], p)
    for nd, r in zip(media[50:80], res):
        single = idx.find(nd, p)
        assert [(x.mediaId, x.score) for x in r] == [(x.mediaId, x.score) for x in single]


# ---- the reference tree is approximate on more than one leaf: tree-compatible mode ------------------------
def _popc(a):
    c = np.zeros(len(a), np.int32)
    for k in range(64):
        c += ((a >> np.uint64(k)) & np.uint64(1)).astype(np.int32)
    return c


def test_leaf_mask_rule_reproduces_real_multileaf_tree(orc):
    """fdct_multileaf.npz holds the REAL HammingTree's candidate sets on a 60k-entry (multi-leaf) tree: the
    restated leaf rule (node (d, prefix) is internal iff > 8192 hashes share the prefix) must select exactly
    those, and the masked voting must reproduce DctFeaturesIndex::find on that tree."""
    g = load_golden("fdct_multileaf.npz")
    h, ids, q, thr = g["hashes"], g["ids"], g["cand_q"], int(g["cand_thresh"])
    mk = orc.htree_leaf_masks(h, q)
    assert (mk > 0).all() and len(set(mk.tolist())) >= 1  # deeper than the root: it really is multi-leaf
    exact_more = 0
    for j, x in enumerate(q):
        xr = h ^ x
        d = _popc(xr)
        sel = (d < thr) & ((xr & mk[j]) == 0)
        a, b = int(g["cand_offs"][j]), int(g["cand_offs"][j + 1])
        want = sorted(zip(g["cand_dist"][a:b].tolist(), g["cand_idx"][a:b].tolist()))
        assert sorted(zip(d[sel].tolist(), ids[sel].tolist())) == want, j
        exact_more += int((d < thr).sum()) - (b - a)
    assert exact_more > 0  # the exact search sees candidates the tree misses: the modes really differ
    for nh, nid, t, wi, ws in _cases(g):
        gi, gs = orc.fdct_find_tree(h, ids, nh, nid, t)
        assert gi.tolist() == wi.tolist() and gs.tolist() == ws.tolist(), (nid, t)


@pytest.mark.gpu
def test_gpu_tree_compatible_mode_equals_real_multileaf_tree(gpu, orc, scan_path, reduce_path):
    g = load_golden("fdct_multileaf.npz")
    h, ids, q, thr = g["hashes"], g["ids"], g["cand_q"], int(g["cand_thresh"])
    raw = gpu.DctHashIndex()
    raw.load(h, np.maximum(ids, 0))
    mk = raw.tree_masks(q)
    assert (mk == orc.htree_leaf_masks(h, q)).all()
    gi, gs, gc = raw.find_batch(q, thr, 64, masks=mk)
    for j in range(len(q)):
        a, b = int(g["cand_offs"][j]), int(g["cand_offs"][j + 1])
        want = sorted((int(d), int(i)) for d, i in zip(g["cand_dist"][a:b], g["cand_idx"][a:b]) if i != 0)
        assert gc[j] == len(want) and list(zip(gs[j, :gc[j]].tolist(), gi[j, :gc[j]].tolist())) == want, j
    # the index class: load everything, remove two media, search with the reference's candidate sets
    idx = gpu.DctFeaturesIndex(tree_compat=True)
    idx.load([(int(i), [int(x)]) for i, x in zip(ids.tolist(), h.tolist())])
    p = gpu.SearchParams()
    for nh, nid, t, wi, ws in _cases(g):
        p.dctThresh = t
        got = idx.find(gpu.Media(id=nid, keyPointHashes=nh.tolist()), p)
        assert [m.mediaId for m in got] == wi.tolist() and [m.score for m in got] == ws.tolist(), (nid, t)
    exact = gpu.DctFeaturesIndex()
    exact.load([(int(i), [int(x)]) for i, x in zip(ids.tolist(), h.tolist())])
    differ = 0
    for nh, nid, t, wi, ws in list(_cases(g))[:30]:
        p.dctThresh = t
        e = exact.find(gpu.Media(id=nid, keyPointHashes=nh.tolist()), p)
        differ += [m.mediaId for m in e] != wi.tolist() or [m.score for m in e] != ws.tolist()
    assert differ > 0  # the default (exact) mode is a different, larger candidate set on this tree
