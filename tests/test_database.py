"""Database::searchIndex / similar (src/database.cpp:1691-1757, 1280-1466) three ways: cbird_amd/database.py's
per-needle route, the C-ABI batch route (cbh_search_index_batch + cbh_filter_groups) and the INDEPENDENT C oracle
oracle/search_index.c -- rule-level cases, an index out of step with the haystack, and the GPU end to end."""
import numpy as np
import pytest


class OracleIndex:
    """find() through the oracle -- a stand-in for DctHashIndex on machines without a GPU"""

    def __init__(self, orc, hashes, ids):
        self.orc, self.h, self.ids = orc, hashes, ids

    def find(self, m, p):
        from cbird_amd import Match

        if not m.dctHash:
            return []
        i, d = self.orc.find64(self.h, self.ids, m.dctHash, p.dctThresh)
        return [Match(int(a), int(b)) for a, b in zip(i, d)]

    def find_batch(self, hashes, thresh, k):
        return self.orc.find64_batch(self.h, self.ids, np.asarray(hashes, np.uint64), thresh, k)


def _media(hashes, ids):
    from cbird_amd import Media

    return [Media(id=int(i), dctHash=int(h), path=f"/img/{int(i):06d}.jpg") for h, i in zip(hashes, ids)]


def test_search_index_rules(orc):
    from cbird_amd import Media, SearchParams
    from cbird_amd.database import search_index

    base = 0x0F0F0F0F0F0F0F00
    hashes = np.array([base, base ^ 2, base ^ 6, base ^ 14, base ^ 30, base ^ 62, base ^ 126, 0xFFFF0000FFFF0000],
                      np.uint64)
    ids = np.arange(1, 9, dtype=np.uint32)
    media = _media(hashes, ids)
    idx = OracleIndex(orc, hashes, ids)
    id_map = {m.id: m for m in media}
    p = SearchParams(dctThresh=4)  # distances to needle 1: 0,1,2,3,4,5,6
    g = search_index(idx, media[0], p, id_map)
    assert [m.id for m in g] == [2, 3, 4] and [m.score for m in g] == [1, 2, 3]  # self filtered, sorted by score
    p.filterSelf = False
    assert [m.id for m in search_index(idx, media[0], p, id_map)] == [1, 2, 3, 4]
    p = SearchParams(dctThresh=8, maxMatches=3)
    assert [m.id for m in search_index(idx, media[0], p, id_map)] == [2, 3, 4]  # cut at maxMatches
    # maxThresh: raise dct threshold until more than minMatches results (database.cpp:1703-1725)
    p = SearchParams(dctThresh=1, maxThresh=3, minMatches=1)
    g = search_index(idx, media[0], p, id_map)
    assert [m.id for m in g] == [2] and g[0].score == 1  # stopped at dht=2: {self, id 2}
    lonely = Media(id=8, dctHash=int(hashes[7]), path="/img/000008.jpg")
    assert search_index(idx, lonely, SearchParams(dctThresh=1, maxThresh=6), id_map) == []
    # an id the caller does not know is skipped with a warning (stale index)
    with pytest.warns(UserWarning):
        g = search_index(idx, media[0], SearchParams(dctThresh=4), {k: v for k, v in id_map.items() if k != 3})
    assert [m.id for m in g] == [2, 4]


PARAM_SETS = [dict(dctThresh=2), dict(dctThresh=5, maxMatches=2), dict(dctThresh=1, maxThresh=4),
              dict(dctThresh=3, minMatches=2, filterSelf=False), dict(dctThresh=4, minMatches=0),
              dict(dctThresh=2, maxThresh=7, minMatches=3, maxMatches=7), dict(dctThresh=6, maxMatches=1)]


def _case(n=2500, seed=8):
    """haystack media + an index that is NOT in step with it: ids the haystack does not know (stale entries, skipped
    like database.cpp:1755), a removed slot (id 0), an item without a hash"""
    from cbird_amd import synth

    h, ids = synth.make_hashes(n, seed=seed, planted_frac=0.3, max_dist=6)
    h[17] = 0  # item without hash: never a needle, never found
    idx_h = np.concatenate([h, h[100:140] ^ np.uint64(2)])          # 40 stale entries close to real ones
    idx_i = np.concatenate([ids, np.arange(900001, 900041, dtype=np.uint32)])
    idx_i[55] = 0                                                    # removed slot
    idx_h[55] = 0
    return h, ids, idx_h, idx_i


def _oracle_groups(orc, h, ids, idx_h, idx_i, p, filter_groups=True):
    rank = np.argsort(np.argsort([f"/img/{int(i):06d}.jpg" for i in ids])).astype(np.int32)
    res = orc.similar_dct(h, ids, rank, idx_h, idx_i, p.dctThresh, p.maxThresh, p.minMatches, p.maxMatches,
                          p.filterSelf, filter_groups)
    return [[(int(ids[j]), -1)] + m for j, m in res]


def _key(groups):
    return [[(m.id, m.score if t else -1) for t, m in enumerate(g)] for g in groups]


def test_python_route_equals_the_independent_c_oracle(orc):
    """cbird_amd/database.py (per-needle route over an oracle-backed find) against oracle/search_index.c, which
    restates src/database.cpp:1691-1757,1400-1463 on its own: two statements of the reference, one answer"""
    import warnings

    from cbird_amd import SearchParams
    from cbird_amd.database import similar

    h, ids, idx_h, idx_i = _case()
    media = _media(h, ids)
    idx = OracleIndex(orc, idx_h, idx_i)
    for kw in PARAM_SETS:
        p = SearchParams(**kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "no media with id" for the stale entries
            got = _key(similar(idx, media, p, batched=False))
        want = _oracle_groups(orc, h, ids, idx_h, idx_i, p)
        assert got == want, kw
        assert len(got) > 5, kw
        assert all(len(g) > max(1, p.minMatches) for g in got)
        sets = [tuple(sorted(i for i, _ in g)) for g in got]
        assert len(sets) == len(set(sets))
        assert not any(i > 900000 or i == 0 for g in got for i, _ in g)  # stale / removed entries never appear


def test_filter_groups_c_abi_on_host(orc):
    """cbh_filter_groups is host code in the C-ABI: acceptance, duplicate groups and order on per-needle results"""
    import ctypes as C
    import warnings

    from cbird_amd import SearchParams, _lib
    from cbird_amd.database import search_index

    h, ids, idx_h, idx_i = _case(800, 3)
    media = _media(h, ids)
    idx = OracleIndex(orc, idx_h, idx_i)
    id_map = {m.id: m for m in media}
    L = _lib.lib()
    for kw in (dict(dctThresh=3), dict(dctThresh=5, minMatches=2, maxMatches=4), dict(dctThresh=4, minMatches=0)):
        p = SearchParams(**kw)
        k = p.maxMatches
        pairs = np.zeros((len(media), k, 2), np.uint32)
        counts = np.zeros(len(media), np.uint32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for j, m in enumerate(media):
                g = search_index(idx, m, p, id_map) if m.dctHash else []
                counts[j] = len(g)
                for t, x in enumerate(g):
                    pairs[j, t] = (x.id, x.score)
        rank = np.argsort(np.argsort([m.path for m in media])).astype(np.uint32)
        by_id = np.argsort(ids)
        ids_sorted, rank_sorted = np.ascontiguousarray(ids[by_id]), np.ascontiguousarray(rank[by_id])
        for fg in (1, 0):
            out = np.zeros(len(media), np.uint32)
            n_out = C.c_size_t(0)
            _lib.check(L.cbh_filter_groups(ids.ctypes.data, pairs.ctypes.data, counts.ctypes.data, len(media), k,
                                           p.minMatches, fg, ids_sorted.ctypes.data, rank_sorted.ctypes.data,
                                           len(media), out.ctypes.data, C.byref(n_out)), "filter_groups")
            got = [[(int(ids[j]), -1)] + [(int(a), int(b)) for a, b in pairs[j, : counts[j]]] for j in out[: n_out.value]]
            assert got == _oracle_groups(orc, h, ids, idx_h, idx_i, p, filter_groups=bool(fg)), (kw, fg)


def _filter_case(seed, n_media=260, n_groups=180):
    """media with directory / zip-member paths and random per-needle results over them"""
    from cbird_amd import Media

    rng = np.random.default_rng(seed)
    dirs = ["/db/a", "/db/a/sub", "/db/b", "/db/b.zip:inner", "/db/b.zip:inner/x", "/db/c.CBZ:", "/db/pics.v2:raw",
            "/other/a", "/db/ab"]
    media = []
    for i in range(n_media):
        d = dirs[int(rng.integers(0, len(dirs)))]
        sep = "" if d.endswith(":") else "/"
        media.append(Media(id=int(1000 + 7 * i), dctHash=1, path=f"{d}{sep}f{int(rng.integers(0, 10 ** 6)):06d}_{i}.jpg"))
    results = []
    for _ in range(n_groups):
        j = int(rng.integers(0, n_media))
        k = int(rng.integers(0, 6))
        others = [int(x) for x in rng.choice(n_media, k, replace=False) if int(x) != j]
        g = [media[j]]
        for o in others:
            m = Media(id=media[o].id, dctHash=1, path=media[o].path)
            m.score = int(rng.integers(0, 9))
            g.append(m)
        g[1:] = sorted(g[1:], key=lambda m: (m.score, m.id))
        results.append(g)
    # a pair found from both ends, and a chain a->b, b->c for the merge
    a, b, c = media[3], media[4], media[5]
    for x, y in ((a, b), (b, a), (b, c)):
        m = Media(id=y.id, dctHash=1, path=y.path)
        m.score = 2
        results.append([x, m])
    return media, results


FILTER_SETS = [dict(), dict(minMatches=2), dict(filterParent=True), dict(path="a", inPath=True),
               dict(path="/db/b", inPath=False), dict(path="b.zip:inner", inPath=True, filterParent=True, minMatches=0),
               dict(mergeGroups=1), dict(expandGroups=True), dict(mergeGroups=1, expandGroups=True, filterGroups=False),
               dict(filterGroups=False, expandGroups=True, filterParent=True, path="/db/a", inPath=False),
               dict(minMatches=0, path="nowhere", inPath=True)]


def test_filter_match_and_filter_matches_three_ways(orc):
    """Database::filterMatch + filterMatches (path / inPath, filterParent, the count rule, filterGroups, mergeGroups,
    expandGroups; src/database.cpp:1209-1278, src/media.cpp:198-208,300-331): the C-ABI's id / attribute form
    (cbh_filter_groups_ex), cbird_amd/database.py on Media objects, and oracle/search_index.c on path strings -- three
    statements, one answer"""
    from cbird_amd import SearchParams
    from cbird_amd.database import dir_path, filter_groups_c_abi, filter_results, parse_archive_path

    assert parse_archive_path("/x/y.zip:a/b.jpg") == ("/x/y.zip", "a/b.jpg")
    assert parse_archive_path("/x/y.zipx:a.jpg") is None and parse_archive_path("C:/x.jpg") is None
    assert parse_archive_path("/x/a.zip:b.epub:c.png") == ("/x/a.zip:b.epub", "c.png")  # the LAST marker wins
    assert dir_path("/x/y.zip:a/b.jpg") == "/x/y.zip" and dir_path("/x/y/b.jpg") == "/x/y" and dir_path("b.jpg") == ""
    for seed in (1, 2):
        media, results = _filter_case(seed)
        paths = [m.path for m in media]
        pos = {m.id: i for i, m in enumerate(media)}
        needle_ids = np.array([g[0].id for g in results], np.uint32)
        k = max(1, max(len(g) - 1 for g in results))
        pairs = np.zeros((len(results), k, 2), np.uint32)
        counts = np.zeros(len(results), np.uint32)
        for j, g in enumerate(results):
            counts[j] = len(g) - 1
            for t, m in enumerate(g[1:]):
                pairs[j, t] = (m.id, np.uint32(m.score))
        for kw in FILTER_SETS:
            p = SearchParams(**kw)
            py = [[(m.id, m.score if t or g[0].score != -1 else -1) for t, m in enumerate(g)]
                  for g in filter_results(p, [list(g) for g in results], "/db")]
            c_abi = filter_groups_c_abi(p, needle_ids, pairs, counts, media, "/db")
            o = orc.filter_groups_paths(paths, [[(pos[m.id], -1 if t == 0 else m.score) for t, m in enumerate(g)]
                                                for g in results], "/db", p.path, p.inPath, p.filterParent,
                                        p.minMatches, p.filterGroups, p.mergeGroups, p.expandGroups)
            o = [[(media[i].id, s) for i, s in g] for g in o]
            assert c_abi == o, (seed, kw)
            assert py == o, (seed, kw)
            assert len(o) > 0 or kw.get("path") == "nowhere", kw
    # the pair found from both ends is reported once, the chain merges into one group
    media, results = _filter_case(1)
    ids3 = {media[3].id, media[4].id, media[5].id}
    needle_ids = np.array([g[0].id for g in results], np.uint32)
    merged = filter_groups_c_abi(SearchParams(mergeGroups=1), needle_ids, pairs * 0 + _pairs_of(results), _counts_of(results),
                                 media, "/db")
    assert any(ids3 <= {i for i, _ in g} for g in merged)


def _pairs_of(results):
    k = max(1, max(len(g) - 1 for g in results))
    pairs = np.zeros((len(results), k, 2), np.uint32)
    for j, g in enumerate(results):
        for t, m in enumerate(g[1:]):
            pairs[j, t] = (m.id, np.uint32(m.score))
    return pairs


def _counts_of(results):
    return np.array([len(g) - 1 for g in results], np.uint32)


@pytest.mark.gpu
def test_similar_behind_the_c_abi_equals_the_oracle(gpu, orc, scan_path):
    """cbh_search_index_batch + cbh_filter_groups (scans, escalation and cut on the device) == oracle/search_index.c
    == the per-needle route, on an index with stale and removed entries; includes needles whose cut needs more
    places than the batch fetched (many stale neighbours: the exact single-needle fallback)"""
    import warnings

    from cbird_amd import SearchParams
    from cbird_amd.database import similar

    h, ids, idx_h, idx_i = _case()
    # twelve stale copies of one hash: its needle (and its planted neighbours) must skip all of them
    idx_h = np.concatenate([idx_h, np.full(12, h[7], np.uint64)])
    idx_i = np.concatenate([idx_i, np.arange(910001, 910013, dtype=np.uint32)])
    media = _media(h, ids)
    idx = gpu.DctHashIndex()
    idx.load(idx_h, idx_i)
    for kw in PARAM_SETS:
        p = SearchParams(**kw)
        got = _key(similar(idx, media, p, batched=True))
        assert got == _oracle_groups(orc, h, ids, idx_h, idx_i, p), kw
    p = SearchParams(dctThresh=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert _key(similar(idx, media, p, batched=False)) == _key(similar(idx, media, p, batched=True))


@pytest.mark.gpu
def test_similar_gpu_configs0_at_full_size(gpu, orc):
    """BASELINE configs[0] as stated: 10 000 synthetic 256x256 grey images, `-p.alg dct -p.dht 2 -similar` -- hash
    kernel -> DctHashIndex -> Database::similar in one batch (cbh_search_index_batch + the group filter) on the GPU,
    against the CPU statement of the same job (oracle hashes, oracle/search_index.c's similar).  Identical hashes,
    identical groups in identical order."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    import bench
    from cbird_amd import SearchParams
    from cbird_amd.database import similar

    n = 10_000
    # bench.py's data set (SURVEY 8d's recipe, generated on the device: synth.make_images takes 100 s for 10k)
    imgs = bench.gen_images(torch, torch.device("cuda:0"), 0, n, n, 1234).cpu().numpy()
    h = gpu.dct_hash64_batch(imgs)
    with ThreadPoolExecutor(8) as ex:  # ctypes releases the GIL
        parts = list(ex.map(orc.dcthash64_fast256_batch, np.array_split(imgs, 32)))
    h_cpu = np.concatenate(parts)
    assert (h == h_cpu).all()
    assert (orc.dcthash64_batch(imgs[:300]) == h[:300]).all()  # and the scalar port on a sample
    ids = np.arange(1, n + 1, dtype=np.uint32)
    p = SearchParams(dctThresh=2)
    want = orc.similar_dct(h_cpu, ids, np.arange(n, dtype=np.int32), h_cpu, ids, 2, 0, p.minMatches, p.maxMatches,
                           True, True)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    got = similar(idx, _media(h, ids), p, batched=True)
    assert len(got) == len(want) >= 100
    for g, (j, members) in zip(got, want):
        assert g[0].id == ids[j] and [(m.id, m.score) for m in g[1:]] == members


@pytest.mark.gpu
def test_similar_gpu_images_end_to_end_three_routes(gpu, orc):
    """configs[0]'s shape at 600 images, so that the per-needle routes can run beside the batch: synthetic 256x256
    images -> hash (GPU) -> DctHashIndex -> -p.dht 2 -similar; equals the same pipeline driven through the oracle."""
    from cbird_amd import SearchParams, synth
    from cbird_amd.database import similar

    imgs = synth.make_images(600, seed=1234)
    h = gpu.dct_hash64_batch(imgs)
    assert (h == orc.dcthash64_batch(imgs)).all()
    ids = np.arange(1, len(h) + 1, dtype=np.uint32)
    media = _media(h, ids)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    p = SearchParams(dctThresh=2)
    got = similar(idx, media, p, batched=True)
    want = similar(OracleIndex(orc, h, ids), media, p, batched=False)
    key = lambda gs: [[(m.id, m.score) for m in g] for g in gs]
    assert key(got) == key(want)
    per_needle = similar(idx, media, p, batched=False)
    assert key(per_needle) == key(want)
    assert len(got) >= 10  # the planted near-duplicates are found
