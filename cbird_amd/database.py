"""The caller contract around Index::find: Database::searchIndex and the -similar driver
(src/database.cpp:1209-1278, 1280-1466, 1691-1757), restated for in-memory media lists.

Only the parts that shape the search RESULT are here (threshold escalation, ordering, self filter, maxMatches
cut, the path / inPath and filterParent filters, minMatches acceptance, duplicate-group filter, mergeGroups /
expandGroups); SQL, negative-match lists and weeds are other tables of the database (out of
scope).

`similar()` has two routes with identical results: the reference's shape (one find() per needle, Python code
below) and, for DctHashIndex, the whole job behind the C-ABI (cbh_search_index_batch + cbh_filter_groups).
The test suite holds both against a third, independent C restatement of the same reference lines
(tests/test_database.py).
"""
from __future__ import annotations

import copy
import warnings

from .index import Match, SearchParams


def _sorted_matches(matches):
    """std::sort(matches) by score (index.h:284, unstable in the reference); ties fixed to ascending
    mediaId -- SURVEY.md section 7, hard part 2."""
    return sorted(matches, key=lambda m: (m.score, m.mediaId))


def _escalate(params: SearchParams, tmp: SearchParams) -> bool:
    """one step of the maxThresh loop (database.cpp:1706-1722); False = stop"""
    if params.algo in (SearchParams.AlgoDCT, SearchParams.AlgoDCTFeatures, SearchParams.AlgoVideo):
        tmp.dctThresh += 1
        return tmp.dctThresh <= params.maxThresh
    if params.algo == SearchParams.AlgoCVFeatures:
        tmp.cvThresh += 5
        return tmp.cvThresh <= params.maxThresh
    if params.algo == SearchParams.AlgoColor:
        return False
    warnings.warn("maxThresh: unsupported algorithm")
    return False


def _group_from_matches(needle, matches, params, id_map):
    group = []
    for match in _sorted_matches(matches):
        if params.filterSelf and int(match.mediaId) == needle.id:
            continue
        if len(group) >= params.maxMatches:
            break
        media = id_map.get(int(match.mediaId))
        if media is not None and media.isValid():
            media = copy.copy(media)
            media.score = match.score
            media.matchRange = match.range
            group.append(media)
        else:
            warnings.warn(f"no media with id: {int(match.mediaId)}, index could be stale or corrupt")
    return group


def search_index(index, needle, params: SearchParams, id_map: dict):
    """Database::searchIndex (database.cpp:1691-1757)"""
    matches = index.find(needle, params)
    if params.maxThresh > 0:
        tmp = copy.copy(params)
        while len(matches) <= params.minMatches:
            if not _escalate(params, tmp):
                break
            matches = index.find(needle, tmp)
    return _group_from_matches(needle, matches, params, id_map)


def search_index_batch(index, needles, params: SearchParams, id_map: dict):
    """Database::searchIndex for a needle batch behind the C-ABI (cbh_*_search_index_batch, searchbatch.hip /
    search.hip): one batched find per threshold level instead of one find per needle and level.  Returns what
    [search_index(index, m, params, id_map) for m in needles] returns."""
    import ctypes as C

    import numpy as np

    from . import _lib
    from .index import DctFeaturesIndex, DctHashIndex, MatchRange

    L = _lib.lib()
    needles = list(needles)
    n, k = len(needles), int(params.maxMatches)
    nid = np.ascontiguousarray([m.id for m in needles], np.uint32)
    valid = np.unique(np.array(sorted(id_map), np.uint32))
    counts = np.zeros(max(n, 1), np.uint32)
    vargs = (valid.ctypes.data, len(valid))
    fs = int(bool(params.filterSelf))
    cls = type(index).__name__
    if isinstance(index, DctHashIndex):
        mi, ms, mc = index.search_index_batch([m.dctHash for m in needles], nid, params, valid_ids=valid)
        rows = [[(int(mi[j, t]), int(ms[j, t]), None) for t in range(int(mc[j]))] for j in range(n)]
    elif isinstance(index, DctFeaturesIndex):
        hs = [np.asarray(list(getattr(m, "keyPointHashes", []) or []), np.uint64) for m in needles]
        hs = [h if len(h) or m.id <= 0 else index.hashesForId(m.id) for h, m in zip(hs, needles)]  # (:270-276)
        offs = np.zeros(n + 1, np.uint64)
        np.cumsum([len(h) for h in hs], out=offs[1:])
        allh = np.ascontiguousarray(np.concatenate(hs) if n else np.zeros(0, np.uint64), np.uint64)
        out = np.zeros((max(n, 1), max(k, 1), 2), np.uint32)
        _lib.check(L.cbh_fdct_search_index_batch(index.handle, allh.ctypes.data, offs.ctypes.data, nid.ctypes.data, n,
                                                 int(params.dctThresh), int(params.maxThresh), int(index.tree_compat),
                                                 int(params.minMatches), k, fs, *vargs, out.ctypes.data,
                                                 counts.ctypes.data), "fdct_search_index_batch")
        sc = out[..., 1].view(np.int32)
        rows = [[(int(out[j, t, 0]), int(sc[j, t]), None) for t in range(int(counts[j]))] for j in range(n)]
    elif cls == "CvFeaturesIndex":
        ds = [index._rows(m.keyPointDescriptors) for m in needles]
        offs = np.zeros(n + 1, np.uint64)
        np.cumsum([len(d) for d in ds], out=offs[1:])
        allr = np.ascontiguousarray(np.concatenate(ds) if n else np.zeros((0, 32), np.uint8))
        out = np.zeros((max(n, 1), max(k, 1), 2), np.uint32)
        _lib.check(L.cbh_idx256_search_index_batch(index.handle, allr.ctypes.data, offs.ctypes.data, nid.ctypes.data, n,
                                                   int(params.cvThresh), int(params.maxThresh), index.KNN,
                                                   int(params.minMatches), k, fs, *vargs, out.ctypes.data,
                                                   counts.ctypes.data), "idx256_search_index_batch")
        sc = out[..., 1].view(np.int32)
        rows = [[(int(out[j, t, 0]), int(sc[j, t]), None) for t in range(int(counts[j]))] for j in range(n)]
    elif cls == "ColorDescIndex":
        from .colordesc import COLOR_DTYPE

        d = np.ascontiguousarray(np.asarray([m.colorDescriptor for m in needles], COLOR_DTYPE).reshape(-1))
        out = np.zeros((max(n, 1), max(k, 1), 2), np.uint32)
        _lib.check(L.cbh_color_search_index_batch(index._h, d.ctypes.data, nid.ctypes.data, n, k, fs, *vargs,
                                                  out.ctypes.data, counts.ctypes.data), "color_search_index_batch")
        sc = out[..., 1].view(np.int32)
        rows = [[(int(out[j, t, 0]), int(sc[j, t]), None) for t in range(int(counts[j]))] for j in range(n)]
    elif cls == "DctVideoIndex":
        index._apply_radix(params)
        f = np.concatenate([np.asarray(m.videoIndex.frames, np.int32) for m in needles] or [np.zeros(0, np.int32)])
        h = np.concatenate([np.asarray(m.videoIndex.hashes, np.uint64) for m in needles] or [np.zeros(0, np.uint64)])
        offs = np.zeros(n + 1, np.uint64)
        np.cumsum([len(m.videoIndex.frames) for m in needles], out=offs[1:])
        out = (_lib.cbh_vmatch * max(1, n * max(k, 1)))()
        _lib.check(L.cbh_vidx_search_index_batch(index._h, f.ctypes.data, h.ctypes.data, offs.ctypes.data, nid.ctypes.data,
                                                 n, int(params.dctThresh), int(params.maxThresh), int(params.skipFrames),
                                                 int(params.minFramesMatched), int(params.minFramesNear),
                                                 int(params.minMatches), k, fs, *vargs, out, counts.ctypes.data),
                   "vidx_search_index_batch")
        rows = [[(out[j * k + t].id, out[j * k + t].score,
                  MatchRange(out[j * k + t].src_in, out[j * k + t].dst_in, out[j * k + t].len))
                 for t in range(int(counts[j]))] for j in range(n)]
    else:
        raise TypeError(f"search_index_batch: unsupported index {cls}")
    groups = []
    for r in rows:
        g = []
        for mid, score, rng in r:
            media = copy.copy(id_map[mid])
            media.score = score
            if rng is not None:
                media.matchRange = rng
            g.append(media)
        groups.append(g)
    return groups


# Media::parseArchivePath (src/media.cpp:1039-1043, 1083-1099): "<zip>:<member>" virtual paths
_ZIP_MARKERS = (".zip:", ".ZIP:", ".cbz:", ".CBZ:", ".epub:", ".EPUB:", ".odt:", ".ODT:", ".ods:", ".ODS:", ".odp:", ".ODP:",
                ".docx:", ".DOCX:", ".pptx:", ".PPTX:", ".xlsx:", ".XLSX:", ".xps:", ".XPS")


def parse_archive_path(path: str):
    """(parent, child) of a zip-member path, None for a plain file"""
    end = path.rfind(":")
    while end > 1:
        for marker in _ZIP_MARKERS:
            start = end - len(marker) + 1
            if start < 0:
                continue
            if path[start:start + len(marker)] == marker:
                cut = start + len(marker)
                return path[:cut - 1], path[cut:]
        end = path.rfind(":", 0, end)  # lastIndexOf(':', end - 1)
    return None


def dir_path(path: str) -> str:
    """Media::dirPath (src/media.cpp:198-208): the archive for a zip member, else everything before the last '/'"""
    a = parse_archive_path(path)
    if a:
        return a[0]
    i = path.rfind("/")
    return "" if i < 0 else path[:i]


def filter_match(params, match, db_path: str = "") -> bool:
    """Database::filterMatch (src/database.cpp:1209-1248) on one group [needle, match...], in place; True = drop it.
    negativeMatch and the weed marks need other tables and are not here."""
    if params.path != "" and len(match) > 1:
        prefix = params.path
        if not prefix.startswith(db_path):
            prefix = db_path + "/" + params.path
        match[:] = [match[0]] + [m for m in match[1:] if (not params.inPath) ^ m.path.startswith(prefix)]
    if params.filterParent and len(match) > 1:
        parent = dir_path(match[0].path)
        match[:] = [match[0]] + [m for m in match[1:] if dir_path(m.path) != parent]
    return not len(match) > params.minMatches


def merge_group_list(groups):
    """Media::mergeGroupList (src/media.cpp:300-324); Media equality is by path (media.h:237); a merged group is
    ordered by score (Media::operator<), equal scores by path"""
    for i in range(len(groups)):
        for j in range(len(groups)):
            if i == j:
                continue
            a, b = groups[i], groups[j]
            if b and any(m.path == b[0].path for m in a):
                for m in b[1:]:
                    if not any(x.path == m.path for x in a):
                        a.append(m)
                del b[:]
                a.sort(key=lambda m: (m.score, m.path))
    return [g for g in groups if g]


def expand_group_list(groups):
    """Media::expandGroupList (src/media.cpp:326-331)"""
    return [[g[0], m] for g in groups for m in g[1:]]


def filter_matches(params, groups):
    """Database::filterMatches (src/database.cpp:1250-1278)"""
    if getattr(params, "filterGroups", True):
        groups = sorted(groups, key=lambda g: g[0].path)  # stable, like Media::sortGroupList
        seen, out = set(), []
        for g in groups:
            key = tuple(sorted(m.path for m in g))
            if key not in seen:
                seen.add(key)
                out.append(g)
        groups = out
    if getattr(params, "mergeGroups", 0):
        groups = merge_group_list(groups)
    elif getattr(params, "expandGroups", False):
        groups = expand_group_list(groups)
    return groups


def filter_results(params, results, db_path: str = ""):
    """the tail of Database::similar (:1446-1463) on [needle, match...] lists: filterMatch per group, filterMatches,
    order by the first member's path"""
    kept = []
    for g in results:
        if len(g) <= 1:  # a needle without a result is no group (:1409)
            continue
        g = list(g)
        if not filter_match(params, g, db_path):
            kept.append(g)
    return sorted(filter_matches(params, kept), key=lambda g: g[0].path)


def media_attributes(media, params, db_path: str = ""):
    """what cbh_filter_groups_ex takes instead of strings: for every media (ascending id) the rank of its path among
    all sorted paths, a number per distinct dirPath(), and whether the path lies under params.path"""
    import numpy as np

    media = sorted(media, key=lambda m: m.id)
    ids = np.array([m.id for m in media], np.uint32)
    order = sorted(range(len(media)), key=lambda i: media[i].path)
    rank = np.zeros(len(media), np.uint32)
    rank[order] = np.arange(len(media), dtype=np.uint32)
    dirs = {}
    dir_id = np.array([dirs.setdefault(dir_path(m.path), len(dirs)) for m in media], np.uint32)
    prefix = params.path
    if prefix != "" and not prefix.startswith(db_path):
        prefix = db_path + "/" + params.path
    under = np.array([1 if (prefix != "" and m.path.startswith(prefix)) else 0 for m in media], np.uint8)
    return ids, rank, dir_id, under


def filter_groups_c_abi(params, needle_ids, pairs, counts, media, db_path: str = ""):
    """cbh_filter_groups_ex on per-needle results (pairs[nq, k, 2] = (id, score), counts[nq]); returns the groups as
    lists of (mediaId, score), the needle first with score -1"""
    import ctypes as C

    import numpy as np

    from . import _lib

    ids, rank, dir_id, under = media_attributes(media, params, db_path)
    needle_ids = np.ascontiguousarray(needle_ids, np.uint32)
    pairs = np.ascontiguousarray(pairs, np.uint32)
    counts = np.ascontiguousarray(counts, np.uint32)
    nq, k = len(needle_ids), pairs.shape[1] if pairs.ndim == 3 else 1
    fp = _lib.cbh_filter_params(int(params.minMatches), int(bool(getattr(params, "filterGroups", True))),
                                int(bool(params.filterParent)), 0 if params.path == "" else (1 if params.inPath else 2),
                                int(params.mergeGroups), int(bool(params.expandGroups)))
    cap_g = cap_m = 0
    while True:
        first = np.zeros(cap_g + 1, np.uint64)
        members = np.zeros((max(cap_m, 1), 2), np.uint32)
        ng, nm = C.c_size_t(0), C.c_size_t(0)
        rc = _lib.lib().cbh_filter_groups_ex(needle_ids.ctypes.data, pairs.ctypes.data, counts.ctypes.data, nq, k,
                                             C.byref(fp), ids.ctypes.data, rank.ctypes.data, dir_id.ctypes.data,
                                             under.ctypes.data, len(ids), first.ctypes.data, cap_g, members.ctypes.data,
                                             cap_m, C.byref(ng), C.byref(nm))
        if rc == _lib.CBH_E_OVERFLOW:
            cap_g, cap_m = ng.value, nm.value
            continue
        _lib.check(rc, "filter_groups_ex")
        sc = members[:, 1].view(np.int32)
        return [[(int(members[t, 0]), int(sc[t])) for t in range(int(first[g]), int(first[g + 1]))]
                for g in range(ng.value)]


def similar(index, haystack, params: SearchParams, batched: bool = True, db_path: str = ""):
    """Database::similar for an in-memory haystack (list of Media with unique ids): every item is searched
    as a needle; returns the accepted groups [needle, match1, ...].

    batched (DctHashIndex): the whole job behind the C-ABI -- cbh_search_index_batch (scans, escalation and the
    per-needle cut on the device, no per-needle loop here) and cbh_filter_groups_ex (path / parent filters, acceptance,
    duplicate groups, merge / expand, order); this function only turns the surviving rows back into Media objects."""
    id_map = {m.id: m for m in haystack}
    if batched and hasattr(index, "search_index_batch") and params.algo == SearchParams.AlgoDCT:
        import ctypes as C

        import numpy as np

        from . import _lib

        hay = list(haystack)
        ids = np.array([m.id for m in hay], np.uint32)
        hashes = np.array([m.dctHash for m in hay], np.uint64)
        mi, ms, mc = index.search_index_batch(hashes, ids, params, valid_ids=ids)
        k = int(params.maxMatches)
        pairs = np.zeros((len(hay), max(k, 1), 2), np.uint32)
        pairs[:, :k, 0], pairs[:, :k, 1] = mi, ms.view(np.uint32) if ms.size else ms
        # paths enter the C-ABI as per-media attributes (rank of the path, directory number, under-the-prefix flag)
        groups = []
        for g in filter_groups_c_abi(params, ids, pairs, mc, hay, db_path):
            out = []
            for t, (mid, score) in enumerate(g):
                media = id_map[mid] if (t == 0 and score == -1) else copy.copy(id_map[mid])
                if not (t == 0 and score == -1):
                    media.score = score
                out.append(media)
            groups.append(out)
        return groups
    results = []
    for m in haystack:
        if params.algo == SearchParams.AlgoDCT and not m.dctHash:
            continue
        results.append([m] + search_index(index, m, params, id_map))
    return filter_results(params, results, db_path)
