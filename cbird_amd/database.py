"""The caller contract around Index::find: Database::searchIndex and the -similar driver
(src/database.cpp:1209-1278, 1280-1466, 1691-1757), restated for in-memory media lists.

Only the parts that shape the search RESULT are here (threshold escalation, ordering, self filter, maxMatches
cut, minMatches acceptance, duplicate-group filter); SQL, negative-match lists, weeds and path filters are
storage/bookkeeping (out of scope).

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


def _accept_and_dedupe(results, params):
    """filterMatch's acceptance (group incl. needle must exceed minMatches, database.cpp:1245; a needle without any
    match never becomes a group, :1409) and the filterGroups pass (same set of paths found more than once is reported
    once, :1252-1272); groups are then ordered by the needle's path (:1463)."""
    groups = [g for g in results if len(g) > 1 and len(g) > params.minMatches]
    groups.sort(key=lambda g: g[0].path)
    if getattr(params, "filterGroups", True):
        seen, out = set(), []
        for g in groups:
            key = tuple(sorted(m.path for m in g))
            if key not in seen:
                seen.add(key)
                out.append(g)
        groups = out
    return groups


def similar(index, haystack, params: SearchParams, batched: bool = True):
    """Database::similar for an in-memory haystack (list of Media with unique ids): every item is searched
    as a needle; returns the accepted groups [needle, match1, ...].

    batched (DctHashIndex): the whole job behind the C-ABI -- cbh_search_index_batch (scans, escalation and the
    per-needle cut on the device, no per-needle loop here) and cbh_filter_groups (acceptance, duplicate groups,
    order); this function only turns the surviving rows back into Media objects."""
    id_map = {m.id: m for m in haystack}
    if batched and hasattr(index, "search_index_batch") and params.algo == SearchParams.AlgoDCT:
        import ctypes as C

        import numpy as np

        from . import _lib

        hay = list(haystack)
        ids = np.array([m.id for m in hay], np.uint32)
        hashes = np.array([m.dctHash for m in hay], np.uint64)
        mi, ms, mc = index.search_index_batch(hashes, ids, params, valid_ids=ids)
        # paths enter the C-ABI as ranks: position of each media's path in the sorted order of all paths
        order = sorted(range(len(hay)), key=lambda i: hay[i].path)
        rank = np.zeros(len(hay), np.uint32)
        rank[order] = np.arange(len(hay), dtype=np.uint32)
        by_id = np.argsort(ids, kind="stable")
        ids_sorted, rank_sorted = np.ascontiguousarray(ids[by_id]), np.ascontiguousarray(rank[by_id])
        k = int(params.maxMatches)
        pairs = np.zeros((len(hay), max(k, 1), 2), np.uint32)
        pairs[:, :k, 0], pairs[:, :k, 1] = mi, ms.view(np.uint32) if ms.size else ms
        out_group = np.zeros(max(1, len(hay)), np.uint32)
        n_out = C.c_size_t(0)
        _lib.check(_lib.lib().cbh_filter_groups(ids.ctypes.data, pairs.ctypes.data, mc.ctypes.data, len(hay), max(k, 1),
                                                int(params.minMatches), int(bool(getattr(params, "filterGroups", True))),
                                                ids_sorted.ctypes.data, rank_sorted.ctypes.data, len(hay),
                                                out_group.ctypes.data, C.byref(n_out)), "filter_groups")
        groups = []
        for j in out_group[: n_out.value].tolist():
            g = [hay[j]]
            for t in range(int(mc[j])):
                media = copy.copy(id_map[int(mi[j, t])])
                media.score = int(ms[j, t])
                g.append(media)
            groups.append(g)
        return groups
    results = []
    for m in haystack:
        if params.algo == SearchParams.AlgoDCT and not m.dctHash:
            continue
        results.append([m] + search_index(index, m, params, id_map))
    return _accept_and_dedupe(results, params)
