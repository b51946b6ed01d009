"""The caller contract around Index::find: Database::searchIndex and the -similar driver
(src/database.cpp:1209-1278, 1280-1466, 1691-1757), restated for in-memory media lists.

Only the parts that shape the search RESULT are here (threshold escalation, ordering, self filter, maxMatches
cut, minMatches acceptance, duplicate-group filter); SQL, negative-match lists, weeds and path filters are
storage/bookkeeping (out of scope).

`similar()` has two routes with identical results: the reference's shape (one find() per needle) and, for an
index that offers `find_batch`, one batched scan per threshold level.
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
    """filterMatch's acceptance (group incl. needle must exceed minMatches, database.cpp:1245) and the
    filterGroups pass (same set of paths found more than once is reported once, :1252-1272); groups are then
    ordered by the needle's path (:1463)."""
    groups = [g for g in results if len(g) > params.minMatches]
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
    as a needle; returns the accepted groups [needle, match1, ...]."""
    id_map = {m.id: m for m in haystack}
    results = []
    if batched and hasattr(index, "find_batch") and params.algo == SearchParams.AlgoDCT:
        # one scan per threshold level instead of one tree walk per needle
        k = params.maxMatches + 1  # room for the self match removed by filterSelf
        pending = [m for m in haystack if m.dctHash]
        found = {}
        tmp = copy.copy(params)
        while pending:
            ids, scores, counts = index.find_batch([m.dctHash for m in pending], tmp.dctThresh, k)
            nxt = []
            for j, m in enumerate(pending):
                n = min(int(counts[j]), k)
                found[m.id] = [Match(int(ids[j, t]), int(scores[j, t])) for t in range(n)]
                # escalation looks at the FULL match count (matches.count() <= minMatches, :1705)
                if params.maxThresh > 0 and int(counts[j]) <= params.minMatches:
                    nxt.append(m)
            if not nxt or not _escalate(params, tmp):
                break
            pending = nxt
        for m in haystack:
            if not m.dctHash:
                continue
            results.append([m] + _group_from_matches(m, found.get(m.id, []), params, id_map))
    else:
        for m in haystack:
            if params.algo == SearchParams.AlgoDCT and not m.dctHash:
                continue
            results.append([m] + search_index(index, m, params, id_map))
    return _accept_and_dedupe(results, params)
