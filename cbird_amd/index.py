"""Host-side mirror of cbird's Index plugin surface for the DCT-hash path.

Same names, argument meaning and error behaviour as the reference classes so the parity tests
read like ``unit/testdcthashindex.cpp`` / ``unit/testindexbase.cpp``:

  SearchParams   src/index.h:36-148   (only the fields the hot path reads)
  Match          src/index.h:157-166  (Index::Match)
  MatchRange     src/media.h:62-78
  Media          src/media.h          (only id / dctHash / path: what DctHashIndex touches)
  DctHashIndex   src/dcthashindex.h:29-67, src/dcthashindex.cpp:30-250

All compute goes through the C-ABI in include/cbird_hip.h (libcbird_hip.so, HIP kernels for
gfx950).  There is no CPU implementation here: without the library or a device the methods
raise ``CbhError``.
"""
from __future__ import annotations

import ctypes as C
import warnings
from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

from . import _lib
from ._lib import CbhError, cbh_match, check


@dataclass
class MatchRange:
    """src/media.h:62-78"""
    srcIn: int = -1
    dstIn: int = -1
    len: int = 0


@dataclass
class Match:
    """Index::Match, src/index.h:157-166"""
    mediaId: int = 0
    score: int = 0
    range: MatchRange = field(default_factory=MatchRange)

    def __lt__(self, other: "Match") -> bool:  # src/index.h:284
        return self.score < other.score


@dataclass
class Media:
    """The slice of cbird's Media a DctHashIndex needle/haystack item carries."""
    id: int = 0
    dctHash: int = 0
    path: str = ""
    keyPointHashes: list = field(default_factory=list)  # KeyPointHashList (src/media.h)
    score: int = -1
    matchRange: MatchRange = field(default_factory=MatchRange)

    def isValid(self) -> bool:
        return self.id != 0


@dataclass
class SearchParams:
    """src/index.h:74-121 (defaults identical)"""
    AlgoDCT = 0
    AlgoDCTFeatures = 1
    AlgoCVFeatures = 2
    AlgoColor = 3
    AlgoVideo = 4

    algo: int = 0
    dctThresh: int = 5
    cvThresh: int = 25
    minMatches: int = 1
    maxMatches: int = 5
    maxThresh: int = 0
    filterSelf: bool = True
    verbose: bool = False
    # what Database::filterMatch / filterMatches read (src/index.h:97-119, defaults identical)
    path: str = ""            # subdirectory to accept / reject results from
    inPath: bool = False      # True = accept results from it, False = reject results from it
    filterGroups: bool = True
    filterParent: bool = False
    expandGroups: bool = False
    mergeGroups: int = 0


def _as_u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _as_u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


class DctHashIndex:
    """Index for 64-bit dct hashes that uses hamming distance (src/dcthashindex.h:26-29),
    resident on one MI355X."""

    def __init__(self, device: int = 0, _handle=None, shards=None) -> None:
        """shards = (device_mask, shards_per_device): one index over several GPUs / logical shards inside this
        process (cbh_idx64_create_sharded); None = one device (or the process default, _lib.set_default_sharding)"""
        self._L = _lib.lib()
        self._device = device
        self._id = SearchParams.AlgoDCT  # dcthashindex.cpp:31
        self._h = _handle if _handle is not None else _lib.create_idx64(device, shards)

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.cbh_idx64_destroy(h)

    # -- Index interface ------------------------------------------------------------------
    def id(self) -> int:
        return self._id

    def isLoaded(self) -> bool:
        return bool(self._L.cbh_idx64_is_loaded(self._h))

    def count(self) -> int:
        return int(self._L.cbh_idx64_count(self._h))

    def memoryUsage(self) -> int:
        return int(self._L.cbh_idx64_memory_usage(self._h))

    def load(self, hashes: Sequence[int], ids: Sequence[int]) -> None:
        """DctHashIndex::load (:70-114).  The reference runs `select id,phash_dct from media
        where type=1`; here the caller passes the two result columns."""
        if self.isLoaded():
            return  # `if (!isLoaded())` (:75)
        h, i = _as_u64(hashes), _as_u32(ids)
        if len(h) != len(i):
            raise ValueError("hashes/ids length mismatch")
        check(self._L.cbh_idx64_load(self._h, h.ctypes.data, i.ctypes.data, len(h)), "load")

    def load_device(self, d_hashes_ptr: int, d_ids_ptr: int, n: int, stream: int = 0) -> None:
        """Adopt device-resident columns (copy device-to-device into the index)."""
        check(self._L.cbh_idx64_load_dev(self._h, d_hashes_ptr, d_ids_ptr, n, stream), "load_dev")

    def add(self, media: Iterable[Media]) -> None:
        """DctHashIndex::add (:158-173)"""
        media = list(media)
        h = _as_u64([m.dctHash for m in media])
        i = _as_u32([m.id for m in media])
        check(self._L.cbh_idx64_add(self._h, h.ctypes.data, i.ctypes.data, len(h)), "add")

    def remove(self, ids: Sequence[int]) -> None:
        """DctHashIndex::remove (:175-191): nullify, never compact."""
        i = _as_u32(list(ids))
        check(self._L.cbh_idx64_remove(self._h, i.ctypes.data, len(i)), "remove")

    def mediaIds(self) -> set[int]:
        """loaded branch of DctHashIndex::mediaIds (:129-133)"""
        n = C.c_size_t(0)
        check(self._L.cbh_idx64_media_ids(self._h, None, 0, C.byref(n)), "mediaIds")
        out = np.zeros(max(1, n.value), np.uint32)
        check(self._L.cbh_idx64_media_ids(self._h, out.ctypes.data, len(out), C.byref(n)),
              "mediaIds")
        return set(int(x) for x in out[: n.value])

    def find(self, m: Media, p: SearchParams) -> list[Match]:
        """DctHashIndex::find (:193-220)"""
        target = int(m.dctHash)
        if not target:
            warnings.warn(f"no hash for needle: {m.path}")
            return []
        if self.count() == 0:
            warnings.warn("empty/null tree")
            return []
        cap = 64
        while True:
            buf = (cbh_match * cap)()
            n = C.c_size_t(0)
            check(self._L.cbh_idx64_find(self._h, target, int(p.dctThresh), buf, cap, C.byref(n)),
                  "find")
            if n.value <= cap:
                return [Match(buf[i].id, buf[i].score) for i in range(n.value)]
            cap = n.value

    def slice(self, mediaIds: Iterable[int]) -> "DctHashIndex":
        """DctHashIndex::slice (:222-250); the caller owns the result."""
        assert self.isLoaded()
        i = _as_u32(sorted(set(int(x) for x in mediaIds)))
        h = self._L.cbh_idx64_slice(self._h, i.ctypes.data, len(i))
        if not h:
            raise CbhError(self._L.cbh_last_error_code() or _lib.CBH_E_HIP, "slice")
        return DctHashIndex(self._device, _handle=h)

    # -- batched entry points (the MI355X-native shape of Database::similar's fan-out) ----------
    def find_batch(self, hashes: Sequence[int], thresh: int, max_per_query: int, masks=None):
        """For every needle hash: first `max_per_query` matches in (score, mediaId) order and the
        full match count.  Returns (ids[nq,k] u32, scores[nq,k] i32, counts[nq] u32).
        masks (optional, one u64 per needle): only entries with ((needle ^ hash) & mask) == 0 qualify."""
        q = _as_u64(hashes)
        nq, k = len(q), int(max_per_query)
        out = np.zeros((nq, max(k, 1), 2), np.uint32)
        counts = np.zeros(nq, np.uint32)
        mk = None if masks is None else _as_u64(masks)
        if mk is not None and len(mk) != nq:
            raise ValueError("one mask per needle")
        check(self._L.cbh_idx64_find_batch_masked(self._h, q.ctypes.data, None if mk is None else mk.ctypes.data,
                                                  nq, int(thresh), k, out.ctypes.data, counts.ctypes.data),
              "find_batch")
        out = out[:, :k, :]
        return out[:, :, 0].copy(), out[:, :, 1].astype(np.int32), counts

    def search_index_batch(self, hashes, needle_ids, p: "SearchParams", valid_ids=None):
        """Database::searchIndex (src/database.cpp:1691-1757) for every needle at once (cbh_search_index_batch):
        escalation to p.maxThresh, (score, mediaId) order, filterSelf, cut at p.maxMatches, ids outside `valid_ids`
        (the caller's idMap; None = all known) skipped.  Returns (ids[nq,maxMatches] u32, scores i32, counts u32)."""
        q, ni = _as_u64(hashes), _as_u32(needle_ids)
        nq, k = len(q), int(p.maxMatches)
        out = np.zeros((nq, max(k, 1), 2), np.uint32)
        counts = np.zeros(nq, np.uint32)
        v = None if valid_ids is None else np.unique(_as_u32(valid_ids))
        check(self._L.cbh_search_index_batch(self._h, q.ctypes.data, ni.ctypes.data, nq, int(p.dctThresh),
                                             int(p.maxThresh), int(p.minMatches), k, int(bool(p.filterSelf)),
                                             None if v is None else v.ctypes.data, 0 if v is None else len(v),
                                             out.ctypes.data, counts.ctypes.data), "search_index_batch")
        if k == 0:
            return np.zeros((nq, 0), np.uint32), np.zeros((nq, 0), np.int32), counts
        return out[:, :k, 0].copy(), out[:, :k, 1].astype(np.int32), counts

    def tree_masks(self, hashes: Sequence[int]) -> np.ndarray:
        """HammingTree leaf masks of needle hashes for the current contents (cbh_idx64_tree_masks)"""
        q = _as_u64(hashes)
        out = np.zeros(len(q), np.uint64)
        check(self._L.cbh_idx64_tree_masks(self._h, q.ctypes.data, len(q), out.ctypes.data), "tree_masks")
        return out

    def download(self):
        n = self.count()
        h = np.zeros(n, np.uint64)
        i = np.zeros(n, np.uint32)
        check(self._L.cbh_idx64_download(self._h, h.ctypes.data, i.ctypes.data, n), "download")
        return h, i

    def set_record_capacity(self, records: int) -> None:
        check(self._L.cbh_idx64_set_record_capacity(self._h, records), "set_record_capacity")

    def shard_count(self) -> int:
        return int(self._L.cbh_idx64_shard_count(self._h))

    def shard_counts(self) -> list:
        """slots held by every shard (a plain index: [count])"""
        return [int(self._L.cbh_idx64_count(self._L.cbh_idx64_shard(self._h, i))) for i in range(self.shard_count())]

    def shard_stats(self) -> "_lib.cbh_shard_stats":
        st = _lib.cbh_shard_stats()
        check(self._L.cbh_idx64_shard_stats(self._h, C.byref(st)), "shard_stats")
        return st

    @property
    def handle(self):
        return self._h

    @property
    def device(self) -> int:
        return self._device


class DctFeaturesIndex:
    """Index of a feature-based matcher using DCT hashes (src/dctfeaturesindex.h:31-38): up to 400
    keypoint hashes per image, stored as (mediaId, hash) entries; `find` votes over the 10 nearest
    entries of every needle hash (src/dctfeaturesindex.cpp:260-358)."""

    def __init__(self, device: int = 0, tree_compat: bool = False, _handle=None, shards=None) -> None:
        self._L = _lib.lib()
        self._device = device
        self._id = SearchParams.AlgoDCTFeatures  # dctfeaturesindex.cpp:84
        # False: exact candidates (a superset of the reference's).  True: every needle hash only sees the
        # entries of its HammingTree leaf (src/tree/hammingtree.h:244-252), i.e. the reference's own
        # approximate candidate sets, also on multi-leaf trees.
        self.tree_compat = bool(tree_compat)
        self._h = _handle if _handle is not None else _lib.create_idx64(device, shards)

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.cbh_idx64_destroy(h)

    def slice(self, mediaIds) -> "DctFeaturesIndex":
        """DctFeaturesIndex::slice (dctfeaturesindex.cpp:239-258): the values whose mediaId is in the set"""
        i = _as_u32(sorted(set(int(x) for x in mediaIds)))
        h = self._L.cbh_idx64_slice(self._h, i.ctypes.data, len(i))
        if not h:
            raise CbhError(self._L.cbh_last_error_code() or _lib.CBH_E_HIP, "slice")
        return DctFeaturesIndex(self._device, self.tree_compat, _handle=h)

    def id(self) -> int:
        return self._id

    def isLoaded(self) -> bool:
        return bool(self._L.cbh_idx64_is_loaded(self._h))

    def count(self) -> int:
        """_tree->size(): every inserted hash, removed ones included (dctfeaturesindex.cpp:93)"""
        return int(self._L.cbh_idx64_count(self._h))

    def memoryUsage(self) -> int:
        return int(self._L.cbh_idx64_memory_usage(self._h))

    @staticmethod
    def _flatten(media):
        ids, hashes = [], []
        for m in media:
            kp = list(getattr(m, "keyPointHashes", []) or [])
            ids += [m.id] * len(kp)
            hashes += kp
        return _as_u64(hashes), _as_u32(ids)

    def load(self, rows) -> None:
        """rows: iterable of (media_id, hashes) as `select media_id,hashes from kphash` yields them
        (dctfeaturesindex.cpp:129-156)."""
        ids, hashes = [], []
        for media_id, hs in rows:
            hs = list(hs)
            ids += [media_id] * len(hs)
            hashes += hs
        h, i = _as_u64(hashes), _as_u32(ids)
        check(self._L.cbh_idx64_load(self._h, h.ctypes.data, i.ctypes.data, len(h)), "load")

    def load_flat(self, hashes, ids) -> None:
        """the (index, hash) values of a HammingTree as stored in its leaves (the dctfeatures.cache file read by
        cbird_amd.indexdir.read_hamming_tree): entries with id 0 are removed ones that keep their hash"""
        h = np.ascontiguousarray(hashes, np.uint64)
        i = np.ascontiguousarray(ids, np.uint32)
        if len(h) != len(i):
            raise ValueError("hashes and ids differ in length")
        check(self._L.cbh_idx64_load(self._h, h.ctypes.data, i.ctypes.data, len(h)), "load")

    def add(self, media) -> None:
        """dctfeaturesindex.cpp:229-238"""
        media = list(media)
        if not media:
            return
        h, i = self._flatten(media)
        check(self._L.cbh_idx64_add(self._h, h.ctypes.data, i.ctypes.data, len(h)), "add")

    def remove(self, ids) -> None:
        """dctfeaturesindex.cpp:240-249 -> HammingTree::remove: index zeroed, hash kept"""
        ids = list(ids)
        if not ids or not self.isLoaded():
            return
        i = _as_u32(ids)
        check(self._L.cbh_idx64_remove_ids_only(self._h, i.ctypes.data, len(i)), "remove")

    def hashesForId(self, media_id: int):
        n = C.c_size_t(0)
        check(self._L.cbh_idx64_hashes_for_id(self._h, media_id, None, 0, C.byref(n)), "findIndex")
        out = np.zeros(max(1, n.value), np.uint64)
        check(self._L.cbh_idx64_hashes_for_id(self._h, media_id, out.ctypes.data, len(out), C.byref(n)),
              "findIndex")
        return out[: n.value]

    def find(self, needle, p: SearchParams) -> list[Match]:
        hashes = _as_u64(list(getattr(needle, "keyPointHashes", []) or []))
        if len(hashes) == 0 and needle.id > 0:
            hashes = self.hashesForId(needle.id)  # _tree->findIndex (:270-276)
        if len(hashes) == 0:
            warnings.warn(f"needle has no hashes {needle.id} {needle.path}")
            return []
        cap = len(hashes) * 10 + 1
        buf = (cbh_match * cap)()
        n = C.c_size_t(0)
        check(self._L.cbh_fdct_find_ex(self._h, hashes.ctypes.data, len(hashes), needle.id,
                                       int(p.dctThresh), int(self.tree_compat), buf, cap, C.byref(n)),
              "fdct_find")
        return [Match(buf[i].id, buf[i].score) for i in range(n.value)]

    def find_batch(self, needles, p: SearchParams):
        """All needles in one scan; returns a list (per needle) of lists of Match."""
        needles = list(needles)
        hs, offs, ids = [], [0], []
        for m in needles:
            kp = list(getattr(m, "keyPointHashes", []) or [])
            hs += kp
            offs.append(len(hs))
            ids.append(m.id)
        h, o, i = _as_u64(hs), _as_u64(offs), _as_u32(ids)
        cap = len(hs) * 10 + 1
        buf = (cbh_match * cap)()
        out_offs = np.zeros(len(needles) + 1, np.uint64)
        check(self._L.cbh_fdct_find_batch_ex(self._h, h.ctypes.data, o.ctypes.data, i.ctypes.data,
                                             len(needles), int(p.dctThresh), int(self.tree_compat), buf, cap,
                                             out_offs.ctypes.data), "fdct_find_batch")
        return [[Match(buf[j].id, buf[j].score) for j in range(int(out_offs[k]), int(out_offs[k + 1]))]
                for k in range(len(needles))]

    @property
    def handle(self):
        return self._h
