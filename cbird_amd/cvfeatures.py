"""Host-side mirror of CvFeaturesIndex (src/cvfeaturesindex.{h,cpp}): 256-bit ORB descriptors, 10 nearest
rows per needle descriptor, median-based score.  Descriptor *generation* (OpenCV ORB, src/media.cpp:859-872)
is outside the hot path: needles and haystack carry pre-packed 32-byte rows."""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import CbhError, cbh_match, check
from .index import Match, SearchParams


class CvFeaturesIndex:
    KNN = 10  # `_index->knnSearch(descriptors, ..., 10)` (cvfeaturesindex.cpp:497)

    def __init__(self, device: int = 0, shards=None) -> None:
        """shards = (device_mask, shards_per_device): one index sharded by image over several GPUs / logical shards in
        this process (cbh_idx256_create_sharded); None = one device (or the process default, _lib.default_sharding)"""
        self._L = _lib.lib()
        self._device = device
        self._id = SearchParams.AlgoCVFeatures
        shards = shards if shards is not None else _lib.default_sharding()
        self._h = self._L.cbh_idx256_create_sharded(shards[0], shards[1]) if shards else self._L.cbh_idx256_create(device)
        if not self._h:
            raise CbhError(_lib.CBH_E_NODEVICE, "cbh_idx256_create")

    def shard_rows(self) -> list:
        return [int(self._L.cbh_idx256_shard_rows(self._h, i)) for i in range(self._L.cbh_idx256_shard_count(self._h))]

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.cbh_idx256_destroy(h)

    def id(self) -> int:
        return self._id

    def isLoaded(self) -> bool:
        return bool(self._L.cbh_idx256_is_loaded(self._h))

    def count(self) -> int:
        return int(self._L.cbh_idx256_count(self._h))

    def memoryUsage(self) -> int:
        return int(self._L.cbh_idx256_memory_usage(self._h))

    @staticmethod
    def _rows(desc) -> np.ndarray:
        d = np.ascontiguousarray(desc, np.uint8)
        if d.ndim != 2 or d.shape[1] != 32:
            raise ValueError("descriptors must be [n, 32] uint8")
        return d

    def add(self, media) -> None:
        """cvfeaturesindex.cpp:122-150: media carry (id, keyPointDescriptors [n,32] u8)"""
        for m in media:
            d = getattr(m, "keyPointDescriptors", None)
            if d is None or len(d) == 0:
                warnings.warn(f"no descriptors for {m.path}")
                continue
            d = self._rows(d)
            check(self._L.cbh_idx256_add(self._h, m.id, d.ctypes.data, len(d)), "add")

    load = add  # load(): one add per `matrix` row, ascending media id (:167-250)

    def remove(self, ids) -> None:
        i = np.ascontiguousarray(list(ids), np.uint32)
        check(self._L.cbh_idx256_remove(self._h, i.ctypes.data, len(i)), "remove")

    def descriptorsForMediaId(self, media_id: int) -> np.ndarray:
        f, c = C.c_size_t(0), C.c_size_t(0)
        check(self._L.cbh_idx256_rows_of(self._h, media_id, C.byref(f), C.byref(c)), "rows_of")
        out = np.zeros((c.value, 32), np.uint8)
        if c.value:
            check(self._L.cbh_idx256_download_rows(self._h, f.value, c.value, out.ctypes.data), "download_rows")
        return out

    def slice(self, mediaIds) -> "CvFeaturesIndex":
        """CvFeaturesIndex::slice (cvfeaturesindex.cpp:285-312): the descriptors of the given media, ascending id"""
        chunk = CvFeaturesIndex(self._device)
        for mid in sorted(set(int(x) for x in mediaIds)):
            d = self.descriptorsForMediaId(mid)
            if len(d):
                check(self._L.cbh_idx256_add(chunk._h, mid, np.ascontiguousarray(d).ctypes.data, len(d)), "add")
        return chunk

    def knn(self, needles, k: int, thresh: int):
        d = self._rows(needles)
        nq = len(d)
        row = np.zeros((nq, k), np.uint32)
        dist = np.zeros((nq, k), np.uint16)
        cnt = np.zeros(nq, np.uint32)
        check(self._L.cbh_idx256_knn(self._h, d.ctypes.data, nq, k, int(thresh), row.ctypes.data,
                                     dist.ctypes.data, cnt.ctypes.data), "knn")
        return row, dist, cnt

    def knn_media(self, needles, k: int, thresh: int):
        """knn + the mediaId of every candidate row (0 = removed): the shard-local step of ShardedCvFeaturesIndex"""
        d = self._rows(needles)
        nq = len(d)
        row = np.zeros((nq, k), np.uint32)
        dist = np.zeros((nq, k), np.uint16)
        media = np.zeros((nq, k), np.uint32)
        cnt = np.zeros(nq, np.uint32)
        check(self._L.cbh_idx256_knn_media(self._h, d.ctypes.data, nq, k, int(thresh), row.ctypes.data,
                                           dist.ctypes.data, media.ctypes.data, cnt.ctypes.data), "knn_media")
        return row, dist, media, cnt

    def radius_match(self, queries, max_dist: int):
        """cv::BFMatcher(NORM_HAMMING).radiusMatch(queries, matches, max_dist) with the index rows as the train set
        (TemplateMatcher, src/templatematcher.cpp:134,217).  Returns (matches int32 [m, 3] = queryIdx, trainIdx,
        distance, grouped by query in ascending (distance, trainIdx) order; first int64 [nq + 1])."""
        d = np.ascontiguousarray(queries, np.uint8).reshape(-1, 32)
        nq = len(d)
        first = np.zeros(nq + 1, np.uint64)
        cap = max(1024, 4 * nq)
        while True:
            out = np.zeros((cap, 3), np.int32)
            rc = self._L.cbh_idx256_radius_match(self._h, d.ctypes.data, nq, int(max_dist), out.ctypes.data, cap,
                                                 first.ctypes.data)
            if rc == _lib.CBH_E_OVERFLOW:
                cap = int(first[-1])
                continue
            check(rc, "radius_match")
            return out[: int(first[-1])].copy(), first.astype(np.int64)

    def find(self, needle, p: SearchParams):
        d = getattr(needle, "keyPointDescriptors", None)
        if d is None or len(d) == 0:
            d = self.descriptorsForMediaId(needle.id)  # (:443)
        if len(d) == 0:
            warnings.warn(f"needle has no descriptors {needle.id} {needle.path}")
            return []
        if self.count() <= 0:
            warnings.warn("empty index")
            return []
        d = self._rows(d)
        cap = len(d) * self.KNN + 1
        buf = (cbh_match * cap)()
        n = C.c_size_t(0)
        check(self._L.cbh_idx256_find(self._h, d.ctypes.data, len(d), int(p.cvThresh), self.KNN, buf, cap,
                                      C.byref(n)), "find")
        return [Match(buf[i].id, buf[i].score) for i in range(n.value)]

    def find_batch(self, needles, p: SearchParams):
        needles = list(needles)
        rows, offs = [], [0]
        for m in needles:
            d = self._rows(m.keyPointDescriptors)
            rows.append(d)
            offs.append(offs[-1] + len(d))
        allr = np.ascontiguousarray(np.concatenate(rows) if rows else np.zeros((0, 32), np.uint8))
        o = np.ascontiguousarray(offs, np.uint64)
        cap = len(allr) * self.KNN + 1
        buf = (cbh_match * cap)()
        out_offs = np.zeros(len(needles) + 1, np.uint64)
        check(self._L.cbh_idx256_find_batch(self._h, allr.ctypes.data, o.ctypes.data, len(needles),
                                            int(p.cvThresh), self.KNN, buf, cap, out_offs.ctypes.data),
              "find_batch")
        return [[Match(buf[j].id, buf[j].score) for j in range(int(out_offs[i]), int(out_offs[i + 1]))]
                for i in range(len(needles))]

    @property
    def handle(self):
        return self._h
