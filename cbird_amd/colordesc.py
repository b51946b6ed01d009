"""Host-side mirror of ColorDescIndex / ColorDescriptor (src/colordescindex.{h,cpp}, src/cvutil.h:57-113).
Descriptors are the reference's 258-byte records; create_descriptors() is ColorDescriptor::create for a batch
(k-means over Luv pixels, src/cvutil.cpp:790-1099; what can and cannot match the cbird binary is stated in
include/cbird_hip.h)."""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import CbhError, cbh_match, check
from .index import Match, SearchParams

DESC_BYTES = 258
COLOR_DTYPE = np.dtype([("colors", np.uint16, (32, 4)), ("numColors", np.uint8), ("_pad", np.uint8)])
assert COLOR_DTYPE.itemsize == DESC_BYTES


def make_descriptor(luvw, num_colors=None) -> np.ndarray:
    """luvw: [n<=32, 4] uint16 rows (l,u,v,w compressed as in DescriptorColor) -> one 258-byte record"""
    d = np.zeros((), COLOR_DTYPE)
    luvw = np.asarray(luvw, np.uint16).reshape(-1, 4)
    d["colors"][: len(luvw)] = luvw
    d["numColors"] = len(luvw) if num_colors is None else num_colors
    return d


def create_descriptors(images, device: int = 0):
    """ColorDescriptor::create (src/cvutil.cpp:790-1099) for a list of uint8 BGR / BGRA images [h, w, 3 | 4] of any
    sizes (one channel count per call).  Returns (descriptors COLOR_DTYPE[n], ok bool[n]); ok is False where the
    reference returns without touching the descriptor ("not enough colors")."""
    n = len(images)
    if n == 0:
        return np.zeros(0, COLOR_DTYPE), np.zeros(0, bool)
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    ch = imgs[0].shape[2] if imgs[0].ndim == 3 else 0
    if ch not in (3, 4) or any(im.ndim != 3 or im.shape[2] != ch or im.size == 0 for im in imgs):
        raise ValueError("expected non-empty uint8 images [h, w, 3] or [h, w, 4] (BGR / BGRA), one channel count")
    sizes = np.array([im.size for im in imgs], np.uint64)
    off = np.zeros(n, np.uint64)
    off[1:] = np.cumsum((sizes[:-1] + np.uint64(15)) // np.uint64(16) * np.uint64(16))
    total = int(off[-1] + sizes[-1])
    buf = np.zeros(total, np.uint8)
    for im, o in zip(imgs, off):
        buf[int(o): int(o) + im.size] = im.reshape(-1)
    w = np.array([im.shape[1] for im in imgs], np.uint32)
    h = np.array([im.shape[0] for im in imgs], np.uint32)
    stride = (w * np.uint32(ch)).astype(np.uint32)
    descs = np.zeros(n, COLOR_DTYPE)
    ok = np.zeros(n, np.uint8)
    check(_lib.lib().cbh_color_descriptors(buf.ctypes.data, total, n, off.ctypes.data, w.ctypes.data, h.ctypes.data,
                                           stride.ctypes.data, ch, descs.ctypes.data, ok.ctypes.data, device),
          "color_descriptors")
    return descs, ok.astype(bool)


class ColorDescIndex:
    def __init__(self, device: int = 0) -> None:
        self._L = _lib.lib()
        self._device = device
        self._id = SearchParams.AlgoColor
        self._h = self._L.cbh_color_create(device)
        if not self._h:
            raise CbhError(_lib.CBH_E_NODEVICE, "cbh_color_create")

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.cbh_color_destroy(h)

    def id(self) -> int:
        return self._id

    def isLoaded(self) -> bool:
        return bool(self._L.cbh_color_is_loaded(self._h))

    def count(self) -> int:
        return int(self._L.cbh_color_count(self._h))

    def memoryUsage(self) -> int:
        return int(self._L.cbh_color_memory_usage(self._h))

    def add(self, media) -> None:
        media = list(media)
        if not media:
            return
        ids = np.ascontiguousarray([m.id for m in media], np.uint32)
        descs = np.ascontiguousarray(np.stack([np.asarray(m.colorDescriptor, COLOR_DTYPE) for m in media]))
        check(self._L.cbh_color_add(self._h, ids.ctypes.data, descs.ctypes.data, len(ids)), "add")

    load = add

    def remove(self, ids) -> None:
        i = np.ascontiguousarray(list(ids), np.uint32)
        check(self._L.cbh_color_remove(self._h, i.ctypes.data, len(i)), "remove")

    def slice(self, mediaIds) -> "ColorDescIndex":
        """ColorDescIndex::slice (colordescindex.cpp:231-248): entries whose mediaId is in the set, index order"""
        n = self.count()
        ids = np.zeros(max(n, 1), np.uint32)
        descs = np.zeros(max(n, 1), COLOR_DTYPE)
        if n:
            check(self._L.cbh_color_download(self._h, ids.ctypes.data, descs.ctypes.data, n), "download")
        keep = np.isin(ids[:n], np.array(sorted(set(int(x) for x in mediaIds)), np.uint32))
        chunk = ColorDescIndex(self._device)
        if keep.any():
            ki, kd = np.ascontiguousarray(ids[:n][keep]), np.ascontiguousarray(descs[:n][keep])
            check(self._L.cbh_color_add(chunk._h, ki.ctypes.data, kd.ctypes.data, len(ki)), "add")
        return chunk

    def findIndexData(self, m) -> bool:
        d = np.zeros((), COLOR_DTYPE)
        buf = np.zeros(DESC_BYTES, np.uint8)
        if self._L.cbh_color_find_index_data(self._h, m.id, buf.ctypes.data) == 1:
            m.colorDescriptor = buf.view(COLOR_DTYPE)[0]
            return True
        return False

    def find(self, m, p: SearchParams | None = None):
        target = getattr(m, "colorDescriptor", None)
        if target is None or int(np.asarray(target, COLOR_DTYPE)["numColors"]) <= 0:
            if not self.findIndexData(m):
                warnings.warn(f"needle has no color descriptor {m.id} {m.path}")
                return []
            target = m.colorDescriptor
            if int(target["numColors"]) <= 0:
                return []
        t = np.ascontiguousarray(np.asarray(target, COLOR_DTYPE).reshape(1))
        cap = max(1, self.count())
        buf = (cbh_match * cap)()
        n = C.c_size_t(0)
        check(self._L.cbh_color_find(self._h, t.ctypes.data, buf, cap, C.byref(n)), "find")
        return [Match(buf[i].id, buf[i].score) for i in range(n.value)]

    def distances(self, descs) -> np.ndarray:
        """ColorDescriptor::distance (cvutil.cpp:682-749) of every given needle descriptor against every index entry
        (add order), as float32 [nq, count]; FLT_MAX where the reference returns FLT_MAX"""
        d = np.ascontiguousarray(descs, COLOR_DTYPE).reshape(-1)
        out = np.zeros((len(d), self.count()), np.float32)
        check(self._L.cbh_color_distances(self._h, d.ctypes.data, len(d), out.ctypes.data), "color_distances")
        return out

    def find_batch(self, descs, k: int):
        d = np.ascontiguousarray(np.asarray(descs, COLOR_DTYPE).reshape(-1))
        nq = len(d)
        out = np.zeros((nq, max(k, 1), 2), np.uint32)
        counts = np.zeros(nq, np.uint32)
        check(self._L.cbh_color_find_batch(self._h, d.ctypes.data, nq, k, out.ctypes.data, counts.ctypes.data),
              "find_batch")
        return out[:, :k, 0].copy(), out[:, :k, 1].astype(np.int32), counts
