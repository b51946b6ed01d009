"""Scanner::processImage for a batch (/root/reference/src/scanner.cpp:828-895) over cbh_index_images: every feature
stage of an image chained on the device from one upload.  Field names follow IndexParams (src/scanner.h:60-75) and
the Media setters processImage calls."""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import check
from .colordesc import COLOR_DTYPE
from .orb import KP_DTYPE

AlgoDCT, AlgoDCTFeatures, AlgoCVFeatures, AlgoColor = 0, 1, 2, 3  # SearchParams::Algo* (src/index.h:41-47)


@dataclass
class IndexParams:
    algos: int = (1 << AlgoDCT) | (1 << AlgoDCTFeatures) | (1 << AlgoCVFeatures) | (1 << AlgoColor)
    autocrop: bool = True
    numFeatures: int = 400
    resizeLongestSide: int = 400


class _Params(C.Structure):
    _fields_ = [("autocrop_range", C.c_int), ("algos", C.c_int), ("resize_longest_side", C.c_int),
                ("num_features", C.c_int), ("kp_cap", C.c_int)]


@dataclass
class IndexResult:
    dctHash: int = 0
    cropRect: tuple = ()           # region autocrop kept: (left, top, right, bottom)
    resizedDims: tuple = (0, 0)    # size of the image the features were taken from
    keyPoints: np.ndarray = field(default_factory=lambda: np.zeros(0, KP_DTYPE))
    keyPointDescriptors: np.ndarray = field(default_factory=lambda: np.zeros((0, 32), np.uint8))
    keyPointHashes: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint64))
    colorDescriptor: np.ndarray | None = None


def process_images(imgs: np.ndarray, params: IndexParams | None = None, device: int = 0) -> list[IndexResult]:
    """imgs: uint8 [n, h, w] (grey) or [n, h, w, 3 | 4] (BGR / BGRA), one geometry.  One IndexResult per image."""
    p = params or IndexParams()
    imgs = np.asarray(imgs)
    if imgs.dtype != np.uint8 or imgs.ndim not in (3, 4):
        raise ValueError("expected uint8 [n, h, w] or [n, h, w, c]")
    imgs = np.ascontiguousarray(imgs)
    n, h, w = imgs.shape[:3]
    ch = 1 if imgs.ndim == 3 else imgs.shape[3]
    if n == 0:
        return []
    # room per image: ties in retainBest can exceed numFeatures; a call that found more is repeated, and the size that
    # worked is remembered (equal buffer sizes from call to call also keep the device's scratch pool reusable)
    cap = max(p.numFeatures + 64, getattr(process_images, "_cap_hint", {}).get(p.numFeatures, 0))
    L = _lib.lib()
    while True:
        cp = _Params(20 if (p.algos and p.autocrop) else -1, p.algos, p.resizeLongestSide, p.numFeatures, cap)
        # output buffers are kept between calls (an indexer calls this in a loop): fresh numpy arrays are untouched
        # pages, and a device-to-host copy into untouched pageable memory pays a page fault per 4 KB
        key = (n, cap)
        bufs = getattr(process_images, "_bufs", None)
        if bufs is None or bufs[0] != key:
            bufs = (key, np.zeros(n, np.uint64), np.zeros((n, 4), np.int32), np.zeros((n, 2), np.int32),
                    np.zeros(n, np.uint32), np.zeros((n, cap), KP_DTYPE), np.zeros((n, cap, 32), np.uint8),
                    np.zeros(n, np.uint32), np.zeros((n, cap), np.uint64), np.zeros(n, COLOR_DTYPE), np.zeros(n, np.uint8))
            for a in bufs[1:]:
                a.fill(0)  # touch
            process_images._bufs = bufs
        _, hashes, rects, dims, kpc, kp, desc, khc, kh, cd, cok = bufs
        hashes.fill(0), kpc.fill(0), khc.fill(0), cok.fill(0)
        t0 = time.perf_counter()
        check(L.cbh_index_images(imgs.ctypes.data, n, w, h, w * ch, w * h * ch, ch, C.byref(cp), hashes.ctypes.data,
                                 rects.ctypes.data, dims.ctypes.data, kpc.ctypes.data, kp.ctypes.data, desc.ctypes.data,
                                 khc.ctypes.data, kh.ctypes.data, cd.ctypes.data, cok.ctypes.data, device), "index_images")
        process_images.last_call_seconds = time.perf_counter() - t0  # the C call alone (the rest is Python unpacking)
        if int(kpc.max()) <= cap:
            break
        cap = (int(kpc.max()) + 63) // 64 * 64
    process_images._cap_hint = {**getattr(process_images, "_cap_hint", {}), p.numFeatures: cap}
    out = []
    for i in range(n):
        c = int(kpc[i])
        r = IndexResult(dctHash=int(hashes[i]), cropRect=tuple(int(v) for v in rects[i]),
                        resizedDims=(int(dims[i, 0]), int(dims[i, 1])), keyPoints=kp[i, :c].copy(),
                        keyPointHashes=kh[i, : int(khc[i])].copy())
        if p.algos & (1 << AlgoCVFeatures):
            r.keyPointDescriptors = desc[i, :c].copy()
        if (p.algos & (1 << AlgoColor)) and cok[i]:
            r.colorDescriptor = cd[i].copy()
        out.append(r)
    return out


def process_image_list(images, params: IndexParams | None = None, device: int = 0) -> list[IndexResult]:
    """A scanner's work list as it comes -- decoded images of ANY sizes and channel counts, in any order: the images
    are grouped by geometry (process_images / cbh_index_images takes one geometry per call; camera folders and
    IDCT-scaled decodes share few), each group goes through the device in one call, and the results come back in the
    order of the input.  uint8 arrays [h, w] or [h, w, 3 | 4]."""
    groups: dict = {}
    for i, im in enumerate(images):
        a = np.asarray(im)
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (3, 4)):
            raise ValueError(f"image {i}: expected uint8 [h, w] or [h, w, 3 | 4]")
        groups.setdefault(a.shape, []).append(i)
    out: list = [None] * len(images)
    for shape, idx in groups.items():
        batch = np.stack([np.asarray(images[i]) for i in idx])
        for i, r in zip(idx, process_images(batch, params, device)):
            out[i] = r
    return out
