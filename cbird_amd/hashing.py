"""dctHash64 over batches of decoded 8-bit grayscale tiles (src/cvutil.cpp:435-545).

``dct_hash64`` mirrors the reference call (one image -> one u64); ``dct_hash64_batch`` is the
MI355X-native shape: many pre-decoded images per launch.  Both go through
``cbh_dcthash_batch`` in libcbird_hip.so; there is no CPU implementation here.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check


def dct_hash64_batch(imgs: np.ndarray, device: int = 0) -> np.ndarray:
    """imgs: u8 array [n, h, w] (C-contiguous rows; row/image strides are taken from the array).
    Returns u64[n]."""
    imgs = np.asarray(imgs)
    if imgs.dtype != np.uint8 or imgs.ndim != 3:
        raise ValueError("expected uint8 array [n, h, w]")
    if imgs.strides[2] != 1:
        imgs = np.ascontiguousarray(imgs)
    n, h, w = imgs.shape
    out = np.zeros(n, np.uint64)
    if n == 0:
        return out
    check(_lib.lib().cbh_dcthash_batch(imgs.ctypes.data, n, w, h, imgs.strides[1], imgs.strides[0],
                                       out.ctypes.data, device), "dcthash_batch")
    return out


def dct_hash64(img: np.ndarray, device: int = 0) -> int:
    """uint64_t dctHash64(const cv::Mat& cvImg) for an 8UC1 image."""
    img = np.asarray(img)
    if img.ndim != 2:
        raise ValueError("expected a single-channel 2-D uint8 image")
    return int(dct_hash64_batch(img[None, ...], device)[0])


def process_images(imgs: np.ndarray, autocrop: int | None = 20, device: int = 0):
    """The hash Scanner::processImage stores for decoded images (src/scanner.cpp:852-862): grayscale ->
    autocrop(gray, 20) when enabled -> dctHash64.  imgs: uint8 [n,h,w] (gray) or [n,h,w,3|4] (BGR/BGRA).
    Returns (hashes u64[n], rects int32[n,4] = left, top, right, bottom of the kept region)."""
    imgs = np.ascontiguousarray(imgs)
    if imgs.dtype != np.uint8 or imgs.ndim not in (3, 4):
        raise ValueError("expected uint8 [n,h,w] or [n,h,w,c]")
    n, h, w = imgs.shape[:3]
    ch = 1 if imgs.ndim == 3 else imgs.shape[3]
    out = np.zeros(n, np.uint64)
    rects = np.zeros((n, 4), np.int32)
    if n:
        check(_lib.lib().cbh_process_images(imgs.ctypes.data, n, w, h, w * ch, w * h * ch, ch,
                                            -1 if autocrop is None else int(autocrop), out.ctypes.data,
                                            rects.ctypes.data, device), "process_images")
    return out, rects
