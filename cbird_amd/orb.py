"""ORB keypoints and descriptors -- host mirror of Media::makeKeyPoints / Media::makeKeyPointDescriptors
(/root/reference/src/media.cpp:859-872) over the C-ABI (cbh_orb*, cbird_amd/csrc/orb.hip).

The 256 rBRIEF test pairs are OpenCV's learned table ``bit_pattern_31_`` (modules/features2d/src/orb.cpp).  It is
not part of cbird's sources and cannot be derived, so it is an input: ``load_pattern(path)`` reads it from an OpenCV
source file (or from a 1024-byte / 1024-integer dump), ``set_pattern`` hands it to the library.  ``synthetic_pattern``
is a seeded stand-in with the same shape for tests and benchmarks -- descriptors made with it are NOT comparable with
a cbird index.
"""
from __future__ import annotations

import re

import numpy as np

from . import _lib
from ._lib import check

KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])  # struct cbh_keypoint


def synthetic_pattern(seed: int = 31) -> np.ndarray:
    """A stand-in for bit_pattern_31_: 256 point pairs drawn like BRIEF's G II layout (isotropic Gaussian, sigma =
    patch/5) and clipped to OpenCV's range [-13, 13].  NOT OpenCV's table."""
    rng = np.random.default_rng(seed)
    xy = np.clip(np.rint(rng.normal(0.0, 31 / 5.0, (256, 4))), -13, 13).astype(np.int8)
    same = (xy[:, 0] == xy[:, 2]) & (xy[:, 1] == xy[:, 3])
    xy[same, 2] = np.where(xy[same, 2] < 13, xy[same, 2] + 1, xy[same, 2] - 1)  # a test needs two distinct points
    return xy.reshape(1024)


def load_pattern(path: str) -> np.ndarray:
    """bit_pattern_31_ from (a) OpenCV's orb.cpp: the initialiser of ``static int bit_pattern_31_[256*4]`` with its
    /*mean ..., correlation ...*/ comments, (b) a text file of 1024 integers, or (c) a raw 1024-byte int8 dump."""
    raw = open(path, "rb").read()
    if len(raw) == 1024:
        return np.frombuffer(raw, np.int8).copy()
    text = raw.decode("utf-8", "replace")
    m = re.search(r"bit_pattern_31_\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", text, re.S)
    body = m.group(1) if m else text
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    body = re.sub(r"//[^\n]*", " ", body)
    vals = [int(v) for v in re.findall(r"-?\d+", body)]
    if len(vals) != 1024:
        raise ValueError(f"{path}: expected 1024 integers (256 x (x0, y0, x1, y1)), found {len(vals)}")
    a = np.array(vals, np.int64)
    if np.abs(a).max() > 15:
        raise ValueError(f"{path}: coordinates outside [-15, 15]")
    return a.astype(np.int8)


def set_pattern(xy) -> None:
    xy = np.ascontiguousarray(xy, np.int8).reshape(1024)
    check(_lib.lib().cbh_orb_set_pattern(xy.ctypes.data), "orb_set_pattern")


def _pack(images):
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    if any(im.ndim != 2 or im.size == 0 for im in imgs):
        raise ValueError("expected non-empty single-channel 2-D uint8 images")
    sizes = np.array([im.size for im in imgs], np.uint64)
    off = np.zeros(len(imgs), np.uint64)
    off[1:] = np.cumsum((sizes[:-1] + np.uint64(15)) // np.uint64(16) * np.uint64(16))
    total = int(off[-1] + sizes[-1])
    buf = np.zeros(total, np.uint8)
    for im, o in zip(imgs, off):
        buf[int(o): int(o) + im.size] = im.reshape(-1)
    w = np.array([im.shape[1] for im in imgs], np.uint32)
    h = np.array([im.shape[0] for im in imgs], np.uint32)
    return buf, total, off, w, h


def orb(images, num_keypoints: int = 400, descriptors: bool = True, device: int = 0, kp_cap: int | None = None):
    """makeKeyPoints (+ makeKeyPointDescriptors) for a list of 2-D uint8 grey images of any sizes.
    Returns a list of (keypoints KP_DTYPE[k], keypoints_xy_after_compute float32[k, 2] | None, descriptors
    uint8[k, 32] | None).  retainBest can keep ties at the cut (include/cbird_hip.h, orb_retain_order), so an image can return more than
    num_keypoints."""
    n = len(images)
    if n == 0:
        return []
    buf, total, off, w, h = _pack(images)
    cap = int(kp_cap) if kp_cap else int(num_keypoints) + 64
    L = _lib.lib()
    while True:
        kp = np.zeros((n, cap), KP_DTYPE)
        after = np.zeros((n, cap, 2), np.float32) if descriptors else None
        desc = np.zeros((n, cap, 32), np.uint8) if descriptors else None
        counts = np.zeros(n, np.uint32)
        check(L.cbh_orb(buf.ctypes.data, total, n, off.ctypes.data, w.ctypes.data, h.ctypes.data, w.ctypes.data,
                        int(num_keypoints), cap, kp.ctypes.data, after.ctypes.data if descriptors else None,
                        desc.ctypes.data if descriptors else None, counts.ctypes.data, device), "orb")
        if int(counts.max()) <= cap:
            break
        cap = int(counts.max())
    out = []
    for i in range(n):
        c = int(counts[i])
        out.append((kp[i, :c].copy(), after[i, :c].copy() if descriptors else None,
                    desc[i, :c].copy() if descriptors else None))
    return out


def make_keypoints(images, num_keypoints: int = 400, device: int = 0):
    """Media::makeKeyPoints for a batch: a list of KP_DTYPE arrays"""
    return [r[0] for r in orb(images, num_keypoints, descriptors=False, device=device)]


def make_keypoint_descriptors(images, keypoints, device: int = 0):
    """Media::makeKeyPointDescriptors (/root/reference/src/media.cpp:868-872) for a batch: keypoints is a list of
    KP_DTYPE arrays (what make_keypoints returned for the same images).  Returns a list of (keypoints as compute()
    leaves them in the reference's non-const list, descriptors uint8[k, 32])."""
    n = len(images)
    if n != len(keypoints):
        raise ValueError("one keypoint array per image")
    if n == 0:
        return []
    buf, total, off, w, h = _pack(images)
    kps = [np.ascontiguousarray(k, KP_DTYPE).reshape(-1) for k in keypoints]
    first = np.zeros(n + 1, np.uint32)
    first[1:] = np.cumsum([len(k) for k in kps])
    nk = int(first[-1])
    kp = np.concatenate(kps) if nk else np.zeros(1, KP_DTYPE)
    out_kp = np.zeros(max(1, nk), KP_DTYPE)
    desc = np.zeros((max(1, nk), 32), np.uint8)
    out_first = np.zeros(n + 1, np.uint32)
    check(_lib.lib().cbh_orb_describe(buf.ctypes.data, total, n, off.ctypes.data, w.ctypes.data, h.ctypes.data,
                                      w.ctypes.data, kp.ctypes.data, first.ctypes.data, out_kp.ctypes.data,
                                      desc.ctypes.data, out_first.ctypes.data, device), "orb_describe")
    return [(out_kp[int(out_first[i]): int(out_first[i + 1])].copy(),
             desc[int(out_first[i]): int(out_first[i + 1])].copy()) for i in range(n)]
