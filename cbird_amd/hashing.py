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


def template_scores(cands: np.ndarray, tmpl: np.ndarray, device: int = 0):
    """TemplateMatcher::match's score (src/templatematcher.cpp:331-374) for n candidate patches against one template:
    cands uint8 [n,h,w] or [n,h,w,3|4] as warpAffine left them (0 outside the patch), tmpl uint8 [h,w] or [h,w,3|4].
    Returns (scores int32[n] = hamm64(candHash, tmplHash), candHashes u64[n], tmplHashes u64[n])."""
    cands = np.ascontiguousarray(cands)
    tmpl = np.ascontiguousarray(tmpl)
    if cands.dtype != np.uint8 or cands.ndim not in (3, 4) or tmpl.dtype != np.uint8 or tmpl.ndim not in (2, 3):
        raise ValueError("expected uint8 cands [n,h,w(,c)] and tmpl [h,w(,c)]")
    n, h, w = cands.shape[:3]
    if tmpl.shape[:2] != (h, w):
        raise ValueError("the candidate patches are template-sized (warpAffine's dsize = tmplImg.size())")
    cc = 1 if cands.ndim == 3 else cands.shape[3]
    tc = 1 if tmpl.ndim == 2 else tmpl.shape[2]
    scores = np.zeros(n, np.int32)
    ch = np.zeros(n, np.uint64)
    th = np.zeros(n, np.uint64)
    if n:
        check(_lib.lib().cbh_template_scores(cands.ctypes.data, n, w, h, w * cc, w * h * cc, cc, tmpl.ctypes.data, w * tc,
                                             tc, ch.ctypes.data, th.ctypes.data, scores.ctypes.data, device),
              "template_scores")
    return scores, ch, th


def process_images_ex(imgs: np.ndarray, autocrop: int | None = 20, resize: int = 400, device: int = 0):
    """process_images plus, from the same upload, sizeLongestSide(cvGray, resize) of every (autocropped) grey image --
    what Scanner::processImage hands to ORB (src/scanner.cpp:876).  Returns (hashes, rects, list of uint8 images)."""
    imgs = np.ascontiguousarray(imgs)
    if imgs.dtype != np.uint8 or imgs.ndim not in (3, 4):
        raise ValueError("expected uint8 [n,h,w] or [n,h,w,c]")
    n, h, w = imgs.shape[:3]
    ch = 1 if imgs.ndim == 3 else imgs.shape[3]
    out = np.zeros(n, np.uint64)
    rects = np.zeros((n, 4), np.int32)
    slots = np.zeros((n, resize * resize), np.uint8)
    dims = np.zeros((n, 2), np.int32)
    if n:
        check(_lib.lib().cbh_process_images_ex(imgs.ctypes.data, n, w, h, w * ch, w * h * ch, ch,
                                               -1 if autocrop is None else int(autocrop), out.ctypes.data,
                                               rects.ctypes.data, int(resize), slots.ctypes.data, dims.ctypes.data,
                                               device), "process_images_ex")
    small = [slots[i, : dims[i, 0] * dims[i, 1]].reshape(dims[i, 1], dims[i, 0]).copy() for i in range(n)]
    return out, rects, small


def keypoint_rects(cols: int, rows: int, keypoints) -> np.ndarray:
    """The rectangles Media::makeKeyPointHashes derives from keypoints (src/media.cpp:880-901): keypoints is
    [k, 3] float32 (pt.x, pt.y, size); returns int32 [m, 3] (x, y, side)."""
    kp = np.ascontiguousarray(keypoints, np.float32).reshape(-1, 3)
    r = np.zeros((max(1, len(kp)), 3), np.int32)
    n = _lib.lib().cbh_keypoint_rects(int(cols), int(rows), kp.ctypes.data, len(kp), r.ctypes.data)
    if n < 0:
        check(int(n), "keypoint_rects")
    return r[:n].copy()


def make_keypoint_hashes(images, keypoints, device: int = 0, return_images: bool = False):
    """Media::makeKeyPointHashes (src/media.cpp:874-923) for a batch: images is a list of 2-D uint8 grey images (any
    sizes), keypoints a list of [k_i, 3] float32 arrays (pt.x, pt.y, size) in detector order.  Returns a list of
    uint64 arrays (one hash per accepted keypoint, in order); with return_images also the images as the in-place
    blurs of dctHash64(sub, inPlace=true) left them."""
    if len(images) != len(keypoints):
        raise ValueError("one keypoint array per image")
    n = len(images)
    if n == 0:
        return ([], []) if return_images else []
    imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
    if any(im.ndim != 2 or im.size == 0 for im in imgs):
        raise ValueError("expected non-empty single-channel 2-D uint8 images")
    kps = [np.ascontiguousarray(k, np.float32).reshape(-1, 3) for k in keypoints]
    sizes = np.array([im.size for im in imgs], np.uint64)
    off = np.zeros(n, np.uint64)
    off[1:] = np.cumsum((sizes[:-1] + np.uint64(15)) // np.uint64(16) * np.uint64(16))  # 16-byte aligned starts
    total = int(off[-1] + sizes[-1])
    buf = np.zeros(total, np.uint8)
    for im, o in zip(imgs, off):
        buf[int(o): int(o) + im.size] = im.reshape(-1)
    w = np.array([im.shape[1] for im in imgs], np.uint32)
    h = np.array([im.shape[0] for im in imgs], np.uint32)
    kp_first = np.zeros(n + 1, np.uint32)
    kp_first[1:] = np.cumsum([len(k) for k in kps])
    kp = np.concatenate(kps) if kp_first[-1] else np.zeros((1, 3), np.float32)
    out = np.zeros(max(1, int(kp_first[-1])), np.uint64)
    out_first = np.zeros(n + 1, np.uint32)
    after = np.zeros(total, np.uint8) if return_images else None
    check(_lib.lib().cbh_keypoint_hashes(buf.ctypes.data, total, n, off.ctypes.data, w.ctypes.data, h.ctypes.data,
                                         w.ctypes.data, kp.ctypes.data, kp_first.ctypes.data, out.ctypes.data,
                                         out_first.ctypes.data, after.ctypes.data if return_images else None,
                                         device), "keypoint_hashes")
    hashes = [out[int(out_first[i]): int(out_first[i + 1])].copy() for i in range(n)]
    if not return_images:
        return hashes
    return hashes, [after[int(o): int(o) + im.size].reshape(im.shape).copy() for im, o in zip(imgs, off)]


def size_longest_side(imgs: np.ndarray, size: int = 400, device: int = 0) -> np.ndarray:
    """sizeLongestSide(img, size) (src/cvutil.cpp:1932-1950, INTER_LANCZOS4) for uint8 grey images [n, h, w] of one
    geometry; returns uint8 [n, h', w'].  size defaults to IndexParams::resizeLongestSide (src/scanner.h:71)."""
    import ctypes as C

    imgs = np.asarray(imgs)
    if imgs.dtype != np.uint8 or imgs.ndim != 3:
        raise ValueError("expected uint8 array [n, h, w]")
    if imgs.strides[2] != 1:
        imgs = np.ascontiguousarray(imgs)
    n, h, w = imgs.shape
    ow, oh = C.c_int(0), C.c_int(0)
    L = _lib.lib()
    L.cbh_longest_side_dims(w, h, int(size), C.byref(ow), C.byref(oh))
    out = np.zeros((n, max(oh.value, 0), max(ow.value, 0)), np.uint8)
    check(L.cbh_size_longest_side(imgs.ctypes.data, n, w, h, imgs.strides[1], imgs.strides[0] if n else 0, int(size),
                                  out.ctypes.data, C.byref(ow), C.byref(oh), device), "size_longest_side")
    return out
