#!/usr/bin/env python3
"""Convert the text output of tools/gen_golden_opencv.cpp (run where OpenCV 2.4.13.7 exists) into
tests/golden/opencv_hash.npz, which tests/test_opencv_golden.py consumes when present.

    python tools/opencv_golden_to_npz.py opencv_hash.txt tests/golden/opencv_hash.npz

Also home of gen_image(), the Python twin of the C++ tool's integer-only image generator: the npz stores outputs
only, the tests regenerate the inputs from (w, h, seed)."""
from __future__ import annotations

import sys

import numpy as np

GEOMETRIES = [(32, 32), (64, 64), (256, 256), (96, 64), (40, 36), (100, 100), (127, 129), (300, 200), (640, 480),
              (1568, 64), (3885, 33), (31, 31), (16, 16), (20, 100), (5, 31), (400, 300), (512, 512), (1920, 1080)]

M32 = 0xFFFFFFFF


def _mix(x):
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & M32
    x = x ^ (x >> 16)
    x = (x * 0x45D9F3B) & M32
    return x ^ (x >> 16)


def _tri(t):
    t = t & 1023
    return np.where(t < 512, t - 256, 768 - t)


def gen_image(w: int, h: int, seed: int) -> np.ndarray:
    """u8 [h, w]; identical to gen_image() in tools/gen_golden_opencv.cpp (all arithmetic in integers)"""
    s = (seed * 2654435761 + 12345) & M32
    par = []
    for _ in range(4):
        row = []
        for mod, add in ((7, 1), (7, 1), (1024, 0), (30, 10)):
            s = (s * 1664525 + 1013904223) & M32
            row.append(add + (s >> 8) % mod)
        par.append(row)
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    acc = np.zeros((h, w), np.int64)
    for fx, fy, ph, amp in par:
        acc += amp * _tri((x * fx * 1024) // w + (y * fy * 1024) // h + ph)
    noise = ((_mix((((y * w + x) & M32) * 2654435761 + seed) & M32) >> 24) & 15) - 8
    v = 128 + np.floor_divide(acc, 256) + noise
    return np.clip(v, 0, 255).astype(np.uint8)


def checksums(img: np.ndarray):
    f = img.reshape(-1).astype(np.uint64)
    return int(f.sum()), int((f * (np.arange(len(f), dtype=np.uint64) % 251 + 1)).sum())


def parse(path: str) -> dict:
    out = {}
    H = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "V":
            out["cv_version"] = np.array(t[1])
        elif t[0] == "H":
            w, h, seed = int(t[1]), int(t[2]), int(t[3])
            H.append((w, h, seed, int(t[4], 16), int(t[5], 16), [int(c, 16) for c in t[6:70]], bytes.fromhex(t[70])))
        elif t[0] == "G":
            w, h = int(t[1]), int(t[2])
            out["gray_whs"] = np.array([w, h, int(t[3])], np.int64)
            out["gray"] = np.frombuffer(bytes.fromhex(t[4]), np.uint8).reshape(h, w)
        elif t[0] == "L":
            ow, oh = int(t[5]), int(t[6])
            out["lanczos_whs_size"] = np.array([int(t[1]), int(t[2]), int(t[3]), int(t[4])], np.int64)
            out["lanczos"] = np.frombuffer(bytes.fromhex(t[7]), np.uint8).reshape(oh, ow)
        elif t[0] == "R":
            n = int(t[4])
            out["rect_whs"] = np.array([int(t[1]), int(t[2]), int(t[3])], np.int64)
            out["rects"] = np.array([[int(t[5 + 4 * i]), int(t[6 + 4 * i]), int(t[7 + 4 * i])] for i in range(n)], np.int32)
            out["rect_hashes"] = np.array([int(t[8 + 4 * i], 16) for i in range(n)], np.uint64)
            out["rect_after_sum"] = np.array([int(t[5 + 4 * n])], np.uint64)
        elif t[0] == "P":
            dw, dh = int(t[4]), int(t[5])
            out["pyr_whs_dims"] = np.array([int(t[1]), int(t[2]), int(t[3]), dw, dh], np.int64)
            out["pyr"] = np.frombuffer(bytes.fromhex(t[6]), np.uint8).reshape(dh, dw)
        elif t[0] == "B":
            w, h = int(t[1]), int(t[2])
            out["gauss_whs"] = np.array([w, h, int(t[3])], np.int64)
            out["gauss"] = np.frombuffer(bytes.fromhex(t[4]), np.uint8).reshape(h, w)
        elif t[0] == "F":
            n = int(t[4])
            out["fast_whs"] = np.array([int(t[1]), int(t[2]), int(t[3])], np.int64)
            out["fast"] = np.array([int(v) for v in t[5: 5 + 3 * n]], np.int32).reshape(n, 3)
        elif t[0] == "O":
            n = int(t[5])
            out["orb_whs_nfeat"] = np.array([int(t[1]), int(t[2]), int(t[3]), int(t[4])], np.int64)
            rec = t[6: 6 + 6 * n]
            out["orb_kp_bits"] = np.array([[int(rec[6 * i + j], 16) for j in range(5)] for i in range(n)], np.uint32).reshape(n, 5)
            out["orb_octave"] = np.array([int(rec[6 * i + 5]) for i in range(n)], np.int32)
            out["orb_desc"] = np.frombuffer(bytes.fromhex(t[6 + 6 * n]) if n else b"", np.uint8).reshape(n, 32)
        elif t[0] == "A":
            n = int(t[1])
            out["atan_bits"] = np.array([int(v, 16) for v in t[2: 2 + 3 * n]], np.uint32).reshape(n, 3)
        elif t[0] == "M":
            cols, rows = int(t[1]), int(t[2])
            out[f"mask_{cols}x{rows}"] = np.frombuffer(bytes.fromhex(t[3]), np.uint8).reshape(rows, cols)
        elif t[0] == "U":
            n = int(t[1])
            rec = t[2: 2 + 6 * n]
            out["luv_bgr"] = np.array([[int(rec[6 * i + j]) for j in range(3)] for i in range(n)], np.uint8)
            out["luv_bits"] = np.array([[int(rec[6 * i + 3 + j], 16) for j in range(3)] for i in range(n)], np.uint32)
        elif t[0] == "K":
            n = int(t[1])
            out["kmeans_labels"] = np.array([int(v) for v in t[3: 3 + n]], np.int32)
            out["kmeans_center_bits"] = np.array([int(v, 16) for v in t[3 + n: 3 + n + 96]], np.uint32).reshape(32, 3)
    if not H:
        return out
    out["hash_whs"] = np.array([(a, b, c) for a, b, c, *_ in H], np.int64)
    out["hashes"] = np.array([x[3] for x in H], np.uint64)
    out["thresh_bits"] = np.array([x[4] for x in H], np.uint32)
    out["coef_bits"] = np.array([x[5] for x in H], np.uint32)
    out["tiles"] = np.stack([np.frombuffer(x[6], np.uint8).reshape(32, 32) for x in H])
    return out


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    d = parse(sys.argv[1])
    np.savez_compressed(sys.argv[2], **d)
    print({k: (v.shape, v.dtype) for k, v in d.items()})
