"""Synthetic workloads of SURVEY.md section 8(d): seeded 64-bit hashes with planted neighbours and
smooth-field 8-bit images with near-duplicates.  numpy only (host); bench.py has the on-device
twin of `make_images` for the 1M-image configuration."""
from __future__ import annotations

import numpy as np


def make_hashes(n: int, seed: int = 1234, planted_frac: float = 0.05, max_dist: int = 8):
    """Uniform u64 hashes shaped like dct hashes (bit 0 clear, non-zero) with `planted_frac` of the
    entries replaced by a copy of an earlier entry with 0..max_dist flipped bits (bits 1..63).
    Returns (hashes u64[n], ids u32[n]) with ids = 1..n (SQLite rowids start at 1)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    h = rng.integers(0, 2**64, n, dtype=np.uint64) & np.uint64(0xFFFFFFFFFFFFFFFE)
    h[h == 0] = np.uint64(2)
    n_pl = int(n * planted_frac)
    if n_pl and n > 1:
        dst = rng.choice(np.arange(1, n), size=min(n_pl, n - 1), replace=False)
        src = (rng.random(len(dst)) * dst).astype(np.int64)  # src < dst
        nflip = rng.integers(0, max_dist + 1, len(dst))
        for d, s, f in zip(dst.tolist(), src.tolist(), nflip.tolist()):
            v = int(h[s])
            if f:
                for b in rng.choice(np.arange(1, 64), size=f, replace=False).tolist():
                    v ^= 1 << b
            h[d] = np.uint64(v if v else 2)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    return h, ids


def make_images(n: int, w: int = 256, h: int = 256, seed: int = 1234, dup_frac: float = 0.10):
    """n u8 images [n,h,w]: smooth random field (8 low-frequency cosines, random phase/amplitude)
    + uniform noise in [-8,8], clipped; the last dup_frac*n images are near-duplicates of earlier
    ones (noise sigma=2 and a brightness shift in [-5,5])."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32),
                         indexing="ij")
    imgs = np.empty((n, h, w), np.uint8)
    n_dup = int(n * dup_frac)
    n_base = n - n_dup
    for i in range(n_base):
        f = np.full((h, w), 128.0, np.float32)
        for _ in range(8):
            fx, fy = rng.uniform(-4, 4, 2)
            amp = rng.uniform(4, 40)
            ph = rng.uniform(0, 2 * np.pi)
            f += amp * np.cos(2 * np.pi * (fx * xx / w + fy * yy / h) + ph)
        f += rng.uniform(-8, 8, (h, w)).astype(np.float32)
        imgs[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    for i in range(n_base, n):
        s = int(rng.integers(0, max(1, n_base)))
        f = imgs[s].astype(np.float32) + rng.normal(0, 2, (h, w)).astype(np.float32)
        f += float(rng.integers(-5, 6))
        imgs[i] = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return imgs
