"""Synthetic clips of SURVEY.md section 8(d) config 5: per clip a random-walk of 64-bit frame hashes (each
next hash flips 0-3 bits of the previous one), strictly increasing frame numbers with gaps 1..30; a
fraction of the clips are sub-clips of earlier ones (optionally with a few bits of noise)."""
from __future__ import annotations

import numpy as np


def make_clips(n_clips: int, n_frames: int = 300, seed: int = 1234, subclip_frac: float = 0.01,
               noise_bits: int = 1, max_gap: int = 30):
    rng = np.random.default_rng(seed)
    clips = []
    n_sub = int(round(n_clips * subclip_frac))
    for c in range(n_clips):
        if c >= n_clips - n_sub and c > 0:
            src_frames, src_hashes = clips[int(rng.integers(0, max(1, n_clips - n_sub)))]
            m = len(src_frames)
            a = int(rng.integers(0, max(1, m // 3)))
            b = int(rng.integers(min(m, a + m // 2), m + 1))
            frames = (src_frames[a:b] - src_frames[a]).astype(np.int32)
            hashes = src_hashes[a:b].copy()
            if noise_bits:
                flip = rng.integers(1, 64, (len(hashes), noise_bits))
                on = rng.random((len(hashes), noise_bits)) < 0.3
                for k in range(noise_bits):
                    hashes ^= np.where(on[:, k], np.uint64(1) << flip[:, k].astype(np.uint64), np.uint64(0))
        else:
            h = np.empty(n_frames, np.uint64)
            cur = int(rng.integers(0, 2**63)) * 2
            for i in range(n_frames):
                for _ in range(int(rng.integers(0, 4))):
                    cur ^= 1 << int(rng.integers(1, 64))
                h[i] = cur
            gaps = rng.integers(1, max_gap + 1, n_frames)
            gaps[0] = 0
            frames = np.cumsum(gaps).astype(np.int32)
            hashes = h
        clips.append((frames, hashes))
    return clips


def make_clips_fast(n_clips: int, n_frames: int = 300, seed: int = 1234, subclip_frac: float = 0.01,
                    noise_bits: int = 1, max_gap: int = 30):
    """Vectorised twin of make_clips for bench-sized sets (10k clips x 300 frames in well under a second); same
    recipe -- random-walk hashes (0-3 single-bit flips per frame, bit 0 never set), frame gaps 1..max_gap, the last
    subclip_frac of the clips are noisy sub-clips of earlier ones -- from a different random stream."""
    rng = np.random.default_rng(seed)
    n_sub = int(round(n_clips * subclip_frac))
    n_base = n_clips - n_sub
    start = (rng.integers(0, 2**63, n_base, dtype=np.uint64) << np.uint64(1))
    nflip = rng.integers(0, 4, (n_base, n_frames))
    step = np.zeros((n_base, n_frames), np.uint64)
    for k in range(3):
        bit = rng.integers(1, 64, (n_base, n_frames)).astype(np.uint64)
        step ^= np.where(nflip > k, np.uint64(1) << bit, np.uint64(0))
    step[:, 0] ^= start
    hashes = np.bitwise_xor.accumulate(step, axis=1)
    gaps = rng.integers(1, max_gap + 1, (n_base, n_frames))
    gaps[:, 0] = 0
    frames = np.cumsum(gaps, axis=1).astype(np.int32)
    clips = [(frames[i], hashes[i]) for i in range(n_base)]
    for _ in range(n_sub):
        f, h = clips[int(rng.integers(0, n_base))]
        a = int(rng.integers(0, n_frames // 3))
        b = int(rng.integers(a + n_frames // 2, n_frames + 1))
        hh = h[a:b].copy()
        if noise_bits:
            on = rng.random(len(hh)) < 0.3
            hh ^= np.where(on, np.uint64(1) << rng.integers(1, 64, len(hh)).astype(np.uint64), np.uint64(0))
        clips.append(((f[a:b] - f[a]).astype(np.int32), hh))
    return clips
