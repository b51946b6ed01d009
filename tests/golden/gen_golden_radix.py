#!/usr/bin/env python3
"""Golden vectors for DctVideoIndex::findFrame from the REAL RadixMap_t (src/tree/radix.h:135-141 indexOf,
:187-210 search, compiled in place by oracle/ref_wrap_qt.cpp), radix 0 (one bucket: exact) and radix 10
(`-p.vradix 10`: a needle only sees the bucket of its bits 1..10).

Only the map is real; the generator keeps everything around it free of restated logic:
  * the clips' hashes all pass DctVideoIndex::insertHashes' filters on their own (skipFrames 0, >= 5 ones and zeros:
    asserted below), so the entry list is simply every frame of every clip in _mediaId order;
  * the nearest-frame-per-video step of findFrame (src/dctvideoindex.cpp:346-356: the FIRST match of the smallest
    distance in the map's own match order) is applied to the real map's matches right here.
Stored per (radix, needle, threshold): the raw matches as sorted (video, frame, distance) triples and the reduced
(mediaId, distance, frame) rows in mediaId order.

    python tests/golden/gen_golden_radix.py      # needs /root/reference (build container); -> radixmap_r0_r10.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cbird_amd import synth_video  # noqa: E402
from oracle import RefRadixMap  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
RADIXES = (0, 10)
THRESHOLDS = (3, 6, 9)

clips = synth_video.make_clips(60, 150, seed=77, subclip_frac=0.2, max_gap=8)
mids = np.arange(700, 700 + len(clips), dtype=np.uint32)
ev = np.concatenate([np.full(len(f), i, np.uint32) for i, (f, _) in enumerate(clips)])
ef = np.concatenate([f for f, _ in clips]).astype(np.uint32)
eh = np.concatenate([h for _, h in clips]).astype(np.uint64)
ones = np.array([bin(int(x)).count("1") for x in eh])
assert ones.min() >= 5 and ones.max() <= 59, "a hash would be dropped by insertHashes' filter"
rng = np.random.default_rng(78)
# needles: indexed frames, the same with one / two bits flipped INSIDE the radix-10 key (bits 1..10: another bucket),
# with bits flipped outside it (same bucket), and unrelated hashes
pick = rng.choice(len(eh), 90, replace=False)
q = eh[pick].copy()
q[30:50] ^= np.uint64(1) << rng.integers(1, 11, 20).astype(np.uint64)
q[50:70] ^= np.uint64(1) << rng.integers(11, 64, 20).astype(np.uint64)
q[70:80] ^= (np.uint64(1) << rng.integers(11, 64, 10).astype(np.uint64)) | (np.uint64(1) << rng.integers(11, 64, 10).astype(np.uint64))
q[80:] = rng.integers(0, 2**63, 10, dtype=np.uint64) << np.uint64(1)

out = {"clip_offs": np.cumsum([0] + [len(f) for f, _ in clips]).astype(np.int64), "frames": ef.astype(np.int32),
       "hashes": eh, "media_ids": mids, "needles": q, "radixes": np.asarray(RADIXES, np.int32),
       "thresholds": np.asarray(THRESHOLDS, np.int32)}
for radix in RADIXES:
    rm = RefRadixMap(radix)
    rm.insert(ev, ef, eh)
    for thr in THRESHOLDS:
        raw_off, raw, red_off, red = [0], [], [0], []
        for x in q.tolist():
            rv, rf, _, rd = rm.search(x, thr)
            nearest = {}
            for v, f, d in zip(rv.tolist(), rf.tolist(), rd.tolist()):
                if v not in nearest or d < nearest[v][0]:
                    nearest[v] = (d, f)
            trip = sorted(zip(rv.tolist(), rf.tolist(), rd.tolist()))
            raw += trip
            raw_off.append(len(raw))
            red += [(int(mids[v]), d, f) for v, (d, f) in sorted(nearest.items())]
            red_off.append(len(red))
        key = f"r{radix}_t{thr}"
        out[key + "_raw_offs"] = np.asarray(raw_off, np.int64)
        out[key + "_raw"] = np.asarray(raw, np.int64).reshape(-1, 3)
        out[key + "_frame_offs"] = np.asarray(red_off, np.int64)
        out[key + "_frame"] = np.asarray(red, np.int64).reshape(-1, 3)
        print(key, "raw matches", len(raw), "findFrame rows", len(red))
np.savez_compressed(os.path.join(HERE, "radixmap_r0_r10.npz"), **out)
