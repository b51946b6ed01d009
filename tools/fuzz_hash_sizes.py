"""Randomised parity soak of dctHash64 over image geometries at batch sizes that take the SHIPPED kernel choice
(k_dcthash_256, the fused / split register-streaming strip kernels, the band kernels): n images of a random w x h
(32..1100, a third of them multiples of 8; a quarter with integer resize ratios, 32..1952 wide) hashed in one call, every hash compared
with the oracle (threads over the host cores).  Prints one JSON line.

    python tools/fuzz_hash_sizes.py [--cases 30] [--seed 1] [--pixels 60000000]
"""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pixels", type=int, default=60_000_000, help="pixels per case (sets the batch size)")
    args = ap.parse_args()
    from cbird_amd.hashing import dct_hash64_batch
    from oracle import Oracle

    orc = Oracle()
    rng = np.random.default_rng(args.seed)
    cores = len(os.sched_getaffinity(0))
    bad, images, geos = [], 0, []
    for c in range(args.cases):
        kind = rng.integers(0, 4)
        if kind == 0:
            w, h = 32 * int(rng.integers(1, 62)), 32 * int(rng.integers(1, 40))  # integer ratios (w <= 1952: k_band_area's block-sum walk)
        elif kind == 1:
            w, h = 8 * int(rng.integers(4, 138)), int(rng.integers(32, 900))
        elif kind == 2 and c % 5 == 0:
            w, h = int(rng.integers(2049, 6000)), int(rng.integers(32, 400))  # wider than one workgroup: column strips
        else:
            w, h = int(rng.integers(32, 1100)), int(rng.integers(32, 900))
        n = int(np.clip(args.pixels // (w * h), 64, 6000))
        imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        if c % 3 == 0:  # smooth content: hashes that are not noise
            yy, xx = np.mgrid[0:h, 0:w]
            base = (127 + 90 * np.sin(xx / rng.uniform(3, 60)) * np.cos(yy / rng.uniform(3, 60))).astype(np.int16)
            imgs = np.clip(base[None] + (imgs.astype(np.int16) - 128) // int(rng.integers(2, 30)), 0, 255).astype(np.uint8)
        got = dct_hash64_batch(imgs)
        with ThreadPoolExecutor(cores) as ex:
            want = np.concatenate(list(ex.map(orc.dcthash64_batch, np.array_split(imgs, min(n, cores * 2)))))
        images += n
        geos.append([w, h, n])
        if not (got == want).all():
            bad.append({"case": c, "w": w, "h": h, "n": n, "differing": int((got != want).sum())})
    print(json.dumps({"cases": args.cases, "images": images, "geometries": geos, "mismatches": bad, "ok": not bad}))
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
