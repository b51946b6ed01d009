"""Randomised parity soak of the 256-bit Hamming scan kernels: cv::BFMatcher::radiusMatch (every match, sorted) through
k_hamm256_small / k_hamm256_mfma3 / k_hamm256_mfma / k_hamm256_scan on the same random index and queries must give the
same list, and that list must equal numpy's brute force on a sample of the queries.  Index rows are random; queries are
rows with a random number of flipped bits around the threshold (so that there are hits just under, at and just over
it), rows of zeros / ones, and unrelated descriptors.  Prints one JSON line.

    python tools/fuzz_scan256.py [--cases 40] [--seed 1]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PATHS = {"shipped": {b"scan256_mfma": 2, b"scan256_small": 1},
         "rows": {b"scan256_mfma": 2, b"scan256_small": 0},  # k_hamm256_mfma3 / k_hamm256_mfma without k_hamm256_small
         "valu": {b"scan256_mfma": 0, b"scan256_small": 1}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from cbird_amd import _lib
    from cbird_amd.cvfeatures import CvFeaturesIndex

    L = _lib.lib()
    rng = np.random.default_rng(args.seed)
    bad, total_matches, checked = [], 0, 0
    for c in range(args.cases):
        n_img = int(rng.integers(8, 400))
        per = int(rng.integers(20, 700))
        idx = CvFeaturesIndex()
        rows = rng.integers(0, 256, (n_img * per, 32), dtype=np.uint8)
        for i in range(n_img):
            _lib.check(L.cbh_idx256_add(idx.handle, i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
        n = len(rows)
        max_dist = int(rng.choice([0, 1, 5, 24, 25, 29, 39, 40, 41, 60, 90]))
        nq = int(rng.choice([1, 31, 32, 33, 95, 96, 97, 500, 512, 513, 1200, int(rng.integers(1, 2000))]))
        q = rows[rng.integers(0, n, nq)].copy()
        bits = np.unpackbits(q, axis=1)
        for j in range(nq):
            kind = rng.integers(0, 10)
            if kind < 7:  # flips around the threshold, anywhere in the 256 bits
                k = int(np.clip(max_dist + rng.integers(-3, 4), 0, 256))
                pos = rng.choice(256, k, replace=False)
                bits[j, pos] ^= 1
            elif kind == 7:
                bits[j] = rng.integers(0, 2, 256)
            elif kind == 8:
                bits[j] = 0
            else:  # all flips in the first 128 bits / in the last 128 bits
                k = int(np.clip(max_dist + rng.integers(-2, 3), 0, 128))
                pos = rng.choice(128, k, replace=False) + (128 if rng.integers(0, 2) else 0)
                bits[j, pos] ^= 1
        q = np.packbits(bits, axis=1)
        got = {}
        for name, knobs in PATHS.items():
            for k, v in knobs.items():
                L.cbh_set_tuning(k, v)
            m, first = idx.radius_match(q, max_dist)
            got[name] = (m, first)
        for k, v in PATHS["shipped"].items():
            L.cbh_set_tuning(k, v)
        L.cbh_set_tuning(b"scan256_mfma", 1)
        base = got["valu"]
        for name, (m, first) in got.items():
            if m.shape != base[0].shape or not (m == base[0]).all() or not (first == base[1]).all():
                bad.append({"case": c, "path": name, "n": n, "nq": nq, "max_dist": max_dist})
        total_matches += len(base[0])
        # numpy brute force on a sample of the queries
        rb = np.unpackbits(rows, axis=1).astype(np.int16)
        for j in rng.choice(nq, min(nq, 12), replace=False):
            d = (rb != bits[j].astype(np.int16)).sum(1)
            hit = np.flatnonzero(d <= max_dist)
            order = np.lexsort((hit, d[hit]))
            want = np.stack([np.full(len(hit), j), hit[order], d[hit][order]], 1).astype(np.int32)
            mine = base[0][int(base[1][j]): int(base[1][j + 1])]
            checked += 1
            if mine.shape != want.shape or not (mine == want).all():
                bad.append({"case": c, "path": "valu vs numpy", "query": int(j), "n": n, "max_dist": max_dist})
        del idx
    print(json.dumps({"cases": args.cases, "paths": list(PATHS), "matches": total_matches,
                      "queries_checked_against_numpy": checked, "mismatches": bad[:10], "ok": not bad}))
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
