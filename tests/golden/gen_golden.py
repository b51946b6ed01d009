#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference search structure.

Runs only in the build container (needs /root/reference to compile oracle/_ref).  For seeded
hash sets with planted neighbours it records, for every needle and every dht in 1..8, the
result of the reference's DctTree::search (src/tree/dcttree.h:124-137 over
src/tree/vptree.h) -- ids and distances -- canonicalised to (distance, id) order because the
reference's tie order is heap order (unspecified), and with mediaId 0 dropped (slots nulled by
DctHashIndex::remove; Database::searchIndex discards them at src/database.cpp:1743-1755 with a
warning, and the brute-force statement of find() skips them, src/dcthashindex.cpp:215).

    python tests/golden/gen_golden.py        # rewrites tests/golden/vptree_*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cbird_amd import synth  # noqa: E402
from oracle import RefTree  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gen(name: str, n: int, nq: int, seed: int, n_removed: int) -> None:
    h, ids = synth.make_hashes(n, seed=seed, planted_frac=0.10, max_dist=8)
    rng = np.random.default_rng(seed + 1)
    # a few exact duplicates, a few low-popcount hashes, and removed (id=0, hash=0) slots
    dup = rng.choice(n, 8, replace=False)
    h[dup[:4]] = h[dup[4:]]
    h[rng.choice(n, 3, replace=False)] = np.array([2, 6, 0x8000000000000000], np.uint64)
    rm = rng.choice(n, n_removed, replace=False)
    h[rm] = 0
    ids[rm] = 0
    # needles: indexed hashes (incl. removed -> 0 needles), perturbed ones and unrelated ones
    qi = rng.choice(n, nq - nq // 4, replace=False)
    q = h[qi].copy()
    extra = rng.integers(0, 2**64, nq // 4, dtype=np.uint64) & np.uint64(0xFFFFFFFFFFFFFFFE)
    q = np.concatenate([q, extra])
    flip = rng.integers(1, 64, len(q))
    mask = rng.random(len(q)) < 0.3
    q[mask] ^= (np.uint64(1) << flip[mask].astype(np.uint64))
    tree = RefTree(h, ids)
    out = {"hashes": h, "ids": ids, "queries": q}
    for dht in range(1, 9):
        offs = [0]
        rid, rd = [], []
        for t in q.tolist():
            if t == 0:  # DctHashIndex::find returns nothing for a null needle
                offs.append(offs[-1])
                continue
            i, d = tree.search(t, dht)
            keep = i != 0
            i, d = i[keep], d[keep]
            order = np.lexsort((i, d))
            rid.append(i[order])
            rd.append(d[order])
            offs.append(offs[-1] + len(i))
        out[f"offs_{dht}"] = np.asarray(offs, np.int64)
        out[f"ids_{dht}"] = np.concatenate(rid).astype(np.uint32) if rid else np.zeros(0, np.uint32)
        out[f"dist_{dht}"] = np.concatenate(rd).astype(np.int8) if rd else np.zeros(0, np.int8)
        print(name, "dht", dht, "matches", offs[-1])
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    gen("vptree_n4096.npz", 4096, 512, 20260101, 16)
    gen("vptree_n32768.npz", 32768, 1024, 20260102, 64)
