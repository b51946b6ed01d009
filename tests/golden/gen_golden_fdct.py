#!/usr/bin/env python3
"""Golden vectors for DctFeaturesIndex::find from the REAL HammingTree (oracle/_ref, conda Qt5 build)
with the voting of src/dctfeaturesindex.cpp:285-358 restated on top of it (oracle/ref_wrap_qt.cpp).
The index is kept to a single tree leaf (<= 8192 entries), where the reference tree is exact, and
cases whose top-10 cut would split a tie of equal distances (unspecified std::sort order in the
reference) are dropped.     python tests/golden/gen_golden_fdct.py  -> tests/golden/fdct_single_leaf.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cbird_amd import synth  # noqa: E402
from oracle import RefHammingTree  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
M, K = 70, 110  # 7700 entries
h, _ = synth.make_hashes(M * K, seed=4242, planted_frac=0.35, max_dist=7)
ids = np.repeat(np.arange(1, M + 1, dtype=np.uint32), K)
rng = np.random.default_rng(9)
perm = rng.permutation(M * K)  # interleave media like incremental adds would
h, ids = h[perm], ids[perm]
removed = np.array([5, 17], np.uint32)
tree = RefHammingTree()
tree.insert(ids[:4000], h[:4000])
tree.insert(ids[4000:], h[4000:])
tree.remove(removed)
ids_after = ids.copy()
ids_after[np.isin(ids, removed)] = 0
needles, offs, nid, thr, res_off, res_id, res_sc = [], [0], [], [], [0], [], []
for needle in range(1, M + 1, 2):
    nh = h[ids == needle][:40]
    for t in (2, 5, 7):
        ok = True
        for x in nh.tolist():  # drop cases with a tie across the 10-cut
            _, _, d = tree.search(x, t)
            if len(d) > 10 and d[9] == d[10]:
                ok = False
        if not ok:
            continue
        ri, rs = tree.fdct_find(nh, needle, t)
        needles.append(nh)
        offs.append(offs[-1] + len(nh))
        nid.append(needle)
        thr.append(t)
        res_id.append(ri)
        res_sc.append(rs)
        res_off.append(res_off[-1] + len(ri))
np.savez_compressed(os.path.join(HERE, "fdct_single_leaf.npz"), hashes=h, ids=ids_after,
                    needle_hashes=np.concatenate(needles), needle_offs=np.asarray(offs, np.int64),
                    needle_ids=np.asarray(nid, np.uint32), thresh=np.asarray(thr, np.int32),
                    res_offs=np.asarray(res_off, np.int64), res_ids=np.concatenate(res_id),
                    res_scores=np.concatenate(res_sc))
print("cases", len(nid), "results", res_off[-1])
