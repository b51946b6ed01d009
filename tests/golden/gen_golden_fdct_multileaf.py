#!/usr/bin/env python3
"""Golden vectors for the HammingTree-compatible search on a MULTI-LEAF tree, from the REAL
HammingTree (oracle/_ref, conda Qt5 build): 60k entries (8+ leaves of <= 8192), inserted in chunks like
DctFeaturesIndex::load does, a few media removed.  Stored: per needle hash the real tree's candidate set
(index, distance) and, per needle image, DctFeaturesIndex::find's result (oracle/ref_wrap_qt.cpp voting
over the real tree).  Cases whose top-10 cut would split a tie of equal distances are dropped (unspecified
std::sort order in the reference).
    python tests/golden/gen_golden_fdct_multileaf.py -> tests/golden/fdct_multileaf.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cbird_amd import synth  # noqa: E402
from oracle import RefHammingTree  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
M, K = 600, 100  # 60000 entries
h, _ = synth.make_hashes(M * K, seed=777, planted_frac=0.4, max_dist=7)
ids = np.repeat(np.arange(1, M + 1, dtype=np.uint32), K)
rng = np.random.default_rng(12)
perm = rng.permutation(M * K)
h, ids = h[perm], ids[perm]
removed = np.array([9, 250], np.uint32)
tree = RefHammingTree()
for c0 in range(0, M * K, 17000):  # several inserts: splits happen at different times
    tree.insert(ids[c0:c0 + 17000], h[c0:c0 + 17000])
tree.remove(removed)
ids_after = ids.copy()
ids_after[np.isin(ids, removed)] = 0

# candidate sets of single needle hashes
cq = np.concatenate([h[rng.integers(0, M * K, 200)], rng.integers(1, 1 << 63, 40, dtype=np.uint64)])
cq[:60] ^= np.uint64(1) << rng.integers(1, 64, 60).astype(np.uint64)
# dct hashes never have bit 0 set (cvutil.cpp:537-538), so the tree's bit-0 child on that side is an EMPTY leaf
# and the reference's unrolled leaf scan reads out of bounds there (`count - 4` on a size_t, hammingtree.h:262);
# needles keep bit 0 clear like every real needle does
cq &= ~np.uint64(1)
c_off, c_idx, c_dist = [0], [], []
THR = 9
for x in cq.tolist():
    i, _, d = tree.search(x, THR, cap=1 << 17)
    o = np.lexsort((i, d))
    c_idx.append(i[o])
    c_dist.append(d[o])
    c_off.append(c_off[-1] + len(i))

# whole find() results
needles, offs, nid, thr, res_off, res_id, res_sc = [], [0], [], [], [0], [], []
for needle in range(1, M + 1, 11):
    nh = h[ids == needle][:40]
    for t in (3, 6):
        ok = True
        for x in nh.tolist():
            _, _, d = tree.search(x, t, cap=1 << 17)
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
np.savez_compressed(os.path.join(HERE, "fdct_multileaf.npz"), hashes=h, ids=ids_after,
                    cand_q=cq, cand_thresh=np.int32(THR), cand_offs=np.asarray(c_off, np.int64),
                    cand_idx=np.concatenate(c_idx), cand_dist=np.concatenate(c_dist).astype(np.int32),
                    needle_hashes=np.concatenate(needles), needle_offs=np.asarray(offs, np.int64),
                    needle_ids=np.asarray(nid, np.uint32), thresh=np.asarray(thr, np.int32),
                    res_offs=np.asarray(res_off, np.int64), res_ids=np.concatenate(res_id),
                    res_scores=np.concatenate(res_sc))
print("candidate needles", len(cq), "candidates", c_off[-1], "find cases", len(nid), "results", res_off[-1])
