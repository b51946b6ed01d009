"""What multi-index hashing would have to compare on the bench's own hashes (an estimate for NOTES 15, nothing is built):
for a threshold t (distance < t, i.e. at most d = t - 1 differing bits) split the 64 bits into m = max(4, t) chunks; two
hashes within distance d agree on at least one whole chunk (pigeonhole), so only pairs that share a chunk value need the
64-bit comparison.  Candidate pairs of the 10^6 x 10^6 self-join = sum over chunks j and values v of H_j[v]^2 -- printed per
threshold beside the 10^12 of the exhaustive scan, for the bench's image-derived hashes and for uniform random ones.
    python tools/ab/mih_estimate.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
import bench

L = _lib.lib()
dev = torch.device("cuda", 0)
N = 1_000_000


def image_hashes():
    out = torch.empty(N, dtype=torch.int64, device=dev)
    for c0 in range(0, N, 100000):
        c1 = min(N, c0 + 100000)
        imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
        _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
        del imgs
    return out.cpu().numpy().view(np.uint64)


def candidates(h, m):
    edges = [round(64 * j / m) for j in range(m + 1)]
    tot = 0
    worst = 0
    for j in range(m):
        lo, hi = edges[j], edges[j + 1]
        key = ((h >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)).astype(np.int64)
        cnt = np.bincount(key, minlength=1 << (hi - lo)).astype(np.float64)
        tot += float((cnt * cnt).sum())
        worst = max(worst, int(cnt.max()))
    return tot, worst


sets = {"bench (image-derived)": image_hashes(),
        "uniform": np.random.default_rng(1).integers(0, 1 << 63, N, dtype=np.uint64) << np.uint64(1)}
for name, h in sets.items():
    print(name)
    for t in range(1, 9):
        m = max(4, t)
        c, worst = candidates(h, m)
        print(f"  dht {t}: {m} chunks of {64 // m}-{-(-64 // m)} bits: {c:.3e} candidate pairs = 1 / {1e12 / c:.0f} of the scan's 1e12; fullest bucket {worst}")
