"""Same-box A/B of the needle chunk a workgroup of the 64-bit scan kernels streams ("scan_mfma_chunk": needle-tile pairs;
0 = the shipped 256 / 172 triples), alternating, on the bench's image-derived hashes, at one prefilter and one full threshold.
    python tools/ab/scan_chunk_ab.py [n=1000000] [rounds=5]"""
import ctypes as C, json, sys
import torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
import bench
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
out = torch.empty(N, dtype=torch.int64, device=dev)
for c0 in range(0, N, 100000):
    c1 = min(N, c0 + 100000)
    imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
    _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
    del imgs
idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx.load_device(out.data_ptr(), ids.data_ptr(), N)
cap = 1 << 24
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
chunks = (0, 64, 128, 512, 1024)
res = {f"dht{t}_chunk{c}": [] for t in (3, 6, 7) for c in chunks}
tot = {}
for r in range(R):
    for t in (3, 6, 7):
        for c in chunks:
            L.cbh_set_tuning(b"scan_mfma_chunk", c)
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, out.data_ptr(), N, t, drec.data_ptr(), cap, dtot.data_ptr(), 4, C.byref(ms)), "t")
            res[f"dht{t}_chunk{c}"].append(round(ms.value, 3))
            tot.setdefault(t, set()).add(int(dtot.item()))
L.cbh_set_tuning(b"scan_mfma_chunk", 0)
res["record_totals_agree"] = all(len(v) == 1 for v in tot.values())
print(json.dumps(res))
