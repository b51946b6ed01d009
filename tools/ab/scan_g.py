"""64-bit matrix-core scan kernels by accumulators in flight (cbh_set_tuning("scan_mfma_g", 1 | 2 | 4)) on the bench's
image-derived hashes: ms per n x n scan at a prefilter threshold (dht 2) and two three-field thresholds (dht 5, 8).
    python tools/scan_g.py [n=1000000]"""
import ctypes as C, json, sys
import torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
import bench
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
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
res = {}
for g in (2, 1, 4, 2):
    L.cbh_set_tuning(b"scan_mfma_g", g)
    for thr in (3, 6, 9):  # "distance < thr"
        best = 1e9
        for _ in range(3):
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, out.data_ptr(), N, thr, drec.data_ptr(), cap, dtot.data_ptr(), 3, C.byref(ms)), "t")
            best = min(best, ms.value)
        res.setdefault(f"g{g}", {})[f"thr{thr}"] = [round(best, 3), int(dtot.item())]
L.cbh_set_tuning(b"scan_mfma_g", 2)
print(json.dumps(res))
