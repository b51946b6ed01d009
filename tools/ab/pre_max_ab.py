"""Where the prefilter kernel stops paying: thresholds 7 and 8 on the bench's image-derived hashes with "scan_mfma_pre_max" 6
(shipped: both take the three-field 64-bit kernel), 7 and 8.  End of round 5: dht 7 17.1 ms full / 26.2 prefilter, dht 8 17.1 / 51.3.
    python tools/ab/pre_max_ab.py"""
import ctypes as C, json, sys
import torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
import bench
L = _lib.lib()
N = 1_000_000
dev = torch.device("cuda", 0)
out = torch.empty(N, dtype=torch.int64, device=dev)
for c0 in range(0, N, 100000):
    c1 = min(N, c0 + 100000)
    imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
    _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
    del imgs
idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx.load_device(out.data_ptr(), ids.data_ptr(), N)
cap = 1 << 25
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
res = {}
for pm in (6, 7, 8):
    L.cbh_set_tuning(b"scan_mfma_pre_max", pm)
    for t in (7, 8):
        for r in range(3):
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, out.data_ptr(), N, t, drec.data_ptr(), cap, dtot.data_ptr(), 4, C.byref(ms)), "t")
            res.setdefault(f"pre_max{pm}_dht{t}", []).append((round(ms.value, 3), int(dtot.item())))
print(json.dumps(res))
