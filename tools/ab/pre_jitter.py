"""Launch-to-launch spread of the low-word prefilter scan (k_hamm64_mfma PRE, dht <= 4) beside the full scan: 1M x 1M,
every launch timed on its own with HIP events; patterns: back to back, alternating with the other kernel, and after an
idle gap.  Prints one JSON line.    python tools/pre_jitter.py [n=1000000]"""
import ctypes as C, json, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from cbird_amd import DctHashIndex, _lib, synth

L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h, ids = synth.make_hashes(n, seed=1234)
idx = DctHashIndex()
idx.load(h, ids)
dq = torch.from_numpy(h.view(np.int64)).cuda()
cap = 1 << 22
rec = torch.zeros(cap, dtype=torch.int64, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
ms = C.c_float(0)


def one(dht):
    tot.zero_()
    _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), n, dht, rec.data_ptr(), cap, tot.data_ptr(), 1,
                                         C.byref(ms)), "time_scan")
    return round(ms.value, 3)


for _ in range(3):
    one(2), one(6)
out = {"n": n}
out["pre_back_to_back"] = [one(2) for _ in range(20)]
out["full_back_to_back"] = [one(6) for _ in range(20)]
alt = []
for _ in range(10):
    alt.append((one(2), one(6)))
out["alternating_pre_full"] = alt
gap = []
for _ in range(6):
    time.sleep(0.25)
    gap.append((one(2), one(2), one(2)))
out["pre_after_250ms_idle_then_two_more"] = gap
by_dht = {d: [one(d) for _ in range(6)] for d in (1, 2, 3, 4)}
out["pre_by_dht"] = by_dht
print(json.dumps(out))
