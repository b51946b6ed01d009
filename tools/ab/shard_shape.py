"""scan timing at the 8-GPU shard shape: 125k slots x 1M needles (development aid)"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth
L = _lib.lib()
h, ids = synth.make_hashes(1_000_000, seed=1234)
dq = torch.from_numpy(h.view(np.int64)).cuda()
cap = 1 << 22
drec = torch.empty(cap, dtype=torch.int64, device="cuda"); dtot = torch.zeros(1, dtype=torch.int64, device="cuda")
ms = C.c_float(0)
for shard in (125_000, 250_000, 500_000, 1_000_000):
    idx = cbird_amd.DctHashIndex(); idx.load(h[:shard], ids[:shard])
    row = []
    for thr in (1, 2, 6, 8):
        _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), len(h), thr, drec.data_ptr(), cap, dtot.data_ptr(), 1, C.byref(ms)), "w")
        _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), len(h), thr, drec.data_ptr(), cap, dtot.data_ptr(), 3, C.byref(ms)), "t")
        row.append(f"dht{thr} {ms.value:7.2f} ms ({shard*1e6/ms.value*1e3:.2e}/s)")
    print(shard, " | ".join(row), flush=True)
