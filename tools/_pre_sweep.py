"""Development aid: the 64-bit matrix-core scan variants (prefilter on/forced/off, tiles per wave, tiles per group)
on 1M uniform hashes at a few thresholds."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth
L = _lib.lib(); N = 1_000_000; dev = torch.device("cuda", 0)
dq = torch.from_numpy(synth.make_hashes(N, seed=1234)[0].view(np.int64)).to(dev)
idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx.load_device(dq.data_ptr(), ids.data_ptr(), N)
cap = 1 << 24
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
for thr in (1, 3, 5, 6):
    for ht, g, pre in ((8, 2, 2), (8, 2, 0)):
        L.cbh_set_tuning(b"scan_mfma_ht", ht); L.cbh_set_tuning(b"scan_mfma_g", g); L.cbh_set_tuning(b"scan_mfma_pre", pre)
        for it in (1, 3):
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap, dtot.data_ptr(), it, C.byref(ms)), "t")
        print(f"thr {thr} ht {ht} g {g} pre {pre}: {ms.value:7.2f} ms  ({int(dtot.item()) // 3} rec)", flush=True)
