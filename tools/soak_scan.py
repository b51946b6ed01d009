"""Repeatability soak of the matrix-core scan kernels (or, third argument 4, of the bucketed join): many launches, every
result set compared with the VALU kernel's (order-independent checksum of the record multiset).  Development aid.
    python tools/soak_scan.py [N=300000] [seconds=60] [scan_mfma=1]"""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # "scan_mfma" of the runs under test: 1 as shipped, 4 the bucketed join for thresholds <= 8
dev = torch.device("cuda", 0)
h = synth.make_hashes(N, seed=99, planted_frac=0.3)[0]
dq = torch.from_numpy(h.view(np.int64)).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx = cbird_amd.DctHashIndex(); idx.load_device(dq.data_ptr(), ids.data_ptr(), N)
cap = 1 << 24
drec = torch.zeros(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
def run(thr):
    dtot.zero_()
    _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap, dtot.data_ptr(), 1, C.byref(ms)), "scan")
    t = int(dtot.item())
    r = drec[:t]
    return t, int(r.sum().item()), int((r * 0x9E3779B97F4A7C15 % (1 << 61)).sum().item() if False else torch.bitwise_xor(r, r >> 17).sum().item())
ref = {}
L.cbh_set_tuning(b"scan_mfma", 0)
for thr in range(1, 13):
    ref[thr] = run(thr)
L.cbh_set_tuning(b"scan_mfma", MODE)
t0 = time.time(); reps = 0; bad = 0
while time.time() - t0 < secs:
    for thr in range(1, 13):
        got = run(thr)
        if got != ref[thr]:
            bad += 1
            print("MISMATCH thr", thr, got, ref[thr], flush=True)
    reps += 1
L.cbh_set_tuning(b"scan_mfma", 1)
print("scan_mfma", MODE, "N", N, "reps", reps, "launches", reps * 12, "mismatches", bad)
