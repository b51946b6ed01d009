"""Compare the matrix-core scan with the VALU scan record by record (development aid)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
h = synth.make_hashes(N, seed=7)[0]
NQ = int(sys.argv[3]) if len(sys.argv) > 3 else N
dq = torch.from_numpy(h.view(np.int64)).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx = cbird_amd.DctHashIndex(); idx.load_device(dq.data_ptr(), ids.data_ptr(), N)
cap = 1 << 24
ms = C.c_float(0)
out = {}
for name, on in (("valu", 0), ("mfma", 2)):
    L.cbh_set_tuning(b"scan_mfma", on)
    drec = torch.zeros(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), NQ, thr, drec.data_ptr(), cap, dtot.data_ptr(), 1, C.byref(ms)), "scan")
    t = int(dtot.item()); r = drec[:t].cpu().numpy().view(np.uint64)
    out[name] = np.sort(r); print(name, t, "records", ms.value, "ms")
a, b = out["valu"], out["mfma"]
miss = np.setdiff1d(a, b); extra = np.setdiff1d(b, a)
print("missing", len(miss), "extra", len(extra))
def show(r):
    q = int(r >> np.uint64(39)); d = int((r >> np.uint64(32)) & np.uint64(0x7f)); i = int(r & np.uint64(0xffffffff)) - 1
    real = bin(int(h[q]) ^ int(h[i])).count("1")
    return f"q={q} (q%64={q%64}) row={i} (row%32={i%32}, tile={i//32}, t={(i//32)%4}) d={d} real={real}"
for r in miss[:40]: print(" miss ", show(r))
for r in extra[:40]: print(" extra", show(r))
