"""Scan-kernel variant sweep on image-derived and uniform hashes (development aid)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth
import bench
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
def hashes_from_images(n):
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for c0 in range(0, n, 100000):
        c1 = min(n, c0 + 100000)
        imgs = bench.gen_images(torch, dev, c0, c1, n, 1234)
        _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
    return out
sets = {"uniform": torch.from_numpy(synth.make_hashes(N, seed=1234)[0].view(np.int64)).to(dev), "images": hashes_from_images(N)}
cap = 1 << 24
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
for name, dq in sets.items():
    idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
    idx.load_device(dq.data_ptr(), ids.data_ptr(), N)
    for thr in (1, 2, 4, 5):
        row = []
        for label, pre, eq, grp in (("mfma8g2full", 0, 0, 0), ("mfma8g2ful3", 0, 0, 0), ("mfma8g2pre", 0, 0, 0)):
            if label == "eq" and thr != 1: continue
            L.cbh_set_tuning(b"scan_mfma", 1 if label.startswith("mfma") else 0)
            if label.startswith("mfma"): L.cbh_set_tuning(b"scan_mfma_ht", int(label[4])); L.cbh_set_tuning(b"scan_mfma_g", int(label[6])); L.cbh_set_tuning(b"scan_mfma_pre", 1 if label.endswith("pre") else 0); L.cbh_set_tuning(b"scan_mfma_full3", 1 if label.endswith("ful3") else 0)
            L.cbh_set_tuning(b"scan_pre_max", pre); L.cbh_set_tuning(b"scan_eq_dht1", eq); L.cbh_set_tuning(b"scan_group", grp)
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap, dtot.data_ptr(), 1, C.byref(ms)), "w")
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap, dtot.data_ptr(), 2, C.byref(ms)), "t")
            row.append(f"{label} {ms.value:7.2f} ms ({int(dtot.item())//2} m)")
        print(name, "dht", thr, " | ".join(row), flush=True)
