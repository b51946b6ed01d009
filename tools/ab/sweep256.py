"""k_hamm256_mfma variant sweep (development aid)."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
from cbird_amd.cvfeatures import CvFeaturesIndex
L = _lib.lib()
n_img, per = 20000, 500
rng = np.random.default_rng(1)
idx = CvFeaturesIndex()
for c0 in range(0, n_img, 2000):
    rows = rng.integers(0, 256, (2000 * per, 32), dtype=np.uint8)
    for i in range(2000):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
needles = np.concatenate([idx.descriptorsForMediaId(i) for i in range(1, 65)])
st = _lib.cbh_stats()
for label, pre, ht in (("full", 0, 6), ("pre6", 1, 106), ("pre8", 1, 108), ("pre12", 1, 112), ("full", 0, 6), ("pre6", 1, 106), ("pre8", 1, 108), ("pre12", 1, 112)):
    L.cbh_set_tuning(b"scan256_pre", pre)
    L.cbh_set_tuning(b"scan256_ht", ht)
    idx.knn(needles[:500], 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0 = st.scan_ms
    r = idx.knn(needles, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); kms = st.scan_ms - ms0
    print(label, "kernel_ms", round(kms, 2), "cmp/s", idx.count() * len(needles) / kms * 1e3, "found", int((r[2] > 0).sum()), flush=True)
