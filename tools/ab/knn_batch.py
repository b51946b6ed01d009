"""configs[3] at the bench's batch shape: the k-nn scan of 64 needle images (32 000 descriptors) against n_img x 500
index rows, by prefilter kernel shape (cbh_set_tuning("scan256_f3", v): 0 = k_hamm256_mfma, one needle tile per
accumulator; HT*10+G = k_hamm256_mfma3 shapes).  Average scan kernel time per search from the handle's statistics.
    python tools/knn_batch.py [n_img=20000]   -> one JSON line"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
from cbird_amd.cvfeatures import CvFeaturesIndex
L = _lib.lib()
n_img, per = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 500
rng = np.random.default_rng(1234)
idx = CvFeaturesIndex()
chunk = 2000
for c0 in range(0, n_img, chunk):
    rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
    for i in range(chunk):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
needle = np.concatenate([idx.descriptorsForMediaId(7 + 11 * j) for j in range(64)]).copy()
needle[::3, 5] ^= 0x11
st = _lib.cbh_stats()
L.cbh_set_tuning(b"scan256_small", 0)
res, base = {"rows": n_img * per, "needle_descriptors": len(needle)}, None
for shape in (0, 62, 63, 82, 122, 123, 1):
    L.cbh_set_tuning(b"scan256_f3", shape)
    r = idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0, l0 = st.scan_ms, st.scan_launches
    for _ in range(3):
        r = idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st))
    ms = (st.scan_ms - ms0) / (st.scan_launches - l0)
    res[f"f3={shape}"] = {"scan_ms": round(ms, 3), "cmp256_per_s": float(f"{n_img * per * len(needle) / ms * 1e3:.4g}"),
                          "frac_of_fp4_peak": round(n_img * per * len(needle) * 256 / ms * 1e3 / 1e16, 3)}
    if base is None:
        base = r
    else:
        assert all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(base, r)), shape
L.cbh_set_tuning(b"scan256_f3", 1)
L.cbh_set_tuning(b"scan256_small", 1)
print(json.dumps(res))
