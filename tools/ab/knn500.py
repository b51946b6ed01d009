import ctypes as C, json, sys, time
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
from cbird_amd.cvfeatures import CvFeaturesIndex
L = _lib.lib()
n_img, per = int(sys.argv[1]), 500
rng = np.random.default_rng(1234)
idx = CvFeaturesIndex()
chunk = 2000
for c0 in range(0, n_img, chunk):
    rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
    for i in range(chunk):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
needle = idx.descriptorsForMediaId(77).copy()
needle[::3, 5] ^= 0x11
st = _lib.cbh_stats()
res = {}
for name, knobs in (("rows_stationary", {b"scan256_small": 0}), ("needles_stationary", {b"scan256_small": 1})):
    for k, v in knobs.items():
        L.cbh_set_tuning(k, v)
    r0 = idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0, l0 = st.scan_ms, st.scan_launches
    for _ in range(8):
        r = idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st))
    res[name] = round((st.scan_ms - ms0) / (st.scan_launches - l0), 4)
    if name == "rows_stationary":
        base = r
    else:
        assert all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(base, r)), name
    L.cbh_set_tuning(b"scan256_small", 1)
print(json.dumps(res))
