"""colour find_batch only (for rocprofv3 kernel stats; development aid)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
from cbird_amd.colordesc import COLOR_DTYPE, ColorDescIndex
L = _lib.lib()
n = 1_000_000
rng = np.random.default_rng(77)
d = np.zeros(n, COLOR_DTYPE)
d["colors"] = rng.integers(0, 65536, (n, 32, 4), dtype=np.uint16)
d["numColors"] = rng.integers(28, 32, n, dtype=np.uint8)
ids = np.arange(1, n + 1, dtype=np.uint32)
ci = ColorDescIndex()
_lib.check(L.cbh_color_add(ci._h, ids.ctypes.data, d.ctypes.data, n), "add")
for fma in (0, 1):  # (1: the fused-square form, within 1e-5 of the reference's floats, not bit-identical)
    L.cbh_set_tuning(b"color_fma", fma)
    ci.find_batch(d[:64], 8)
    t0 = time.time(); ci.find_batch(d[:64], 8); print("color_fma", fma, "batch64 s", time.time() - t0, flush=True)
L.cbh_set_tuning(b"color_fma", 0)
