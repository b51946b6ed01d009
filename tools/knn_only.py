"""configs[3] scan kernels alone (for rocprofv3 passes): n_img x 500 rows resident, then `reps` searches with ONE needle
image (500 descriptors: k_hamm256_small<16>) and `reps` with 64 needle images (32 000 descriptors: k_hamm256_mfma3).
    python tools/knn_only.py [n_img=100000] [reps=3]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
from cbird_amd.cvfeatures import CvFeaturesIndex
L = _lib.lib()
n_img, per = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 500
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(1234)
idx = CvFeaturesIndex()
chunk = 2000
for c0 in range(0, n_img, chunk):
    rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
    for i in range(chunk):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
one = idx.descriptorsForMediaId(77).copy()
one[::3, 5] ^= 0x11
many = np.concatenate([idx.descriptorsForMediaId(7 + 11 * j) for j in range(64)]).copy()
many[::3, 5] ^= 0x11
for _ in range(reps):
    idx.knn(one, 10, 25)
for _ in range(reps):
    idx.knn(many, 10, 25)
print("ok", idx.count())
