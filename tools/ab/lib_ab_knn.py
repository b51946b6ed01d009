"""Same-box A/B of two BUILDS (see lib_ab.py) on the 256-bit scan: one ORB needle image (500 descriptors, k_hamm256_small)
and 64 needle images (k_hamm256_mfma3) against 100 000 x 500 rows.
    python tools/ab/lib_ab_knn.py [rounds=3] [other=cbird_amd/libcbird_hip.so.prev]"""
import json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r'''
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, ".")
from cbird_amd import _lib
if sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
from cbird_amd.cvfeatures import CvFeaturesIndex
L = _lib.lib()
n_img, per = 100000, 500
rng = np.random.default_rng(1234)
idx = CvFeaturesIndex()
chunk = 2000
for c0 in range(0, n_img, chunk):
    rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
    for i in range(chunk):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
st = _lib.cbh_stats()
def timed(needle, reps):
    idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0, l0 = st.scan_ms, st.scan_launches
    out = None
    for _ in range(reps):
        out = idx.knn(needle, 10, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st))
    return (st.scan_ms - ms0) / max(1, st.scan_launches - l0), int(sum(len(x) for x in out[0])) if isinstance(out, tuple) else 0
one = idx.descriptorsForMediaId(77).copy(); one[::3, 5] ^= 0x11
many = np.concatenate([idx.descriptorsForMediaId(100 + 7 * i) for i in range(64)]); many[::5, 9] ^= 0x21
a, _ = timed(one, 12)
b, _ = timed(many, 3)
print(json.dumps({"one_needle_image_ms": round(a, 4), "64_needle_images_ms": round(b, 3)}))
'''


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    other = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "cbird_amd", "libcbird_hip.so.prev"))
    out = {"this": [], "other": []}
    for _ in range(rounds):
        for name, path in (("this", "-"), ("other", other)):
            r = subprocess.run([sys.executable, "-c", CHILD, path], cwd=ROOT, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[name].append(json.loads(line[-1]) if line else {"error": r.stderr[-400:]})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
