"""Same-box A/B of two BUILDS of the library (a kernel changed in source, not behind a knob): child processes alternate
between cbird_amd/libcbird_hip.so and another file (default cbird_amd/libcbird_hip.so.prev, built from an earlier
commit's source and linked with the current objects), each timing the 1M x 1M scan at the given thresholds on the bench's
image-derived hashes.  Box-to-box variance on this pool is +-4 %; only alternation on one box resolves a 2 % change.
    python tools/ab/lib_ab.py [rounds=3] [thresholds=2,5,6,7] [other=cbird_amd/libcbird_hip.so.prev]"""
import json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r'''
import ctypes as C, json, sys
sys.path.insert(0, ".")
from cbird_amd import _lib
if sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
import torch, cbird_amd, bench
L = _lib.lib()
N = 1000000
T = [int(x) for x in sys.argv[2].split(",")]
dev = torch.device("cuda", 0)
out = torch.empty(N, dtype=torch.int64, device=dev)
for c0 in range(0, N, 100000):
    c1 = min(N, c0 + 100000)
    imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
    _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
    del imgs
idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx.load_device(out.data_ptr(), ids.data_ptr(), N)
cap = 1 << 25
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
res = {}
for rep in range(3):
    for t in T:
        _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, out.data_ptr(), N, t, drec.data_ptr(), cap, dtot.data_ptr(), 4, C.byref(ms)), "t")
        res.setdefault(str(t), []).append(round(ms.value, 3))
print(json.dumps({k: min(v) for k, v in res.items()}))
'''


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    thr = sys.argv[2] if len(sys.argv) > 2 else "2,5,6,7"
    other = os.path.abspath(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "cbird_amd", "libcbird_hip.so.prev"))
    out = {"this": [], "other": []}
    for _ in range(rounds):
        for name, path in (("this", "-"), ("other", other)):
            r = subprocess.run([sys.executable, "-c", CHILD, path, thr], cwd=ROOT, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[name].append(json.loads(line[-1]) if line else {"error": r.stderr[-300:]})
    best = {n: {t: min(x[t] for x in v if t in x) for t in thr.split(",")} for n, v in out.items()}
    print(json.dumps({"min_ms": best, "ratio_this_over_other": {t: round(best["this"][t] / best["other"][t], 4) for t in thr.split(",")},
                      "runs": out}))


if __name__ == "__main__":
    main()
