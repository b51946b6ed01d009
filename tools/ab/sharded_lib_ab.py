"""Same-box A/B of two BUILDS of the library on the bench's sharded leg: the dht sweep (cbh_idx64_find_batch_dev: scan +
cut) through ONE plain handle and through ONE handle over 8 logical shards of device 0, alternating child processes
between cbird_amd/libcbird_hip.so and another file (default cbird_amd/libcbird_hip.so.r05, the round-5 build).  Pure
ctypes on entry points both builds export, so a library of an older C-ABI version can stand in.
    python tools/ab/sharded_lib_ab.py [rounds=2] [other=cbird_amd/libcbird_hip.so.r05] [dht=1,2,3,4,5,6,7,8]"""
import json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r'''
import ctypes as C, json, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from cbird_amd import synth
L = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
vp, sz = C.c_void_p, C.c_size_t
L.cbh_idx64_create.restype = vp; L.cbh_idx64_create.argtypes = [C.c_int]
L.cbh_idx64_create_sharded.restype = vp; L.cbh_idx64_create_sharded.argtypes = [C.c_uint32, C.c_int]
L.cbh_idx64_load.argtypes = [vp, vp, vp, sz]
L.cbh_idx64_find_batch_dev.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_uint64), vp]
L.cbh_idx64_destroy.argtypes = [vp]
dhts = [int(x) for x in sys.argv[2].split(",")]
n, topk = 1000000, 8
dev = torch.device("cuda", 0)
h, ids = synth.make_hashes(n, seed=1234)
dq = torch.from_numpy(h.view(np.int64)).to(dev)
out = torch.empty((n, topk, 2), dtype=torch.int32, device=dev)
cnt = torch.empty(n, dtype=torch.int32, device=dev)
def sweep(hnd):
    tot = C.c_uint64(0); per = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for d in dhts:
        t1 = time.perf_counter()
        rc = L.cbh_idx64_find_batch_dev(hnd, dq.data_ptr(), n, d, topk, out.data_ptr(), cnt.data_ptr(), C.byref(tot), None)
        assert rc == 0, rc
        per[d] = round((time.perf_counter() - t1) * 1e3, 3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, per
res = {}
for name, make in (("one", lambda: L.cbh_idx64_create(0)), ("sharded8", lambda: L.cbh_idx64_create_sharded(1, 8))):
    hnd = make(); assert hnd
    assert L.cbh_idx64_load(hnd, h.ctypes.data, ids.ctypes.data, n) == 0
    sweep(hnd)
    best, per = min((sweep(hnd) for _ in range(3)), key=lambda x: x[0])
    res[name] = {"sweep_ms": round(best, 3), "per_dht_ms": per}
    L.cbh_idx64_destroy(hnd)
print(json.dumps(res))
'''


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    other = sys.argv[2] if len(sys.argv) > 2 else "cbird_amd/libcbird_hip.so.r05"
    dht = sys.argv[3] if len(sys.argv) > 3 else "1,2,3,4,5,6,7,8"
    libs = {"current": "cbird_amd/libcbird_hip.so", "other": other}
    out = {k: [] for k in libs}
    for _ in range(rounds):
        for k, path in libs.items():
            r = subprocess.run([sys.executable, "-c", CHILD, path, dht], cwd=ROOT, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(f"{k}: {r.stderr[-800:]}")
            out[k].append(json.loads(r.stdout.strip().splitlines()[-1]))
    print(json.dumps({"libs": libs, "runs": out}))


if __name__ == "__main__":
    main()
