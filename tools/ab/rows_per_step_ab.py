"""hash_rows_per_step (k_blur_area_regs<7>: 14 / 21 / 28 source rows per step): equal hashes under every value, GB/s per
geometry (4 GB batches).    python tools/ab/rows_per_step_ab.py [WxH ...]"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
GEOS = [(400, 300), (320, 240), (512, 512), (533, 400), (640, 480), (800, 600), (1024, 768), (1280, 720), (1366, 768), (1440, 900),
        (1536, 1024), (1920, 1080), (2560, 1440), (3000, 2000), (3840, 2160), (4000, 3000), (4096, 2304), (5000, 3000), (8000, 6000)]
if len(sys.argv) > 1:
    GEOS = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (w, h) in GEOS:
    n = max(64, min(40000, int(4e9 // (w * h))))
    imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    VALS = (0, 1, 21, 28)
    outs, best, line = {}, {v: 1e9 for v in VALS}, f"{w}x{h}:"
    for rep in range(4):  # alternating: the first launches on a geometry run 3-5 % slower whatever the knob says
        for v in VALS:
            L.cbh_set_tuning(b"hash_rows_per_step", v)
            out = torch.empty(n, dtype=torch.int64, device=dev)
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms)), "hash")
            best[v] = min(best[v], ms.value)
            outs[v] = out
    for v in VALS:
        line += f"  {v:2d}: {n * w * h / best[v] * 1e-6:7.1f}"
    ok = all(bool((outs[0] == outs[v]).all()) for v in VALS[1:])
    print(line, " GB/s  equal" if ok else "  DIFFERENT", flush=True)
    del imgs
L.cbh_set_tuning(b"hash_rows_per_step", 1)
