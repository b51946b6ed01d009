"""dctHash64 throughput by image geometry (development aid): which kernel path each size takes and its GB/s"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
for kv in sys.argv[1:]:  # tuning overrides: key=value (and geos=WxH,WxH,...)
    k, v = kv.split("=")
    if k in ("geos", "ab", "bytes"):
        continue
    L.cbh_set_tuning(k.encode(), int(v))
GEOS = ((256, 256), (128, 128), (512, 512), (1024, 1024), (320, 240), (640, 480), (1024, 768), (300, 200), (1920, 1080), (3840, 2160), (4000, 3000))
if any(a.startswith("geos=") for a in sys.argv[1:]):
    GEOS = tuple(tuple(int(v) for v in g.split("x")) for a in sys.argv[1:] if a.startswith("geos=") for g in a[5:].split(","))
BYTES = float(next((a[6:] for a in sys.argv[1:] if a.startswith("bytes=")), 2e9))  # pixels per batch (2 GB: ~1 ms launches)
AB = [a[3:].split(",") for a in sys.argv[1:] if a.startswith("ab=")]  # ab=key:v1:v2:...  -> A/B the knob per geometry
for (w, h) in GEOS:
    n = max(64, min(int(20000 * BYTES / 2e9), int(BYTES // (w * h))))
    imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    if AB:
        key, *vals = AB[0][0].split(":")
        line = f"{w}x{h}:"
        for v in vals:
            L.cbh_set_tuning(key.encode(), int(v))
            best = 1e9
            for _ in range(3):
                rc = L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms))
                best = min(best, ms.value)
            line += f"  {key}={v}: {n * w * h / best * 1e-6:7.1f} GB/s"
        print(line, flush=True)
        del imgs
        continue
    rc = L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 2, C.byref(ms))
    rc = L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms))
    print(f"{w}x{h}: rc {rc} n {n} {ms.value:8.3f} ms {n / ms.value * 1e3:10.3e} img/s {n * w * h / ms.value * 1e-6:8.1f} GB/s", flush=True)
    del imgs
