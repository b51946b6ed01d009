import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
for (w, h) in ((400, 300), (800, 600), (320, 240), (256, 256)):
    n = int(6e9 // (w * h))
    imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    res = {}
    for name, stride in (("hbm", w * h), ("aliased", 0)):
        best = 1e9
        for _ in range(3):
            L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, stride, out.data_ptr(), 0, 3, C.byref(ms))
            best = min(best, ms.value)
        res[name] = round(best, 3)
    print(f"{w}x{h} n {n}: {res}  (GB/s hbm {n*w*h/res['hbm']*1e-6:.0f}, aliased-equivalent {n*w*h/res['aliased']*1e-6:.0f})", flush=True)
    del imgs
