"""general-geometry hash kernels only (for rocprofv3 --pmc passes): N resident images of WxH, ONE launch group.
    python tools/hash_geo_only.py W H [N]"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
w, h = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else max(64, min(20000, int(2e9 // (w * h))))
dev = torch.device("cuda", 0)
imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
out = torch.empty(n, dtype=torch.int64, device=dev)
ms = C.c_float(0)
_lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 1, C.byref(ms)), "h")
print(f"{w}x{h} n {n} bytes {n * w * h} {ms.value:.3f} ms {n * w * h / ms.value * 1e-6:.1f} GB/s")
