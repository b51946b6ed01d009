"""one geometry, a few launches (for rocprofv3 kernel stats; development aid)"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
w, h = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
n = max(64, min(20000, int(2e9 // (w * h))))
imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
out = torch.empty(n, dtype=torch.int64, device=dev)
ms = C.c_float(0)
for _ in range(3):
    L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 2, C.byref(ms))
print(w, h, n, ms.value)
