"""hash kernel only (for rocprofv3 PMC passes): N images resident, a few launches"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
import bench
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = torch.device("cuda", 0)
imgs = bench.gen_images(torch, dev, 0, n, n, 1234)
out = torch.empty(n, dtype=torch.int64, device=dev)
ms = C.c_float(0)
for _ in range(2):
    _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, 3, C.byref(ms)), "h")
print(f"dcthash {n} imgs: {ms.value:.3f} ms  {n/ms.value*1e3:.3e} img/s  {n*65544/ms.value*1e-6:.1f} GB/s")
