"""k_dcthash_256 with every image aliased to image 0 (img_stride 0: 64 KB working set, all loads hit L1/L2): what the
kernel costs when HBM delivers nothing -- beside the normal launch and the loads-only figure of tools/ubench/read_pattern.
    python tools/ab/hash_compute_only.py [images]"""
import ctypes as C, sys, json
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
import bench
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = torch.device("cuda", 0)
imgs = bench.gen_images(torch, dev, 0, n, n, 1234)
out = torch.empty(n, dtype=torch.int64, device=dev)
ms = C.c_float(0)
res = {}
for div in (0, 2, 3):
    L.cbh_set_tuning(b"hash_div", div)
    for name, stride in (("hbm", 65536), ("aliased", 0)):
        for _ in range(2):
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, 256, 256, 256, stride, out.data_ptr(), 0, 5, C.byref(ms)), "h")
        res[f"div{div}_{name}_ms"] = round(ms.value, 3)
print(json.dumps({"images": n, **res}))
