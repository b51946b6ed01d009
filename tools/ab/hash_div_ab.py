"""k_dcthash_256 under the forms of its divide-by-49 step ("hash_div" 0 / 1 / 2): same hashes, time per launch.
    python tools/ab/hash_div_ab.py [images]"""
import ctypes as C, sys, json
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
import bench
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = torch.device("cuda", 0)
imgs = bench.gen_images(torch, dev, 0, n, n, 1234)
# plus adversarial tiles: extremes and pure noise
g = torch.Generator(device=dev).manual_seed(3)
imgs[:2000] = torch.randint(0, 256, (2000, 256, 256), dtype=torch.uint8, device=dev, generator=g)
imgs[2000:2100] = 255
imgs[2100:2200] = 0
ms = C.c_float(0)
res = {}
ref = None
for div in (0, 1, 2, 3, 0, 2, 3):
    L.cbh_set_tuning(b"hash_div", div)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for _ in range(2):
        _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, 5, C.byref(ms)), "h")
    if ref is None:
        ref = out.clone()
    res.setdefault(str(div), []).append(round(ms.value, 3))
    assert bool((out == ref).all()), f"hash_div {div} changes hashes"
print(json.dumps({"images": n, "ms_per_launch": res, "GBps": {k: round(n * 65544 / min(v) * 1e-6, 1) for k, v in res.items()},
                  "hashes_equal": True}))
