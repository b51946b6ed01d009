"""k_dcthash_256 (VALU) beside k_dcthash_256_band (horizontal box sums on the matrix cores, "hash_mfma" 2): same hashes
on the bench's images + adversarial tiles, time per launch, and the compute-only time (images aliased, img_stride 0).
    python tools/ab/hash_band_ab.py [images]"""
import ctypes as C, sys, json
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
import bench
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = torch.device("cuda", 0)
imgs = bench.gen_images(torch, dev, 0, n, n, 1234)
g = torch.Generator(device=dev).manual_seed(3)
imgs[:4000] = torch.randint(0, 256, (4000, 256, 256), dtype=torch.uint8, device=dev, generator=g)
imgs[4000:4100] = 255
imgs[4100:4200] = 0
imgs[4200:4300, ::2] = 255   # row stripes
imgs[4300:4400, :, ::2] = 255  # column stripes
imgs[4400:4500, :, :3] = 255; imgs[4400:4500, :, -3:] = 1  # the reflected borders
ms = C.c_float(0)
res, ref = {}, None
tiles_ref = None
for name, knob, nw in (("valu", 0, 2), ("band1", 2, 1), ("band2", 2, 2), ("valu", 0, 2), ("band1", 2, 1), ("band2", 2, 2)):
    L.cbh_set_tuning(b"hash_mfma", knob)
    L.cbh_set_tuning(b"hash_band_waves", nw)
    out = torch.empty(n, dtype=torch.int64, device=dev)
    for stride, tag in ((65536, "hbm"), (0, "aliased")):
        for _ in range(2):
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, 256, 256, 256, stride, out.data_ptr(), 0, 5, C.byref(ms)), "h")
        res.setdefault(f"{name}_{tag}_ms", []).append(round(ms.value, 3))
    _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, None), "h")
    if ref is None:
        ref = out.clone()
    bad = int((out != ref).sum().item())
    res[f"{name}_differs_from_first"] = bad
    # the 32x32 tiles of the first 6000 images (stage-level comparison)
    m = 6000
    t = torch.empty((m, 1024), dtype=torch.uint8, device=dev)
    o2 = torch.empty(m, dtype=torch.int64, device=dev)
    _lib.check(L.cbh_dcthash_tiles_dev(imgs.data_ptr(), m, 256, 256, 256, 65536, o2.data_ptr(), t.data_ptr(), 0, None), "t")
    if tiles_ref is None:
        tiles_ref = t.clone()
    res[f"{name}_tile_bytes_differing"] = int((t != tiles_ref).sum().item())
L.cbh_set_tuning(b"hash_mfma", 2)
L.cbh_set_tuning(b"hash_band_waves", 1)
print(json.dumps({"images": n, **res}))
