import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cbird_amd import _lib
from oracle import Oracle
L = _lib.lib(); o = Oracle()
rng = np.random.default_rng(1)
n = 9
imgs = rng.integers(0, 256, (n, 256, 256), dtype=np.uint8)
imgs[1] = 200
imgs[2] = (np.arange(256)[None, :] * np.ones((256, 1))).astype(np.uint8)
imgs[3] = (np.arange(256)[:, None] * np.ones((1, 256))).astype(np.uint8)
imgs[4] = 0; imgs[4, 100:140, 60:90] = 255
L.cbh_set_tuning(b"hash_mfma", 1)
d = torch.from_numpy(imgs).cuda(); out = torch.zeros(n, dtype=torch.int64, device="cuda")
tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
_lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), tiles.data_ptr(), 0, None), "t")
t = tiles.cpu().numpy(); hv = out.cpu().numpy().view(np.uint64)
for i in range(n):
    w = o.tile32(imgs[i]); diff = t[i].astype(int) - w.astype(int)
    print(i, "tile mismatches", int((diff != 0).sum()), "maxdiff", int(np.abs(diff).max()), "hash ok", int(hv[i]) == o.dcthash64(imgs[i]))
    if (diff != 0).any():
        ys, xs = np.nonzero(diff); print("   rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:12]); print(diff[:3, :10]); print(t[i][:2,:8], w[:2,:8])
out2 = torch.zeros(n, dtype=torch.int64, device="cuda")
_lib.check(L.cbh_dcthash_batch_dev(d.data_ptr(), n, 256, 256, 256, 65536, out2.data_ptr(), 0, None), "t")
print("nodump hashes equal:", (out2.cpu().numpy().view(np.uint64) == o.dcthash64_batch(imgs)).all())
nb = 100000
big = torch.randint(0, 256, (nb, 256, 256), dtype=torch.uint8, device="cuda")
ob = torch.empty(nb, dtype=torch.int64, device="cuda")
ms = C.c_float(0)
for mode in (0, 1):
    L.cbh_set_tuning(b"hash_mfma", mode)
    _lib.check(L.cbh_time_dcthash_dev(big.data_ptr(), nb, 256, 256, 256, 65536, ob.data_ptr(), 0, 1, C.byref(ms)), "h")
    _lib.check(L.cbh_time_dcthash_dev(big.data_ptr(), nb, 256, 256, 256, 65536, ob.data_ptr(), 0, 3, C.byref(ms)), "h")
    print("mode", mode, f"{ms.value:.3f} ms {nb/ms.value*1e3:.3e} img/s {nb*65544/ms.value*1e-6:.0f} GB/s")
    if mode == 0: ref = ob.clone()
print("mfma == valu kernel on 100k random images:", bool((ob == ref).all()))
