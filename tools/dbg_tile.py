import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cbird_amd import _lib
from oracle import Oracle
L = _lib.lib(); o = Oracle()
rng = np.random.default_rng(1)
imgs = rng.integers(0, 256, (9, 256, 256), dtype=np.uint8)
imgs[1] = 200; imgs[2] = (np.arange(256)[None, :] * np.ones((256, 1))).astype(np.uint8)
imgs[3] = (np.arange(256)[:, None] * np.ones((1, 256))).astype(np.uint8)
d = torch.from_numpy(imgs).cuda(); out = torch.zeros(9, dtype=torch.int64, device="cuda")
tiles = torch.zeros((9, 32, 32), dtype=torch.uint8, device="cuda")
_lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), 9, 256, 256, 256, 65536, out.data_ptr(), tiles.data_ptr(), 0, None), "t")
t = tiles.cpu().numpy(); hv = out.cpu().numpy().view(np.uint64)
for i in range(9):
    w = o.tile32(imgs[i]); diff = (t[i].astype(int) - w.astype(int))
    hw = o.dcthash64(imgs[i]); ht = o.hash_from_tile32(t[i])
    print(i, "tile mismatches", int((diff != 0).sum()), "maxdiff", int(np.abs(diff).max()), "hash ok", int(hv[i]) == hw, "hash(gpu tile) by oracle == gpu hash", ht == int(hv[i]))
    if (diff != 0).any():
        ys, xs = np.nonzero(diff); print("   rows", sorted(set(ys.tolist()))[:10], "cols", sorted(set(xs.tolist()))[:10]); print(diff[:4, :8]); print(diff[-2:, -8:])
print("---- non-dump variant")
for n in (9, 24, 64):
    imgs = rng.integers(0, 256, (n, 256, 256), dtype=np.uint8)
    d = torch.from_numpy(imgs).cuda(); out = torch.zeros(n, dtype=torch.int64, device="cuda"); out2 = torch.zeros(n, dtype=torch.int64, device="cuda")
    tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
    _lib.check(L.cbh_dcthash_batch_dev(d.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, None), "t")
    _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, 256, 256, 256, 65536, out2.data_ptr(), tiles.data_ptr(), 0, None), "t")
    a = out.cpu().numpy().view(np.uint64); b = out2.cpu().numpy().view(np.uint64); w = o.dcthash64_batch(imgs)
    print(n, "nodump==oracle", int((a == w).sum()), "dump==oracle", int((b == w).sum()), "of", n)
    bad = np.nonzero(a != w)[0]
    print("  bad idx", bad[:16], [bin(int(x)).count("1") for x in (a ^ w)[bad][:8]])
    import cbird_amd
    h = cbird_amd.dct_hash64_batch(imgs)
    print("  host api == oracle", int((h == w).sum()))
print("---- smooth images")
from cbird_amd import synth
imgs = synth.make_images(24, w=256, h=256, seed=512, dup_frac=0.25)
n = len(imgs)
d = torch.from_numpy(imgs).cuda(); out2 = torch.zeros(n, dtype=torch.int64, device="cuda")
tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
_lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, 256, 256, 256, 65536, out2.data_ptr(), tiles.data_ptr(), 0, None), "t")
b = out2.cpu().numpy().view(np.uint64); w = o.dcthash64_batch(imgs); t = tiles.cpu().numpy()
for i in range(n):
    wt = o.tile32(imgs[i])
    hv, co, th = o.hash_from_tile32(t[i], with_coefs=True)
    print(i, "tile ok", bool((wt == t[i]).all()), "hash ok", int(b[i]) == int(w[i]), hex(int(b[i]) ^ int(w[i])), "min|c-thr|", float(np.abs(co[1:] - th).min()))
