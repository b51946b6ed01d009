import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
n = 8
imgs = np.zeros((n, 256, 256), np.uint8)
imgs[0] = 100
imgs[1] = np.arange(256, dtype=np.uint8)[None, :]          # horizontal ramp
imgs[2] = np.arange(256, dtype=np.uint8)[:, None]          # vertical ramp
imgs[3, 128, 128] = 255                                     # impulse
imgs[4, 0, 0] = 255
imgs[5] = np.random.default_rng(1).integers(0, 256, (256, 256), dtype=np.uint8)
imgs[6, :, 64:] = 200
imgs[7, 64:, :] = 200
d = torch.from_numpy(imgs).to(dev)
def run(knob):
    L.cbh_set_tuning(b"hash_mfma", knob)
    t = torch.empty((n, 1024), dtype=torch.uint8, device=dev)
    o = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, 256, 256, 256, 65536, o.data_ptr(), t.data_ptr(), 0, None), "t")
    return t.cpu().numpy().reshape(n, 32, 32), o.cpu().numpy()
a, ha = run(0)
b, hb = run(2)
for i in range(n):
    diff = (a[i] != b[i])
    print(i, "tile diffs", int(diff.sum()), "hash eq", ha[i] == hb[i])
    if diff.sum():
        ys, xs = np.nonzero(diff)
        print("  first diffs (y,x,valu,band):", [(int(y), int(x), int(a[i][y, x]), int(b[i][y, x])) for y, x in list(zip(ys, xs))[:10]])
        print("  rows with diffs", sorted(set(ys.tolist()))[:40])
        print("  cols with diffs", sorted(set(xs.tolist()))[:40])
print("valu tile0 row0", a[1][0][:12], "band", b[1][0][:12])
print("valu vr col0", a[2][:12, 0], "band", b[2][:12, 0])
