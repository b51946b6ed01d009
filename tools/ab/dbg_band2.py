import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
n = 4
imgs = np.zeros((n, 256, 256), np.uint8)
imgs[0] = np.arange(256, dtype=np.uint8)[:, None]
imgs[1, 64:, :] = 200
imgs[2] = (np.arange(256)[:, None] * 37 % 251).astype(np.uint8)
imgs[3] = np.random.default_rng(1).integers(0, 256, (256, 256), dtype=np.uint8)
d = torch.from_numpy(imgs).to(dev)
L.cbh_set_tuning(b"hash_mfma", 2)
t = torch.zeros((n, 1024), dtype=torch.uint8, device=dev)
o = torch.empty(n, dtype=torch.int64, device=dev)
_lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, 256, 256, 256, 65536, o.data_ptr(), t.data_ptr(), 0, None), "t")
dbg = t.cpu().numpy().view(np.uint16).reshape(n, 512)[:, :264]
def refl(v): 
    v = -v if v < 0 else v
    return 510 - v if v > 255 else v
for i in range(n):
    im = imgs[i].astype(np.int64)
    col = 83
    want = []
    for w in range(264):
        y = w - 8
        s = 0
        for dy in range(-3, 4):
            r = refl(y + dy) if -8 <= y + dy else None
            for dx in range(-3, 4):
                s += im[refl(y + dy), refl(col + dx)] if y + dy >= -5 else 0
        want.append(s)
    want = np.array(want)
    got = dbg[i].astype(np.int64)
    bad = np.nonzero((got != (want & 0xffff)) & (np.arange(264) >= 8))[0]
    print(i, "mismatching virtual rows:", bad[:20], "count", len(bad))
    for w in bad[:6]:
        print("   w", w, "got", got[w], "want", want[w], "diff", got[w] - want[w])
f = t.cpu().numpy()[2].view(np.uint32)[:132].reshape(66, 2).astype(np.int64)
im = imgs[1].astype(np.int64)
for tt in range(14, 20):
    qs = []
    for r in range(4):
        y = 4 * tt + r - 8
        s = sum(7 * im[refl(y + dy), 83] for dy in range(-3, 4))
        qs.append((2 * s + 49) // 98)
    print("step", tt, "f.x-2^23", f[tt, 0], "f.y-2^23", f[tt, 1], "  q of rows", qs, " multiples of 171196:", f[tt, 0] / 171196, f[tt, 1] / 171196)
