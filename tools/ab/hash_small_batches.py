"""dctHash64 latency of small and medium batches: the automatic kernel choice against the strip kernels forced, split and
fused.    python tools/hash_small_batches.py [WxH,WxH,... n,n,...]"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
GEOS = ((400, 300), (640, 480), (533, 400), (1024, 768))
NS = (16, 64, 128, 256, 512, 1024)
if len(sys.argv) > 2:
    GEOS = tuple(tuple(int(v) for v in g.split('x')) for g in sys.argv[1].split(','))
    NS = tuple(int(v) for v in sys.argv[2].split(','))
for (w, h) in GEOS:
    for n in NS:
        imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        line = f"{w}x{h} n={n}:"
        for key, val in ((b"hash_stream", 1), (b"hash_stream", 8)):
            L.cbh_set_tuning(key, val)
            for fuse in ((1,) if val == 1 else (0, 2)):
                L.cbh_set_tuning(b"hash_fuse", fuse)
                best = 1e9
                for _ in range(5):
                    L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms))
                    best = min(best, ms.value)
                line += f"  stream={val},fuse={fuse}: {best*1e3:7.1f} us"
        L.cbh_set_tuning(b"hash_stream", 1); L.cbh_set_tuning(b"hash_fuse", 1)
        print(line, flush=True)
