"""hash_cell_pad (k_blur_area_regs, integer ratios): the hashes with the pad dword per LDS cell equal those without (both knob
values 1 and 2 against 0), and what it buys per geometry.    python tools/ab/cell_pad_ab.py"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
GEOS = ((512, 512), (768, 576), (1024, 768), (1024, 1024), (1280, 720), (1536, 1024), (2048, 1536), (2560, 1440), (3072, 2048),
        (3840, 2160), (4096, 2304), (640, 480), (256, 192), (512, 288))
for (w, h) in GEOS:
    n = max(64, min(20000, int(2e9 // (w * h))))
    imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    outs, line = [], f"{w}x{h} (cell {w // 32} px):"
    for v in (0, 1, 2):
        L.cbh_set_tuning(b"hash_cell_pad", v)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        best = 1e9
        for _ in range(3):
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms)), "hash")
            best = min(best, ms.value)
        outs.append(out.clone())
        line += f"  pad={v}: {n * w * h / best * 1e-6:7.1f} GB/s"
    ok = bool((outs[0] == outs[1]).all()) and bool((outs[0] == outs[2]).all())
    print(line, "  equal" if ok else "  DIFFERENT", flush=True)
    del imgs
L.cbh_set_tuning(b"hash_cell_pad", 1)
