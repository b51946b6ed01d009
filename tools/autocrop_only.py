"""autocrop kernels only (for rocprofv3 --pmc / --kernel-trace passes): N resident grey frames WxH with BAR-row bars,
one cbh_autocrop_dev call.    python tools/autocrop_only.py W H BAR [N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from cbird_amd import _lib
L = _lib.lib()
w, h, bar = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
frames = torch.full((n, h, w), 16, dtype=torch.uint8, device=dev)
frames += torch.randint(0, 3, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
frames[:, bar:h - bar, :] = torch.randint(40, 256, (n, h - 2 * bar, w), dtype=torch.uint8, device=dev, generator=g)
rects = torch.zeros((n, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
_lib.check(L.cbh_autocrop_dev(frames.data_ptr(), n, w, h, w, w * h, 20, rects.data_ptr(), 0, None), "autocrop")
torch.cuda.synchronize()
print(w, h, bar, n, "bytes", n * w * h, "rect0", rects[0].tolist())
