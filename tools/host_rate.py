"""PCIe-inclusive rates of the host-buffer entry points (what a caller holding decoded images in host memory sees):
cbh_dcthash_batch on pageable and on pinned memory, against the device-resident kernel rate."""
import ctypes as C, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from cbird_amd import _lib
L = _lib.lib()
for (w, h, n) in ((256, 256, 16384), (1920, 1080, 512)):
    imgs = np.random.default_rng(0).integers(0, 256, (n, h, w), dtype=np.uint8)
    out = np.zeros(n, np.uint64)
    pinned = torch.from_numpy(imgs).pin_memory()
    for name, ptr in (("pageable", imgs.ctypes.data), ("pinned", pinned.data_ptr())):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            _lib.check(L.cbh_dcthash_batch(ptr, n, w, h, w, w * h, out.ctypes.data, 0), "hash")
            best = min(best, time.perf_counter() - t0)
        print(f"{w}x{h} n={n} {name:9s}: {best * 1e3:8.1f} ms  {n / best:10.3e} img/s  {n * w * h / best / 1e9:6.1f} GB/s", flush=True)
