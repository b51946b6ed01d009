"""ColorDescriptor::create throughput (cbird_amd/csrc/colordesc_create.hip): n BGR images resident on the device.
The clustering kernel runs ONE LANE PER IMAGE (its sums are order-dependent, see the kernel's header), so the rate
grows with the batch until every SIMD holds a wave: --images 4096 / 16384 / 65536 show the curve.

    python tools/color_create_bench.py [--images 4096] [--w 256 --h 192]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--w", type=int, default=256)
    ap.add_argument("--h", type=int, default=192)
    ap.add_argument("--cpu-images", type=int, default=8)
    args = ap.parse_args()
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(1)
    n, w, h = args.images, args.w, args.h
    base = rng.integers(0, 256, (32, h, w, 3), dtype=np.uint8)
    for b in base:
        for _ in range(40):
            x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
            b[y: y + int(rng.integers(3, h // 3)), x: x + int(rng.integers(3, w // 3))] = rng.integers(0, 256, 3)
    dev = torch.device("cuda", 0)
    d_base = torch.from_numpy(base).to(dev)
    d = d_base.repeat((n + 31) // 32, 1, 1, 1)[:n].contiguous()
    off = np.arange(n, dtype=np.uint64) * np.uint64(w * h * 3)
    ww, hh, ss = np.full(n, w, np.uint32), np.full(n, h, np.uint32), np.full(n, 3 * w, np.uint32)
    d_desc = torch.zeros((n, 258), dtype=torch.uint8, device=dev)
    d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream()

    def run():
        _lib.check(L.cbh_color_descriptors_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data,
                                               ss.ctypes.data, 3, d_desc.data_ptr(), d_ok.data_ptr(), 0,
                                               C.c_void_p(stream.cuda_stream)), "color_descriptors")

    run()
    t0 = time.time()
    run()
    dt = time.time() - t0
    out = {"workload": f"{n} BGR images {w}x{h}", "s": dt, "images_per_s": n / dt}
    from oracle import ColorCreateOracle

    o = ColorCreateOracle()
    m = min(args.cpu_images, n)
    got = d_desc[:m].cpu().numpy()
    t0 = time.time()
    ok = True
    for i in range(m):
        want, _ = o.create(base[i % 32])
        ok &= want is not None and (got[i] == want).all()
    out["cpu_oracle_images_per_s_1core"] = m / (time.time() - t0)
    out["sample_bit_exact"] = bool(ok)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
