"""ORB detect + describe throughput under both retainBest orders (cbh_set_tuning("orb_retain_order", 0 | 1)):
   python tools/orb_retain_timing.py [images=512]   -> one JSON line"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from cbird_amd import _lib, orb  # noqa: E402


def scene(rng, w, h):
    img = np.full((h, w), 128, np.int32)
    for _ in range(w * h // 1000):
        x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        rw, rh = (int(v) for v in rng.integers(4, max(6, min(w, h) // 4), 2))
        img[y: y + rh, x: x + rw] = int(rng.integers(0, 256))
    return (img + rng.integers(-4, 5, img.shape)).clip(0, 255).astype(np.uint8)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rng = np.random.default_rng(1)
    base = [scene(rng, 400, 300) for _ in range(32)]
    imgs = [base[i % 32] for i in range(n)]
    orb.set_pattern(orb.synthetic_pattern())
    L = _lib.lib()
    buf, _total, off, w, h = orb._pack(imgs)
    stride = w
    d = torch.from_numpy(buf).cuda()
    cap = 640
    d_kp = torch.zeros(n * cap * 6, dtype=torch.float32, device="cuda")
    d_desc = torch.zeros(n * cap * 32, dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = {"images": n, "size": "400x300", "num_keypoints": 400}
    for mode, name in ((0, "canonical"), (1, "libstdcxx")):
        L.cbh_set_tuning(b"orb_retain_order", mode)
        ts = []
        for it in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(L.cbh_orb_dev(d.data_ptr(), n, off.ctypes.data, w.ctypes.data, h.ctypes.data, stride.ctypes.data,
                                     400, cap, d_kp.data_ptr(), None, d_desc.data_ptr(), d_cnt.data_ptr(), 0, None),
                       "cbh_orb_dev")
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        out[name] = {"ms": round(t * 1e3, 3), "images_per_s": round(n / t), "keypoints": int(d_cnt.sum().item())}
    L.cbh_set_tuning(b"orb_retain_order", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
