"""ORB throughput (Media::makeKeyPoints + makeKeyPointDescriptors, cbird_amd/csrc/orb.hip): n grey images of 400x300
(cbird's sizeLongestSide(400) output) resident on the device, 400 keypoints asked for (IndexParams::numFeatures).
Prints images/s and keypoints/s, the oracle's single-core rate on a sample, and checks that sample bit for bit.

    python tools/orb_bench.py [--images 4096] [--w 400 --h 300] [--detect-only]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def scene(rng, w, h):
    img = np.full((h, w), 128, np.int32)
    for _ in range((w * h) // 1000):
        x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        rw, rh = (int(v) for v in rng.integers(4, max(6, min(w, h) // 4), 2))
        img[y: y + rh, x: x + rw] = int(rng.integers(0, 256))
    return (img + rng.integers(-4, 5, img.shape)).clip(0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--w", type=int, default=400)
    ap.add_argument("--h", type=int, default=300)
    ap.add_argument("--kp", type=int, default=400)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cpu-images", type=int, default=16)
    ap.add_argument("--detect-only", action="store_true")
    args = ap.parse_args()
    import torch

    from cbird_amd import _lib, orb

    L = _lib.lib()
    pat = orb.synthetic_pattern()
    orb.set_pattern(pat)
    rng = np.random.default_rng(1)
    n, w, h = args.images, args.w, args.h
    base = np.stack([scene(rng, w, h) for _ in range(64)])
    imgs = np.concatenate([base] * ((n + 63) // 64))[:n].copy()
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(imgs).to(dev)
    cap = args.kp + 112
    off = (np.arange(n, dtype=np.uint64) * np.uint64(w * h))
    ww = np.full(n, w, np.uint32)
    hh = np.full(n, h, np.uint32)
    d_kp = torch.zeros((n, cap, 6), dtype=torch.float32, device=dev)
    d_after = torch.zeros((n, cap, 2), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    desc_ptr = None if args.detect_only else d_desc.data_ptr()
    after_ptr = None if args.detect_only else d_after.data_ptr()

    def run():
        _lib.check(L.cbh_orb_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data, ww.ctypes.data,
                                 args.kp, cap, d_kp.data_ptr(), after_ptr, desc_ptr, d_cnt.data_ptr(), 0,
                                 C.c_void_p(stream.cuda_stream)), "orb")

    with torch.cuda.stream(stream):
        run()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.iters):
            run()
        e1.record(stream)
        stream.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    cnt = d_cnt.cpu().numpy()
    assert cnt.max() <= cap, "raise cap"
    nk = int(cnt.sum())
    out = {"workload": f"{n} images {w}x{h}, {args.kp} keypoints asked, detect{'' if args.detect_only else ' + describe'}",
           "ms": ms, "images_per_s": n / ms * 1e3, "keypoints_per_s": nk / ms * 1e3, "keypoints_per_image": nk / n,
           "pixels_GBps": n * w * h / ms * 1e-6}
    # CPU beside it: the oracle on one core, same images
    from oracle import OrbOracle

    o = OrbOracle()
    o.set_pattern(pat)
    m = min(args.cpu_images, n)
    kp_h = d_kp[:m].cpu().numpy().view(np.uint32)
    desc_h = d_desc[:m].cpu().numpy()
    t0 = time.time()
    ok = True
    for i in range(m):
        k = o.detect(imgs[i], args.kp)
        if not args.detect_only:
            k2, dd = o.compute(imgs[i], k)
        c = int(cnt[i])
        got = kp_h[i, :c]
        want = np.stack([k[f].view(np.uint32) for f in ("x", "y", "size", "angle", "response", "octave")], 1)
        ok &= len(k) == c and (got == want).all()
        if not args.detect_only:
            ok &= (desc_h[i, :c] == dd).all()
    dt = time.time() - t0
    out["cpu_oracle_images_per_s_1core"] = m / dt
    out["sample_bit_exact"] = bool(ok)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
