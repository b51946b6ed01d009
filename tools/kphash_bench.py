"""Keypoint-hash throughput (Media::makeKeyPointHashes, k_rect_hashes): n grey images of 400x300 with 400 ORB-like
keypoints each (sizes 31 * 1.2^level, level frequencies falling by 1/1.2 per level like ORB's per-level feature
budget, centres where ORB can detect them), images resident on the device.  Prints hashes/s, and the oracle's single-core rate on a sample.

    python tools/kphash_bench.py [--images 4096]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--kp", type=int, default=400)
    ap.add_argument("--cpu-images", type=int, default=8)
    ap.add_argument("--level", type=int, default=-1, help="force one pyramid level (diagnostics)")
    ap.add_argument("--lds-side", type=int, default=0, help="tuning: largest square processed in LDS")
    ap.add_argument("--blur-side", type=int, default=0, help="tuning: largest square blurred into LDS")
    ap.add_argument("--size", type=float, default=0, help="force one keypoint size (diagnostics)")
    args = ap.parse_args()
    import ctypes as C

    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    if args.blur_side:
        L.cbh_set_tuning(b"kp_blur_side", args.blur_side)
    if args.lds_side:
        L.cbh_set_tuning(b"kp_lds_side", args.lds_side)
    rng = np.random.default_rng(1)
    n, w, h, k = args.images, 400, 300, args.kp
    base = rng.integers(0, 256, (64, h, w), dtype=np.uint8)
    imgs = np.concatenate([base] * ((n + 63) // 64))[:n].copy()
    # ORB's per-level feature budget falls by 1/scaleFactor per level; a level-l keypoint lies at least
    # edgeThreshold * 1.2^l from the border (its centre -- cbird then anchors the square's corner there)
    p = (1.0 / 1.2) ** np.arange(12)
    p /= p.sum()
    kps = np.zeros((n, k, 3), np.float32)
    lv = rng.choice(12, (n, k), p=p)
    if args.level >= 0:
        lv[:] = args.level
    scale = 1.2 ** lv
    size = (31.0 * scale).astype(np.float32)
    if args.size > 0:
        size[:] = args.size
    kps[:, :, 2] = size
    if args.level >= 0 or args.size > 0:  # diagnostics: every keypoint passes the inside-image rule
        kps[:, :, 0] = (rng.uniform(0, 1, (n, k)) * np.maximum(1, w - 3 - size)).astype(np.float32) + 0.5
        kps[:, :, 1] = (rng.uniform(0, 1, (n, k)) * np.maximum(1, h - 3 - size)).astype(np.float32) + 0.5
    else:
        m = 31.0 * scale
        kps[:, :, 0] = (m + rng.uniform(0, 1, (n, k)) * np.maximum(0, w - 2 * m)).astype(np.float32)
        kps[:, :, 1] = (m + rng.uniform(0, 1, (n, k)) * np.maximum(0, h - 2 * m)).astype(np.float32)
    off = (np.arange(n, dtype=np.uint64) * np.uint64(w * h))
    ww = np.full(n, w, np.uint32)
    hh = np.full(n, h, np.uint32)
    kp_first = (np.arange(n + 1, dtype=np.uint32) * np.uint32(k))
    out_first = np.zeros(n + 1, np.uint32)
    d_src = torch.from_numpy(imgs).cuda()
    d_out = torch.zeros(n * k, dtype=torch.int64, device="cuda")
    best = 1e9
    for it in range(4):
        d = d_src.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(L.cbh_keypoint_hashes_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data,
                                             ww.ctypes.data, kps.ctypes.data, kp_first.ctypes.data, d_out.data_ptr(),
                                             out_first.ctypes.data, 0, None), "kp")
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    total = int(out_first[-1])
    print(f"GPU: {n} images, {total} hashes ({total / n:.0f}/image) in {best * 1e3:.1f} ms = {total / best:.3e} hashes/s, "
          f"{n / best:.3e} images/s")
    try:
        from oracle import Oracle

        orc = Oracle()
        m = args.cpu_images
        t0 = time.perf_counter()
        cnt = 0
        got = d_out.cpu().numpy().view(np.uint64)
        ok = True
        for i in range(m):
            hs, _ = orc.keypoint_hashes(imgs[i], kps[i])
            cnt += len(hs)
            ok &= bool((hs == got[out_first[i]: out_first[i + 1]]).all())
        dt = time.perf_counter() - t0
        print(f"oracle (1 core): {cnt} hashes in {dt * 1e3:.0f} ms = {cnt / dt:.3e} hashes/s; agrees with the GPU: {ok}")
    except Exception as e:  # the oracle is optional here
        print("oracle not available:", e)


if __name__ == "__main__":
    main()
