"""Randomised parity soak for the indexer's feature stages: ORB (cbh_orb) and ColorDescriptor::create
(cbh_color_descriptors) against their oracles on images of random sizes and kinds.  Prints one JSON line.

    python tools/fuzz_features.py [--orb 300] [--color 120] [--seed 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def image(rng, w, h, kind, ch=1):
    shape = (h, w) if ch == 1 else (h, w, ch)
    if kind == 0:
        img = rng.integers(0, 256, shape)
    elif kind == 1:
        img = np.full(shape, int(rng.integers(0, 256)), np.int64)
        for _ in range(int(rng.integers(5, 200))):
            x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y: y + int(rng.integers(1, max(2, h // 2))), x: x + int(rng.integers(1, max(2, w // 2)))] = \
                rng.integers(0, 256, ch if ch > 1 else None)
        img = img + rng.integers(-int(rng.integers(0, 12)) - 1, int(rng.integers(0, 12)) + 2, shape)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 100 * np.sin(xx / rng.uniform(2, 40)) * np.cos(yy / rng.uniform(2, 40))
        img = base if ch == 1 else np.stack([np.roll(base, int(rng.integers(0, 50)), i % 2) for i in range(ch)], -1)
        img = img + rng.normal(0, rng.uniform(0, 10), shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orb", type=int, default=300)
    ap.add_argument("--color", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from cbird_amd import orb as gorb
    from cbird_amd.colordesc import create_descriptors
    from oracle import ColorCreateOracle, OrbOracle

    rng = np.random.default_rng(args.seed)
    oo, co = OrbOracle(), ColorCreateOracle()
    out = {"seed": args.seed}
    t0 = time.time()
    bad, nk = [], 0
    for b0 in range(0, args.orb, 32):
        pat = rng.integers(-13, 14, 1024).astype(np.int8)
        oo.set_pattern(pat)
        gorb.set_pattern(pat)
        imgs = [image(rng, int(rng.integers(40, 700)), int(rng.integers(40, 700)), int(rng.integers(0, 3)))
                for _ in range(min(32, args.orb - b0))]
        nfeat = int(rng.choice([400, 500, 100, 1000, 7]))
        for i, (img, (k, a, d)) in enumerate(zip(imgs, gorb.orb(imgs, nfeat))):
            wk = oo.detect(img, nfeat)
            wk2, wd = oo.compute(img, wk)
            nk += len(wk)
            if not (len(k) == len(wk) and (k == wk).all() and (d == wd).all() and (a[:, 0] == wk2["x"]).all()
                    and (a[:, 1] == wk2["y"]).all()):
                bad.append((b0 + i, img.shape, nfeat))
    out["orb"] = {"images": args.orb, "keypoints": nk, "mismatching_images": bad}
    bad = []
    nok = 0
    for b0 in range(0, args.color, 16):
        ch = int(rng.choice([3, 4]))
        imgs = [image(rng, int(rng.integers(8, 900)), int(rng.integers(8, 900)), int(rng.integers(0, 3)), ch)
                for _ in range(min(16, args.color - b0))]
        if b0 == 0:
            imgs[0][:] = 2                      # too dark
            imgs[1][:] = (200, 10, 90, 255)[:ch]  # one colour
        descs, ok = create_descriptors(imgs)
        for i, img in enumerate(imgs):
            want, _ = co.create(img)
            got = np.frombuffer(descs[i].tobytes(), np.uint8)
            nok += int(ok[i])
            if bool(ok[i]) != (want is not None) or (want is not None and not (got == want).all()):
                bad.append((b0 + i, img.shape))
    out["color"] = {"images": args.color, "descriptors": nok, "mismatching_images": bad}
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
