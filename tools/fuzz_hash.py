"""Fuzz of the build path over random geometries: cbh_process_images (gray -> autocrop(20) -> dctHash64 of the kept
VIEW, any width / height / letterbox) against the oracle's processImage hash and rectangle, image by image.
Prints one JSON line.

    python tools/fuzz_hash.py [--cases 400] [--seed 1] [--max-side 1400]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-side", type=int, default=1400)
    args = ap.parse_args()
    from cbird_amd.hashing import process_images
    from oracle import PrestageOracle

    po = PrestageOracle()
    rng = np.random.default_rng(args.seed)
    bad = []
    images = cropped = 0
    for c in range(args.cases):
        w = int(rng.integers(32, args.max_side + 1))
        h = int(rng.integers(32, args.max_side + 1))
        if rng.random() < 0.3:  # the round numbers real material has
            w, h = [(640, 480), (1280, 720), (400, 300), (512, 512), (1024, 768), (854, 480), (320, 240)][int(rng.integers(0, 7))]
        n = int(rng.integers(1, 6))
        gray = np.zeros((n, h, w), np.uint8)
        for i in range(n):
            border = int(rng.integers(0, 40))
            img = np.clip(border + rng.integers(0, 4, (h, w)), 0, 255).astype(np.uint8)
            t = b = le = r = 0
            k = rng.random()
            if k < 0.4:
                t = b = int(rng.integers(0, h // 6 + 1))
            elif k < 0.6:
                le = r = int(rng.integers(0, w // 6 + 1))
            elif k < 0.75:
                t, b = int(rng.integers(0, h // 6 + 1)), int(rng.integers(0, h // 6 + 1))
            # smooth-ish content: block noise upsampled, so hashes are not all-noise
            bh, bw = max(1, (h - t - b) // 16 + 1), max(1, (w - le - r) // 16 + 1)
            blocks = rng.integers(70, 256, (bh, bw)).astype(np.uint8)
            inner = np.kron(blocks, np.ones((16, 16), np.uint8))[: h - t - b, : w - le - r]
            inner = np.clip(inner.astype(np.int16) + rng.integers(-6, 7, inner.shape), 62, 255).astype(np.uint8)
            img[t:h - b, le:w - r] = inner
            gray[i] = img
        for ac in (20, None):
            got, rects = process_images(gray, ac)
            for i in range(n):
                wh, wr = po.process_image(gray[i], ac)
                images += 1
                cropped += int(wr.tolist() != [0, 0, w, h])
                if int(got[i]) != wh or rects[i].tolist() != wr.tolist():
                    bad.append({"case": c, "w": w, "h": h, "i": i, "autocrop": ac, "rect": rects[i].tolist(),
                                "want_rect": wr.tolist(), "bits": bin(int(got[i]) ^ wh).count("1")})
    print(json.dumps({"cases": args.cases, "images": images, "cropped": cropped, "n_mismatches": len(bad),
                      "mismatches": bad[:10]}))


if __name__ == "__main__":
    main()
