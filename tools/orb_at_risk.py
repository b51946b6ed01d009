"""How much of an ORB result rests on choices the restatement had to make (oracle/orb_oracle.c header) or on the last
bit of a float -- the counterpart of tools/hash_at_risk.py for SURVEY section 8 row a11.  CPU only (the oracle).

  ties     keypoints that survive only because retainBest keeps ALL responses equal to the n-th best (OpenCV keeps
           whichever of them std::nth_element left in front): per cut (FAST score, Harris response), how often the cut
           binds, how many tied keypoints it admits, and how many FINAL keypoints carry a tied FAST score
  bits     descriptor tests decided by a grey-level difference of 0 or 1 (one rounding step of the fixed-point
           Gaussian or of the resize would flip them)
  rounding rotated test coordinates within 1e-4 of a .5 boundary (a last-ulp difference in (float)cos / sin would move
           the sample by a pixel)

    python tools/orb_at_risk.py [--images 48] > profiles/r02_orb_at_risk.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def scene(rng, w, h, kind):
    if kind == 0:    # flat patches + mild noise (sharp corners, many equal FAST scores)
        img = np.full((h, w), 128, np.int32)
        for _ in range((w * h) // 1000):
            x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
            img[y: y + int(rng.integers(4, h // 4)), x: x + int(rng.integers(4, w // 4))] = int(rng.integers(0, 256))
        img = img + rng.integers(-4, 5, img.shape)
    elif kind == 1:  # smooth texture: low-pass filtered noise, photograph-like statistics
        f = rng.normal(0, 1, (h, w))
        F = np.fft.rfft2(f)
        ky, kx = np.meshgrid(np.fft.fftfreq(h), np.fft.rfftfreq(w), indexing="ij")
        F /= (1e-3 + np.hypot(kx, ky)) ** 1.2
        img = np.fft.irfft2(F, (h, w))
        img = (img - img.min()) / (img.max() - img.min()) * 255
    else:            # texture + hard edges
        img = scene(rng, w, h, 1).astype(np.int32) // 2 + scene(rng, w, h, 0).astype(np.int32) // 2
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=48)
    ap.add_argument("--kp", type=int, default=400)
    args = ap.parse_args()
    from cbird_amd.orb import synthetic_pattern
    from oracle import OrbOracle

    o = OrbOracle()
    pat = synthetic_pattern()
    o.set_pattern(pat)
    P = pat.reshape(256, 2, 2).astype(np.float32)
    rng = np.random.default_rng(7)
    per = o.features_per_level(args.kp)
    st = dict(images=0, keypoints=0, levels=0, fast_cut_binds=0, fast_tied_admitted=0, fast_kept=0, harris_levels_with_ties_at_the_cut=0,
              harris_tied_admitted=0, final_with_tied_fast_score=0, bits=0, bits_diff0=0, bits_diff1=0, coords=0,
              coords_near_half=0)
    for i in range(args.images):
        img = scene(rng, 400, 300, i % 3)
        kps = o.detect(img, args.kp)
        st["images"] += 1
        st["keypoints"] += len(kps)
        for l in range(12):
            lw, lh = o.level_size(400, 300, l)
            if lw <= 62 or lh <= 62:
                break
            lvl = o.pyramid_level(img, l)
            sc = o.fast_nms_scores(lvl)[31: lh - 31, 31: lw - 31]
            vals = np.sort(sc[sc > 0])[::-1]
            st["levels"] += 1
            n2 = 2 * int(per[l])
            cut = 0
            if len(vals) > n2 > 0:
                cut = int(vals[n2 - 1])
                kept = int((vals >= cut).sum())
                st["fast_cut_binds"] += 1
                st["fast_tied_admitted"] += kept - n2
                st["fast_kept"] += kept
            sel = kps[kps["octave"] == l]
            s = o.scale(l)
            xs = np.rint(sel["x"] / (s if l else 1)).astype(int)
            ys = np.rint(sel["y"] / (s if l else 1)).astype(int)
            full = o.fast_nms_scores(lvl)
            if cut:
                st["final_with_tied_fast_score"] += int((full[ys, xs] == cut).sum())
            if len(sel) > per[l] > 0:
                st["harris_levels_with_ties_at_the_cut"] += 1
                st["harris_tied_admitted"] += len(sel) - int(per[l])
            # descriptor tests of this level
            if len(sel):
                blurred = o.gauss7_blur(lvl).astype(np.int32)
                ang = sel["angle"] * np.float32(np.pi / 180.0)
                a = np.cos(ang.astype(np.float64)).astype(np.float32)[:, None, None]
                b = np.sin(ang.astype(np.float64)).astype(np.float32)[:, None, None]
                fx = P[None, :, :, 0] * a - P[None, :, :, 1] * b       # float32, the reference's expression order
                fy = P[None, :, :, 0] * b + P[None, :, :, 1] * a
                frac = np.abs(np.abs(np.stack([fx, fy]) - np.floor(np.stack([fx, fy]))) - 0.5)
                st["coords"] += frac.size
                st["coords_near_half"] += int((frac < 1e-4).sum())
                ix = np.rint(fx).astype(int) + xs[:, None, None]
                iy = np.rint(fy).astype(int) + ys[:, None, None]
                t = blurred[iy, ix]
                d = np.abs(t[:, :, 0] - t[:, :, 1])
                st["bits"] += d.size
                st["bits_diff0"] += int((d == 0).sum())
                st["bits_diff1"] += int((d == 1).sum())
    out = dict(st)
    out["tied_fast_admitted_per_binding_cut"] = st["fast_tied_admitted"] / max(1, st["fast_cut_binds"])
    out["final_keypoints_with_tied_fast_score_frac"] = st["final_with_tied_fast_score"] / max(1, st["keypoints"])
    out["bits_decided_by_0_or_1_grey_levels_frac"] = (st["bits_diff0"] + st["bits_diff1"]) / max(1, st["bits"])
    out["coords_within_1e-4_of_rounding_boundary_frac"] = st["coords_near_half"] / max(1, st["coords"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
