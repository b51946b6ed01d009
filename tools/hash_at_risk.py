#!/usr/bin/env python3
"""At-risk-bit statistics of dctHash64 (SURVEY.md 7 hard part 1).

cbird's hash thresholds 63 float DCT coefficients against their mean (src/cvutil.cpp:528-538).  The coefficients come
from cv::dct, whose float arithmetic lives in OpenCV 2.4.13.7 and cannot be run here, so a hash bit whose coefficient
sits within float-rounding distance of the threshold may come out differently in cbird than in any restatement.
This tool MEASURES that set on the bench workload (bench.py's 1M synthetic 256x256 images) and on the golden-stage
images:

  * three evaluations of stages 3-6 on the same 32x32 tiles: canonical matrix form (variant 0), cv::dct as recalled
    (variant 1, the default), float64 with the exact basis (yardstick);
  * per hash: do the evaluations agree; the smallest float64 margin |coef - thresh| over bits 1..63, in absolute
    units and in float ulps at the threshold's magnitude; the largest deviation of a float evaluation's
    (coef - thresh) from the float64 value (eps);
  * the bound: a bit can only differ between two float evaluations whose errors stay below eps when its float64
    margin is below 2*eps.  With eps_max = the largest deviation observed over all tiles and both float variants,
    the fraction of hashes with a margin < 2*eps_max is an upper bound on the hashes that any such evaluation --
    cv::dct's included, if its error is of the size of the two we can run -- can change, and each of those hashes
    changes in at most the bits inside that band (counted too).

On the GPU box the tiles come from the HIP kernels (and their hashes under both variants are compared with the
oracle's on all images: a full-size parity check); with --cpu the oracle produces the tiles (small n only).

    python tools/hash_at_risk.py [--images 1000000] [--out profiles/r02_hash_at_risk.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def popcount64(a):
    a = a.astype(np.uint64)
    return np.unpackbits(a.view(np.uint8)).reshape(len(a), 64).sum(1)


def bits_of(a):
    """per-bit-position counts over an array of u64 (bit i of the hash -> index i)"""
    return np.unpackbits(a.astype("<u8").view(np.uint8), bitorder="little").reshape(len(a), 64).sum(0)


def summarise(r, label):
    n = len(r["h0"])
    x01, x02, x12 = r["h0"] ^ r["h1"], r["h0"] ^ r["h2"], r["h1"] ^ r["h2"]
    eps_max = float(max(r["err0"].max(), r["err1"].max()))
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(r["thr64"]), 1e-30))) - 23)
    margin_ulps = r["min_margin"] / ulp
    out = {
        "label": label,
        "tiles": int(n),
        "hashes_differ": {
            "canonical_vs_cvdct": int((x01 != 0).sum()),
            "canonical_vs_f64": int((x02 != 0).sum()),
            "cvdct_vs_f64": int((x12 != 0).sum()),
        },
        "bits_differ": {
            "canonical_vs_cvdct": int(popcount64(x01).sum()),
            "canonical_vs_f64": int(popcount64(x02).sum()),
            "cvdct_vs_f64": int(popcount64(x12).sum()),
        },
        "max_bits_differ_in_one_hash": int(max(popcount64(x01).max(), popcount64(x02).max(), popcount64(x12).max())),
        "bit_positions_that_flipped": {
            "canonical_vs_cvdct": np.nonzero(bits_of(x01))[0].tolist(),
            "cvdct_vs_f64": np.nonzero(bits_of(x12))[0].tolist(),
        },
        "float_error_of_coef_minus_thresh": {
            "canonical_max": float(r["err0"].max()), "canonical_p99": float(np.quantile(r["err0"], 0.99)),
            "cvdct_max": float(r["err1"].max()), "cvdct_p99": float(np.quantile(r["err1"], 0.99)),
            "eps_max": eps_max,
        },
        "min_margin_f64": {
            "min": float(r["min_margin"].min()),
            "quantiles_1e-4_1e-3_1e-2_0.1_0.5": [float(np.quantile(r["min_margin"], q)) for q in (1e-4, 1e-3, 1e-2, 0.1, 0.5)],
            "in_float_ulps_at_thresh_quantiles_1e-4_1e-3_1e-2": [float(np.quantile(margin_ulps, q)) for q in (1e-4, 1e-3, 1e-2)],
        },
        "fraction_of_hashes_with_margin_below": {
            f"{m:g}": float((r["min_margin"] < m).mean()) for m in (1e-4, 3e-4, 1e-3, 3e-3, 1e-2)
        },
    }
    band = 2 * eps_max
    at_risk = r["min_margin"] < band
    out["upper_bound"] = {
        "band": band,
        "hashes_at_risk": int(at_risk.sum()),
        "fraction_of_hashes_at_risk": float(at_risk.mean()),
        "statement": "a float evaluation of the same transform with |error| <= eps_max can differ from another such "
                     "evaluation only in hashes whose smallest float64 margin is below 2*eps_max",
    }
    return out


def tiles_on_gpu(n, seed, chunk):
    """bench.py's synthetic images, hashed on the device under both variants; yields (tiles u8[m,1024], h_canon,
    h_cvdct) per chunk"""
    import torch

    import bench
    from cbird_amd import _lib

    import cbird_amd

    cbird_amd.require_device()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        imgs = bench.gen_images(torch, dev, a, b, n, seed)
        m = b - a
        tiles = torch.empty((m, 1024), dtype=torch.uint8, device=dev)
        hs = []
        for variant in (0, 1):
            L.cbh_set_tuning(b"hash_dct", variant)
            out = torch.empty(m, dtype=torch.int64, device=dev)
            _lib.check(L.cbh_dcthash_tiles_dev(imgs.data_ptr(), m, 256, 256, 256, 65536, out.data_ptr(),
                                               tiles.data_ptr(), 0, None), "tiles")
            # and the kernel without the tile dump (the one bench.py times) must agree
            out2 = torch.empty(m, dtype=torch.int64, device=dev)
            _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), m, 256, 256, 256, 65536, out2.data_ptr(), 0, None), "hash")
            torch.cuda.synchronize()
            assert bool((out == out2).all())
            hs.append(out.cpu().numpy().view(np.uint64))
        L.cbh_set_tuning(b"hash_dct", 1)
        yield tiles.cpu().numpy(), hs[0], hs[1]
        del imgs, tiles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--cpu", action="store_true", help="tiles from the oracle (no device; keep --images small)")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    from oracle import Oracle

    orc = Oracle()
    t0 = time.time()
    parts, gpu_mismatch = [], {"canonical": 0, "cvdct": 0}
    if args.cpu:
        from cbird_amd import synth

        imgs = synth.make_images(args.images, seed=args.seed)
        tiles = np.stack([orc.tile32(i) for i in imgs]).reshape(-1, 1024)
        parts.append(orc.hash_tiles_risk(tiles))
    else:
        for tiles, hg0, hg1 in tiles_on_gpu(args.images, args.seed, args.chunk):
            r = orc.hash_tiles_risk(tiles)
            gpu_mismatch["canonical"] += int((r["h0"] != hg0).sum())
            gpu_mismatch["cvdct"] += int((r["h1"] != hg1).sum())
            parts.append(r)
            print(f"  {sum(len(p['h0']) for p in parts)} tiles, {time.time() - t0:.0f} s", file=sys.stderr)
    r = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    res = {"tool": "tools/hash_at_risk.py", "workload": f"bench.py synthetic 256x256 images, seed {args.seed}",
           "bench": summarise(r, "bench images")}
    if not args.cpu:
        res["gpu_vs_oracle_hash_mismatches_over_all_images"] = gpu_mismatch
    # the golden-stage images (every resize branch), tiles from the oracle
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "tests", "golden", "gen_golden_hash_stages.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gt = np.stack([orc.tile32(gen.image(w, h, 100 + i)) for i, (w, h) in enumerate(gen.GEOMETRIES)]).reshape(-1, 1024)
    res["golden_stage_images"] = summarise(orc.hash_tiles_risk(gt), "tests/golden/hash_stages.npz geometries")
    res["seconds"] = round(time.time() - t0, 1)
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
