#!/usr/bin/env python3
"""Stage-by-stage comparison of the restated OpenCV pieces with real OpenCV 2.4.13.7 goldens
(tests/golden/opencv_hash.npz, made by tools/pin_with_opencv.sh).  Prints one line per stage -- ok / DIFFERS / no
golden -- in pipeline order, names the first stage that disagrees and the files that restate it, and exits 1 if any does.
The restatements live in oracle/ (the GPU kernels are bit-exact to them by the -m gpu tests, so a stage that is right here
is right there; a stage that is wrong has to be corrected in both places named)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import opencv_golden_to_npz as conv  # noqa: E402

# (CBH_PIN_GOLD: the self-test of tests/test_opencv_golden.py points the report at a throw-away file)
GOLD = os.environ.get("CBH_PIN_GOLD", os.path.join(ROOT, "tests", "golden", "opencv_hash.npz"))
PATTERN = os.path.join(ROOT, "tests", "golden", "orb_bit_pattern_31.txt")


def main():
    if not os.path.exists(GOLD):
        print("no tests/golden/opencv_hash.npz: run tools/pin_with_opencv.sh <opencv-2.4.13.7-prefix> first")
        return 2
    from oracle import ColorCreateOracle, Oracle, OrbOracle

    g = np.load(GOLD)
    orc = Oracle()
    results = []  # (stage, status, detail, files)

    def stage(name, files, fn, needs=()):
        if any(k not in g.files for k in needs):
            results.append((name, "no golden", "", files))
            return
        try:
            bad = fn()
        except Exception as e:  # a stage that cannot even run is a stage that differs
            bad = [repr(e)[:200]]
        results.append((name, "ok" if not bad else "DIFFERS", str(bad[:4]) if bad else "", files))

    H = g["hash_whs"].tolist()
    imgs = [conv.gen_image(w, h, s) for w, h, s in H]
    stage("cv::blur + cv::resize(INTER_AREA / bilinear) -> 32x32 tile (src/cvutil.cpp:446-471)",
          "oracle/cbird_oracle.c (box_blur, area_resize), cbird_amd/csrc/dcthash.hip",
          lambda: [tuple(H[i]) for i in range(len(H)) if not (orc.tile32(imgs[i]) == g["tiles"][i]).all()])
    stage("cv::dct 32x32, first 9x9 coefficients, bitwise (src/cvutil.cpp:475-482)",
          "oracle/cv_dct32.c, cbird_amd/csrc/cv_dct32_dev.h",
          lambda: [tuple(H[i]) for i in range(len(H))
                   if not (orc.hash_from_tile32_v(g["tiles"][i], 1, with_coefs=True)[1].view(np.uint32) == g["coef_bits"][i]).all()])
    stage("cv::sum threshold + bits -> the 64-bit hash (src/cvutil.cpp:513-545)",
          "oracle/cv_dct32.c (cv_sum64), oracle/cbird_oracle.c (hash_from_tile)",
          lambda: [tuple(H[i]) for i in range(len(H))
                   if orc.hash_from_tile32_v(g["tiles"][i], 1, with_coefs=True)[0] != int(g["hashes"][i])])

    def gray():
        w, h, s = g["gray_whs"].tolist()
        bgr = np.stack([conv.gen_image(w, h, s + c) for c in range(3)], -1)
        return [] if (orc.bgr2gray(bgr) == g["gray"]).all() else [(w, h, s)]

    stage("cvtColor(BGR2GRAY) (src/cvutil.cpp:1265-1283)", "oracle/cbird_oracle.c (bgr2gray), cbird_amd/csrc/prestage.hip", gray,
          ("gray",))

    def lanczos():
        w, h, s, size = g["lanczos_whs_size"].tolist()
        return [] if (orc.size_longest_side(conv.gen_image(w, h, s), size) == g["lanczos"]).all() else [(w, h, s, size)]

    stage("resize(INTER_LANCZOS4) of sizeLongestSide (src/cvutil.cpp:1932-1950)",
          "oracle/cbird_oracle.c (lanczos4), cbird_amd/csrc/prestage.hip", lanczos, ("lanczos",))

    def rects():
        w, h, s = g["rect_whs"].tolist()
        hs, after = orc.keypoint_hashes(conv.gen_image(w, h, s), g["rects"].astype(np.float32))
        ok = hs.tolist() == g["rect_hashes"].tolist() and int(after.astype(np.uint64).sum()) == int(g["rect_after_sum"][0])
        return [] if ok else [(w, h, s)]

    stage("Media::makeKeyPointHashes: in-place squares (src/media.cpp:874-923)",
          "oracle/cbird_oracle.c (keypoint_hashes), cbird_amd/csrc/dcthash.hip (k_kp_hashes)", rects, ("rects",))
    # ---- ORB
    o = OrbOracle()
    if "pyr" in g.files:
        w, h, s, dw, dh = g["pyr_whs_dims"].tolist()
        img = conv.gen_image(w, h, s)
        stage("ORB pyramid step: resize(INTER_LINEAR)", "oracle/orb_oracle.c (resize_linear), cbird_amd/csrc/orb.hip",
              lambda: [] if (o.resize_linear(img, dw, dh) == g["pyr"]).all() else ["pyr"])
        stage("ORB GaussianBlur 7x7 sigma 2", "oracle/orb_oracle.c (gauss7), orb.hip",
              lambda: [] if (o.gauss7_blur(img) == g["gauss"]).all() else ["gauss"])

        def fast():
            sc = o.fast_nms_scores(img)
            ys, xs = np.nonzero(sc)
            want = g["fast"]
            ok = len(want) == len(xs) and (want[:, 0] == xs).all() and (want[:, 1] == ys).all() and (want[:, 2] == sc[ys, xs]).all()
            return [] if ok else [f"{len(xs)} corners vs {len(want)}"]

        stage("cv::FAST(20, nonmax)", "oracle/orb_oracle.c (fast9_16), orb.hip (k_orb_fast)", fast)
        stage("cv::fastAtan2", "oracle/orb_oracle.c (fast_atan2), orb.hip",
              lambda: [(int(a), int(b)) for a, b, r in g["atan_bits"]
                       if np.float32(o.fast_atan2(np.uint32(a).view(np.float32), np.uint32(b).view(np.float32))).view(np.uint32) != r])

        def detect():
            kp = o.detect(img, int(g["orb_whs_nfeat"][3]))
            ok = len(kp) == len(g["orb_octave"]) and (kp["octave"] == g["orb_octave"]).all() and \
                (kp["angle"].view(np.uint32) == g["orb_kp_bits"][:, 3]).all() and \
                (kp["response"].view(np.uint32) == g["orb_kp_bits"][:, 4]).all()
            return [] if ok else [f"{len(kp)} keypoints vs {len(g['orb_octave'])}"]

        stage("ORB detect: Harris, retainBest (libstdc++ order), orientation", "oracle/orb_oracle.c, oracle/retain_stl.cpp, orb.hip",
              detect)
        if os.path.exists(PATTERN):
            from cbird_amd.orb import load_pattern

            def describe():
                o.set_pattern(load_pattern(PATTERN))
                kp = o.detect(img, int(g["orb_whs_nfeat"][3]))
                _, desc = o.compute(img, kp)
                return [] if desc.shape == g["orb_desc"].shape and (desc == g["orb_desc"]).all() else ["descriptors"]

            stage("ORB compute: rBRIEF with bit_pattern_31_", "oracle/orb_oracle.c (compute), orb.hip (k_orb_describe)", describe)
        else:
            results.append(("ORB compute: rBRIEF with bit_pattern_31_", "no golden",
                            "give pin_with_opencv.sh the OpenCV source tree", "tests/golden/orb_bit_pattern_31.txt"))
    else:
        results.append(("ORB stages", "no golden", "", "tools/gen_golden_opencv.cpp records P B F O A"))
    # ---- the library pieces under ColorDescriptor::create
    if "luv_bits" in g.files:
        c = ColorCreateOracle()
        stage("cv::ellipse mask", "oracle/colordesc_oracle.c (ellipse), colordesc_create.hip",
              lambda: [n for n in g.files if n.startswith("mask_") and
                       not (c.ellipse_mask(*(int(v) for v in n[5:].split("x"))) == g[n]).all()])
        s255 = np.float32(1.0 / 255.0)
        stage("cvtColor(BGR2Luv) on floats", "oracle/colordesc_oracle.c (bgr2luv), colordesc_create.hip",
              lambda: [tuple(int(v) for v in bgr) for bgr, bits in zip(g["luv_bgr"], g["luv_bits"])
                       if not (c.bgr2luv(*(np.float32(v) * s255 for v in bgr)).view(np.uint32) == bits).all()])

        def kmeans():
            w, h, seed = 120, 90, 950
            planes = [conv.gen_image(w, h, seed + i).astype(np.float32) * s255 for i in range(3)]
            samples = np.array([c.bgr2luv(planes[0][y, x], planes[1][y, x], planes[2][y, x]) for y in range(h)
                                for x in range(w)], np.float32)
            labels, centers, _ = c.kmeans(samples)
            ok = (labels == g["kmeans_labels"]).all() and (centers.view(np.uint32) == g["kmeans_center_bits"]).all()
            return [] if ok else ["labels / centres"]

        stage("cv::kmeans (k-means++ seeding, cv::RNG)", "oracle/colordesc_oracle.c (kmeans), colordesc_create.hip", kmeans)
    else:
        results.append(("ColorDescriptor::create pieces", "no golden", "", "tools/gen_golden_opencv.cpp records M U K"))

    print("OpenCV", str(g["cv_version"]))
    first = None
    for name, status, detail, files in results:
        print(f"  [{status:9s}] {name}" + (f"   {detail}" if detail else ""))
        if status == "DIFFERS" and first is None:
            first = (name, files)
    if first:
        print(f"\nFIRST stage that disagrees with OpenCV: {first[0]}\n  restated in: {first[1]}\n  (stages after it consume "
              "its output where the pipeline chains them: fix this one, run again)")
        return 1
    print("\nevery stage with a golden agrees with OpenCV " + str(g["cv_version"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
