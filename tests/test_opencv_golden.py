"""Real OpenCV 2.4.13.7 goldens for dctHash64 and its pre-stages -- WHEN THEY EXIST.

tools/gen_golden_opencv.cpp has to be run where that library is installed (it is not in this image: the hash side of
the oracle is "parity unpinned", NOTES.md section 4); tools/opencv_golden_to_npz.py turns its output into
tests/golden/opencv_hash.npz.  Until that file is committed the golden tests below SKIP, loudly; what always runs is
the check that the C++ tool and its Python twin generate identical input images (the tool is compiled here with
-DNO_OPENCV, which leaves only the generator)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "opencv_hash.npz")
_spec = importlib.util.spec_from_file_location("opencv_golden_to_npz", os.path.join(ROOT, "tools", "opencv_golden_to_npz.py"))
conv = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(conv)

needs_golden = pytest.mark.skipif(not os.path.exists(GOLD), reason="tests/golden/opencv_hash.npz absent: nobody has run "
                                  "tools/gen_golden_opencv.cpp against OpenCV 2.4.13.7 yet -- hash parity UNPINNED")


def test_generator_twin_matches_the_cpp_tool(tmp_path):
    exe = tmp_path / "gen_selftest"
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-DNO_OPENCV", os.path.join(ROOT, "tools", "gen_golden_opencv.cpp"),
                           "-o", str(exe)])
    lines = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    assert len(lines) == len(conv.GEOMETRIES)
    for i, ((w, h), line) in enumerate(zip(conv.GEOMETRIES, lines)):
        t = line.split()
        assert (int(t[1]), int(t[2]), int(t[3])) == (w, h, 100 + i)
        assert conv.checksums(conv.gen_image(w, h, 100 + i)) == (int(t[4]), int(t[5])), (w, h)


def test_converter_round_trip(tmp_path, orc):
    """the text format the C++ tool prints, fed with the ORACLE's outputs: parse() recovers them (so a real file
    will be read correctly); also documents that such a self-made file is NOT a pin and must not be committed"""
    w, h, seed = 64, 64, 101
    img = conv.gen_image(w, h, seed)
    hv, co, th = orc.hash_from_tile32(orc.tile32(img), with_coefs=True)
    line = "H %d %d %d %016x %08x %s %s\n" % (w, h, seed, hv, np.float32(th).view(np.uint32),
                                            " ".join("%08x" % c for c in co.view(np.uint32)), orc.tile32(img).tobytes().hex())
    p = tmp_path / "t.txt"
    p.write_text("V self-made\n" + line)
    d = conv.parse(str(p))
    assert int(d["hashes"][0]) == hv and (d["tiles"][0] == orc.tile32(img)).all()
    assert (d["coef_bits"][0] == co.view(np.uint32)).all()


def test_pin_report_names_the_first_stage_that_differs(tmp_path, orc):
    """tools/pin_report.py (the last step of tools/pin_with_opencv.sh) on a throw-away golden file made from the ORACLE's
    own outputs: every stage agrees; with one tile byte changed the blur / resize stage is the first named, with the
    cv::dct restatement right behind it ok (it is judged on the golden's tile, not on the oracle's)"""
    recs = ["V 2.4.13.7-selfmade"]
    for w, h, seed in ((64, 64, 101), (300, 200, 107), (256, 256, 102)):
        img = conv.gen_image(w, h, seed)
        tile = orc.tile32(img)
        hv, co, th = orc.hash_from_tile32(tile, with_coefs=True)
        recs.append("H %d %d %d %016x %08x %s %s" % (w, h, seed, hv, np.float32(th).view(np.uint32),
                                                       " ".join("%08x" % c for c in co.view(np.uint32)), tile.tobytes().hex()))
    txt, npz = tmp_path / "g.txt", tmp_path / "g.npz"
    txt.write_text("\n".join(recs) + "\n")
    np.savez(npz, **conv.parse(str(txt)))
    env = dict(os.environ, CBH_PIN_GOLD=str(npz))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_report.py")], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "every stage with a golden agrees" in out.stdout, out.stdout + out.stderr
    d = dict(np.load(npz))
    d["tiles"] = d["tiles"].copy()
    d["tiles"][1, 3, 4] ^= 1
    np.savez(npz, **d)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_report.py")], capture_output=True, text=True, env=env)
    assert out.returncode == 1, out.stdout + out.stderr
    assert "FIRST stage that disagrees with OpenCV: cv::blur" in out.stdout and "(300, 200, 107)" in out.stdout


@needs_golden
def test_oracle_matches_opencv(orc):
    g = np.load(GOLD)
    assert str(g["cv_version"]).startswith("2.4.13"), "goldens must come from the version cbird pins"
    bad_tiles, bad_hash, bad_coef = [], [], []
    for i, (w, h, seed) in enumerate(g["hash_whs"].tolist()):
        img = conv.gen_image(w, h, seed)
        tile = orc.tile32(img)
        if not (tile == g["tiles"][i]).all():
            bad_tiles.append((w, h, seed))
            tile = g["tiles"][i]  # judge stages 3-6 on OpenCV's own tile
        hv, co, th = orc.hash_from_tile32_v(tile, 1, with_coefs=True)
        if hv != int(g["hashes"][i]):
            bad_hash.append((w, h, seed, hex(hv ^ int(g["hashes"][i]))))
        if not (co.view(np.uint32) == g["coef_bits"][i]).all():
            bad_coef.append((w, h, seed))
    assert not bad_tiles, f"stage 1-2 (blur/resize) differ from OpenCV: {bad_tiles[:10]}"
    assert not bad_coef, f"stage 3 (cv::dct restatement, oracle/cv_dct32.c) differs bitwise: {bad_coef[:10]}"
    assert not bad_hash, f"hash differs: {bad_hash[:10]}"
    w, h, seed = g["gray_whs"].tolist()
    bgr = np.stack([conv.gen_image(w, h, seed + c) for c in range(3)], -1)
    assert (orc.bgr2gray(bgr) == g["gray"]).all(), "cvtColor(BGR2GRAY)"
    w, h, seed, size = g["lanczos_whs_size"].tolist()
    assert (orc.size_longest_side(conv.gen_image(w, h, seed), size) == g["lanczos"]).all(), "resize(INTER_LANCZOS4)"
    w, h, seed = g["rect_whs"].tolist()
    hs, after = orc.keypoint_hashes(conv.gen_image(w, h, seed), g["rects"].astype(np.float32))
    assert hs.tolist() == g["rect_hashes"].tolist(), "in-place keypoint squares"
    assert int(after.astype(np.uint64).sum()) == int(g["rect_after_sum"][0])


@needs_golden
@pytest.mark.gpu
def test_gpu_matches_opencv(gpu):
    from cbird_amd.hashing import make_keypoint_hashes

    g = np.load(GOLD)
    for i, (w, h, seed) in enumerate(g["hash_whs"].tolist()):
        assert gpu.dct_hash64(conv.gen_image(w, h, seed)) == int(g["hashes"][i]), (w, h, seed)
    w, h, seed = g["rect_whs"].tolist()
    hs = make_keypoint_hashes([conv.gen_image(w, h, seed)], [g["rects"].astype(np.float32)])[0]
    assert hs.tolist() == g["rect_hashes"].tolist()


# ---- ORB and the library pieces under ColorDescriptor::create (records P, B, F, O, A, M, U, K) ------------------------
def _has(key):
    return os.path.exists(GOLD) and key in np.load(GOLD).files


def test_converter_parses_the_orb_and_colour_records(tmp_path):
    """the text format of the new records, fed with the ORACLE's outputs: parse() recovers them (a self-made file like
    this is not a pin and must not be committed)"""
    from oracle import ColorCreateOracle, OrbOracle

    o, c = OrbOracle(), ColorCreateOracle()
    img = conv.gen_image(96, 80, 900)
    lvl = o.resize_linear(img, 80, 67)
    bl = o.gauss7_blur(img)
    sc = o.fast_nms_scores(img)
    ys, xs = np.nonzero(sc)
    mask = c.ellipse_mask(40, 30)
    luv = c.bgr2luv(np.float32(10 / 255), np.float32(20 / 255), np.float32(30 / 255))
    lines = ["V self-made",
             "P 96 80 900 80 67 " + lvl.tobytes().hex(),
             "B 96 80 900 " + bl.tobytes().hex(),
             "F 96 80 900 %d %s" % (len(xs), " ".join("%d %d %d" % (x, y, sc[y, x]) for y, x in zip(ys, xs))),
             "M 40 30 " + mask.tobytes().hex(),
             "U 1 10 20 30 " + " ".join("%08x" % v for v in luv.view(np.uint32))]
    p = tmp_path / "t.txt"
    p.write_text("\n".join(lines) + "\n")
    d = conv.parse(str(p))
    assert (d["pyr"] == lvl).all() and (d["gauss"] == bl).all() and len(d["fast"]) == len(xs)
    assert (d["mask_40x30"] == mask).all() and (d["luv_bits"][0] == luv.view(np.uint32)).all()


@pytest.mark.skipif(not _has("pyr"), reason="no OpenCV goldens for ORB yet -- ORB parity UNPINNED")
def test_orb_stages_match_opencv():
    from oracle import OrbOracle

    g = np.load(GOLD)
    o = OrbOracle()
    w, h, seed, dw, dh = g["pyr_whs_dims"].tolist()
    img = conv.gen_image(w, h, seed)
    assert (o.resize_linear(img, dw, dh) == g["pyr"]).all(), "cv::resize INTER_LINEAR"
    assert (o.gauss7_blur(img) == g["gauss"]).all(), "GaussianBlur 7x7 sigma 2"
    sc = o.fast_nms_scores(img)
    ys, xs = np.nonzero(sc)
    want = g["fast"]
    assert len(want) == len(xs) and (want[:, 0] == xs).all() and (want[:, 1] == ys).all() and (want[:, 2] == sc[ys, xs]).all()
    ab = g["atan_bits"]
    for yb, xb, rb in ab:
        y, x = np.uint32(yb).view(np.float32), np.uint32(xb).view(np.float32)
        assert np.float32(o.fast_atan2(y, x)).view(np.uint32) == rb, (y, x)
    # the detector as a whole, first under the canonical retainBest rule (every tie kept: a superset of what any C++
    # library leaves) -- every OpenCV keypoint must be one of ours, with identical response and angle
    o.set_retain_order(0)
    mine = o.detect(img, int(g["orb_whs_nfeat"][3]))
    o.set_retain_order(1)
    key = {(int(k["octave"]), np.float32(k["x"]).view(np.uint32).item(), np.float32(k["y"]).view(np.uint32).item()): k
           for k in mine}
    # (pt of the goldens went through compute(): compare on the level coordinates)
    missing = 0
    for bits, octave in zip(g["orb_kp_bits"], g["orb_octave"]):
        s = o.scale(int(octave))
        x, y = np.uint32(bits[0]).view(np.float32), np.uint32(bits[1]).view(np.float32)
        lx, ly = int(np.rint(x / (s if octave else 1))), int(np.rint(y / (s if octave else 1)))
        hit = [k for k in mine if k["octave"] == octave and int(np.rint(k["x"] / (s if octave else 1))) == lx
               and int(np.rint(k["y"] / (s if octave else 1))) == ly]
        if not hit:
            missing += 1
            continue
        assert np.float32(hit[0]["angle"]).view(np.uint32) == bits[3] and np.float32(hit[0]["response"]).view(np.uint32) == bits[4]
    assert missing == 0, f"{missing} OpenCV keypoints are not in the oracle's (superset) result"
    assert len(key) >= len(g["orb_octave"])
    # then in libstdc++'s order (the default; goldens made by an OpenCV built against libstdc++): the same keypoints in
    # the same ORDER
    stl = o.detect(img, int(g["orb_whs_nfeat"][3]))
    assert len(stl) == len(g["orb_octave"]) and (stl["octave"] == g["orb_octave"]).all()
    assert (stl["angle"].view(np.uint32) == g["orb_kp_bits"][:, 3]).all()
    assert (stl["response"].view(np.uint32) == g["orb_kp_bits"][:, 4]).all()


@pytest.mark.skipif(not _has("luv_bits"), reason="no OpenCV goldens for ColorDescriptor::create yet -- parity UNPINNED")
def test_colour_create_pieces_match_opencv():
    from oracle import ColorCreateOracle

    g = np.load(GOLD)
    c = ColorCreateOracle()
    for name in g.files:
        if name.startswith("mask_"):
            cols, rows = (int(v) for v in name[5:].split("x"))
            assert (c.ellipse_mask(cols, rows) == g[name]).all(), name
    s255 = np.float32(1.0 / 255.0)
    for (b, gg, r), bits in zip(g["luv_bgr"], g["luv_bits"]):
        got = c.bgr2luv(np.float32(b) * s255, np.float32(gg) * s255, np.float32(r) * s255)
        assert (got.view(np.uint32) == bits).all(), (b, gg, r)
    w, h, seed = 120, 90, 950
    planes = [conv.gen_image(w, h, seed + i).astype(np.float32) * s255 for i in range(3)]
    samples = np.array([c.bgr2luv(planes[0][y, x], planes[1][y, x], planes[2][y, x]) for y in range(h) for x in range(w)],
                       np.float32)
    labels, centers, _ = c.kmeans(samples)
    assert (labels == g["kmeans_labels"]).all() and (centers.view(np.uint32) == g["kmeans_center_bits"]).all()
