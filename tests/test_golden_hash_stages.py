"""tests/golden/hash_stages.npz: frozen outputs of the image-side stages (dctHash64 over every resize branch,
makeKeyPointHashes, sizeLongestSide).  The vectors come from this repository's oracle, not from cbird (no OpenCV
2.4 here: "parity unpinned") -- they pin the oracle against accidental change (CPU test) and give the GPU a second,
frozen reference (GPU test)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import load_golden

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("gen_hash_stages", os.path.join(HERE, "golden", "gen_golden_hash_stages.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.fixture(params=["cvdct", "canon"])
def variant(request, orc):
    """both evaluations of stages 3/5 (oracle/cv_dct32.c and the canonical matrix form); golden key suffix"""
    orc.set_hash_variant(1 if request.param == "cvdct" else 0)
    yield "" if request.param == "cvdct" else "_canon"
    orc.set_hash_variant(1)


def test_oracle_reproduces_the_frozen_vectors(orc, variant):
    g = load_golden("hash_stages.npz")
    assert g["geometries"].tolist() == [list(x) for x in gen.GEOMETRIES]
    for i, (w, h) in enumerate(gen.GEOMETRIES):
        img = gen.image(w, h, 100 + i)
        assert orc.dcthash64(img) == int(g["hashes" + variant][i]), (w, h)
        assert (orc.tile32(img) == g["tiles"][i]).all(), (w, h)
    for i, (w, h) in enumerate([(400, 300), (200, 150)]):
        hs, after = orc.keypoint_hashes(gen.image(w, h, 500 + i), gen.keypoints(w, h, 120, 600 + i))
        assert hs.tolist() == g[f"kp_hashes_{i}" + variant].tolist()
        assert int(after.astype(np.uint64).sum()) == int(g["kp_after_sum"][i])
        assert (orc.tile32(after) == g[f"kp_after_tile_{i}"]).all()
    assert (orc.size_longest_side(gen.image(500, 300, 700), 128) == g["resized"]).all()


@pytest.mark.gpu
def test_gpu_reproduces_the_frozen_vectors(gpu):
    from cbird_amd.hashing import make_keypoint_hashes, size_longest_side

    variant = ""  # the product ships ONE evaluation of stages 3 / 5: cv::dct / cv::sum as OpenCV 2.4 runs them
    g = load_golden("hash_stages.npz")
    for i, (w, h) in enumerate(gen.GEOMETRIES):
        assert gpu.dct_hash64(gen.image(w, h, 100 + i)) == int(g["hashes" + variant][i]), (w, h)
    imgs = [gen.image(400, 300, 500), gen.image(200, 150, 501)]
    kps = [gen.keypoints(400, 300, 120, 600), gen.keypoints(200, 150, 120, 601)]
    hs, after = make_keypoint_hashes(imgs, kps, return_images=True)
    for i in range(2):
        assert hs[i].tolist() == g[f"kp_hashes_{i}" + variant].tolist()
        assert int(after[i].astype(np.uint64).sum()) == int(g["kp_after_sum"][i])
    assert (size_longest_side(gen.image(500, 300, 700)[None], 128)[0] == g["resized"]).all()


def test_one_fma_divides_rounds_and_accumulates_exactly():
    """The division step of k_dcthash_256_band and of k_band_area, exhaustively: the column sum
    S of a 7x7 window (0 .. 49 * 255) lives as the integer 0x4B000000 + S = the float 2^23 + S; with c = 42799 * 2^-21,
    fma(2^23 + S, c, acc) adds 171196 + nearest(S / 49) to an integer-valued acc below 2^24 -- the blur's
    round-to-nearest (cv::blur, src/cvutil.cpp:463) and the running sum of an 8x8 cell in one instruction."""
    c32 = np.float32(42799 / 2 ** 21)
    assert float(c32) == 42799 / 2 ** 21                      # c is a float
    S = np.arange(0, 49 * 255 + 1, dtype=np.int64)
    want = (2 * S + 49) // 98                                # nearest(S / 49); 49 is odd, no ties
    x = (S.astype(np.uint32) + np.uint32(0x4B000000)).view(np.float32)
    assert (x.astype(np.float64) == 2.0 ** 23 + S).all()
    # the product is exact in double (24 x 16 significant bits), the sum too: rounding it once to float is the fma
    for acc in (2.0 ** 23, 2.0 ** 23 + 3 * 171196 + 3 * 255, 2.0 ** 23 + 7 * 171196 + 7 * 255):
        y = (x.astype(np.float64) * float(c32) + acc).astype(np.float32)
        assert (y.astype(np.float64) - acc - 171196 == want).all()
        assert float(y.max()) < 2.0 ** 24
    frac = (S * 42799) % 2 ** 21 / 2 ** 21                    # S * c is never closer than 0.01 to a tie
    assert np.abs(frac - 0.5).min() > 0.01
