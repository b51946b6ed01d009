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
def test_gpu_reproduces_the_frozen_vectors(gpu, hash_dct):
    from cbird_amd.hashing import make_keypoint_hashes, size_longest_side

    variant = "" if hash_dct == "cvdct" else "_canon"
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
