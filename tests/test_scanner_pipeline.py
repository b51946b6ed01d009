"""Scanner::processImage for a batch (cbh_index_images, /root/reference/src/scanner.cpp:828-895): the chained device
pipeline equals the stages of the oracle applied one after the other in the reference's order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _images(rng, n, w, h, border):
    imgs = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        img = np.full((h, w, 3), 120, np.int32)
        for _ in range(60):
            x, y = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
            img[y: y + int(rng.integers(6, h // 3)), x: x + int(rng.integers(6, w // 3))] = rng.integers(0, 256, 3)
        img = (img + rng.integers(-4, 5, img.shape)).clip(0, 255)
        if border and i % 2 == 0:   # letterbox: autocrop removes it, every later stage works on the kept region
            img[: 20 + i] = 0
            img[h - 25:] = 0
        imgs[i] = img
    return imgs


@pytest.mark.parametrize("w,h,border", [(480, 360, True), (300, 420, False)])
def test_pipeline_equals_the_stages_in_order(gpu, orc, w, h, border):
    from cbird_amd import orb as gorb
    from cbird_amd.scanner import IndexParams, process_images
    from oracle import ColorCreateOracle, OrbOracle, PrestageOracle

    rng = np.random.default_rng(w)
    imgs = _images(rng, 5, w, h, border)
    pat = gorb.synthetic_pattern(2)
    gorb.set_pattern(pat)
    oo, co, po = OrbOracle(), ColorCreateOracle(), PrestageOracle()
    oo.set_pattern(pat)
    res = process_images(imgs, IndexParams())
    ncrop = 0
    for img, r in zip(imgs, res):
        gray = po.bgr2gray(img)
        hsh, rect = po.process_image(img, 20)
        assert r.dctHash == hsh and r.cropRect == tuple(int(v) for v in rect)
        ncrop += r.cropRect != (0, 0, w, h)
        kept = np.ascontiguousarray(gray[rect[1]: rect[3], rect[0]: rect[2]])
        small = orc.size_longest_side(kept, 400)
        assert r.resizedDims == (small.shape[1], small.shape[0])
        kp = oo.detect(small, 400)
        kp2, desc = oo.compute(small, kp)
        assert len(kp2) > 100 and (r.keyPoints == kp2).all() and (r.keyPointDescriptors == desc).all()
        tri = np.stack([kp2["x"], kp2["y"], kp2["size"]], 1)
        want_h, _ = orc.keypoint_hashes(small, tri)
        assert len(want_h) > 10 and (r.keyPointHashes == want_h).all()
        want_c, _ = co.create(img)
        assert (np.frombuffer(r.colorDescriptor.tobytes(), np.uint8) == want_c).all()
    assert ncrop > 0 or not border
    # fdct alone: the hashes start from the keypoints as DETECTED (no descriptor pass rewrote them)
    only = process_images(imgs[:2], IndexParams(algos=1 << 1))
    for img, r in zip(imgs[:2], only):
        gray = po.bgr2gray(img)
        _, rect = po.process_image(img, 20)
        small = orc.size_longest_side(np.ascontiguousarray(gray[rect[1]: rect[3], rect[0]: rect[2]]), 400)
        kp = oo.detect(small, 400)
        assert (r.keyPoints == kp).all() and r.dctHash == 0 and r.colorDescriptor is None
        want_h, _ = orc.keypoint_hashes(small, np.stack([kp["x"], kp["y"], kp["size"]], 1))
        assert (r.keyPointHashes == want_h).all() and len(r.keyPointDescriptors) == 0


def test_pipeline_grey_input_and_arguments(gpu, orc):
    from cbird_amd import _lib
    from cbird_amd.scanner import IndexParams, process_images

    rng = np.random.default_rng(5)
    grey = rng.integers(0, 256, (3, 200, 260), dtype=np.uint8)
    res = process_images(grey, IndexParams(algos=(1 << 0) | (1 << 3), autocrop=False))
    for img, r in zip(grey, res):
        assert r.dctHash == orc.dcthash64(img) and r.colorDescriptor is None   # "passed a grayscale image"
        assert r.cropRect == (0, 0, 260, 200) and len(r.keyPoints) == 0
    assert process_images(np.zeros((0, 10, 10), np.uint8)) == []
    with pytest.raises(ValueError):
        process_images(np.zeros((2, 10), np.uint8))
    L = _lib.lib()
    assert L.cbh_index_images(None, 1, 4, 4, 4, 16, 1, None, None, None, None, None, None, None, None, None, None, None,
                              0) == _lib.CBH_E_INVAL


@pytest.mark.gpu
def test_large_batches_are_split_over_host_threads_with_the_same_results(gpu):
    """cbh_index_images cuts a batch of >= 1024 images without the colour leg into sub-batches on their own host threads
    and streams (upload of one overlapping the kernels of another): image by image the results of the unsplit calls"""
    from cbird_amd import orb
    from cbird_amd.scanner import IndexParams, process_images

    orb.set_pattern(orb.synthetic_pattern())
    rng = np.random.default_rng(12)
    base = []
    for _ in range(25):
        blocks = rng.integers(20, 256, (8, 10)).astype(np.uint8)
        base.append(np.kron(blocks, np.ones((12, 12), np.uint8)) + rng.integers(0, 5, (96, 120)).astype(np.uint8))
    imgs = np.stack([np.roll(base[i % 25], i // 25, axis=1) for i in range(1100)])
    p = IndexParams(algos=7, numFeatures=60)
    whole = process_images(imgs, p)                       # 1100 images: two sub-batches of 550
    parts = process_images(imgs[:600], p) + process_images(imgs[600:], p)  # both below the splitting size
    assert len(whole) == len(parts) == 1100
    for a, b in zip(whole, parts):
        assert a.dctHash == b.dctHash and a.cropRect == b.cropRect and a.resizedDims == b.resizedDims
        assert (a.keyPoints == b.keyPoints).all() and (a.keyPointDescriptors == b.keyPointDescriptors).all()
        assert (a.keyPointHashes == b.keyPointHashes).all()
    assert sum(len(r.keyPoints) for r in whole) > 1100 * 10


@pytest.mark.gpu
def test_mixed_geometries_are_grouped_and_returned_in_input_order(gpu):
    from cbird_amd import orb
    from cbird_amd.scanner import IndexParams, process_image_list, process_images

    orb.set_pattern(orb.synthetic_pattern())
    rng = np.random.default_rng(21)
    shapes = [(120, 160), (120, 160, 3), (200, 150, 3), (120, 160), (96, 96, 4), (200, 150, 3), (120, 160, 3)]
    imgs = []
    for sh in shapes:
        blocks = rng.integers(20, 256, (sh[0] // 8 + 1, sh[1] // 8 + 1) + sh[2:]).astype(np.uint8)
        rep = (8, 8) + ((1,) if len(sh) == 3 else ())
        imgs.append(np.ascontiguousarray(np.kron(blocks, np.ones(rep, np.uint8))[: sh[0], : sh[1]]))
    p = IndexParams(algos=15, numFeatures=80)
    got = process_image_list(imgs, p)
    assert len(got) == len(imgs)
    for im, r in zip(imgs, got):
        one = process_images(im[None], p)[0]
        assert r.dctHash == one.dctHash and r.cropRect == one.cropRect
        assert (r.keyPoints == one.keyPoints).all() and (r.keyPointDescriptors == one.keyPointDescriptors).all()
        assert (r.keyPointHashes == one.keyPointHashes).all()
        assert (r.colorDescriptor is None) == (one.colorDescriptor is None)
        if r.colorDescriptor is not None:
            assert r.colorDescriptor.tobytes() == one.colorDescriptor.tobytes()
    assert process_image_list([], p) == []
