"""Pre-stages of Scanner::processImage (src/scanner.cpp:852-862): grayscale + autocrop (src/cvutil.cpp:1265-1402)
followed by the hash of the kept region."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def po():
    from oracle import PrestageOracle

    return PrestageOracle()


def letterboxed(rng, h, w, top, bottom, left, right, border=16, jitter=3):
    img = np.full((h, w), border, np.uint8)
    img += rng.integers(0, jitter + 1, (h, w)).astype(np.uint8)
    inner = rng.integers(60, 256, (h - top - bottom, w - left - right), dtype=np.uint8)
    img[top:h - bottom, left:w - right] = inner
    img[0, 0] = border
    return img


def test_autocrop_rules(po):
    rng = np.random.default_rng(1)
    # symmetric horizontal letterbox -> cropped
    img = letterboxed(rng, 300, 400, 40, 40, 0, 0)
    assert po.autocrop(img).tolist() == [0, 40, 400, 260]
    # pillarbox
    img = letterboxed(rng, 300, 400, 0, 0, 50, 50)
    assert po.autocrop(img).tolist() == [50, 0, 350, 300]
    # off-centre bars beyond 5 %: centred on the lesser margin (cvutil.cpp:1377-1393)
    img = letterboxed(rng, 300, 400, 30, 60, 0, 0)
    assert po.autocrop(img).tolist() == [0, 30, 400, 270]
    # no border at all -> untouched
    img = rng.integers(0, 256, (200, 200), dtype=np.uint8)
    assert po.autocrop(img).tolist() == [0, 0, 200, 200]
    # a crop that would remove more than 35 % is refused (:1398-1399)
    img = letterboxed(rng, 300, 400, 70, 70, 0, 0)
    assert po.autocrop(img).tolist() == [0, 0, 400, 300]
    # threshold: a border that differs from the corner colour by more than `range` is content
    img = letterboxed(rng, 300, 400, 40, 40, 0, 0, border=16, jitter=3)
    img[10:30, :] = 60
    r = po.autocrop(img, 20)
    assert r.tolist() == [0, 40, 400, 260]  # the scan stops at the first bar row from the centre
    assert po.autocrop(img, 0).tolist() == [0, 0, 400, 300]


def test_gray_weights(po):
    bgr = np.zeros((1, 3, 3), np.uint8)
    bgr[0, 0] = [255, 0, 0]
    bgr[0, 1] = [0, 255, 0]
    bgr[0, 2] = [0, 0, 255]
    g = po.bgr2gray(bgr)[0]
    assert g.tolist() == [(255 * 1868 + 8192) >> 14, (255 * 9617 + 8192) >> 14, (255 * 4899 + 8192) >> 14]
    assert g.tolist() == [29, 150, 76]
    white = np.full((2, 2, 4), 255, np.uint8)
    assert (po.bgr2gray(white) == 255).all()


@pytest.mark.gpu
def test_gpu_process_images_vs_oracle(gpu, po):
    from cbird_amd.hashing import process_images

    rng = np.random.default_rng(5)
    h, w = 300, 400
    gray = np.stack([letterboxed(rng, h, w, 40, 40, 0, 0), letterboxed(rng, h, w, 0, 0, 50, 50),
                     rng.integers(0, 256, (h, w), dtype=np.uint8), letterboxed(rng, h, w, 30, 60, 0, 0),
                     letterboxed(rng, h, w, 70, 70, 0, 0), letterboxed(rng, h, w, 20, 20, 30, 30)])
    got, rects = process_images(gray, 20)
    for i in range(len(gray)):
        wh, wr = po.process_image(gray[i], 20)
        assert rects[i].tolist() == wr.tolist(), i
        assert int(got[i]) == wh, i
    got2, rects2 = process_images(gray, None)
    assert (rects2 == np.array([0, 0, w, h])).all()
    for i in range(len(gray)):
        assert int(got2[i]) == po.process_image(gray[i], None)[0]
    # colour input: BGR and BGRA
    for ch in (3, 4):
        col = rng.integers(0, 256, (4, 256, 256, ch), dtype=np.uint8)
        col[1, :30] = 10
        col[1, -30:] = 10
        got, rects = process_images(col, 20)
        for i in range(4):
            wh, wr = po.process_image(col[i], 20)
            assert rects[i].tolist() == wr.tolist() and int(got[i]) == wh, (ch, i)
