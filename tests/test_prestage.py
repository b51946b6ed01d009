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


# ---- sizeLongestSide (src/cvutil.cpp:1932-1950): the resize in front of ORB detection -------------------------

def test_longest_side_dims_and_lanczos_table(orc):
    """float aspect + truncation (:1934-1942); Lanczos-4 weights: unit DC gain within the fixed-point rounding,
    symmetric pairs for mirrored phases, identity at integer positions"""
    assert orc.longest_side_dims(4000, 3000, 400) == (400, 300)
    assert orc.longest_side_dims(3000, 4000, 400) == (300, 400)
    assert orc.longest_side_dims(300, 500, 400) == (240, 400)
    assert orc.longest_side_dims(500, 500, 400) == (400, 400)   # w > h is false: h = size, w = int(1.0 * 400)
    assert orc.longest_side_dims(1000, 3, 400) == (400, 1)
    assert orc.longest_side_dims(1000, 1, 400)[1] == 0          # the reference throws std::invalid_argument here
    for ssize, dsize in ((1000, 400), (4000, 300), (123, 400), (400, 400), (37, 400)):
        ofs, c = orc.lanczos4_tab(ssize, dsize)
        assert np.abs(c.astype(int).sum(1) - 2048).max() <= 3
        assert (np.diff(ofs) >= 0).all() and ofs[0] >= -1 and ofs[-1] <= ssize - 1
    ofs, c = orc.lanczos4_tab(400, 400)
    assert (ofs == np.arange(400)).all() and (c[:, 3] == 2048).all() and (np.delete(c, 3, 1) == 0).all()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    assert (orc.resize_lanczos4(img, 80, 60) == img).all()
    # against a float evaluation of the same separable filter (weights from the table, no fixed-point rounding)
    img = rng.integers(0, 256, (300, 500), dtype=np.uint8)
    out = orc.size_longest_side(img, 400).astype(np.float64)
    xo, xc = orc.lanczos4_tab(500, 400)
    yo, yc = orc.lanczos4_tab(300, 240)
    wx = np.zeros((400, 500))
    for d in range(400):
        for j in range(8):
            wx[d, min(max(xo[d] - 3 + j, 0), 499)] += xc[d, j] / 2048
    wy = np.zeros((240, 300))
    for d in range(240):
        for j in range(8):
            wy[d, min(max(yo[d] - 3 + j, 0), 299)] += yc[d, j] / 2048
    ref = np.clip(wy @ img.astype(np.float64) @ wx.T, 0, 255)
    assert np.abs(out - ref).max() <= 0.5 + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,size", [(500, 300, 400), (300, 500, 400), (1920, 1080, 400), (123, 77, 400),
                                      (400, 400, 400), (4000, 3000, 400), (37, 1000, 256), (640, 480, 64)])
def test_size_longest_side_matches_oracle(gpu, orc, w, h, size):
    from cbird_amd.hashing import size_longest_side

    rng = np.random.default_rng(w * 7 + h)
    n = 2 if w * h > 4_000_000 else 3
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = np.stack([np.clip(128 + 90 * np.sin(xx / (9.0 + i)) * np.cos(yy / 13.0) + rng.normal(0, 25, (h, w)), 0, 255)
                     .astype(np.uint8) for i in range(n)])
    got = size_longest_side(imgs, size)
    for i in range(n):
        want = orc.size_longest_side(imgs[i], size)
        assert got[i].shape == want.shape
        assert (got[i] == want).all(), (i, np.abs(got[i].astype(int) - want.astype(int)).max())
    # strided input view (row padding)
    pad = np.zeros((n, h, w + 5), np.uint8)
    pad[:, :, :w] = imgs
    assert (size_longest_side(pad[:, :, :w], size) == got).all()


@pytest.mark.gpu
def test_size_longest_side_errors(gpu):
    from cbird_amd import _lib
    from cbird_amd.hashing import size_longest_side

    with pytest.raises(gpu.CbhError) as e:
        size_longest_side(np.zeros((1, 1, 1000), np.uint8), 400)  # computed height 0
    assert e.value.code == _lib.CBH_E_INVAL
    assert size_longest_side(np.zeros((0, 30, 40), np.uint8), 400).shape == (0, 300, 400)


@pytest.mark.gpu
def test_autocropped_hash_uses_the_parent_border(gpu, po, orc):
    """autocrop() narrows cvGray to a colRange/rowRange VIEW (cvutil.cpp:1397-1401) and cv::blur on a view takes its
    border pixels from the parent image: the hash of a cropped image is the hash of the view, not of an isolated
    copy of the kept region.  Wide images (two column workgroups of the fused kernel) and small ones."""
    from cbird_amd import _lib
    from cbird_amd.hashing import process_images

    L = _lib.lib()
    rng = np.random.default_rng(6)
    differs = cropped = 0
    cases = ((300, 400, 40, 40, 0, 0), (700, 2600, 90, 90, 0, 0), (1200, 900, 0, 0, 100, 120), (64, 90, 8, 8, 0, 0),
             (480, 640, 60, 60, 0, 0), (1080, 1920, 140, 140, 0, 0), (450, 601, 50, 37, 0, 0), (300, 400, 0, 75, 0, 0),
             (120, 160, 14, 14, 0, 0), (400, 533, 33, 0, 0, 0), (300, 400, 30, 30, 40, 0),
             # pillarbox / window: the kept region's vertical edges lie inside the parent (interior edge lanes), at
             # aligned and odd offsets, any width mod 8; margins under 4 pixels or too small for the last lane's overhang
             # stay on the band kernels
             (480, 640, 0, 0, 80, 80), (360, 641, 20, 20, 33, 47), (400, 600, 0, 0, 6, 6), (300, 500, 25, 25, 64, 9),
             (720, 1280, 0, 0, 160, 160), (240, 427, 12, 12, 51, 50), (300, 400, 0, 0, 3, 40), (300, 400, 0, 0, 40, 3),
             # views at integer ratios whose cells are an even number of dwords (the padded LDS rows)
             (896, 1024, 64, 64, 0, 0), (512, 672, 0, 0, 80, 80), (616, 868, 20, 20, 50, 50), (320, 2048, 32, 32, 0, 0))
    # as shipped (small batches: the band kernels), then with the strip kernels forced -- a view that spans the parent's
    # width (letterbox) takes the register-streaming kernel with the parent's rows above and below it, split and fused;
    # any other view must still come out right (it stays on the band kernels)
    try:
        # (band: views whose vertical edges are the parent's or lie well inside it take k_band_area, round 5; 0 = the
        # kernels that took them before, still the path of the other views)
        for (stream, fuse, band) in ((1, 1, 1), (1, 1, 0), (3, 0, 1), (8, 2, 0)):
            L.cbh_set_tuning(b"hash_band_area", band)
            L.cbh_set_tuning(b"hash_stream", stream)
            L.cbh_set_tuning(b"hash_fuse", fuse)
            for (h, w, t, b, le, r) in cases:
                gray = np.stack([letterboxed(rng, h, w, t, b, le, r) for _ in range(2)])
                got, rects = process_images(gray, 20)
                for i in range(2):
                    wh, wr = po.process_image(gray[i], 20)
                    assert rects[i].tolist() == wr.tolist() and int(got[i]) == wh, (h, w, i, stream, fuse)
                    x0, y0, x1, y1 = wr.tolist()
                    cropped += (x0, y0, x1, y1) != (0, 0, w, h)  # (one-sided bars are not always cropped)
                    differs += int(orc.dcthash64(gray[i][y0:y1, x0:x1]) != wh)
    finally:
        L.cbh_set_tuning(b"hash_stream", 1)
        L.cbh_set_tuning(b"hash_fuse", 1)
        L.cbh_set_tuning(b"hash_band_area", 1)
    assert differs >= 3 and cropped >= 3 * 2 * 14


@pytest.mark.gpu
def test_process_images_ex_also_returns_the_orb_input(gpu, po, orc):
    """cbh_process_images_ex: hash + kept region as cbh_process_images, plus sizeLongestSide(cvGray, size) of the
    kept region (a view: the resize does not look outside it) from the same upload"""
    from cbird_amd.hashing import process_images, process_images_ex

    rng = np.random.default_rng(11)
    h, w = 300, 500
    gray = np.stack([letterboxed(rng, h, w, 40, 40, 0, 0), letterboxed(rng, h, w, 40, 40, 0, 0),
                     rng.integers(0, 256, (h, w), dtype=np.uint8), letterboxed(rng, h, w, 0, 0, 60, 60)])
    for size in (400, 128):
        got, rects, small = process_images_ex(gray, 20, size)
        ref_h, ref_r = process_images(gray, 20)
        assert (got == ref_h).all() and (rects == ref_r).all()
        for i in range(len(gray)):
            x0, y0, x1, y1 = rects[i].tolist()
            want = orc.size_longest_side(gray[i][y0:y1, x0:x1], size)
            assert small[i].shape == want.shape and (small[i] == want).all(), (size, i)
    col = rng.integers(0, 256, (3, 200, 260, 3), dtype=np.uint8)
    col[1, :25] = 7
    col[1, -25:] = 7
    got, rects, small = process_images_ex(col, 20, 100)
    for i in range(3):
        g = po.bgr2gray(col[i])
        x0, y0, x1, y1 = rects[i].tolist()
        assert (small[i] == orc.size_longest_side(g[y0:y1, x0:x1], 100)).all(), i
        assert int(got[i]) == po.process_image(col[i], 20)[0]
    # no autocrop, no images
    _, r2, s2 = process_images_ex(gray[:1], None, 64)
    assert r2.tolist() == [[0, 0, w, h]] and s2[0].shape == (38, 64)
    assert process_images_ex(np.zeros((0, 40, 40), np.uint8), 20, 64)[2] == []


@pytest.mark.gpu
def test_autocrop_dev_equals_oracle_over_geometries_and_strides(gpu, po):
    """cbh_autocrop_dev on its own (the rectangles, before any hashing): widths around the 256-byte segment and the
    4-byte load boundaries, heights around the 8-row steps, odd row strides (unaligned rows), bars of every kind,
    content inside the bars (subtitles, a logo in a side bar), images that are all border or all content"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(99)
    widths = [1, 2, 3, 4, 5, 7, 63, 64, 65, 255, 256, 257, 300, 511, 513, 777]
    heights = [1, 2, 3, 7, 8, 9, 15, 17, 64, 65, 100, 203]
    cases = 0
    for w in widths:
        for h in heights:
            n = 6
            stride = w + int(rng.integers(0, 6))
            buf = np.zeros((n, h, stride), np.uint8)
            for i in range(n):
                border = int(rng.integers(0, 256))
                img = np.clip(border + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
                kind = i % 6
                t = b = l = r = 0
                if kind in (0, 3):
                    t, b = int(rng.integers(0, h // 3 + 1)), int(rng.integers(0, h // 3 + 1))
                if kind in (1, 3):
                    l, r = int(rng.integers(0, w // 3 + 1)), int(rng.integers(0, w // 3 + 1))
                if kind == 2:  # symmetric bars (the balanced case that actually crops)
                    t = b = int(rng.integers(0, h // 5 + 1))
                if kind != 4 and h - t - b > 0 and w - l - r > 0:  # kind 4: nothing but border
                    inner = rng.integers(0, 256, (h - t - b, w - l - r))
                    far = np.abs(inner - border) <= 20  # push most of the content away from the border colour
                    inner = np.where(far & (rng.random(inner.shape) < 0.9), (border + 128) % 256, inner)
                    img[t:h - b, l:w - r] = inner.astype(np.uint8)
                if kind == 5 and h > 4 and w > 8:  # a subtitle inside the bottom bar
                    img[h - 2, w // 3: w // 2] = (border + 100) % 256
                if kind == 1 and l > 2 and i % 12 == 1 and w % 2:  # a logo inside the left bar: that bar is walked after all
                    img[h // 2, l // 2] = (border + 100) % 256
                buf[i, :, :w] = img
            d = torch.from_numpy(buf).to(dev)
            rects = torch.full((n, 4), -7, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            _lib.check(L.cbh_autocrop_dev(d.data_ptr(), n, w, h, stride, h * stride, 20, rects.data_ptr(), 0, None),
                       "autocrop_dev")
            torch.cuda.synchronize()
            got = rects.cpu().numpy()
            for i in range(n):
                assert got[i].tolist() == po.autocrop(np.ascontiguousarray(buf[i, :, :w])).tolist(), (w, h, stride, i)
                cases += 1
    assert cases == len(widths) * len(heights) * 6
