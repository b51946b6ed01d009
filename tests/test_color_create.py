"""SURVEY section 8 row a14 -- ColorDescriptor::create (/root/reference/src/cvutil.cpp:790-1099).  CPU: the oracle's
stages against independent statements; GPU: cbh_color_descriptors byte for byte against the oracle
(oracle/colordesc_oracle.c: parity unpinned versus OpenCV / the cbird binary, and why)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cc():
    from oracle import ColorCreateOracle

    return ColorCreateOracle()


def _photo(rng, w, h, ch=3, blocks=40):
    img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    for _ in range(blocks):
        x, y = int(rng.integers(0, max(1, w - 4))), int(rng.integers(0, max(1, h - 4)))
        img[y: y + int(rng.integers(3, max(4, h // 3))), x: x + int(rng.integers(3, max(4, w // 3)))] = \
            rng.integers(0, 256, ch)
    return img


# ---- oracle stages (CPU) ---------------------------------------------------------------------------------------------
def test_oracle_cbrt_and_luv(cc):
    xs = np.linspace(1e-4, 1.5, 4001)
    assert max(abs(cc.cbrt(x) - np.cbrt(x)) / np.cbrt(x) for x in xs) < 4e-7  # "error < 2^-24" + float rounding
    assert cc.cbrt(0.0) == 0.0 and cc.cbrt(-8.0) == -2.0 and cc.cbrt(27.0) == 3.0

    def luv64(b, g, r):
        lin = lambda c: c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4
        R, G, B = lin(r), lin(g), lin(b)
        X = 0.412453 * R + 0.357580 * G + 0.180423 * B
        Y = 0.212671 * R + 0.715160 * G + 0.072169 * B
        Z = 0.019334 * R + 0.119193 * G + 0.950227 * B
        L = 116 * np.cbrt(Y) - 16 if Y > 0.008856 else 903.3 * Y
        d = X + 15 * Y + 3 * Z
        if d < 1e-12:
            return L, 0.0, 0.0
        wn = 0.950456 + 15 + 3 * 1.088754
        return L, 13 * L * (4 * X / d - 4 * 0.950456 / wn), 13 * L * (9 * Y / d - 9 / wn)

    rng = np.random.default_rng(0)
    for _ in range(500):
        b, g, r = (rng.integers(0, 256, 3) / 255).tolist()
        got = cc.bgr2luv(np.float32(b), np.float32(g), np.float32(r))
        assert np.abs(got - np.array(luv64(b, g, r))).max() < 0.02   # the accuracy of the library's spline tables
    assert np.abs(cc.bgr2luv(1, 1, 1) - [100, 0, 0]).max() < 0.02 and cc.bgr2luv(0, 0, 0)[0] == 0


def test_oracle_mask_and_dims(cc):
    assert cc.resized_dims(200, 100) == (200, 100) and cc.resized_dims(256, 256) == (256, 256)
    assert cc.resized_dims(400, 300) == (256, 192) and cc.resized_dims(300, 400) == (192, 256)
    assert cc.resized_dims(1000, 10) == (256, 2)
    for cols, rows in ((256, 192), (192, 256), (100, 100), (40, 30), (255, 131)):
        m = cc.ellipse_mask(cols, rows)
        assert set(np.unique(m)) <= {0, 255}
        area = (m > 0).sum()
        ideal = np.pi * 0.45 * cols * 0.45 * rows
        assert ideal * 0.98 < area < ideal * 1.0 + 2.2 * (cols + rows)    # the polygon + its drawn outline
        yy, xx = np.indices((rows, cols))
        inside = ((xx - cols * 0.5) / (0.45 * cols - 1.5)) ** 2 + ((yy - rows * 0.5) / (0.45 * rows - 1.5)) ** 2 < 1
        outside = ((xx - cols * 0.5) / (0.45 * cols + 1.5)) ** 2 + ((yy - rows * 0.5) / (0.45 * rows + 1.5)) ** 2 > 1
        assert (m[inside] == 255).all() and (m[outside] == 0).all()
        assert m[0].sum() == 0 and m[:, 0].sum() == 0   # corners and sides are dropped: pure black is filtered later
        for r in range(rows):                            # convex: every row is one run
            nz = np.flatnonzero(m[r])
            assert len(nz) == 0 or nz[-1] - nz[0] + 1 == len(nz)


def test_oracle_kmeans_invariants(cc):
    rng = np.random.default_rng(3)
    blobs = rng.normal(0, 3, (40, 400, 3)) + rng.uniform(-80, 80, (40, 1, 3)) + [50, 0, 0]
    data = blobs.reshape(-1, 3).astype(np.float32)
    rng.shuffle(data)
    labels, centers, it = cc.kmeans(data)
    assert 2 <= it <= 100 and labels.min() == 0 and labels.max() == 31 and len(np.unique(labels)) == 32
    # the centres returned are the means (float sums in sample order) of the label sets returned
    for k in range(32):
        sel = data[labels == k]
        acc = np.zeros(3, np.float32)
        for row in sel:
            acc += row
        assert (acc * np.float32(1.0 / np.float32(len(sel))) == centers[k]).all()
    # deterministic: a fresh generator per call
    l2, c2, it2 = cc.kmeans(data)
    assert (l2 == labels).all() and (c2 == centers).all() and it2 == it
    # fewer distinct colours than clusters: duplicates are seeded, empty clusters get split off, nothing breaks
    few = np.repeat(rng.uniform(0, 100, (5, 3)).astype(np.float32), 50, axis=0)
    l3, c3, it3 = cc.kmeans(few)
    assert len(l3) == 250 and np.isfinite(c3).all() and len(np.unique(l3)) == 32


def test_oracle_descriptor_properties(cc):
    rng = np.random.default_rng(4)
    img = _photo(rng, 400, 300)
    desc, (cols, rows, n, it) = cc.create(img)
    assert (cols, rows) == (256, 192) and 20000 < n < cols * rows and it >= 2
    col = desc[:256].view(np.uint16).reshape(32, 4)
    nc = int(desc[256]) + 1              # numColors holds the index of the last colour
    assert 1 <= nc <= 32 and col[0, 3] == 65535 and (np.diff(col[:nc, 3].astype(int)) <= 0).all()
    assert (col[nc:] == 0).all()
    # a dark image has no sample with l > 4: the reference leaves the descriptor alone
    none, st = cc.create(np.full((64, 64, 3), 3, np.uint8))
    assert none is None and st[2] == 0
    # BGRA: the alpha byte is dropped
    bgra = np.dstack([img, rng.integers(0, 256, img.shape[:2], dtype=np.uint8)])
    d4, _ = cc.create(bgra)
    assert (d4 == desc).all()


# ---- GPU: byte for byte against the oracle ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_mask_equals_oracle(gpu, cc):
    from cbird_amd import _lib

    L = _lib.lib()
    for cols, rows in ((256, 192), (192, 256), (100, 100), (40, 30), (255, 131), (7, 5), (1, 1), (256, 2)):
        m = np.zeros((rows, cols), np.uint8)
        assert L.cbh_color_ellipse_mask(cols, rows, m.ctypes.data) == 0
        assert (m == cc.ellipse_mask(cols, rows)).all(), (cols, rows)
    a, b = C.c_int(0), C.c_int(0)
    for w, h in ((400, 300), (300, 400), (1000, 10), (256, 256), (257, 256), (31, 900)):
        L.cbh_color_descriptor_dims(w, h, C.byref(a), C.byref(b))
        assert (a.value, b.value) == cc.resized_dims(w, h)


@pytest.mark.gpu
def test_gpu_color_descriptors_equal_oracle(gpu, cc):
    from cbird_amd.colordesc import create_descriptors

    rng = np.random.default_rng(8)
    imgs = [_photo(rng, w, h) for (w, h) in ((400, 300), (300, 400), (256, 256), (640, 480), (200, 120), (97, 131),
                                             (64, 64), (33, 47), (1024, 200))]
    flat = np.zeros((120, 160, 3), np.uint8)
    flat[:] = (40, 90, 200)                                    # one colour: 31 duplicate seeds, empty clusters
    imgs.append(flat)
    two = np.zeros((100, 100, 3), np.uint8)
    two[:, :50] = (250, 30, 30)
    two[:, 50:] = (20, 240, 20)
    imgs.append(two)
    grad = np.zeros((180, 240, 3), np.uint8)
    grad[..., 0] = np.linspace(0, 255, 240).astype(np.uint8)[None, :]
    grad[..., 2] = np.linspace(255, 0, 180).astype(np.uint8)[:, None]
    imgs.append(grad)
    imgs.append(np.full((64, 64, 3), 3, np.uint8))             # too dark: not enough colours
    imgs.append(np.full((8, 8, 3), 200, np.uint8))             # too small: fewer than 32 samples
    descs, ok = create_descriptors(imgs)
    for i, img in enumerate(imgs):
        want, st = cc.create(img)
        assert bool(ok[i]) == (want is not None), (i, st)
        got = np.frombuffer(descs[i].tobytes(), np.uint8)
        if want is None:
            assert not got.any()
        else:
            assert (got == want).all(), (i, img.shape, st)
    # BGRA input, and more images than one wave of the clustering kernel holds (lane-per-image, 64 per wave)
    many = [_photo(rng, int(rng.integers(40, 200)), int(rng.integers(40, 200)), ch=4, blocks=10) for _ in range(70)]
    d4, ok4 = create_descriptors(many)
    for i in (0, 1, 31, 63, 64, 69):
        want, _ = cc.create(many[i])
        assert ok4[i] and (np.frombuffer(d4[i].tobytes(), np.uint8) == want).all(), i


@pytest.mark.gpu
def test_gpu_color_descriptor_arguments(gpu):
    from cbird_amd import _lib
    from cbird_amd.colordesc import create_descriptors

    d, ok = create_descriptors([])
    assert len(d) == 0 and len(ok) == 0
    with pytest.raises(ValueError):
        create_descriptors([np.zeros((10, 10), np.uint8)])
    with pytest.raises(ValueError):
        create_descriptors([np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10, 4), np.uint8)])
    L = _lib.lib()
    assert L.cbh_color_ellipse_mask(0, 5, None) == _lib.CBH_E_INVAL
