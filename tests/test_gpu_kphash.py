"""Media::makeKeyPointHashes (src/media.cpp:874-923) and images with a side < 32 on the GPU (k_rect_hashes),
bit-exact against the oracle: hashes AND the images as the in-place blurs leave them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keypoints(rng, w, h, k):
    """ORB-like keypoints: sizes 31 * 1.2^level, positions anywhere (some fail the inside-image rule)"""
    lv = rng.integers(0, 12, k)
    size = (31.0 * 1.2 ** lv).astype(np.float32)
    x = rng.uniform(-5, w, k).astype(np.float32)
    y = rng.uniform(-5, h, k).astype(np.float32)
    return np.stack([x, y, size], 1)


def test_keypoint_rects_rule_matches_oracle(gpu, orc):
    from cbird_amd.hashing import keypoint_rects

    rng = np.random.default_rng(1)
    for (w, h) in ((400, 300), (200, 150), (64, 64), (40, 500)):
        kp = _keypoints(rng, w, h, 300)
        kp[:20, 2] = rng.uniform(29, 33, 20).astype(np.float32)  # around the size >= 31 cut
        assert (keypoint_rects(w, h, kp) == orc.keypoint_rects(w, h, kp)).all()
    assert len(keypoint_rects(400, 300, np.zeros((0, 3), np.float32))) == 0


def test_keypoint_hashes_match_oracle(gpu, orc):
    """a batch of differently sized images, up to 400 keypoints each (scanner.h:70), overlapping squares of every
    blur class (31 px: none and the bilinear 31 -> 32 resize; <= 64: 3x3; <= 128: 5x5; larger: 7x7; 32, 64 and
    96-pixel squares for the copy / integer-block paths)"""
    from cbird_amd.hashing import make_keypoint_hashes

    rng = np.random.default_rng(2)
    shapes = [(300, 400), (400, 267), (150, 200), (64, 64), (33, 500), (400, 400), (120, 90)]
    images, kps = [], []
    for i, (h, w) in enumerate(shapes):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = 128 + 60 * np.sin(xx / 17.0 + i) * np.cos(yy / 23.0) + rng.normal(0, 12, (h, w))
        images.append(np.clip(smooth, 0, 255).astype(np.uint8))
        kp = _keypoints(rng, w, h, 400 if i < 3 else 60)
        kp[:6, 2] = [32.0, 64.0, 96.0, 31.0, 31.5, 128.0]
        kp[:6, :2] = rng.uniform(1, 20, (6, 2)).astype(np.float32)
        kps.append(kp)
    kps[3] = kps[3][:0]  # an image without keypoints
    got, after = make_keypoint_hashes(images, kps, return_images=True)
    n_hashes = 0
    for i in range(len(images)):
        want, want_img = orc.keypoint_hashes(images[i], kps[i])
        assert len(got[i]) == len(want), i
        assert (got[i] == want).all(), (i, np.nonzero(got[i] != want)[0][:5])
        assert (after[i] == want_img).all(), i
        n_hashes += len(want)
    assert n_hashes > 300 and len(got[3]) == 0
    # without return_images the same hashes
    again = make_keypoint_hashes(images, kps)
    assert all((a == b).all() for a, b in zip(again, got))


def test_keypoint_hashes_order_dependence(gpu, orc):
    """two overlapping squares: swapping them changes the result the same way it does in the oracle"""
    from cbird_amd.hashing import make_keypoint_hashes

    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (200, 200), dtype=np.uint8)
    kp = np.array([[10.2, 12.7, 53.6], [30.0, 33.0, 64.3], [20.0, 20.0, 134.0]], np.float32)
    for order in ([0, 1, 2], [2, 1, 0], [1, 2, 0]):
        got = make_keypoint_hashes([img], [kp[order]])[0]
        want, _ = orc.keypoint_hashes(img, kp[order])
        assert (got == want).all(), order
    a = make_keypoint_hashes([img], [kp])[0]
    b = make_keypoint_hashes([img], [kp[[2, 1, 0]]])[0]
    assert sorted(a.tolist()) != sorted(b.tolist())


def test_keypoint_hashes_many_images(gpu, orc):
    """more images than workgroups in one launch (grid-stride over images), sampled against the oracle"""
    from cbird_amd.hashing import make_keypoint_hashes

    rng = np.random.default_rng(4)
    n = 2500
    images = [rng.integers(0, 256, (int(rng.integers(40, 90)), int(rng.integers(40, 90))), dtype=np.uint8)
              for _ in range(n)]
    kps = [_keypoints(rng, im.shape[1], im.shape[0], 6) for im in images]
    for k in kps:
        k[:, 2] = rng.choice([31.0, 32.0, 37.2], len(k)).astype(np.float32)
        k[:, :2] = rng.uniform(0.5, 8, (len(k), 2)).astype(np.float32)
    got = make_keypoint_hashes(images, kps)
    assert sum(len(g) for g in got) > n
    for i in list(range(0, n, 97)) + [n - 1]:
        want, _ = orc.keypoint_hashes(images[i], kps[i])
        assert (got[i] == want).all(), i


def test_small_images_bilinear_path(gpu, orc):
    """a side < 32: cv::resize(INTER_AREA) enlarges and runs its bilinear emulation; hashes and tiles"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(5)
    for (w, h) in ((31, 31), (16, 16), (1, 1), (5, 31), (31, 5), (20, 100), (100, 20), (31, 32), (32, 31), (8, 4000),
                   (24, 24), (2, 2)):
        n = 5
        imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        want = orc.dcthash64_batch(imgs)
        assert (gpu.dct_hash64_batch(imgs) == want).all(), (w, h)
        d = torch.from_numpy(imgs).cuda()
        out = torch.zeros(n, dtype=torch.int64, device="cuda")
        tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
        _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None),
                   "tiles")
        t = tiles.cpu().numpy()
        for i in range(n):
            assert (t[i] == orc.tile32(imgs[i])).all(), (w, h, i)
        assert (out.cpu().numpy().view(np.uint64) == want).all()


def test_keypoint_hashes_fuzz(gpu, orc):
    """arbitrary (non-ORB) keypoint sizes -- every side length from 31 up to the image, i.e. all resize modes
    (bilinear 31, copy 32, integer blocks 64/96/128, weighted tables), the LDS path (<= 134) and the global-memory
    path (larger) -- on images of random size, with heavily overlapping squares"""
    from cbird_amd.hashing import make_keypoint_hashes

    rng = np.random.default_rng(77)
    images, kps = [], []
    for i in range(60):
        h, w = int(rng.integers(36, 330)), int(rng.integers(36, 330))
        images.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
        k = int(rng.integers(0, 25))
        m = min(w, h) - 3
        size = rng.uniform(29, max(31.5, m), k).astype(np.float32)
        size[rng.random(k) < 0.3] = np.float32(rng.choice([31, 32, 64, 96, 128, 134, 135]))
        x = rng.uniform(0.1, np.maximum(0.2, w - 2.5 - size)).astype(np.float32)
        y = rng.uniform(0.1, np.maximum(0.2, h - 2.5 - size)).astype(np.float32)
        kps.append(np.stack([x, y, size], 1).reshape(-1, 3))
    got, after = make_keypoint_hashes(images, kps, return_images=True)
    total = 0
    for i in range(len(images)):
        want, want_img = orc.keypoint_hashes(images[i], kps[i])
        assert len(got[i]) == len(want) and (got[i] == want).all(), (i, images[i].shape, kps[i][:, 2])
        assert (after[i] == want_img).all(), i
        total += len(want)
    assert total > 300


def test_hash_entry_points_from_concurrent_threads(gpu, orc):
    """Scanner workers call the hash functions from a QThreadPool: the stateless entry points (and the per-geometry
    table caches behind them) must tolerate concurrent callers"""
    import threading

    from cbird_amd.hashing import make_keypoint_hashes, size_longest_side

    rng = np.random.default_rng(9)
    geos = [(300, 200), (640, 480), (257, 129), (31, 31), (1000, 40), (256, 256), (96, 96), (2050, 64)]
    imgs = {g: rng.integers(0, 256, (4, g[1], g[0]), dtype=np.uint8) for g in geos}
    want = {g: orc.dcthash64_batch(imgs[g]) for g in geos}
    kimg = rng.integers(0, 256, (200, 260), dtype=np.uint8)
    kp = _keypoints(rng, 260, 200, 80)
    kwant, _ = orc.keypoint_hashes(kimg, kp)
    rwant = orc.size_longest_side(kimg, 100)
    errs = []

    def worker(seed):
        try:
            order = np.random.default_rng(seed).permutation(len(geos))
            for _ in range(3):
                for j in order:
                    g = geos[j]
                    if not (gpu.dct_hash64_batch(imgs[g]) == want[g]).all():
                        errs.append(("hash", g))
                if not (make_keypoint_hashes([kimg], [kp])[0] == kwant).all():
                    errs.append("kp")
                if not (size_longest_side(kimg[None], 100)[0] == rwant).all():
                    errs.append("resize")
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(s,)) for s in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:5]


def test_entry_point_argument_errors(gpu):
    """error conventions of the C-ABI (negative codes, never abort): empty batches are fine, inconsistent descriptors
    are CBH_E_INVAL"""
    import ctypes as C

    from cbird_amd import _lib

    L = _lib.lib()
    img = np.zeros((50, 60), np.uint8)
    off = np.zeros(1, np.uint64)
    w = np.array([60], np.uint32)
    h = np.array([50], np.uint32)
    kp = np.array([[5, 5, 31]], np.float32)
    kpf = np.array([0, 1], np.uint32)
    out = np.zeros(4, np.uint64)
    of = np.zeros(2, np.uint32)

    def call(**kw):
        a = dict(imgs=img.ctypes.data, nbytes=img.size, n=1, off=off.ctypes.data, w=w.ctypes.data, h=h.ctypes.data,
                 st=w.ctypes.data, kp=kp.ctypes.data, kpf=kpf.ctypes.data, out=out.ctypes.data, of=of.ctypes.data)
        a.update(kw)
        return L.cbh_keypoint_hashes(a["imgs"], a["nbytes"], a["n"], a["off"], a["w"], a["h"], a["st"], a["kp"], a["kpf"],
                                     a["out"], a["of"], None, 0)

    assert call() == 0 and of.tolist() == [0, 1]
    assert call(n=0) == 0 and of[0] == 0
    assert call(nbytes=img.size - 1) == _lib.CBH_E_INVAL           # image reaches past the buffer
    bad_first = np.array([1, 0], np.uint32)
    assert call(kpf=bad_first.ctypes.data) == _lib.CBH_E_INVAL     # keypoint ranges must not decrease
    narrow = np.array([59], np.uint32)
    assert call(st=narrow.ctypes.data) == _lib.CBH_E_INVAL         # row stride below the width
    assert call(of=None) == _lib.CBH_E_INVAL
    # rectangles outside their image
    rects = np.array([[30, 30, 40, 10]], np.int32)
    rf = np.array([0, 1], np.uint32)
    rc = L.cbh_dcthash_rects(img.ctypes.data, img.size, 1, off.ctypes.data, w.ctypes.data, h.ctypes.data, w.ctypes.data,
                             rects.ctypes.data, rf.ctypes.data, 0, out.ctypes.data, None, 0)
    assert rc == _lib.CBH_E_INVAL
    rects[0] = [10, 10, 40, 35]
    rc = L.cbh_dcthash_rects(img.ctypes.data, img.size, 1, off.ctypes.data, w.ctypes.data, h.ctypes.data, w.ctypes.data,
                             rects.ctypes.data, rf.ctypes.data, 0, out.ctypes.data, None, 0)
    assert rc == 0 and out[0] == 1  # a constant rectangle hashes to 1
    # resize to nothing / radius match with a tiny output buffer
    ow, oh = C.c_int(0), C.c_int(0)
    assert L.cbh_size_longest_side(img.ctypes.data, 1, 60, 50, 60, 3000, 0, img.ctypes.data, C.byref(ow), C.byref(oh),
                                   0) == _lib.CBH_E_INVAL
    from cbird_amd.cvfeatures import CvFeaturesIndex

    rows = np.random.default_rng(0).integers(0, 256, (100, 32), dtype=np.uint8)
    ix = CvFeaturesIndex()
    ix.add([type("M", (), dict(id=1, keyPointDescriptors=rows, path=""))()])
    first = np.zeros(101, np.uint64)
    m = np.zeros((5, 3), np.int32)
    rc = L.cbh_idx256_radius_match(ix.handle, rows.ctypes.data, 100, 0, m.ctypes.data, 5, first.ctypes.data)
    assert rc == _lib.CBH_E_OVERFLOW and first[-1] == 100           # every row matches itself; cap was 5
