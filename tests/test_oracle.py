"""CPU suite: pins the oracle (oracle/cbird_oracle.c) before anything is compared against it.

 * search semantics against golden vectors produced by the real reference VP-tree
   (tests/golden/gen_golden.py) and, when oracle/_ref is present, against the tree itself;
 * hash pipeline: closed-form known-answer tests derived from the definition of the transform
   (single DCT basis image -> exactly one hash bit), stage-level cross-checks, and the structural
   facts the reference source states (zig-zag table, bit 0, kernel-size rule).
"""
import numpy as np
import pytest

from conftest import golden_cases, load_golden

# (row, col) of the coefficient behind hash bit i, as listed in SURVEY.md 8(a1) from the
# reference's zigZag table (src/cvutil.cpp:491-495, positions 6..69)
BIT_RC_HEAD = {0: (3, 0), 1: (2, 1), 2: (1, 2), 3: (0, 3), 4: (0, 4), 5: (1, 3), 6: (2, 2),
               7: (3, 1), 8: (4, 0), 9: (5, 0), 63: (7, 5)}
# the reference's table itself, for the structural check only (81 small integers are data)
ZIGZAG_REF = [0, 9, 1, 2, 10, 18, 27, 19, 11, 3, 4, 12, 20, 28, 36, 45, 37, 29, 21, 13, 5, 6, 14, 22,
              30, 38, 46, 54, 63, 55, 47, 39, 31, 23, 15, 7, 8, 16, 24, 32, 40, 48, 56, 64, 72, 73, 65,
              57, 49, 41, 33, 25, 17, 26, 34, 42, 50, 58, 66, 74, 75, 67, 59, 51, 43, 35, 44, 52, 60,
              68, 76, 77, 69, 61, 53, 62, 70, 78, 79, 71, 80]


def test_hamm64(orc):
    assert orc.hamm64(0, 0) == 0
    assert orc.hamm64(0, 2**64 - 1) == 64
    assert orc.hamm64(0xF0F0, 0x0F0F) == 16
    rng = np.random.default_rng(0)
    for a, b in rng.integers(0, 2**63, (100, 2)).tolist():
        assert orc.hamm64(a, b) == bin(a ^ b).count("1")


@pytest.mark.parametrize("name", golden_cases())
def test_scan_matches_reference_golden(orc, name):
    g = load_golden(name)
    h, ids, q = g["hashes"], g["ids"], g["queries"]
    for dht in range(1, 9):
        offs, gid, gd = g[f"offs_{dht}"], g[f"ids_{dht}"], g[f"dist_{dht}"]
        for j, t in enumerate(q.tolist()):
            i, d = orc.find64(h, ids, t, dht)
            a, b = offs[j], offs[j + 1]
            assert i.tolist() == gid[a:b].tolist(), (name, dht, j)
            assert d.tolist() == gd[a:b].tolist(), (name, dht, j)


def test_find_batch_truncation(orc):
    g = load_golden("vptree_n4096.npz")
    h, ids, q = g["hashes"], g["ids"], g["queries"][:64]
    oi, od, cnt = orc.find64_batch(h, ids, q, 8, 3)
    for j, t in enumerate(q.tolist()):
        i, d = orc.find64(h, ids, t, 8)
        assert cnt[j] == len(i)
        m = min(3, len(i))
        assert oi[j, :m].tolist() == i[:m].tolist() and od[j, :m].tolist() == d[:m].tolist()
        assert (oi[j, m:] == 0).all()


def test_oracle_vs_real_vptree(orc):
    """Direct check against the compiled reference tree (skipped where oracle/_ref is absent)."""
    import oracle

    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    from cbird_amd import synth

    h, ids = synth.make_hashes(50000, seed=99, planted_frac=0.1)
    tree = oracle.RefTree(h, ids)
    rng = np.random.default_rng(5)
    for t in h[rng.choice(len(h), 300, replace=False)].tolist():
        for dht in (1, 2, 5, 8, 12):
            ri, rd = tree.search(t, dht)
            assert (np.diff(rd) >= 0).all()  # ascending by distance (vptree.h:50-69)
            oi, od = orc.find64(h, ids, t, dht)
            order = np.lexsort((ri, rd))
            assert ri[order].tolist() == oi.tolist() and rd[order].tolist() == od.tolist()
    tot, cnt = tree.search_many(h[:2000], 5, threads=4, want_counts=True)
    assert tot == orc.count64_pairs(h, ids, h[:2000], 5) == int(cnt.sum())


def test_null_needle_and_removed_slots(orc):
    h = np.array([0, 2, 6, 0], np.uint64)
    ids = np.array([0, 7, 8, 9], np.uint32)  # slot 0 removed; slot 3: hash never computed, id valid
    assert orc.find64(h, ids, 0, 65)[0].tolist() == []  # dcthashindex.cpp:196-200
    i, d = orc.find64(h, ids, 2, 3)
    # the hash-0 slot with a valid id IS returned (only id 0 is skipped, :215)
    assert list(zip(i.tolist(), d.tolist())) == [(7, 0), (8, 1), (9, 1)]


# ---- hash pipeline -------------------------------------------------------------------------

def test_zigzag_is_the_reference_table(orc):
    zz = orc.zigzag81().tolist()
    assert zz == ZIGZAG_REF
    for bit, (r, c) in BIT_RC_HEAD.items():
        assert zz[6 + bit] == r * 9 + c


def test_kernel_size_rule(orc):
    # cvutil.cpp:446-455 (by input AREA)
    assert orc.blur_ksize(32, 32) == 0
    assert orc.blur_ksize(64, 16) == 0
    assert orc.blur_ksize(64, 64) == 3
    assert orc.blur_ksize(128, 128) == 5
    assert orc.blur_ksize(128, 129) == 7
    assert orc.blur_ksize(256, 256) == 7


def test_dct_table_is_orthonormal(orc):
    c = orc.dct9_table().astype(np.float64)
    assert np.allclose(c @ c.T, np.eye(9), atol=1e-6)
    assert np.allclose(c[0], np.sqrt(1 / 32))


@pytest.mark.parametrize("k", [3, 5, 7])
def test_box_blur_separable_equals_direct(orc, k):
    rng = np.random.default_rng(k)
    for shape in [(64, 64), (96, 160), (33, 47), (8, 300)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        assert (orc.box_blur(img, k) == orc.box_blur(img, k, direct=True)).all()


def test_box_blur_against_numpy_definition(orc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    for k in (3, 5, 7):
        r = k // 2
        p = np.pad(img.astype(np.int64), r, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
        s = sum(p[dy:dy + 40, dx:dx + 56] for dy in range(k) for dx in range(k))
        want = np.floor(s / (k * k) + 0.5).astype(np.uint8)
        assert (orc.box_blur(img, k) == want).all()


def test_area_resize_rounding(orc):
    # 256 -> 32: 8x8 block mean, ties to even; 64 -> 32: (s+2)>>2
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (256, 256), dtype=np.uint8)
    blur = orc.box_blur(img, 7)
    s = blur.reshape(32, 8, 32, 8).astype(np.int64).sum(axis=(1, 3))
    want = np.rint(s / 64.0).astype(np.uint8)  # np.rint is half-to-even, s/64 exact
    assert (orc.tile32(img) == want).all()
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    blur = orc.box_blur(img, 3)
    s = blur.reshape(32, 2, 32, 2).astype(np.int64).sum(axis=(1, 3))
    assert (orc.tile32(img) == ((s + 2) >> 2).astype(np.uint8)).all()
    t = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    assert (orc.tile32(t) == t).all()  # 32x32: no blur, resize is a no-op


def _basis_image(u, v, n=32, amp=100.0):
    y = np.arange(n)[:, None]
    x = np.arange(n)[None, :]
    f = 128 + amp * np.cos(np.pi * (2 * y + 1) * u / (2 * n)) * np.cos(np.pi * (2 * x + 1) * v / (2 * n))
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def test_single_basis_function_sets_exactly_its_bit(orc):
    """Known answers from the definition: a 32x32 image that is one DCT basis function (u,v) has
    one dominant coefficient; with the mean threshold only that coefficient's bit is set.
    Bit 0's coefficient (3,0) is never encoded (loop starts at 1, cvutil.cpp:537) -> hash 0 -> 1."""
    zz = orc.zigzag81()
    for bit in range(64):
        u, v = divmod(int(zz[6 + bit]), 9)
        hv = orc.dcthash64(_basis_image(u, v))
        assert hv == ((1 << bit) if bit else 1), (bit, u, v, hex(hv))
        # negative amplitude: the coefficient is below the mean, every other one is above it
        hv = orc.dcthash64(_basis_image(u, v, amp=-100.0))
        full = (2**64 - 2)
        assert hv == (full & ~(1 << bit) if bit else full), (bit, hex(hv))


def test_low_frequency_basis_through_blur_and_resize(orc):
    zz = orc.zigzag81()
    for bit in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        u, v = divmod(int(zz[6 + bit]), 9)
        img = _basis_image(u, v, n=256)
        assert orc.dcthash64(img) == 1 << bit


def test_area_tables_cover_every_source_pixel_with_unit_weight(orc):
    """computeResizeAreaTab: per destination cell the weights sum to 1 and every source pixel is used"""
    import ctypes as C

    for ssize in (33, 47, 100, 257, 600, 1000):
        si = np.zeros(ssize + 70, np.int32)
        di = np.zeros(ssize + 70, np.int32)
        al = np.zeros(ssize + 70, np.float32)
        f = orc.L.orc_resize_area_tab
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        k = f(ssize, 32, si.ctypes.data, di.ctypes.data, al.ctypes.data)
        si, di, al = si[:k], di[:k], al[:k]
        assert (np.diff(di) >= 0).all() and set(di.tolist()) == set(range(32))
        for d in range(32):
            assert abs(al[di == d].sum() - 1.0) < 1e-5
        assert set(si.tolist()) == set(range(ssize))
        cover = np.zeros(ssize)
        np.add.at(cover, si, al * (ssize / 32))
        assert np.allclose(cover, 1.0, atol=2e-3)  # the 1e-3 cut-off of the reference drops slivers


def test_resize_scale_double_rounding(orc):
    """cv::resize computes scale = 1./((double)32/w): for w = 32*m with m in {49, 93, 98, ...} that is not m, so the
    integer block path (|scale - round(scale)| < DBL_EPSILON) is not taken; the weighted tables then still
    describe the same block mean"""
    import ctypes as C
    import sys

    f = orc.L.orc_resize_area_fast
    f.argtypes = [C.c_int, C.c_int]
    f.restype = C.c_int
    odd = [m for m in range(1, 257) if abs(1.0 / (32.0 / (32 * m)) - m) >= sys.float_info.epsilon]
    assert odd[:5] == [49, 93, 98, 99, 103]
    for m in range(1, 257):
        assert f(32 * m, 32 * m) == (0 if m in odd else 1), m
    assert f(256, 256) == 1 and f(256, 1568) == 0 and f(100, 64) == 0 and f(33, 33) == 0
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (1568, 1568), dtype=np.uint8)
    b = orc.box_blur(img, 7)
    mean = b.reshape(32, 49, 32, 49).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(orc.tile32(img).astype(np.float64) - mean).max() <= 0.5 + 1e-3


def test_general_area_resize_is_the_rounded_area_mean(orc):
    rng = np.random.default_rng(12)
    for h, w in ((100, 100), (33, 47), (400, 600), (255, 256)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        k = orc.blur_ksize(w, h)
        b = orc.box_blur(img, k) if k else img

        def wmat(n):
            m = np.zeros((32, n))
            sc = n / 32
            for d in range(32):
                a, e = d * sc, (d + 1) * sc
                for s in range(int(np.floor(a)), min(n, int(np.ceil(e)))):
                    m[d, s] = max(0.0, min(e, s + 1) - max(a, s)) / sc
            return m

        exact = wmat(h) @ b.astype(np.float64) @ wmat(w).T
        assert np.abs(orc.tile32(img).astype(np.float64) - exact).max() <= 0.5 + 1e-3


def test_hash_never_zero_and_bit0(orc):
    rng = np.random.default_rng(8)
    for _ in range(50):
        hv = orc.dcthash64(rng.integers(0, 256, (32, 32), dtype=np.uint8))
        assert hv != 0 and (hv & 1) == 0


def test_hash_is_scale_stable(orc):
    """Behavioural property the reference's index tests rely on (unit/testindexbase.cpp:112-146,
    data set 40x5-sizes): the same picture at another scale lands within the default threshold."""
    from cbird_amd import synth

    imgs = synth.make_images(6, w=512, h=512, seed=3, dup_frac=0)
    for im in imgs:
        big = orc.dcthash64(im)
        half = orc.dcthash64(im.reshape(256, 2, 256, 2).mean(axis=(1, 3)).round().astype(np.uint8))
        assert orc.hamm64(big, half) < 5


def test_small_images_take_the_bilinear_emulation(orc):
    """a side shorter than 32 enlarges: cv::resize(INTER_AREA) runs its 2-tap fixed-point resizer with area-mode
    coefficients.  Checks of the restated tables and arithmetic: coefficients sum to 2048, offsets stay inside the
    source, a constant image stays constant, 16 -> 32 duplicates pixels, and 31 -> 32 stays within one grey level
    of true bilinear-area weights."""
    for ssize in list(range(1, 32)) + [40, 100]:
        for is_x in (0, 1):
            ofs, c0, c1 = orc.resize_linear_tab(ssize, is_x)
            assert (c0.astype(int) + c1 == 2048).all()
            assert (ofs >= 0).all() and (ofs <= ssize - 1).all() and (np.diff(ofs) >= 0).all()
            if is_x:
                assert (c1[ofs == ssize - 1] == 0).all()
    rng = np.random.default_rng(3)
    for h, w in ((31, 31), (20, 100), (100, 20), (1, 1), (5, 31), (16, 16)):
        assert (orc.tile32(np.full((h, w), 77, np.uint8)) == 77).all()
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        t = orc.tile32(img)  # no blur below 32*32 pixels; (20,100) and (100,20) blur 3x3 first
        assert t.shape == (32, 32)
        assert orc.dcthash64(img) == orc.hash_from_tile32(t)
    img = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    assert (orc.tile32(img) == np.repeat(np.repeat(img, 2, 0), 2, 1)).all()  # f = 0 everywhere: pixel doubling
    img = rng.integers(0, 256, (31, 31), dtype=np.uint8)
    ofs, c0, c1 = orc.resize_linear_tab(31, 1)
    wx = np.zeros((32, 31))
    for d in range(32):
        wx[d, ofs[d]] += c0[d] / 2048
        wx[d, min(30, ofs[d] + 1)] += c1[d] / 2048
    assert np.abs(orc.tile32(img).astype(float) - wx @ img.astype(float) @ wx.T).max() <= 1.0


def test_keypoint_rects_rule(orc):
    """media.cpp:882-901: size >= 31, strictly inside (0, cols-2) x (0, rows-2), Rect(floor, floor, ceil)"""
    kp = np.array([[10.5, 20.25, 31.0],    # kept: (10, 20, 31)
                   [10.5, 20.25, 30.99],   # too small
                   [0.0, 5.0, 31.0],       # x0 > 0 fails
                   [5.0, 0.0, 40.0],       # y0 > 0 fails
                   [100.0, 50.0, 37.2],    # kept: ceil -> 38
                   [160.0, 50.0, 38.0],    # x1 = 198 < 198 fails (cols = 200)
                   [159.9, 50.0, 38.0],    # kept
                   [50.0, 110.0, 38.0],    # y1 = 148 < 148 fails (rows = 150)
                   ], np.float32)
    r = orc.keypoint_rects(200, 150, kp)
    assert r.tolist() == [[10, 20, 31], [100, 50, 38], [159, 50, 38]]


def test_keypoint_hashes_are_sequential_and_in_place(orc):
    """dctHash64(sub, inPlace=true): a blurred rectangle is written back, later overlapping rectangles see it; the
    blur takes its border pixels from the parent image, not by reflection at the rectangle's edge"""
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (150, 200), dtype=np.uint8)
    kp = np.array([[20.3, 30.7, 44.6], [30.0, 40.0, 53.6], [100.2, 20.1, 31.0], [25.0, 35.0, 37.2],
                   [60.0, 10.0, 134.0]], np.float32)
    hashes, after = orc.keypoint_hashes(img, kp)
    assert len(hashes) == 5
    # replay by hand with the single-rectangle entry point
    work = img.copy()
    want = [orc.dcthash64_rect_inplace(work, x, y, s, s) for x, y, s in orc.keypoint_rects(200, 150, kp).tolist()]
    assert hashes.tolist() == want and (work == after).all()
    # the 31-pixel rectangle is not blurred (area <= 32*32), the others are: their pixels changed
    _, after4 = orc.keypoint_hashes(img, kp[:4])  # (the fifth, large rectangle covers that one)
    assert (after4[20:51, 100:131] == img[20:51, 100:131]).all()
    assert (after4[30:75, 20:65] != img[30:75, 20:65]).any()
    # order matters (overlap), and the first rectangle's hash differs from hashing an isolated copy of it,
    # because the isolated copy reflects at its own border
    h_rev, _ = orc.keypoint_hashes(img, kp[[1, 0, 2, 3, 4]])
    assert sorted(h_rev.tolist()) != sorted(hashes.tolist())
    x, y, s = 20, 30, 45
    direct = orc.box_blur(img, 3)[y:y + s, x:x + s]  # blur of the whole image = parent neighbours for interior rects
    assert (after_first_rect(orc, img, x, y, s) == direct).all()


def after_first_rect(orc, img, x, y, s):
    work = img.copy()
    orc.dcthash64_rect_inplace(work, x, y, s, s)
    return work[y:y + s, x:x + s]


def test_fast_cpu_hash_equals_the_port(orc):
    """oracle/fast_hash.c (bench.py's CPU baseline for the hash leg: running column sums, vectorised) gives the hashes of
    the per-pixel restatement, under both evaluations of stages 3/5, incl. flat, extreme and striped images"""
    from cbird_amd import synth

    rng = np.random.default_rng(5)
    imgs = np.concatenate([synth.make_images(40, seed=9), rng.integers(0, 256, (24, 256, 256), dtype=np.uint8)])
    imgs[0] = 0
    imgs[1] = 255
    imgs[2, :, ::2] = 255
    imgs[3] = (np.arange(256)[None, :] * 7 + np.arange(256)[:, None] * 3) % 256
    try:
        for v in (1, 0):
            orc.set_hash_variant(v)
            assert (orc.dcthash64_fast256_batch(imgs) == orc.dcthash64_batch(imgs)).all(), v
    finally:
        orc.set_hash_variant(1)
