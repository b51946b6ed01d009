"""SURVEY section 8 row a11 -- ORB keypoints / rBRIEF descriptors (Media::makeKeyPoints, makeKeyPointDescriptors,
/root/reference/src/media.cpp:859-872).  CPU: the oracle's stages against independent numpy statements and known
values; GPU: cbh_orb bit for bit against the oracle (oracle/orb_oracle.c: parity unpinned versus OpenCV itself)."""
import numpy as np
import pytest


def _scene(rng, w, h, nrect=None, noise=4):
    """flat patches with sharp corners + mild noise: plenty of FAST corners at every pyramid level"""
    img = np.full((h, w), 128, np.int32)
    for _ in range(nrect or (w * h) // 1000):
        x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
        rw, rh = (int(v) for v in rng.integers(4, max(6, min(w, h) // 4), 2))
        img[y: y + rh, x: x + rw] = int(rng.integers(0, 256))
    img = img + rng.integers(-noise, noise + 1, img.shape)
    return img.clip(0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def orb_orc():
    from oracle import OrbOracle

    return OrbOracle()


@pytest.fixture(params=["canonical", "libstdcxx"])
def retain_order(request, gpu, orb_orc):
    """KeyPointsFilter::retainBest's order (include/cbird_hip.h, cbh_orb): the canonical one (ties kept, raster order)
    and the one libstdc++'s nth_element + partition leave, the product and the oracle switched together."""
    from cbird_amd import _lib

    v = 1 if request.param == "libstdcxx" else 0
    _lib.lib().cbh_set_tuning(b"orb_retain_order", v)
    orb_orc.set_retain_order(v)
    yield request.param
    _lib.lib().cbh_set_tuning(b"orb_retain_order", 1)  # the default
    orb_orc.set_retain_order(1)


def _retain_cases(rng):
    """(responses, n_points) pairs: heavy ties (FAST scores are small integers), distinct floats, constant, sorted
    either way, organ pipe, sizes around the 256-thread chunk and the <= 3 tail, n at both ends"""
    out = []
    for cnt in (0, 1, 2, 3, 4, 5, 7, 255, 256, 257, 511, 513, 1000, 5000, 20011):
        fam = [rng.integers(21, 60, cnt).astype(np.float32), rng.standard_normal(cnt).astype(np.float32),
               np.full(cnt, 33, np.float32), np.arange(cnt, dtype=np.float32), np.arange(cnt, dtype=np.float32)[::-1],
               np.minimum(np.arange(cnt), np.arange(cnt)[::-1]).astype(np.float32),
               (rng.integers(0, 4, cnt) * 1e-7).astype(np.float32)]
        for r in fam:
            for n in sorted({0, 1, 2, 3, cnt // 7, cnt // 2, cnt - 2, cnt - 1, cnt, cnt + 5}):
                if n >= 0:
                    out.append((np.array(r, np.float32, order="C", copy=True), int(n)))
    return out


# ---- the oracle's stages (CPU) --------------------------------------------------------------------------------------
def test_retain_best_on_the_real_nth_element(orb_orc):
    """oracle/retain_stl.cpp (the real std::nth_element + std::partition): the survivors are the best n plus, possibly,
    ties of the response left in place n - 1; every strictly better response survives; nothing is duplicated; the
    heap-select branch (depth limit 0) selects a valid best-n as well"""
    rng = np.random.default_rng(3)
    for r, n in _retain_cases(rng):
        for depth in (-1, 0, 2):
            k = orb_orc.retain_best_stl(r, n, depth)
            cnt = len(r)
            if cnt <= n:
                assert k.tolist() == list(range(cnt))
                continue
            if n == 0:
                assert len(k) == 0
                continue
            assert len(set(k.tolist())) == len(k) >= n
            nth = np.sort(r)[::-1][n - 1]
            assert (r[k[:n]] >= nth).all() and set(np.flatnonzero(r > nth).tolist()) <= set(k.tolist())
            amb = r[k[n - 1]]
            rest = np.setdiff1d(np.arange(cnt), k[:n])
            assert sorted(k[n:].tolist()) == sorted(rest[r[rest] >= amb].tolist())
    # the canonical rule keeps a superset
    r = rng.integers(21, 40, 3000).astype(np.float32)
    k = orb_orc.retain_best_stl(r, 500)
    assert set(k.tolist()) <= set(np.flatnonzero(r >= np.sort(r)[::-1][499]).tolist())


def test_oracle_tables(orb_orc):
    o = orb_orc
    # u_max for half patch 15 as every ORB implementation prints it
    assert o.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # 400 features over 12 levels at 1.2: the geometric series, rounded, remainder on the last level
    f = o.features_per_level(400)
    assert f.sum() == 400 and f[0] == 75 and all(f[i] >= f[i + 1] for i in range(11))
    assert o.features_per_level(500).sum() == 500
    # the fixed-point Gaussian: round(256 * g) of the normalised sigma-2 kernel; symmetric; sum 257 (not renormalised)
    g = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0)
    assert o.gauss7_kernel().tolist() == np.rint(256 * g / g.sum()).astype(int).tolist() == [18, 34, 49, 55, 49, 34, 18]
    # level sizes: cvRound(side / 1.2^l)
    assert [o.level_size(400, 300, l) for l in (0, 1, 2, 11)] == [(400, 300), (333, 250), (278, 208), (54, 40)]
    assert abs(o.scale(3) - 1.2 ** 3) < 1e-6
    # fastAtan2: degrees in [0, 360), within 0.3 degrees of atan2 (the published accuracy)
    for y, x in ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (3, 7), (-5, 2)):
        want = np.degrees(np.arctan2(y, x)) % 360
        assert abs(o.fast_atan2(y, x) - want) < 0.3


def test_oracle_fast_is_the_segment_test(orb_orc):
    """orc_fast_nms_scores against a direct statement of FAST-9: a corner has 9 contiguous circle pixels all darker
    than v - 20 or all brighter than v + 20; its score is the largest t for which that still holds with threshold t;
    keypoints are strict 3x3 maxima of the score"""
    rng = np.random.default_rng(5)
    img = _scene(rng, 96, 80, nrect=40)
    got = orb_orc.fast_nms_scores(img)
    ox = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    oy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    h, w = img.shape
    im = img.astype(int)

    def is_corner(y, x, t):
        v = im[y, x]
        d = [v - im[y + oy[k], x + ox[k]] for k in range(16)]
        for sgn in (1, -1):
            m = [sgn * e > t for e in d]
            m2 = m + m
            run = 0
            for b in m2:
                run = run + 1 if b else 0
                if run >= 9:
                    return True
        return False

    raw = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(y, x, 20):
                t = 20
                while is_corner(y, x, t + 1):
                    t += 1
                raw[y, x] = t
    want = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = raw[y, x]
            if s and all(s > raw[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dy or dx):
                want[y, x] = s
    assert (want > 0).sum() > 20
    assert (got == want).all()


def test_oracle_resize_blur_harris_angle(orb_orc):
    rng = np.random.default_rng(6)
    img = _scene(rng, 200, 150)
    o = orb_orc
    # bilinear resize: within 1 grey level of the float formula (fixed-point coefficients)
    small = o.resize_linear(img, 167, 125)
    sx, sy = 200 / 167, 150 / 125
    fx = np.clip((np.arange(167) + 0.5) * sx - 0.5, 0, 199)
    fy = np.clip((np.arange(125) + 0.5) * sy - 0.5, 0, 149)
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    x1, y1 = np.minimum(x0 + 1, 199), np.minimum(y0 + 1, 149)
    ax, ay = fx - x0, fy - y0
    f = img.astype(float)
    ref = ((f[y0][:, x0] * (1 - ax) + f[y0][:, x1] * ax) * (1 - ay)[:, None]
           + (f[y1][:, x0] * (1 - ax) + f[y1][:, x1] * ax) * ay[:, None])
    assert np.abs(small.astype(float) - ref).max() <= 1.0
    assert (o.pyramid_level(img, 1) == o.resize_linear(img, *o.level_size(200, 150, 1))).all()
    # Gaussian: the integer kernel applied separably with reflect-101 borders
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    pad = np.pad(img.astype(int), 3, mode="reflect")
    rows = sum(k[t] * pad[:, t: t + 200] for t in range(7))
    full = sum(k[t] * rows[t: t + 150] for t in range(7))
    assert (o.gauss7_blur(img) == np.clip((full + 32768) >> 16, 0, 255)).all()
    # Harris: float32 evaluation of the published formula on the 7x7 block
    im = img.astype(int)
    for (x, y) in ((40, 40), (100, 75), (160, 110)):
        a = b = c = 0
        for i in range(-3, 4):
            for j in range(-3, 4):
                yy, xx = y + i, x + j
                ix = (im[yy, xx + 1] - im[yy, xx - 1]) * 2 + (im[yy - 1, xx + 1] - im[yy - 1, xx - 1]) + (im[yy + 1, xx + 1] - im[yy + 1, xx - 1])
                iy = (im[yy + 1, xx] - im[yy - 1, xx]) * 2 + (im[yy + 1, xx - 1] - im[yy - 1, xx - 1]) + (im[yy + 1, xx + 1] - im[yy - 1, xx + 1])
                a, b, c = a + ix * ix, b + iy * iy, c + ix * iy
        sc = 1.0 / (4 * 7 * 255.0)
        want = (float(a) * b - float(c) * c - 0.04 * (a + b) ** 2) * sc ** 4
        assert abs(o.harris(img, x, y) - want) <= 1e-5 * max(1.0, abs(want))
    # orientation: the intensity centroid of the circular patch
    um = o.umax()
    for (x, y) in ((50, 50), (120, 70)):
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-um[abs(v)], um[abs(v)] + 1):
                m10 += u * im[y + v, x + u]
                m01 += v * im[y + v, x + u]
        assert abs(o.ic_angle(img, x, y) - o.fast_atan2(m01, m10)) == 0


def test_oracle_detect_compute_properties(orb_orc):
    from cbird_amd.orb import synthetic_pattern

    rng = np.random.default_rng(7)
    img = _scene(rng, 400, 300)
    o = orb_orc
    o.set_pattern(synthetic_pattern())
    o.set_retain_order(1)
    kp_stl = o.detect(img, 400)
    o.set_retain_order(0)
    kp = o.detect(img, 400)
    o.set_retain_order(1)  # the default
    assert 200 < len(kp) <= 400 + 50
    # the library's order: nearly the same keypoints with identical values (the two rules differ in the ties at the two
    # cuts only), level by level, not in raster order
    as_set = {tuple(k.tolist()) for k in kp}
    assert 200 < len(kp_stl) <= 400 + 50 and sum(tuple(k.tolist()) in as_set for k in kp_stl) > 0.9 * len(kp_stl)
    assert (np.diff(kp_stl["octave"]) >= 0).all()
    l0 = kp_stl[kp_stl["octave"] == 0]
    assert not (np.diff(np.rint(l0["y"]).astype(int) * 10000 + np.rint(l0["x"]).astype(int)) > 0).all()
    per = o.features_per_level(400)
    for l in range(12):
        sel = kp[kp["octave"] == l]
        lw, lh = o.level_size(400, 300, l)
        if lw <= 62 or lh <= 62:
            assert len(sel) == 0
            continue
        s = o.scale(l)
        assert np.allclose(sel["size"], 31 * s)
        # inside the 31-pixel border of the level, integer level coordinates, raster order
        lx, ly = sel["x"] / (s if l else 1), sel["y"] / (s if l else 1)
        assert (np.abs(lx - np.rint(lx)) < 1e-3).all() and (np.rint(lx) >= 31).all() and (np.rint(lx) < lw - 31).all()
        assert (np.rint(ly) >= 31).all() and (np.rint(ly) < lh - 31).all()
        key = np.rint(ly).astype(int) * 10000 + np.rint(lx).astype(int)
        assert (np.diff(key) > 0).all()
        assert len(sel) >= min(per[l], len(sel))
    assert ((kp["angle"] >= 0) & (kp["angle"] < 360)).all()
    kp2, desc = o.compute(img, kp)
    assert len(kp2) == len(kp) == len(desc) and (kp2["octave"] == kp["octave"]).all()
    assert np.abs(kp2["x"] - kp["x"]).max() < 1e-3
    bits = np.unpackbits(desc, axis=1)
    assert 0.3 < bits.mean() < 0.7
    # a keypoint's descriptor is a function of the blurred level and its angle only
    lvl = o.gauss7_blur(o.pyramid_level(img, 2))
    s = o.scale(2)
    for j in np.flatnonzero(kp["octave"] == 2)[:5]:
        cx, cy = int(np.rint(kp2["x"][j] / s)), int(np.rint(kp2["y"][j] / s))
        assert (o.descriptor(lvl, cx, cy, kp["angle"][j]) == desc[j]).all()


def test_pattern_loader(tmp_path):
    from cbird_amd.orb import load_pattern, synthetic_pattern

    p = synthetic_pattern(3)
    assert p.shape == (1024,) and np.abs(p).max() <= 13
    lines = ["static int bit_pattern_31_[256*4] =", "{"]
    for i in range(256):
        a = p[4 * i: 4 * i + 4]
        lines.append(f"    {a[0]},{a[1]}, {a[2]},{a[3]}/*mean ({i * 1e-5:g}), correlation (0.{i})*/,")
    lines += ["};", "static void other() { int x[3] = {1, 2, 3}; }"]
    f = tmp_path / "orb.cpp"
    f.write_text("\n".join(lines))
    assert (load_pattern(str(f)) == p).all()
    g = tmp_path / "pat.bin"
    g.write_bytes(p.tobytes())
    assert (load_pattern(str(g)) == p).all()


# ---- GPU: bit-exact against the oracle ------------------------------------------------------------------------------
def _compare(o, imgs, nfeat, res):
    for img, (kp, after, desc) in zip(imgs, res):
        want = o.detect(img, nfeat)
        assert len(kp) == len(want), (img.shape, len(kp), len(want))
        for f in ("x", "y", "size", "angle", "response"):
            assert (kp[f].view(np.uint32) == want[f].view(np.uint32)).all(), (img.shape, f)
        assert (kp["octave"] == want["octave"]).all()
        if desc is not None:
            w2, d2 = o.compute(img, want)
            assert len(w2) == len(want)
            assert (after[:, 0].view(np.uint32) == w2["x"].view(np.uint32)).all()
            assert (after[:, 1].view(np.uint32) == w2["y"].view(np.uint32)).all()
            assert (desc == d2).all(), img.shape


@pytest.mark.gpu
def test_gpu_retain_best_equals_libstdcxx(gpu, orb_orc):
    """cbh_orb_retain_best_dev (the workgroup restatement of introselect + partition in orb.hip) against the real
    library: the same survivors in the same order, for every case family, through nth_element's own depth limit and
    through forced limits 0..3 (heap select after 0..3 partition rounds)"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(5)
    cases = _retain_cases(rng)
    checked = 0
    for r, n in cases:
        cnt = len(r)
        d_r = torch.from_numpy(r if cnt else np.zeros(1, np.float32)).cuda()
        d_o = torch.zeros(max(cnt, 1), dtype=torch.int32, device="cuda")
        d_c = torch.zeros(1, dtype=torch.int32, device="cuda")
        for depth in (-1, 0, 1, 3):
            if depth >= 0 and cnt > 6000:
                continue  # the forced heap branch is one thread: keep it to the sizes that take milliseconds
            _lib.check(L.cbh_orb_retain_best_dev(d_r.data_ptr(), cnt, n, depth, d_o.data_ptr(), d_c.data_ptr(), 0, None),
                       "cbh_orb_retain_best_dev")
            k = int(d_c.item())
            want = orb_orc.retain_best_stl(r, n, depth)
            assert k == len(want), (cnt, n, depth, k, len(want))
            assert (d_o[:k].cpu().numpy() == want).all(), (cnt, n, depth)
            checked += 1
    assert checked > 2000


@pytest.mark.gpu
def test_gpu_orb_equals_oracle(gpu, orb_orc, retain_order):
    from cbird_amd import orb

    pat = orb.synthetic_pattern()
    orb_orc.set_pattern(pat)
    orb.set_pattern(pat)
    rng = np.random.default_rng(11)
    sizes = [(400, 300), (300, 400), (400, 400), (400, 225), (333, 250), (128, 96), (63, 63), (64, 70), (62, 200),
             (75, 63), (401, 267), (200, 150), (30, 30), (5, 3), (640, 480)]
    imgs = [_scene(rng, w, h) for (w, h) in sizes]
    imgs.append(np.zeros((120, 160), np.uint8))                                   # nothing to find
    imgs.append(rng.integers(0, 256, (150, 200), dtype=np.uint8))                 # pure noise: thousands of corners
    chk = (np.indices((160, 200)).sum(0) // 8 % 2 * 255).astype(np.uint8)        # checkerboard: ties everywhere
    imgs.append(chk)
    for nfeat in (400, 500, 37, 0):
        _compare(orb_orc, imgs, nfeat, orb.orb(imgs, nfeat))
    # detection alone needs no pattern and returns the same keypoints
    kps = orb.make_keypoints(imgs[:4], 400)
    for img, k in zip(imgs[:4], kps):
        assert (k == orb_orc.detect(img, 400)).all()


@pytest.mark.gpu
def test_gpu_make_keypoint_descriptors_on_provided_keypoints(gpu, orb_orc):
    """the extractor alone (makeKeyPointDescriptors as the reference calls it, on the keypoints makeKeyPoints
    returned): equal to the oracle's compute() -- also for a list that was shuffled, thinned and given keypoints too
    close to the border (dropped) -- and equal to the fused detect + describe call"""
    from cbird_amd import orb

    pat = orb.synthetic_pattern(5)
    orb_orc.set_pattern(pat)
    orb.set_pattern(pat)
    rng = np.random.default_rng(21)
    imgs = [_scene(rng, w, h) for (w, h) in ((400, 300), (320, 400), (150, 110), (64, 64), (40, 40))]
    kps = orb.make_keypoints(imgs, 400)
    got = orb.make_keypoint_descriptors(imgs, kps)
    fused = orb.orb(imgs, 400)
    for img, k, (k2, d), (fk, fa, fd) in zip(imgs, kps, got, fused):
        w2, d2 = orb_orc.compute(img, k)
        assert (k2 == w2).all() and (d == d2).all()
        assert (fd == d).all() and (fa[:, 0] == k2["x"]).all() and (fa[:, 1] == k2["y"]).all()
    # a modified list: shuffled, every third dropped, two points near the border added
    k = kps[0].copy()
    k = k[rng.permutation(len(k))][::3]
    extra = np.zeros(2, orb.KP_DTYPE)
    extra["x"], extra["y"], extra["octave"], extra["angle"] = [5.0, 395.0], [100.0, 100.0], [0, 1], [10.0, 20.0]
    k = np.concatenate([k[:10], extra, k[10:]])
    (k2, d), = orb.make_keypoint_descriptors(imgs[:1], [k])
    w2, d2 = orb_orc.compute(imgs[0], k)
    assert len(k2) == len(k) - 2 and (k2 == w2).all() and (d == d2).all()
    # a keypoint whose octave does not exist in its image is an error, not a read outside the level
    bad = np.zeros(1, orb.KP_DTYPE)
    bad["x"], bad["y"], bad["octave"] = 32.0, 32.0, 5
    with pytest.raises(orb._lib.CbhError):
        orb.make_keypoint_descriptors(imgs[3:4], [bad])


@pytest.mark.gpu
def test_gpu_orb_cos_sin_match_libm(gpu, orb_orc, retain_order):
    """the one place the device's libm meets the host's: (float)cos(angle), (float)sin(angle) of the descriptor
    rotation.  Every keypoint of a large batch gives the oracle's descriptor, i.e. no rounding difference surfaced."""
    from cbird_amd import orb

    pat = orb.synthetic_pattern(9)
    orb_orc.set_pattern(pat)
    orb.set_pattern(pat)
    rng = np.random.default_rng(12)
    imgs = [_scene(rng, 400, 300) for _ in range(24)]
    res = orb.orb(imgs, 400)
    _compare(orb_orc, imgs, 400, res)
    assert sum(len(r[0]) for r in res) > 5000


@pytest.mark.gpu
def test_gpu_orb_large_images(gpu, orb_orc, retain_order):
    """the sizes the interface admits, not just the 400-pixel images cbird feeds: 12 MP, the widest row (8192), more
    keypoints asked than cbird ever does"""
    from cbird_amd import orb

    pat = orb.synthetic_pattern()
    orb_orc.set_pattern(pat)
    orb.set_pattern(pat)
    rng = np.random.default_rng(13)
    for (w, h, nf) in ((4000, 3000, 400), (8192, 300, 1500)):
        img = _scene(rng, w, h, nrect=400)
        _compare(orb_orc, [img], nf, orb.orb([img], nf))


@pytest.mark.gpu
def test_gpu_orb_arguments(gpu):
    from cbird_amd import _lib, orb

    L = _lib.lib()
    assert L.cbh_orb_set_pattern(None) == _lib.CBH_E_INVAL
    bad = np.zeros(1024, np.int8)
    bad[5] = 16
    assert L.cbh_orb_set_pattern(bad.ctypes.data) == _lib.CBH_E_INVAL
    assert orb.orb([], 400) == []
    with pytest.raises(ValueError):
        orb.orb([np.zeros((4, 4, 3), np.uint8)], 400)
    assert L.cbh_orb_retain_best_dev(None, 5, 2, -1, None, None, 0, None) == _lib.CBH_E_INVAL
    assert L.cbh_orb_retain_best_dev(None, 5, 2, -1, None, None, 30, None) == _lib.CBH_E_NODEVICE
    assert L.cbh_set_tuning(b"orb_retain_order", 2) == _lib.CBH_E_INVAL
    # truncation is reported, not silent: counts above kp_cap
    rng = np.random.default_rng(1)
    img = _scene(rng, 400, 300)
    orb.set_pattern(orb.synthetic_pattern())
    full = orb.orb([img], 400)[0][0]
    part = orb.orb([img], 400, kp_cap=len(full))[0][0]
    assert (part == full).all()
