"""oracle/cv_dct32.c: the restatement of OpenCV 2.4's cv::dct (32-point float path of dxt.cpp) and cv::sum
(stat.cpp) that dctHash64 calls at src/cvutil.cpp:476-477 and :528.  OpenCV is not available here, so what these
CPU tests can pin is (a) that the factorised algorithm IS the orthonormal DCT-II (to float rounding, against a float64
matrix evaluation), (b) its tables, (c) cv::sum's grouping, (d) the closed-form known answers through both
evaluations of the hash, and (e) how rarely the two evaluations and float64 disagree on a bit."""
import numpy as np
import pytest


def dct_matrix(n=32):
    k = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    return np.sqrt(np.where(k == 0, 1.0, 2.0) / n) * np.cos(np.pi * (2 * j + 1) * k / (2 * n))


def test_factorised_transform_is_the_orthonormal_dct2(orc):
    C = dct_matrix()
    rng = np.random.default_rng(1)
    for _ in range(50):
        x = rng.integers(0, 256, 32).astype(np.float32)
        got = orc.cv_dct32_1d(x)
        want = C @ x.astype(np.float64)
        # 16-point FFT + two rotations in f32: a handful of ulps of the largest term (|X0| <= 255*sqrt(32))
        assert np.abs(got - want).max() <= 2e-4, np.abs(got - want).max()
    m = rng.integers(0, 256, (32, 32)).astype(np.float32)
    got2 = orc.cv_dct32x32(m)
    want2 = C @ m.astype(np.float64) @ C.T
    assert np.abs(got2 - want2).max() <= 2e-3
    # linearity in exact cases: a constant row has only a DC term, and it is exact (255*32*0.25*sin45 rounds once)
    y = orc.cv_dct32_1d(np.full(32, 200, np.float32))
    assert abs(y[0] - 200 * np.sqrt(32)) < 1e-3 and np.abs(y[1:]).max() < 1e-3


def test_tables_are_the_double_recurrences(orc):
    import ctypes as C

    dft = np.zeros(64, np.float32)
    dct = np.zeros(34, np.float32)
    orc.L.orc_cv_dct32_tables.argtypes = [C.c_void_p, C.c_void_p]
    orc.L.orc_cv_dct32_tables(dft.ctypes.data, dct.ctypes.data)
    k = np.arange(32)
    w = np.exp(-2j * np.pi * k / 32)
    assert np.abs(dft[0::2] - w.real).max() < 1e-7 and np.abs(dft[1::2] - w.imag).max() < 1e-7
    assert dft[0] == 1.0 and dft[1] == 0.0 and dft[32] == -1.0 and dft[33] == 0.0  # set exactly, not by recurrence
    k = np.arange(17)
    w = 0.25 * np.exp(-1j * np.pi * k / 64)
    assert np.abs(dct[0::2] - w.real).max() < 1e-7 and np.abs(dct[1::2] - w.imag).max() < 1e-7
    assert dct[0] == 0.25


def test_cv_sum_groups_of_four_in_float(orc):
    # ((a+b)+c)+d in float loses the small terms next to 2^24; a sequential double sum does not
    x = np.array([16777216.0, 1.0, 1.0, 1.0] + [0.0] * 60, np.float32)
    assert orc.cv_sum_f32(x) == 16777216.0
    assert float(np.sum(x.astype(np.float64))) == 16777219.0
    # groups are accumulated in double: 16 groups of (2^24, 0, 0, 0) + one lone 1.0 survives
    y = np.zeros(64, np.float32)
    y[0::4] = 16777216.0
    y[62] = 1.0
    assert orc.cv_sum_f32(y) == 16 * 16777216.0  # the 1.0 is absorbed inside its float group
    y[62] = 0.0
    y[63] = 0.0
    z = np.concatenate([y[:60], np.array([1.0, 0, 0, 0], np.float32)])
    assert orc.cv_sum_f32(z) == 15 * 16777216.0 + 1.0
    rng = np.random.default_rng(3)
    v = rng.normal(0, 300, 64).astype(np.float32)
    want = 0.0
    for g in range(16):
        a, b, c, d = v[4 * g: 4 * g + 4]
        want += float(np.float32(np.float32(np.float32(a + b) + c) + d))
    assert orc.cv_sum_f32(v) == want


@pytest.mark.parametrize("variant", [0, 1])
def test_single_basis_function_sets_exactly_its_bit_both_variants(orc, variant):
    zz = orc.zigzag81()
    y = np.arange(32)[:, None]
    x = np.arange(32)[None, :]
    for bit in range(64):
        u, v = divmod(int(zz[6 + bit]), 9)
        f = 128 + 100 * np.cos(np.pi * (2 * y + 1) * u / 64) * np.cos(np.pi * (2 * x + 1) * v / 64)
        tile = np.clip(np.rint(f), 0, 255).astype(np.uint8)
        assert orc.hash_from_tile32_v(tile, variant) == ((1 << bit) if bit else 1)


def test_variant_switch_reaches_every_entry_point(orc):
    from cbird_amd import synth

    imgs = synth.make_images(8, seed=21)
    try:
        orc.set_hash_variant(0)
        h0 = orc.dcthash64_batch(imgs)
        c0 = [orc.hash_from_tile32(orc.tile32(i), with_coefs=True)[1] for i in imgs]
        orc.set_hash_variant(1)
        assert orc.hash_variant() == 1
        h1 = orc.dcthash64_batch(imgs)
        c1 = [orc.hash_from_tile32(orc.tile32(i), with_coefs=True)[1] for i in imgs]
    finally:
        orc.set_hash_variant(1)
    # same transform, different rounding: coefficients agree to float precision but not bit for bit
    assert any((a != b).any() for a, b in zip(c0, c1))
    assert all(np.abs(a - b).max() < 5e-3 for a, b in zip(c0, c1))
    assert (h0 == h1).all()  # these eight images have no coefficient within 5e-3 of its threshold


def test_evaluations_disagree_only_on_bits_at_the_threshold(orc):
    """the at-risk statistic in miniature (tools/hash_at_risk.py runs it over the 1M bench images): whenever two
    evaluations give different hashes, the float64 margin of that tile is within float rounding of zero"""
    from cbird_amd import synth

    imgs = synth.make_images(300, seed=77)
    tiles = np.stack([orc.tile32(i) for i in imgs])
    h0, _ = orc.hash_tiles_stats(tiles, 0)
    h1, _ = orc.hash_tiles_stats(tiles, 1)
    h2, m2 = orc.hash_tiles_stats(tiles, 2)
    for a in (h0, h1):
        diff = a != h2
        assert diff.sum() <= 3
        assert (m2[diff] < 2e-3).all()
