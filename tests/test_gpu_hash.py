"""GPU parity suite for the hash build path (dcthash kernels) through the C-ABI.  Bit-exact
against oracle/cbird_oracle.c on the same inputs."""
import numpy as np
import pytest

# every test with the fractional-ratio geometries of <= 1920 columns on k_band_area (the default) as well as on the VALU
# kernels (k_blur_area_regs / k_blur_area: what serves every other geometry, odd views, and a device whose band tables
# cannot be made)
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("band_area")]


@pytest.fixture(params=["band_area", "regs"])
def band_area(request, gpu):
    from cbird_amd import _lib

    _lib.lib().cbh_set_tuning(b"hash_band_area", 1 if request.param == "band_area" else 0)
    yield request.param
    _lib.lib().cbh_set_tuning(b"hash_band_area", 1)


@pytest.mark.parametrize("w,h", [(32, 32), (64, 64), (64, 32), (128, 128), (128, 160), (256, 256),
                                 (512, 256), (320, 480), (1024, 1024)])
def test_hash_matches_oracle_random_and_smooth(gpu, orc, w, h):
    from cbird_amd import synth

    rng = np.random.default_rng(w * 131 + h)
    n = 24 if w * h <= 256 * 256 else 6
    noise = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    smooth = synth.make_images(n, w=w, h=h, seed=w + h, dup_frac=0.25)
    for imgs in (noise, smooth):
        got = gpu.dct_hash64_batch(imgs)
        want = orc.dcthash64_batch(imgs)
        assert (got == want).all(), (w, h, [hex(int(x)) for x in (got ^ want)])


@pytest.mark.parametrize("w,h", [(100, 100), (33, 47), (600, 400), (257, 256), (400, 533), (1280, 720), (2048, 1536),
                                 (32, 33), (640, 64)])
def test_hash_any_size_general_area_path(gpu, orc, w, h):
    """sizes whose ratio to 32 is not an integer: cv::resize's weighted INTER_AREA path"""
    from cbird_amd import _lib
    import torch

    rng = np.random.default_rng(w * 7 + h)
    n = 5
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[1] = (128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0)).astype(np.uint8)
    got = gpu.dct_hash64_batch(imgs)
    want = orc.dcthash64_batch(imgs)
    assert (got == want).all(), [hex(int(x)) for x in got ^ want]
    # stage level: the 32x32 tiles agree byte for byte
    L = _lib.lib()
    d = torch.from_numpy(imgs).cuda()
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
    _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None), "t")
    t = tiles.cpu().numpy()
    for i in range(n):
        assert (t[i] == orc.tile32(imgs[i])).all()


@pytest.mark.parametrize("w,h", [(3840, 2160), (4000, 3000), (3000, 2000), (2049, 64), (2051, 97), (4096, 4096), (5000, 40),
                                 (8192, 64), (8191, 33), (6001, 100)])
def test_hash_images_wider_than_one_workgroup(gpu, orc, w, h):
    """images wider than 2048 pixels: k_blur_area_regs on 2 or 4 column strips (each a view of the parent making its share
    of the 32 output cells; large batches, forced here with "hash_stream" 4) == the LDS band kernel k_blur_area (small
    batches; "hash_stream" 0) == oracle, hashes and 32 x 32 tiles; integer and fractional ratios, strips that end on and
    off 8-pixel boundaries"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(w + 3 * h)
    n = 3 if w * h > 4_000_000 else 9  # enough workgroups for the strip kernels to be chosen for the small ones too
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[0] = (128 + 100 * np.sin(xx / 97.0) * np.cos(yy / 41.0)).astype(np.uint8)
    imgs[1, :, : w // 2] = 255  # an edge on the strip boundary of two strips
    want = orc.dcthash64_batch(imgs)
    tile0 = orc.tile32(imgs[0])
    d = torch.from_numpy(imgs).cuda()
    try:
        for knob in (4, 0):
            L.cbh_set_tuning(b"hash_stream", knob)  # strips of four steps whatever the batch size / never strips
            assert (gpu.dct_hash64_batch(imgs) == want).all(), (w, h, knob)
            out = torch.zeros(n, dtype=torch.int64, device="cuda")
            tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
            _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None), "t")
            assert (tiles[0].cpu().numpy() == tile0).all(), (w, h, knob)
    finally:
        L.cbh_set_tuning(b"hash_stream", 1)


@pytest.mark.parametrize("w,h", [(512, 160), (768, 96), (1024, 128), (1280, 64), (2048, 96), (4096, 64), (256, 192), (3072, 64)])
def test_hash_integer_ratios_with_padded_cells(gpu, orc, w, h):
    """integer ratios whose cells are an even number of dwords: k_blur_area_regs keeps a pad dword behind every cell of a
    blurred LDS row where the 32 lanes reading their cells together would share a bank 4 ways or more (512, 1024, 2048 px
    ...; not 768, 1280): hashes and tiles == oracle, whole images (fused and strip-split forms) and column strips of wide
    ones"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(7 * w + h)
    n = 12
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[0] = (128 + 100 * np.sin(xx / 53.0) * np.cos(yy / 29.0)).astype(np.uint8)
    want = orc.dcthash64_batch(imgs)
    tile0 = orc.tile32(imgs[0])
    d = torch.from_numpy(imgs).cuda()
    try:
        for pad in (1,):
            for fuse, stream in ((2, 4), (0, 4)):  # whole image per workgroup; strips of four steps + k_tile_hash
                L.cbh_set_tuning(b"hash_fuse", fuse)
                L.cbh_set_tuning(b"hash_stream", stream)
                assert (gpu.dct_hash64_batch(imgs) == want).all(), (w, h, pad, fuse)
                out = torch.zeros(n, dtype=torch.int64, device="cuda")
                tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
                _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None),
                           "t")
                assert (tiles[0].cpu().numpy() == tile0).all(), (w, h, pad, fuse)
    finally:
        L.cbh_set_tuning(b"hash_fuse", 1)
        L.cbh_set_tuning(b"hash_stream", 1)


@pytest.mark.parametrize("w,h", [(400, 300), (1280, 200), (1366, 130), (1031, 257), (1536, 96), (2560, 120), (3000, 200), (5000, 90),
                                 (8000, 64), (704, 576)])
def test_hash_rows_per_step_14_and_21(gpu, orc, w, h):
    """k_blur_area_regs<7> walks an image 14 or 21 source rows per step (by how the rows fill the area phase's turns: 21 at
    1280 / 1366 / 1536 px and on column strips, 14 at 400 px), in strips whose length is chosen for the fewest rows
    processed: hashes and tiles == oracle, fused and strip-split, whole images and column strips, and the band kernel"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(11 * w + h)
    n = 6
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    imgs[0] = (128 + 100 * np.sin(xx / 71.0) * np.cos(yy / 23.0)).astype(np.uint8)
    want = orc.dcthash64_batch(imgs)
    tile0 = orc.tile32(imgs[0])
    d = torch.from_numpy(imgs).cuda()
    try:
        for rows in (1,):
            for fuse, stream in ((2, 4), (0, 4), (0, 0)):
                L.cbh_set_tuning(b"hash_fuse", fuse)
                L.cbh_set_tuning(b"hash_stream", stream)
                assert (gpu.dct_hash64_batch(imgs) == want).all(), (w, h, rows, fuse, stream)
                out = torch.zeros(n, dtype=torch.int64, device="cuda")
                tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
                _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None),
                           "t")
                assert (tiles[0].cpu().numpy() == tile0).all(), (w, h, rows, fuse, stream)
    finally:
        L.cbh_set_tuning(b"hash_fuse", 1)
        L.cbh_set_tuning(b"hash_stream", 1)


@pytest.mark.parametrize("n", [1, 2, 5, 33])
def test_fused_tiles_are_hashed_two_per_wave(gpu, orc, n):
    """stages 3-6 of the fused strip kernel's tiles: k_tiles_hash2 (two images per 64-thread workgroup); odd counts leave
    half a workgroup empty; the tiles handed back are the oracle's"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    w, h = 400, 300
    rng = np.random.default_rng(5 + n)
    imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    want = orc.dcthash64_batch(imgs)
    d = torch.from_numpy(imgs).cuda()
    try:
        L.cbh_set_tuning(b"hash_fuse", 2)
        L.cbh_set_tuning(b"hash_stream", 4)
        for knob in (1,):
            assert (gpu.dct_hash64_batch(imgs) == want).all(), (n, knob)
            out = torch.zeros(n, dtype=torch.int64, device="cuda")
            tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
            _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w, w * h, out.data_ptr(), tiles.data_ptr(), 0, None), "t")
            t = tiles.cpu().numpy()
            assert all((t[i] == orc.tile32(imgs[i])).all() for i in range(n)), (n, knob)
            assert (out.cpu().numpy().view(np.uint64) == want).all(), (n, knob)
    finally:
        L.cbh_set_tuning(b"hash_fuse", 1)
        L.cbh_set_tuning(b"hash_stream", 1)


def test_hash_edge_images(gpu, orc, hash256_kernel):
    imgs = np.zeros((6, 256, 256), np.uint8)
    imgs[1] = 255
    imgs[2, ::2] = 255
    imgs[3, :, ::2] = 255
    imgs[4, :128] = 255
    imgs[5] = (np.arange(256)[None, :] + np.arange(256)[:, None]) // 2
    got = gpu.dct_hash64_batch(imgs)
    assert (got == orc.dcthash64_batch(imgs)).all()
    assert (got != 0).all()


def test_hash_strided_views_and_single(gpu, orc):
    rng = np.random.default_rng(77)
    big = rng.integers(0, 256, (5, 300, 400), dtype=np.uint8)
    view = big[:, 10:266, 40:296]  # row stride 400, image stride 120000
    got = gpu.dct_hash64_batch(view)
    want = orc.dcthash64_batch(np.ascontiguousarray(view))
    assert (got == want).all()
    assert gpu.dct_hash64(view[2]) == int(want[2])


def test_hash_known_answers(gpu, orc, hash256_kernel):
    """single DCT basis function -> exactly its bit (see tests/test_oracle.py)"""
    zz = orc.zigzag81()
    imgs32, imgs256, bits = [], [], []
    for bit in range(64):
        u, v = divmod(int(zz[6 + bit]), 9)
        for n, dst in ((32, imgs32), (256, imgs256)):
            y = np.arange(n)[:, None]
            x = np.arange(n)[None, :]
            f = 128 + 100 * np.cos(np.pi * (2 * y + 1) * u / (2 * n)) * np.cos(np.pi * (2 * x + 1) * v / (2 * n))
            dst.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
        bits.append(bit)
    got = gpu.dct_hash64_batch(np.stack(imgs32))
    assert got.tolist() == [(1 << b) if b else 1 for b in bits]
    got = gpu.dct_hash64_batch(np.stack(imgs256))
    assert got.tolist()[1:10] == [1 << b for b in range(1, 10)]


def test_hash_unsupported_geometry(gpu):
    from cbird_amd import _lib

    with pytest.raises(gpu.CbhError) as e:
        gpu.dct_hash64_batch(np.zeros((1, 20, 8200), np.uint8))  # wider than 8192
    assert e.value.code == _lib.CBH_E_UNSUPPORTED
    assert len(gpu.dct_hash64_batch(np.zeros((0, 256, 256), np.uint8))) == 0


def test_hash_large_batch_sampled(gpu, orc, hash256_kernel):
    """8192 tiles resident on the device (512 MiB), hashed in one launch; a 256-image sample is
    checked against the oracle and duplicates of the same tile hash identically."""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randint(0, 256, (n // 2, 256, 256), dtype=torch.uint8, device="cuda", generator=g)
    # low-pass so that hashes are not pure noise: 4x4 mean via avg_pool then upsample
    lp = torch.nn.functional.avg_pool2d(base.float()[:, None], 8, 8)
    lp = torch.nn.functional.interpolate(lp, scale_factor=8, mode="bilinear")[:, 0]
    base = (0.7 * lp + 0.3 * base.float()).round().clamp(0, 255).to(torch.uint8)
    imgs = torch.cat([base, base]).contiguous()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, None),
               "dcthash_batch_dev")
    got = out.cpu().numpy().view(np.uint64)
    assert (got[: n // 2] == got[n // 2:]).all()
    sample = np.random.default_rng(1).choice(n, 256, replace=False)
    want = orc.dcthash64_batch(imgs[torch.from_numpy(sample).cuda()].cpu().numpy())
    assert (got[sample] == want).all()


def test_band_kernel_every_quotient_borders_and_ragged_batches(gpu, orc):
    """k_dcthash_256_band against the oracle where its own machinery could slip: batches that do not fill the last
    wave's four images, images that drive the 7x7 sums through every quotient and both extremes, structure confined to
    the three border rows / columns that REFLECT_101 folds back (the first and last column tile's band matrices, the
    row addresses), single bright pixels in every corner, and the 32 x 32 tiles byte for byte"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(12)
    imgs = rng.integers(0, 256, (23, 256, 256), dtype=np.uint8)
    yy, xx = np.mgrid[0:256, 0:256]
    imgs[0] = ((xx * 7 + yy * 13) % 256).astype(np.uint8)
    imgs[1] = 0
    imgs[2] = 255
    imgs[3] = np.where((xx < 3) | (xx > 252), 255, 0)
    imgs[4] = np.where((yy < 3) | (yy > 252), 255, 0)
    imgs[5] = np.where((xx + yy) % 2 == 0, 255, 0)
    imgs[6] = 0
    for y, x in ((0, 0), (0, 255), (255, 0), (255, 255), (3, 3), (252, 252), (1, 254)):
        imgs[6, y, x] = 255
    imgs[7] = (yy // 8 * 8 + xx // 8) % 256          # constant 8x8 cells
    imgs[8] = np.minimum(255, (xx // 16) * 17)       # steps on the column tile boundaries
    imgs[9] = rng.integers(127, 130, (256, 256))     # around the i8 sign change
    for n in (1, 2, 3, 4, 5, 7, 23):
        got = gpu.dct_hash64_batch(imgs[:n])
        assert (got == orc.dcthash64_batch(imgs[:n])).all(), n
    d = torch.from_numpy(imgs).cuda()
    out = torch.zeros(len(imgs), dtype=torch.int64, device="cuda")
    tiles = torch.zeros((len(imgs), 32, 32), dtype=torch.uint8, device="cuda")
    _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), len(imgs), 256, 256, 256, 65536, out.data_ptr(), tiles.data_ptr(), 0,
                                       None), "tiles")
    t = tiles.cpu().numpy()
    for i in range(len(imgs)):
        assert (t[i] == orc.tile32(imgs[i])).all(), i
    # rows that are not 16-byte aligned take k_dcthash_256 (same answers)
    big = rng.integers(0, 256, (3, 300, 400), dtype=np.uint8)
    view = big[:, 7:263, 9:265]
    assert (gpu.dct_hash64_batch(view) == orc.dcthash64_batch(np.ascontiguousarray(view))).all()


@pytest.mark.parametrize("knob", [0])
def test_mfma_variant_is_bit_identical(gpu, orc, knob):
    """k_dcthash_256 (all VALU, "hash_mfma" 0) == the default kernel (k_dcthash_256_band: box filter on the matrix cores) == oracle, tiles included"""
    import torch

    from cbird_amd import _lib, synth

    L = _lib.lib()
    imgs = np.concatenate([synth.make_images(37, seed=3), np.random.default_rng(4).integers(0, 256, (30, 256, 256),
                                                                                          dtype=np.uint8)])
    imgs[5] = 0
    imgs[6] = 255
    want = orc.dcthash64_batch(imgs)
    try:
        L.cbh_set_tuning(b"hash_mfma", knob)
        got = gpu.dct_hash64_batch(imgs)
        d = torch.from_numpy(imgs).cuda()
        out = torch.zeros(len(imgs), dtype=torch.int64, device="cuda")
        tiles = torch.zeros((len(imgs), 32, 32), dtype=torch.uint8, device="cuda")
        _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), len(imgs), 256, 256, 256, 65536, out.data_ptr(),
                                           tiles.data_ptr(), 0, None), "tiles")
    finally:
        L.cbh_set_tuning(b"hash_mfma", 2)
    assert (got == want).all()
    t = tiles.cpu().numpy()
    for i in range(0, len(imgs), 7):
        assert (t[i] == orc.tile32(imgs[i])).all()
    assert (gpu.dct_hash64_batch(imgs) == want).all()


def test_register_streaming_kernel(gpu, orc):
    """k_blur_area_regs (blur input straight from global memory) at the geometries it accepts --
    widths 32..2048: multiples of 8 with aligned strides (aligned 8-byte loads) and any width / stride / base address
    (the GEN form) -- in strips of 3 and 8 steps, all three blur sizes,
    integer and fractional resize ratios, heights around the step and strip boundaries: hashes and tiles equal the
    oracle and the LDS-staged band kernel k_blur_area ("hash_stream" 0) that small batches take"""
    from cbird_amd import _lib
    import torch

    L = _lib.lib()
    rng = np.random.default_rng(99)
    geos = [(640, 480), (400, 300), (600, 450), (1024, 768), (1920, 1080), (2048, 96), (512, 512), (64, 64), (72, 56),
            (96, 120), (128, 100), (200, 57), (256, 250), (8, 200), (2040, 33), (320, 41), (320, 42), (320, 43),
            (1000, 1000), (264, 136)]
    # the any-width / any-alignment form (GEN): every residue of w mod 8 (the last lane owns 1..7 real pixels; for
    # residues <= 3 the lane before it takes its right halo from the row's last dword), odd strides and an odd base
    gen = [(641, 480), (533, 400), (401, 301), (33, 64), (34, 40), (35, 33), (36, 50), (37, 47), (38, 32), (39, 90),
           (43, 35), (2047, 40), (2041, 37), (1023, 100), (999, 77), (333, 500), (1366, 768), (640, 480), (256, 255)]
    try:
        for gi, (w, h) in enumerate(geos + gen):
            n = 3
            unaligned = gi >= len(geos)
            unit = 1 if unaligned else 8
            row_stride = w + unit * int(rng.integers(0, 3 if not unaligned else 6))
            img_stride = h * row_stride + unit * int(rng.integers(0, 5))
            base = int(rng.integers(1, 8)) if unaligned else 0
            buf = rng.integers(0, 256, (n, img_stride), dtype=np.uint8)
            if (w, h) == (640, 480):  # flat and extreme images too
                buf[0] = 0
                buf[1] = 255
            imgs = np.stack([buf[i, : h * row_stride].reshape(h, row_stride)[:, :w] for i in range(n)])
            want = orc.dcthash64_batch(np.ascontiguousarray(imgs))
            dflat = torch.zeros(buf.size + 16, dtype=torch.uint8, device="cuda")
            dflat[base : base + buf.size] = torch.from_numpy(buf.reshape(-1)).cuda()
            d = dflat[base:]
            # fuse 2: the whole-image form with the vertical INTER_AREA pass and the tile inside the kernel
            # (k_blur_area_regs<.., FUSE> + k_tiles_hash), forced for any batch size
            for regs, steps, fuse in ((1, 3, 0), (1, 8, 0), (1, 3, 2), (0, 0, 0)):
                L.cbh_set_tuning(b"hash_stream", steps)
                L.cbh_set_tuning(b"hash_fuse", fuse)
                out = torch.zeros(n, dtype=torch.int64, device="cuda")
                tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
                _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, row_stride, img_stride, out.data_ptr(),
                                                   tiles.data_ptr(), 0, None), "tiles")
                got = out.cpu().numpy().view(np.uint64)
                t = tiles.cpu().numpy()
                for i in range(n):
                    assert (t[i] == orc.tile32(np.ascontiguousarray(imgs[i]))).all(), (w, h, regs, steps, fuse, i)
                assert (got == want).all(), (w, h, regs, steps, fuse)
    finally:
        L.cbh_set_tuning(b"hash_stream", 1)
        L.cbh_set_tuning(b"hash_fuse", 1)


def test_hash_random_geometries_and_strides(gpu, orc):
    """Random widths/heights (every blur kernel size, widths around the 8-pixel lane groups and the 2048-column
    workgroups, integer and fractional resize ratios) and padded row/image strides, on the band kernel k_blur_area
    ("hash_stream" 0) and on strips of 3 and 8 steps: bit-exact against the oracle, hashes and 32x32 tiles."""
    from cbird_amd import _lib
    import torch

    L = _lib.lib()
    rng = np.random.default_rng(4321)
    geos = [(256, 256), (40, 36), (63, 65), (64, 96), (127, 129), (130, 128), (255, 257), (264, 100), (1000, 37), (2047, 33),
            (2049, 40), (2056, 34), (4100, 64), (96, 4097), (2304, 1728), (8190, 33), (33, 8192),
            # cv::resize's scale = 1/(32/w) misses w/32 by an ulp: 32*49 and 32*93 leave the integer path,
            # 3885 gets a different weight table (oracle: cv_resize_scale)
            (1568, 1568), (1568, 64), (64, 2976), (3885, 33),
            # fused kernel: 1 / 2 / 4 column workgroups, widths around their limits, integer ratios
            (512, 40), (2048, 64), (2050, 33), (4096, 35), (4097, 32), (6000, 48), (8192, 33), (1024, 1024), (4064, 96)]
    geos += [(int(rng.integers(32, 700)), int(rng.integers(32, 700))) for _ in range(10)]
    try:
        for (w, h) in geos:
            n = 3
            pad_x, pad_img = int(rng.integers(0, 9)), int(rng.integers(0, 50))
            if (w, h) == (256, 256):
                pad_x, pad_img = 3, 5  # 256x256 with unaligned rows: not the k_dcthash_256 layout
            buf = rng.integers(0, 256, (n, h * (w + pad_x) + pad_img), dtype=np.uint8)
            imgs = np.stack([buf[i, : h * (w + pad_x)].reshape(h, w + pad_x)[:, :w] for i in range(n)])
            want = orc.dcthash64_batch(np.ascontiguousarray(imgs))
            d = torch.from_numpy(buf).cuda()
            # blur + area kernel per 16-row band (0) or walking down strips of 3 / 8 steps (forced: the automatic choice
            # needs thousands of images)
            for fast, fused, stream_steps in ((1, 1, 0), (1, 1, 3), (1, 1, 8)):
                L.cbh_set_tuning(b"hash_stream", stream_steps)
                out = torch.zeros(n, dtype=torch.int64, device="cuda")
                tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
                _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, w + pad_x, buf.shape[1], out.data_ptr(),
                                                   tiles.data_ptr(), 0, None), "tiles")
                got = out.cpu().numpy().view(np.uint64)
                t = tiles.cpu().numpy()
                for i in range(n):
                    assert (t[i] == orc.tile32(np.ascontiguousarray(imgs[i]))).all(), (w, h, fast, fused, stream_steps, i)
                assert (got == want).all(), (w, h, fast, fused, stream_steps)
    finally:
        L.cbh_set_tuning(b"hash_stream", 1)


def test_band_area_kernel_geometries_strides_and_ragged_batches(gpu, orc, band_area):
    """k_band_area (matrix-core blur + four-row area walks) at the edges of what it accepts: widths 64 .. 960 with every
    tile count and both kinds of last strip, cells of 2 .. 30 columns, heights that are not multiples of the four-row step
    and barely above the blur's reach, batches that are not multiples of the four images of a wave, padded row and image
    strides, odd base addresses -- hashes AND 32 x 32 tiles equal the oracle."""
    if band_area != "band_area":
        pytest.skip("the new kernel only")
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(2025)
    geos = [(64, 301), (65, 33), (66, 32), (79, 129), (96, 97), (100, 35), (127, 200), (160, 121), (200, 150), (239, 37),
            (241, 241), (300, 200), (333, 250), (400, 300), (401, 299), (479, 361), (480, 270), (533, 400), (600, 401),
            (640, 481), (641, 480), (700, 99), (720, 405), (799, 601), (854, 480), (900, 34), (959, 540), (960, 541), (961, 100), (1000, 40), (1280, 721), (1366, 70), (1601, 99), (1919, 50), (1920, 1081), (1921, 40)]
    # integer ratios on both axes (round 6: the kernel's block-sum walk): cells of 2 .. 60 columns and 1 .. 34 rows, every
    # rows-per-lane variant (16 / 11 cells per strip up to 672 columns, 8 up to 960, 4 up to 1920)
    geos += [(128, 160), (160, 128), (256, 128), (64, 288), (320, 96), (480, 64), (512, 512), (640, 480), (672, 96),
             (704, 32), (960, 64), (1024, 768), (1280, 704), (1600, 32), (1920, 1088)]
    for gi, (w, h) in enumerate(geos):
        n = int(rng.integers(1, 10))
        row_stride = w + int(rng.integers(0, 7))
        img_stride = h * row_stride + int(rng.integers(0, 33))
        base = int(rng.integers(0, 5))
        buf = rng.integers(0, 256, (n, img_stride), dtype=np.uint8)
        if gi % 5 == 0:  # smooth content too: the tiles then sit near rounding boundaries less often than noise does
            ramp = np.linspace(0, 230, row_stride, dtype=np.float32)
            buf[:, : h * row_stride] = (buf[:, : h * row_stride].reshape(n, h, row_stride) // 8 + ramp[None, None, :]).astype(
                np.uint8).reshape(n, -1)
        if gi == 3:
            buf[0] = 0
            buf[-1] = 255
        imgs = np.stack([buf[i, : h * row_stride].reshape(h, row_stride)[:, :w] for i in range(n)])
        want = orc.dcthash64_batch(np.ascontiguousarray(imgs))
        dflat = torch.zeros(buf.size + 16, dtype=torch.uint8, device="cuda")
        dflat[base: base + buf.size] = torch.from_numpy(buf.reshape(-1)).cuda()
        d = dflat[base:]
        out = torch.zeros(n, dtype=torch.int64, device="cuda")
        tiles = torch.zeros((n, 32, 32), dtype=torch.uint8, device="cuda")
        _lib.check(L.cbh_dcthash_tiles_dev(d.data_ptr(), n, w, h, row_stride, img_stride, out.data_ptr(), tiles.data_ptr(), 0,
                                           None), "tiles")
        t = tiles.cpu().numpy()
        for i in range(n):
            assert (t[i] == orc.tile32(np.ascontiguousarray(imgs[i]))).all(), (w, h, i)
        assert (out.cpu().numpy().view(np.uint64) == want).all(), (w, h)


@pytest.mark.gpu
def test_band_area_row_bands_agree_across_batch_sizes(gpu, orc):
    """k_band_area splits a (group of four images, strip) into 1, 2, 4 or 8 row bands by how many waves the batch makes
    (dcthash.hip, launch_dcthash): the same images hashed in one large batch (one band), in pieces that take 2 and 4
    bands, and in small pieces (8 bands) give the same hashes, and a sample of them equals the oracle.  (Letterbox views on
    the same kernel: tests/test_prestage.py, test_autocropped_hash_uses_the_parent_border.)"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    for (w, h, n) in ((200, 150, 11000), (1280, 45, 2600), (640, 480, 1400), (1024, 96, 3000)):
        g = torch.Generator(device="cuda").manual_seed(w * 7 + h)
        imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda", generator=g)
        imgs[::3] //= 7  # darker, smoother images too
        whole = torch.zeros(n, dtype=torch.int64, device="cuda")
        _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), n, w, h, w, w * h, whole.data_ptr(), 0, None), "whole")
        for piece in (n // 2 + 1, n // 4 + 3, 1000, 61):
            part = torch.zeros(n, dtype=torch.int64, device="cuda")
            for i0 in range(0, n if piece > 100 else 10 * piece, piece):
                m = min(piece, n - i0)
                _lib.check(L.cbh_dcthash_batch_dev(imgs[i0:].data_ptr(), m, w, h, w, w * h, part[i0:].data_ptr(), 0, None),
                           "piece")
            k = n if piece > 100 else 10 * piece
            assert torch.equal(part[:k], whole[:k]), (w, h, piece)
        sample = imgs[:48].cpu().numpy()
        assert (whole[:48].cpu().numpy().view(np.uint64) == orc.dcthash64_batch(np.ascontiguousarray(sample))).all(), (w, h)
