"""GPU parity suite for the Hamming find path (k_hamm64_scan + record sort/select) through the
C-ABI, against the oracle and the reference-generated golden vectors.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_cases, load_golden

# every test runs on both scan kernels (conftest.scan_path): k_hamm64_mfma forced, and k_hamm64_scan
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("scan_path")]


def _matches(ms):
    return [(m.mediaId, m.score) for m in ms]


@pytest.mark.parametrize("name", golden_cases())
def test_find_matches_reference_golden(gpu, name):
    """DctHashIndex::find == the real VP-tree's answers (canonical (score,id) order)."""
    g = load_golden(name)
    idx = gpu.DctHashIndex()
    idx.load(g["hashes"], g["ids"])
    q = g["queries"]
    for dht in range(1, 9):
        offs, gid, gd = g[f"offs_{dht}"], g[f"ids_{dht}"], g[f"dist_{dht}"]
        gi, gs, gc = idx.find_batch(q, dht, 16)
        for j in range(len(q)):
            a, b = int(offs[j]), int(offs[j + 1])
            assert gc[j] == b - a, (name, dht, j)
            m = min(16, b - a)
            assert gi[j, :m].tolist() == gid[a:a + m].tolist(), (name, dht, j)
            assert gs[j, :m].tolist() == gd[a:a + m].tolist(), (name, dht, j)
            assert (gi[j, m:] == 0).all()
    # the single-needle entry point, every needle of the small case at the default threshold
    if name == "vptree_n4096.npz":
        p = gpu.SearchParams()
        offs, gid, gd = g["offs_5"], g["ids_5"], g["dist_5"]
        for j, t in enumerate(q.tolist()):
            if t == 0:
                continue
            got = _matches(idx.find(gpu.Media(id=0, dctHash=t), p))
            a, b = int(offs[j]), int(offs[j + 1])
            assert got == list(zip(gid[a:b].tolist(), gd[a:b].tolist()))


@pytest.mark.parametrize("n,nq,seed", [(1, 1, 1), (7, 3, 2), (2047, 9, 3), (2048, 8, 4), (2049, 17, 5),
                                       (10000, 1000, 6), (70001, 4099, 7)])
def test_find_batch_vs_oracle_ragged_sizes(gpu, orc, n, nq, seed):
    from cbird_amd import synth

    h, ids = synth.make_hashes(n, seed=seed, planted_frac=0.2)
    rng = np.random.default_rng(seed)
    q = h[rng.integers(0, n, nq)].copy()
    q[::5] ^= np.uint64(1) << np.uint64(17)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    for dht in (1, 2, 5, 6, 8, 13):
        gi, gs, gc = idx.find_batch(q, dht, 6)
        wi, ws, wc = orc.find64_batch(h, ids, q, dht, 6)
        assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), (n, nq, dht)


def test_random_shapes_and_thresholds(gpu, orc, cases=24):
    """Tile-padding edges of the matrix-core kernels (32-row tiles, 64/96-needle groups, 192-needle scratch
    padding) and every threshold class, on random shapes; nulls on both sides."""
    from cbird_amd import synth

    rng = np.random.default_rng(2024)
    for case in range(cases):
        n = int(rng.integers(1, 6000))
        nq = int(rng.integers(1, 800))
        dht = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 17, 32, 33, 64, 65]))
        h, ids = synth.make_hashes(n, seed=1000 + case, planted_frac=0.3)
        q = h[rng.integers(0, n, nq)].copy()
        flip = rng.integers(0, 64, nq).astype(np.uint64)
        q[::3] ^= (np.uint64(1) << flip[::3])
        q[rng.integers(0, nq, max(1, nq // 50))] = 0  # null needles never match
        z = rng.integers(0, n, max(1, n // 40))
        h[z] = 0
        ids[z] = 0  # removed slots
        idx = gpu.DctHashIndex()
        idx.load(h, ids)
        gi, gs, gc = idx.find_batch(q, dht, 5)
        wi, ws, wc = orc.find64_batch(h, ids, q, dht, 5)
        assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), (case, n, nq, dht)


def test_distance_extremes_all_thresholds(gpu, orc):
    """Distances 0 and 64 (complement) and the neighbourhood of every field boundary of the packed
    accumulators: identical, complemented and k-bit-flipped copies, every threshold 1..65."""
    rng = np.random.default_rng(77)
    base = rng.integers(1, 1 << 63, 600, dtype=np.uint64) << np.uint64(1)
    h = np.concatenate([base, ~base & ~np.uint64(1)])  # bit 0 kept clear like real dct hashes
    ids = np.arange(1, len(h) + 1, dtype=np.uint32)
    q = [base[:40], ~base[:40]]
    for k in (1, 2, 31, 32, 33, 62, 63):
        m = np.zeros(40, np.uint64)
        for j in range(40):
            bits = rng.choice(64, k, replace=False)
            m[j] = np.bitwise_or.reduce(np.uint64(1) << bits.astype(np.uint64))
        q.append(base[40:80] ^ m)
    q = np.concatenate(q)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    for dht in list(range(1, 9)) + [16, 31, 32, 33, 34, 48, 63, 64, 65]:
        gi, gs, gc = idx.find_batch(q, dht, 4)
        wi, ws, wc = orc.find64_batch(h, ids, q, dht, 4)
        assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), dht


def test_thresholds_full_range(gpu, orc):
    """dht valid range is 0..65 (src/index.cpp:77); 65 matches everything, <=0 nothing."""
    from cbird_amd import synth

    h, ids = synth.make_hashes(3000, seed=21)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    q = h[:5]
    for dht in (0, 1, 20, 33, 64, 65):
        gi, gs, gc = idx.find_batch(q, dht, 4)
        wi, ws, wc = orc.find64_batch(h, ids, q, dht, 4)
        assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), dht
    assert (idx.find_batch(q, 65, 1)[2] == 3000).all()
    m = idx.find(gpu.Media(dctHash=int(h[0])), gpu.SearchParams(dctThresh=65))
    assert len(m) == 3000 and _matches(m) == list(zip(*[x.tolist() for x in orc.find64(h, ids, h[0], 65)]))


def test_null_needle_empty_index_removed_slots(gpu, orc):
    idx = gpu.DctHashIndex()
    assert not idx.isLoaded() and idx.count() == 0 and idx.memoryUsage() == 0
    idx.load([], [])
    assert idx.isLoaded() and idx.count() == 0
    with pytest.warns(UserWarning):
        assert idx.find(gpu.Media(id=1, dctHash=2), gpu.SearchParams()) == []  # empty/null tree
    idx2 = gpu.DctHashIndex()
    h = np.array([0, 2, 6, 0], np.uint64)
    ids = np.array([0, 7, 8, 9], np.uint32)
    idx2.load(h, ids)
    with pytest.warns(UserWarning):
        assert idx2.find(gpu.Media(id=1, dctHash=0), gpu.SearchParams()) == []  # null needle
    assert _matches(idx2.find(gpu.Media(dctHash=2), gpu.SearchParams(dctThresh=3))) == [(7, 0), (8, 1), (9, 1)]
    gi, gs, gc = idx2.find_batch(np.array([0, 2], np.uint64), 3, 4)
    assert gc.tolist() == [0, 3]
    assert idx2.mediaIds() == {7, 8}  # hash != 0 only (dcthashindex.cpp:129-133)


def test_add_remove_slice_like_testindexbase(gpu, orc):
    """unit/testindexbase.cpp:148-218 (remove 3, verify gone, re-add, same results) and
    unit/testdcthashindex.cpp:27-30 (memoryUsage == 12 B * count)."""
    from cbird_amd import synth

    h, ids = synth.make_hashes(5000, seed=31, planted_frac=0.3)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    assert idx.memoryUsage() == (8 + 4) * idx.count() == 12 * 5000
    p = gpu.SearchParams(dctThresh=9)
    # pick 3 items that have matches
    _, _, cnt = idx.find_batch(h, 9, 1)
    victims = [int(ids[i]) for i in np.nonzero(cnt > 1)[0][:3]]
    before = {v: _matches(idx.find(gpu.Media(id=v, dctHash=int(h[v - 1])), p)) for v in victims}
    idx.remove(victims)
    assert idx.count() == 5000  # slots are nulled, not compacted
    h2, ids2 = idx.download()
    assert all(ids2[v - 1] == 0 and h2[v - 1] == 0 for v in victims)
    for v in victims:
        got = _matches(idx.find(gpu.Media(id=v, dctHash=int(h[v - 1])), p))
        assert all(i not in victims for i, _ in got)
        oi, od = orc.find64(h2, ids2, h[v - 1], 9)
        assert got == list(zip(oi.tolist(), od.tolist()))
    idx.add([gpu.Media(id=v, dctHash=int(h[v - 1])) for v in victims])
    assert idx.count() == 5003 and idx.memoryUsage() == 12 * 5003
    for v in victims:
        assert _matches(idx.find(gpu.Media(id=v, dctHash=int(h[v - 1])), p)) == before[v]
    # slice(): subset index, original order, caller-owned
    want = set(int(x) for x in ids[100:400:3])
    sub = idx.slice(want)
    sh, si = sub.download()
    keep = np.isin(ids2, list(want))
    assert si.tolist() == ids2[keep].tolist() + [v for v in victims if v in want]
    for t in h[100:130].tolist():
        oi, od = orc.find64(sh, si, t, 9)
        assert _matches(sub.find(gpu.Media(dctHash=t), p)) == list(zip(oi.tolist(), od.tolist()))


def test_record_buffer_grows_instead_of_truncating(gpu, orc):
    from cbird_amd import synth

    h, ids = synth.make_hashes(6000, seed=41)
    h[:] = h[0]  # every pair matches at distance 0: 6000*64 records for 64 needles
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    idx.set_record_capacity(1000)
    gi, gs, gc = idx.find_batch(h[:64], 1, 3)
    assert (gc == 6000).all() and (gi == np.array([1, 2, 3], np.uint32)).all() and (gs == 0).all()
    m = idx.find(gpu.Media(dctHash=int(h[0])), gpu.SearchParams(dctThresh=1))
    assert [x.mediaId for x in m] == list(range(1, 6001))


def test_raw_scan_records_and_concurrent_readers(gpu, orc):
    """cbh_idx64_scan_dev record format + thread-safety of find() (QThreadPool readers)."""
    import threading

    import torch

    from cbird_amd import _lib, synth

    L = _lib.lib()
    h, ids = synth.make_hashes(30000, seed=51, planted_frac=0.2)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    q = h[:3000]
    dq = torch.from_numpy(q.view(np.int64)).cuda()
    cap = 1 << 16
    drec = torch.zeros(cap, dtype=torch.int64, device="cuda")
    dtot = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(L.cbh_idx64_scan_dev(idx.handle, dq.data_ptr(), len(q), 7, drec.data_ptr(), cap,
                                    dtot.data_ptr(), None), "scan")
    tot = int(dtot.item())
    rec = drec[:tot].cpu().numpy().view(np.uint64)
    got = sorted((int(r >> 39), int((r >> 32) & 0x7F), int(r & 0xFFFFFFFF)) for r in rec.tolist())
    want = []
    for j, t in enumerate(q.tolist()):
        oi, od = orc.find64(h, ids, t, 7)
        want += [(j, int(d), int(i)) for i, d in zip(oi, od)]
    assert got == sorted(want)

    p = gpu.SearchParams(dctThresh=7)
    errs = []

    def worker(lo, hi):
        try:
            for j in range(lo, hi):
                m = _matches(idx.find(gpu.Media(dctHash=int(q[j])), p))
                oi, od = orc.find64(h, ids, q[j], 7)
                if m != list(zip(oi.tolist(), od.tolist())):
                    errs.append(j)
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(k * 40, k * 40 + 40)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs


def test_all_pairs_full_size_properties(gpu, orc):
    """BASELINE config 2 size (1M x 1M): properties that do not need a CPU all-pairs run --
    (i) every non-null needle that is in the index finds itself at distance 0, so counts >= 1;
    (ii) the pair relation is symmetric: sum of counts == number of records, and the multiset of
         (needle,match) pairs is closed under swapping (checked through an order-free checksum);
    (iii) raising the threshold only adds matches; (iv) a 2k-needle sample equals the oracle."""
    import torch

    from cbird_amd import _lib, synth

    L = _lib.lib()
    n = 1_000_000
    h, ids = synth.make_hashes(n, seed=1234)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    dq = torch.from_numpy(h.view(np.int64)).cuda()
    k = 4
    dout = torch.empty((n, k, 2), dtype=torch.int32, device="cuda")
    dcnt = torch.empty(n, dtype=torch.int32, device="cuda")
    prev = None
    for dht in (2, 5):
        tot = C.c_uint64(0)
        _lib.check(L.cbh_idx64_find_batch_dev(idx.handle, dq.data_ptr(), n, dht, k, dout.data_ptr(),
                                              dcnt.data_ptr(), C.byref(tot), None), "find_batch_dev")
        cnt = dcnt.cpu().numpy().astype(np.int64)
        out = dout.cpu().numpy()
        assert cnt.min() >= 1 and cnt.sum() == tot.value
        assert (out[:, 0, 1] == 0).all()  # best match is at distance 0 (itself or an exact duplicate)
        # symmetry checksum over the pairs that fit in k slots for needles with count <= k
        small = cnt <= k
        a = np.repeat(np.arange(1, n + 1, dtype=np.uint64)[small], cnt[small])
        mask = (np.arange(k)[None, :] < cnt[small][:, None])
        b = out[small][:, :, 0].astype(np.uint64)[mask]
        assert len(a) == len(b)
        big_ids = set((np.nonzero(~small)[0] + 1).tolist())
        keep = np.array([x not in big_ids for x in b.tolist()]) if big_ids else np.ones(len(b), bool)
        f = lambda x, y: int(((x * np.uint64(0x9E3779B97F4A7C15)) ^ (y * np.uint64(0xC2B2AE3D27D4EB4F))).sum())
        assert f(a[keep], b[keep]) == f(b[keep], a[keep])
        if prev is not None:
            assert (cnt >= prev).all()
        prev = cnt
        sample = np.random.default_rng(dht).choice(n, 2000, replace=False)
        wi, ws, wc = orc.find64_batch(h, ids, h[sample], dht, k)
        assert (cnt[sample] == wc).all()
        assert (out[sample][:, :, 0].astype(np.uint32) == wi).all() and (out[sample][:, :, 1] == ws).all()


def test_prefilter_pair_boundaries_and_low_word_collisions(gpu, orc):
    """The 32-bit prefilter (the "mfma_pre" leg runs it at every threshold) steps two needle pairs (128 needles) at a time, four 6-bit fields
    per accumulator with the top field flagging through the exponent: needle counts around every pair / step /
    chunk boundary, and a haystack full of entries whose LOW words are within the threshold of a needle's while the
    high words are not (false positives in every field, including the carrying one, that the 64-bit re-check must
    drop) next to true matches."""
    rng = np.random.default_rng(4242)
    n = 3000
    lo = rng.integers(0, 1 << 32, 40, dtype=np.uint64) & ~np.uint64(1)
    h = (rng.integers(0, 1 << 32, n, dtype=np.uint64) << np.uint64(32)) | lo[rng.integers(0, 40, n)]
    flips = rng.integers(0, 4, n)
    for i in range(n):  # 0..3 flipped low-word bits
        for b in rng.choice(np.arange(1, 32), int(flips[i]), replace=False):
            h[i] ^= np.uint64(1) << np.uint64(b)
    h[h == 0] = 2
    ids = np.arange(1, n + 1, dtype=np.uint32)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    for nq in (1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 257, 16383, 16385, 32769):
        q = h[rng.integers(0, n, nq)].copy()
        q[::2] ^= rng.integers(0, 1 << 32, len(q[::2]), dtype=np.uint64) << np.uint64(32)  # other high word
        q[1::7] ^= np.uint64(1) << np.uint64(40)                                          # one high bit
        for dht in ((1, 2, 3, 4) if nq < 1000 else (2, 4)):
            gi, gs, gc = idx.find_batch(q, dht, 3)
            wi, ws, wc = orc.find64_batch(h, ids, q, dht, 3)
            assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), (nq, dht)


def test_fold_prefilter_and_deferred_recheck(gpu, orc):
    """Round 5's prefilter (thresholds <= 6) compares lo ^ hi and defers the re-check: candidates found in up to four
    lanes of a group are parked, listed and drained 64 at a time; denser groups go through the queue path after the
    pending list has been drained.  The haystack mixes (a) entries whose FOLD equals a needle's while the hash does not
    (x ^ (r | r << 32): false positives in every field, the carrying one included, that the 64-bit re-check drops unless
    2 popc(r) is under the threshold), (b) true neighbours at 0..7 flipped bits, (c) 300 exact duplicates of one entry
    (every lane of a group holds a flag: the dense path, entered with descriptors pending) and (d) random entries;
    needle counts straddle the pair / step / chunk boundaries."""
    rng = np.random.default_rng(555)
    base = (rng.integers(1, 1 << 62, 48, dtype=np.uint64) << np.uint64(1))
    parts = [base]
    r = rng.integers(0, 1 << 32, 1500, dtype=np.uint64)
    few = rng.random(1500) < 0.5  # half of them with only 0..3 bits set in r: 64-bit distance 0, 2, 4, 6
    for i in np.nonzero(few)[0]:
        r[i] = np.bitwise_or.reduce(np.uint64(1) << rng.choice(np.arange(1, 32), int(rng.integers(0, 4)), replace=False)
                                    .astype(np.uint64)) if rng.random() < 0.9 else np.uint64(0)
    parts.append(base[rng.integers(0, 48, 1500)] ^ (r | (r << np.uint64(32))))
    near = base[rng.integers(0, 48, 1200)].copy()
    for i in range(len(near)):
        for b in rng.choice(np.arange(1, 64), int(rng.integers(0, 8)), replace=False):
            near[i] ^= np.uint64(1) << np.uint64(b)
    parts.append(near)
    parts.append(np.full(300, base[7], np.uint64))
    parts.append(rng.integers(1, 1 << 63, 2000, dtype=np.uint64) << np.uint64(1))
    h = np.concatenate(parts)
    h[h == 0] = 2
    h = h[rng.permutation(len(h))]
    ids = np.arange(1, len(h) + 1, dtype=np.uint32)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    for nq in (1, 64, 65, 128, 129, 193, 511, 4097, 16385):
        q = h[rng.integers(0, len(h), nq)].copy()
        q[::3] ^= np.uint64(1) << rng.integers(1, 64, len(q[::3])).astype(np.uint64)
        q[5::11] = 0  # null needles
        for dht in ((1, 3, 4, 5, 6, 7) if nq < 1000 else (5, 6)):
            gi, gs, gc = idx.find_batch(q, dht, 4)
            wi, ws, wc = orc.find64_batch(h, ids, q, dht, 4)
            assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), (nq, dht)


def test_prefilter_beyond_its_range(gpu, orc, scan_path):
    """The prefilter kernel forced for thresholds 7..16 ("scan_mfma_pre_max"): on random hashes one pair in 4000 .. 2 is a
    candidate, so every group parks several hit lanes, in several rounds -- the re-check must still be exact."""
    if scan_path == "valu":
        pytest.skip("matrix-core kernels only")
    from cbird_amd import _lib, synth

    L = _lib.lib()
    h, ids = synth.make_hashes(12000, seed=808, planted_frac=0.3, max_dist=14)
    q = h[:1500].copy()
    q[::4] ^= np.uint64(0x10100)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    L.cbh_set_tuning(b"scan_mfma_pre_max", 16)
    try:
        for dht in (7, 8, 9, 10, 12, 16):
            gi, gs, gc = idx.find_batch(q, dht, 5)
            wi, ws, wc = orc.find64_batch(h, ids, q, dht, 5)
            assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), dht
    finally:
        L.cbh_set_tuning(b"scan_mfma_pre_max", {"mfma_pre": 32, "mfma_full": 0}.get(scan_path, -1))


def test_bucketed_join_equals_the_scan_and_steps_aside_on_skewed_data(gpu, orc, scan_path):
    """hamm64_join.hip ("scan_mfma" 3: the join where its candidate count says it is cheaper, 4: forced): on 400 000 x
    400 000 hashes the thresholds 1..8 give the cut lists and record totals of the matrix-core scan -- on near-uniform
    hashes and with half the slots exact copies in clusters (the join runs: "scan_joins" counts it), and with half the
    slots sharing their low 16 bits (4 x 10^10 candidate pairs on one chunk value: the sampled count sends the call back to
    the scan); removed slots and null needles on both sides."""
    import torch

    from cbird_amd import _lib, synth

    if scan_path != "join":
        pytest.skip("sets the kernel family itself: once is enough")
    L = _lib.lib()
    n, k = 400_000, 6
    rng = np.random.default_rng(77)
    base, ids = synth.make_hashes(n, seed=4321, planted_frac=0.2)
    dup = base.copy()
    dup[rng.permutation(n)[: n // 2]] = dup[rng.integers(0, n, n // 2)]
    skew = base.copy()
    half = rng.permutation(n)[: n // 2]
    skew[half] = (skew[half] & ~np.uint64(0xFFFF)) | np.uint64(0x1234)

    def joins():
        v = C.c_longlong(0)
        assert L.cbh_get_tuning(b"scan_joins", C.byref(v)) == 0
        return v.value

    try:
        for name, h, modes, thresholds in (("uniform", base, (2, 3, 4), (1, 3, 5, 6, 7, 8)), ("dup50", dup, (2, 3), (2, 4, 6, 8)),
                                           ("skew", skew, (2, 3), (1, 4, 8))):
            idz = ids.copy()
            idz[rng.integers(0, n, 500)] = 0  # removed slots
            q = h.copy()
            q[rng.integers(0, n, 300)] = 0  # null needles
            idx = gpu.DctHashIndex()
            idx.load(h, idz)
            dq = torch.from_numpy(q.view(np.int64)).cuda()
            outs = {}
            for mode in modes:
                L.cbh_set_tuning(b"scan_mfma", mode)
                j0 = joins()
                for dht in thresholds:
                    dout = torch.empty((n, k, 2), dtype=torch.int32, device="cuda")
                    dcnt = torch.empty(n, dtype=torch.int32, device="cuda")
                    tot = C.c_uint64(0)
                    _lib.check(L.cbh_idx64_find_batch_dev(idx.handle, dq.data_ptr(), n, dht, k, dout.data_ptr(),
                                                          dcnt.data_ptr(), C.byref(tot), None), "find_batch_dev")
                    outs.setdefault(dht, {})[mode] = (int(tot.value), dcnt.cpu().numpy(), dout.cpu().numpy())
                ran = joins() - j0
                if mode == 2:
                    assert ran == 0
                elif name == "skew":
                    assert ran < len(thresholds), "the join took a call with 4e10 candidate pairs on one value"
                elif mode == 4:
                    assert ran == len(thresholds), (name, mode, ran)
                else:  # (at this size the largest threshold sits near the cost model's break-even)
                    assert ran >= len(thresholds) - 1, (name, mode, ran)
            for dht, by in outs.items():
                ref = by[2]
                for mode, got in by.items():
                    assert got[0] == ref[0] and (got[1] == ref[1]).all(), (name, dht, mode)
                    m = np.arange(k)[None, :] < np.minimum(ref[1], k)[:, None]
                    assert (got[2][m] == ref[2][m]).all(), (name, dht, mode)
            del idx
    finally:
        L.cbh_set_tuning(b"scan_mfma", 1)
