"""One index over several shards inside ONE process, behind the C-ABI (cbh_idx64_create_sharded, sharded.hip).

cbird registers each Index once and fans find() out from its own thread pool (src/engine.cpp:38-45,
src/database.cpp:1400-1432), so the drop-in shards inside the handle.  This pool has one GPU per box: the shards are
logical ones on device 0 (their own streams, device-to-device copies as the exchange), and the inter-device transport
-- one grouped ncclAllGather of the per-device blocks, librccl called directly -- runs on a one-rank communicator
("shard_force_rccl").  Everything a sharded handle returns must equal what the one-device index returns, which the
suites of test_gpu_hamm / test_fdct / test_video / test_database pin to the oracle and to the reference's golden
vectors: those suites are re-run here, unchanged, with every index they create sharded."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import test_cvfeatures as TC
import test_database as TD
import test_fdct as TF
import test_gpu_hamm as TH
import test_video as TV
from conftest import load_golden


def _shapes():
    """(name, (device_mask, shards_per_device), exchange, force_rccl, fault_rccl).  On a one-GPU box: logical shards
    on device 0.  On a box with several GPUs (tools/first_contact.sh) two more shapes join in, through the same suites:
    every device x 1 shard with both exchanges, and two devices x 2 shards."""
    vdev = int(os.environ.get("CBH_VDEV", "0"))
    if vdev > 1:
        # tests/shim/vdev.c is preloaded (test_virtual_devices.py): the one GPU answers to `vdev` ordinals.  Only the
        # shapes that need several ordinals; the collective shape with librccl treated as absent (RCCL refuses two ranks
        # on one physical GPU), i.e. the fall-back to copies between DIFFERENT ordinals.
        full = (1 << vdev) - 1
        return [("alldev_copies", (full, 1), 1, 0, 0), ("alldev_norccl", (full, 1), 0, 0, 1), ("dev2x2", (0x3, 2), 1, 0, 0),
                ("dev3x1_sparse", (0x5 | (1 << (vdev - 1)), 1), 1, 0, 0)]
    shapes = [("shards5", (1, 5), 1, 0, 0), ("rccl3", (1, 3), 0, 1, 0), ("shards2x", (1, 2), 1, 0, 0),
              ("norccl3", (1, 3), 0, 1, 1)]
    try:
        import torch

        ndev = torch.cuda.device_count()  # (counting does not initialise the GPU)
    except Exception:
        ndev = 0
    if ndev > 1:
        full = (1 << ndev) - 1
        shapes += [("alldev_copies", (full, 1), 1, 0, 0), ("alldev_rccl", (full, 1), 0, 0, 0),
                   ("dev2x2", (0x3, 2), 1, 0, 0)]
    return shapes


_SHAPES = {s[0]: s for s in _shapes()}


def _R(name):
    mask, per = _SHAPES[name][1]
    return bin(mask).count("1") * per


@pytest.fixture(params=list(_SHAPES))
def sharded(request, gpu):
    """"shards5": five logical shards, copies only (the default exchange); "rccl3": three, their concatenated block
    through ncclAllGather ("shard_exchange" 0 + "shard_force_rccl"); "shards2x": two shards, copies; "norccl3": as
    rccl3 with librccl treated as absent ("fault_rccl"): the exchange must fall back to copies and say so."""
    from cbird_amd import _lib

    L = _lib.lib()
    name, shape, exchange, force, fault = _SHAPES[request.param]
    _lib.set_default_sharding(shape)
    L.cbh_set_tuning(b"shard_force_rccl", force)
    L.cbh_set_tuning(b"shard_exchange", exchange)
    L.cbh_set_tuning(b"fault_rccl", fault)
    yield request.param
    _lib.set_default_sharding(None)
    L.cbh_set_tuning(b"shard_force_rccl", 0)
    L.cbh_set_tuning(b"shard_exchange", 1)
    L.cbh_set_tuning(b"fault_rccl", 0)


pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("sharded")]


@pytest.fixture(params=["mfma", "valu"])
def scan_path(request, gpu):
    """(overrides conftest's four kernel families for this module: which 64-bit kernel scans a shard is orthogonal to how
    the shards are put together -- the plain-index suites run every family; here the shipped choice and the popcount
    kernel)"""
    from cbird_amd import _lib

    L = _lib.lib()
    L.cbh_set_tuning(b"scan_mfma", 0 if request.param == "valu" else 2)
    yield request.param
    L.cbh_set_tuning(b"scan_mfma", 1)


@pytest.fixture(params=["device"])
def reduce_path(request, gpu):
    """(likewise: where the per-needle reductions of fdct / video run is orthogonal to the sharding; the plain-index suites
    run both, here the device reductions, which are what a sharded handle's batches take)"""
    from cbird_amd import _lib

    _lib.lib().cbh_set_tuning(b"fdct_host_vote", 2)
    _lib.lib().cbh_set_tuning(b"video_host_reduce", 2)
    yield request.param
    _lib.lib().cbh_set_tuning(b"fdct_host_vote", 0)
    _lib.lib().cbh_set_tuning(b"video_host_reduce", 0)


# ---- the one-device suites, every index sharded ---------------------------------------------------------------------
test_find_matches_reference_golden = TH.test_find_matches_reference_golden
test_find_batch_vs_oracle_ragged_sizes = TH.test_find_batch_vs_oracle_ragged_sizes


def test_random_shapes_and_thresholds(gpu, orc, sharded):
    # (a handle that exchanges through ncclAllGather makes a communicator of its own: eight of the 24 shapes there)
    TH.test_random_shapes_and_thresholds(gpu, orc, cases=8 if "rccl" in sharded else 24)


test_distance_extremes_all_thresholds = TH.test_distance_extremes_all_thresholds
test_thresholds_full_range = TH.test_thresholds_full_range
test_null_needle_empty_index_removed_slots = TH.test_null_needle_empty_index_removed_slots
test_add_remove_slice_like_testindexbase = TH.test_add_remove_slice_like_testindexbase
test_record_buffer_grows_instead_of_truncating = TH.test_record_buffer_grows_instead_of_truncating
test_prefilter_pair_boundaries_and_low_word_collisions = TH.test_prefilter_pair_boundaries_and_low_word_collisions
test_fdct_golden_and_oracle = TF.test_gpu_matches_golden_and_oracle
test_fdct_add_remove_findindex_batch = TF.test_gpu_add_remove_findindex_batch
test_fdct_tree_compatible_mode = TF.test_gpu_tree_compatible_mode_equals_real_multileaf_tree
test_video_find_video_and_frame = TV.test_gpu_find_video_and_frame_vs_oracle
test_video_batch_remove_add = TV.test_gpu_video_batch_remove_add
test_video_radix_compatible = TV.test_gpu_radix_compatible_mode_equals_bucket_search
test_similar_behind_the_c_abi = TD.test_similar_behind_the_c_abi_equals_the_oracle
test_cvfeatures_knn_and_find = TC.test_gpu_knn_and_find_vs_oracle
test_cvfeatures_radius_match = TC.test_gpu_radius_match_vs_oracle
vorc = TV.vorc
cvo = TC.cvo


# ---- what only a sharded handle has ---------------------------------------------------------------------------------
def test_shares_follow_the_shard_range_rule_and_the_order_is_global(gpu, sharded):
    from cbird_amd import synth

    R = _R(sharded)
    for n in (0, 1, R - 1, R, 1000, 12345):
        h, ids = synth.make_hashes(max(n, 1), seed=7 + n)
        h, ids = h[:n], ids[:n]
        idx = gpu.DctHashIndex()
        idx.load(h, ids)
        assert idx.shard_count() == R and idx.count() == n and idx.memoryUsage() == 12 * n
        assert idx.shard_counts() == [(s + 1) * n // R - s * n // R for s in range(R)]  # dist.py shard_range
        dh, di = idx.download()
        assert (dh == h).all() and (di == ids).all()
    # add(): appended runs keep the global order whatever shard they land on
    h, ids = synth.make_hashes(5000, seed=99)
    idx = gpu.DctHashIndex()
    idx.load(h[:1000], ids[:1000])
    for a, b in ((1000, 1001), (1001, 1700), (1700, 1700), (1700, 4000), (4000, 5000)):
        idx.add([gpu.Media(id=int(i), dctHash=int(x)) for x, i in zip(h[a:b], ids[a:b])])
    dh, di = idx.download()
    assert (dh == h).all() and (di == ids).all() and sum(idx.shard_counts()) == 5000
    assert max(idx.shard_counts()) - min(idx.shard_counts()) < 2400  # the emptiest shard takes each batch
    assert idx.mediaIds() == set(int(i) for i, x in zip(ids, h) if x)
    st = idx.shard_stats()
    mask = _SHAPES[sharded][1][0]
    assert st.shards == R and st.devices == bin(mask).count("1") and st.device_mask == mask and st.segments >= R


def test_bucketed_join_on_every_shard(gpu, orc, sharded):
    """"scan_mfma" 4 on a sharded handle: every shard answers its slots by the join (hamm64_join.hip), appending into the
    root block like the scans do -- random shapes, thresholds on both sides of 8, removed slots and null needles"""
    from cbird_amd import _lib

    if sharded not in ("shards5", "shards2x", "dev2x2", "alldev_copies"):
        pytest.skip("the copy exchange's shapes: how the records travel is the other tests' subject")
    L = _lib.lib()
    L.cbh_set_tuning(b"scan_mfma", 4)
    try:
        TH.test_random_shapes_and_thresholds(gpu, orc)
        TH.test_null_needle_empty_index_removed_slots(gpu, orc)
    finally:
        L.cbh_set_tuning(b"scan_mfma", 1)


def test_exchange_route_and_overflow_redo_are_what_the_shape_says(gpu, orc, sharded):
    """the counters of cbh_idx64_shard_stats: "rccl3" really goes through ncclAllGather, the others never; a shard
    whose own block overflows is the only one that scans again -- under the copy exchange the shards of the ROOT device
    have no block of their own (they append straight into the root block, no copy into place): when that overflows they
    scan again together"""
    from cbird_amd import _lib, synth

    L = _lib.lib()
    R = _R(sharded)
    ndev = bin(_SHAPES[sharded][1][0]).count("1")
    h, ids = synth.make_hashes(40000, seed=5, planted_frac=0.3)
    # every needle matches the whole of shard 0's share at distance 0: that shard overflows, the others do not
    share0 = 40000 // R
    h[:share0] = h[0]
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    idx.set_record_capacity(4096)
    q = np.concatenate([h[:8], h[share0:share0 + 100]])
    s0 = idx.shard_stats()
    gi, gs, gc = idx.find_batch(q, 3, 7)
    s1 = idx.shard_stats()
    wi, ws, wc = orc.find64_batch(h, ids, q, 3, 7)
    assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all()
    name, (mask, per), exchange, force, fault = _SHAPES[sharded]
    collective_asked = exchange == 0 and (ndev > 1 or force)  # (also when librccl then turns out to be absent)
    again = 1 if collective_asked else per  # shard 0 lives on the root device
    assert s1.scans - s0.scans == R + again and s1.rescans - s0.rescans == again
    if sharded in ("rccl3", "alldev_rccl"):
        assert s1.collectives - s0.collectives == 1 and s1.collective_fallbacks == 0
    else:
        assert s1.collectives == 0
    if sharded in ("norccl3", "alldev_norccl"):  # no communicator: copies, counted and explained
        assert s1.collective_fallbacks - s0.collective_fallbacks == 1
        assert b"RCCL unavailable" in L.cbh_last_error()
    if ndev == 1:
        if collective_asked:
            assert 1 <= s1.local_copies - s0.local_copies <= R
        else:
            assert s1.local_copies == s0.local_copies  # appended in place
        assert s1.peer_copies == 0  # one device: nothing crosses xGMI here
    else:  # needles out to every other device; records back by peer copies unless the collective carried them
        assert s1.peer_copies - s0.peer_copies >= ndev - 1


def test_many_reader_threads_on_one_sharded_handle(gpu, orc, sharded):
    """Index::find from QThreadPool workers (src/database.cpp:1400-1432): the plain find and the caller-combining
    find, 8 threads, one handle; each call leases its own workspace per shard"""
    from cbird_amd import _lib, synth

    L = _lib.lib()
    h, ids = synth.make_hashes(30000, seed=51, planted_frac=0.2)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    q = h[:320]
    p = gpu.SearchParams(dctThresh=7)
    want = []
    for t in q.tolist():
        oi, od = orc.find64(h, ids, t, 7)
        want.append(list(zip(oi.tolist(), od.tolist())))
    errs = []

    def worker(lo, hi, coalesced):
        try:
            buf = (_lib.cbh_match * 4096)()
            n = C.c_size_t(0)
            for j in range(lo, hi):
                if coalesced:
                    _lib.check(L.cbh_idx64_find_coalesced(idx.handle, int(q[j]), 7, buf, 4096, C.byref(n)), "find")
                    m = [(buf[i].id, buf[i].score) for i in range(n.value)]
                else:
                    m = [(x.mediaId, x.score) for x in idx.find(gpu.Media(dctHash=int(q[j])), p)]
                if m != want[j]:
                    errs.append(j)
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    for coalesced in (False, True):
        th = [threading.Thread(target=worker, args=(k * 40, k * 40 + 40, coalesced)) for k in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, (coalesced, errs[:5])


def test_self_join_cache_of_a_sharded_index_and_its_size_limit(gpu, orc, sharded):
    """cbh_idx64_find_coalesced builds its whole-index self-join from the shards (needles = the host mirror of the
    slots in global order); an index whose self-join would not fit is served by combined scans and allocates nothing
    of that size"""
    from cbird_amd import _lib, synth

    L = _lib.lib()
    h, ids = synth.make_hashes(20000, seed=77, planted_frac=0.3)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    buf = (_lib.cbh_match * 4096)()
    n = C.c_size_t(0)
    st = (C.c_uint64 * 5)()
    for rep in range(3):
        for j in range(0, 20000, 7):
            if not h[j]:
                continue
            _lib.check(L.cbh_idx64_find_coalesced(idx.handle, int(h[j]), 5, buf, 4096, C.byref(n)), "find")
            if j % 700 == 0:
                oi, od = orc.find64(h, ids, h[j], 5)
                assert [(buf[i].id, buf[i].score) for i in range(n.value)] == list(zip(oi.tolist(), od.tolist()))
    _lib.check(L.cbh_idx64_coalesce_stats(idx.handle, st), "stats")
    assert st[4] >= 1 and st[1] > 0  # a self-join was built and hit


def test_sharded_slice_is_sharded_and_video_index_takes_the_shape(gpu, sharded):
    from cbird_amd import _lib, synth

    L = _lib.lib()
    h, ids = synth.make_hashes(3000, seed=3)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    sl = idx.slice(ids[100:900].tolist())
    assert sl.shard_count() == idx.shard_count() and sl.count() == 800
    dh, di = sl.download()
    assert (di == ids[100:900]).all() and (dh == h[100:900]).all()
    # the raw shard-local step is a shard's, not the parent's
    assert L.cbh_idx64_scan_dev(idx.handle, 1, 1, 2, 1, 1, 1, None) == _lib.CBH_E_UNSUPPORTED
    assert L.cbh_idx64_shard(idx.handle, idx.shard_count()) is None
    assert L.cbh_idx64_device_mask(idx.handle) == _SHAPES[sharded][1][0]
    # unusable masks are refused outright (no silent narrowing to the devices that exist)
    assert L.cbh_idx64_create_sharded(0, 1) is None
    assert L.cbh_idx64_create_sharded(1 << 30, 1) is None


def test_cvfeatures_index_sharded_by_image(gpu, sharded):
    """cbh_idx256_create_sharded: media stay whole on one shard, runs of 16384 rows; rows_of / download_rows speak
    global row numbers; the knn table (rows, distances, counts) equals the one-device index's"""
    from cbird_amd import _lib

    R = _R(sharded)
    rng = np.random.default_rng(11)
    n_media, per = 160, 700  # 112000 rows: several runs per shard
    rows = rng.integers(0, 256, (n_media * per, 32), dtype=np.uint8)
    rows[per * 3 + 5] = rows[per * 90 + 7]  # equal rows on different shards: the (distance, global row) tie-break
    from cbird_amd.cvfeatures import CvFeaturesIndex
    sh = CvFeaturesIndex()
    _lib.set_default_sharding(None)
    one = CvFeaturesIndex()
    L = _lib.lib()
    for i in range(n_media):
        blk = np.ascontiguousarray(rows[i * per:(i + 1) * per])
        for ix in (sh, one):
            _lib.check(L.cbh_idx256_add(ix._h, i + 1, blk.ctypes.data, per), "add")
    assert L.cbh_idx256_shard_count(sh._h) == R and sh.count() == one.count() == n_media * per
    sr = sh.shard_rows()
    assert sum(sr) == n_media * per and all(x % per == 0 for x in sr) and min(sr) > 0
    got = np.zeros((per * 3, 32), np.uint8)
    _lib.check(L.cbh_idx256_download_rows(sh._h, per * 22 + 13, per * 3, got.ctypes.data), "download")
    assert (got == rows[per * 22 + 13: per * 25 + 13]).all()
    q = rows[rng.integers(0, len(rows), 600)].copy()
    q[::3, 5] ^= 0x11
    a = sh.knn(q, 10, 30)
    b = one.knn(q, 10, 30)
    for x, y in zip(a, b):
        assert (np.asarray(x) == np.asarray(y)).all()
    st = _lib.cbh_shard_stats()
    _lib.check(L.cbh_idx256_shard_stats(sh._h, C.byref(st)), "stats")
    assert st.shards == R and st.scans >= R and (st.collectives >= 1) == (sharded in ("rccl3", "alldev_rccl"))


def test_virtual_ordinals_were_really_used(gpu, orc, sharded):
    """under tests/shim/vdev.c only (test_virtual_devices.py): the shim saw the library switch to non-zero ordinals and
    copy between ordinals -- the shapes did not quietly collapse onto device 0"""
    if not os.environ.get("CBH_VDEV"):
        pytest.skip("needs the virtual-device shim")
    from cbird_amd import synth

    shim = C.CDLL(None)
    shim.vdev_stat.restype = C.c_long
    before = [shim.vdev_stat(i) for i in range(3)]
    h, ids = synth.make_hashes(5000, seed=21, planted_frac=0.3)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    gi, gs, gc = idx.find_batch(h[:300], 4, 7)
    wi, ws, wc = orc.find64_batch(h, ids, h[:300], 4, 7)
    assert (gc == wc).all() and (gi == wi).all() and (gs == ws).all()
    after = [shim.vdev_stat(i) for i in range(3)]
    ndev = bin(_SHAPES[sharded][1][0]).count("1")
    assert shim.vdev_stat(3) == int(os.environ["CBH_VDEV"])
    assert after[0] - before[0] >= ndev - 1   # hipSetDevice(d != 0)
    assert after[1] - before[1] >= ndev - 1   # hipMemcpyPeer[Async] between ordinals
    st = idx.shard_stats()
    assert st.devices == ndev and st.peer_copies >= ndev - 1
    # device discipline over everything this process has done so far: no kernel went to a stream of an ordinal that was
    # not current, no event was recorded on another ordinal's stream
    assert shim.vdev_stat(6) > 0 and shim.vdev_stat(4) == 0 and shim.vdev_stat(5) == 0
