"""Error paths of the C-ABI (include/cbird_hip.h: "return 0 / negative codes, never abort").

cbh_set_tuning("fault_alloc_after", k) makes the k-th allocation from now fail once with out-of-memory -- device arrays,
pinned words and arena scratch all pass the same gate (cbird_amd/csrc/cbh_internal.h).  For one entry point of every
family the test walks k = 0, 1, 2, ... over every allocation the call makes on a FRESH handle, until a call makes fewer
than k + 1.  Each time the call must either absorb the failure (same result as without it) or come back with
CBH_E_NOMEM / CBH_E_OVERFLOW (a create: NULL), and the very next, un-faulted call must give the right result.  At
the end nothing may be left behind: no arena block still handed out, and after cbh_trim the device has its memory back.
"""
import ctypes as C
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tuning(L, key):
    v = C.c_longlong(-2)
    assert L.cbh_get_tuning(key, C.byref(v)) == 0
    return int(v.value)


def _free_bytes(L):
    import torch

    rel = C.c_ulonglong(0)
    assert L.cbh_trim(0, C.byref(rel)) == 0
    return int(torch.cuda.mem_get_info(0)[0])


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return isinstance(b, (tuple, list)) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return isinstance(b, np.ndarray) and a.shape == b.shape and bool((a == b).all())
    return a == b


def _walk(L, call, max_sites=300):
    """-> (allocation sites that failed, how many of those failures the call absorbed)"""
    from cbird_amd import _lib

    want = call()
    assert _same(call(), want)  # (the call is repeatable: what follows compares against it)
    failed = absorbed = 0
    for k in range(max_sites):
        fired0 = _tuning(L, b"fault_fired")
        L.cbh_set_tuning(b"fault_alloc_after", k)
        err, got = None, None
        try:
            got = call()
        except _lib.CbhError as e:
            err = e
        finally:
            L.cbh_set_tuning(b"fault_alloc_after", -1)
        if _tuning(L, b"fault_fired") == fired0:
            assert err is None and _same(got, want)
            break
        failed += 1
        if err is None:
            assert _same(got, want), f"allocation {k} failed and the call returned a different result"
            absorbed += 1
        else:
            assert err.code in (_lib.CBH_E_NOMEM, _lib.CBH_E_OVERFLOW, _lib.CBH_E_NODEVICE), (k, err)
        assert _same(call(), want), f"after the failure of allocation {k} the same call no longer answers"
    else:
        pytest.fail("the call never ran out of allocations to fail")
    return failed, absorbed


def _cases(gpu):
    from cbird_amd import hashing, orb, synth, synth_video
    from cbird_amd.colordesc import ColorDescIndex, create_descriptors
    from cbird_amd.cvfeatures import CvFeaturesIndex
    from cbird_amd.index import DctFeaturesIndex, SearchParams
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoIndexer, VideoSearchParams
    from test_color import synth_descriptors

    rng = np.random.default_rng(5)
    h, ids = synth.make_hashes(30000, seed=3, planted_frac=0.2)
    tiles = synth.make_images(24, seed=2)
    photos = rng.integers(0, 256, (3, 300, 400), dtype=np.uint8)
    cases = {}

    cases["dcthash_256"] = lambda: hashing.dct_hash64_batch(tiles)
    cases["dcthash_general"] = lambda: hashing.dct_hash64_batch(photos)
    cases["process_images"] = lambda: tuple(np.asarray(x) for x in hashing.process_images(photos, autocrop=20))

    def idx64(shards=None, thresh=5, nq=2048):
        def run():
            idx = gpu.DctHashIndex(shards=shards)
            idx.load(h, ids)
            a = idx.find_batch(h[:nq], thresh, 6)
            m = idx.find(gpu.Media(id=1, dctHash=int(h[7])), SearchParams(dctThresh=9))
            sl = idx.slice(ids[100:400].tolist())
            return a, [(x.mediaId, x.score) for x in m], sl.count()
        return run

    cases["idx64_mfma_full"] = idx64()
    cases["idx64_mfma_pre"] = idx64(thresh=3)
    cases["idx64_valu"] = idx64(nq=40)
    cases["idx64_sharded"] = idx64(shards=(1, 3))

    def search_index():
        idx = gpu.DctHashIndex()
        idx.load(h, ids)
        p = SearchParams(dctThresh=2, maxThresh=6, minMatches=1, maxMatches=4)
        return idx.search_index_batch(h[:1500], ids[:1500], p)

    cases["search_index_batch"] = search_index

    kp_h, kp_ids = synth.make_hashes(20000, seed=9, planted_frac=0.3)

    class M:
        pass

    def fdct():
        idx = DctFeaturesIndex()
        idx.load_flat(kp_h, (np.arange(20000) // 40 + 1).astype(np.uint32))
        needles = []
        for i in range(6):
            m = M()
            m.id, m.path, m.keyPointHashes = i + 1, "", [int(x) for x in kp_h[i * 40:(i + 1) * 40]]
            needles.append(m)
        return [[(x.mediaId, x.score) for x in r] for r in idx.find_batch(needles, SearchParams(dctThresh=6))]

    cases["fdct_find_batch"] = fdct

    clips = synth_video.make_clips(60, 120, seed=4, subclip_frac=0.2, max_gap=6)

    def video():
        idx = DctVideoIndex()
        media = []
        for i, (f, hh) in enumerate(clips):
            m = M()
            m.id, m.path, m.videoIndex, m.dctHash = i + 1, "", VideoIndex(f, hh), 0
            media.append(m)
        idx.add(media)
        p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=10, minFramesNear=20)
        return [[(x.mediaId, x.score, x.range.srcIn, x.range.dstIn, x.range.len) for x in r]
                for r in idx.find_videos_batch(media[-12:], p)]

    cases["video_find_batch"] = video

    rows = rng.integers(0, 256, (40 * 300, 32), dtype=np.uint8)

    def idx256(nq):
        def run():
            from cbird_amd import _lib

            idx = CvFeaturesIndex()
            for i in range(40):
                _lib.check(_lib.lib().cbh_idx256_add(idx.handle, i + 1, rows[i * 300:(i + 1) * 300].ctypes.data, 300), "add")
            q = rows[5:5 + nq].copy()
            q[::2, 3] ^= 0x21
            return tuple(np.asarray(x) for x in idx.knn(q, 6, 30))
        return run

    cases["idx256_knn_small"] = idx256(300)
    cases["idx256_knn_batch"] = idx256(2000)

    cd, cids = synth_descriptors(3000, 8)

    def color():
        idx = ColorDescIndex()
        media = []
        for d, i in zip(cd, cids):
            m = M()
            m.id, m.colorDescriptor = int(i), d
            media.append(m)
        idx.add(media)
        return tuple(np.asarray(x) for x in idx.find_batch(cd[:48], 5)), idx.distances(cd[:3])

    cases["color_find_batch"] = color

    scene = np.full((200, 260), 128, np.int32)
    for _ in range(50):
        x, y = int(rng.integers(0, 250)), int(rng.integers(0, 190))
        scene[y: y + int(rng.integers(4, 60)), x: x + int(rng.integers(4, 60))] = int(rng.integers(0, 256))
    scene = (scene + rng.integers(-3, 4, scene.shape)).clip(0, 255).astype(np.uint8)
    orb.set_pattern(orb.synthetic_pattern())

    def run_orb():
        (k, a, d), = orb.orb([scene], 150)
        return np.asarray(k), np.asarray(d)

    cases["orb"] = run_orb
    colour = rng.integers(0, 256, (2, 90, 120, 3), dtype=np.uint8)
    colour[:, 20:70, 30:90] = (30, 200, 90)
    cases["color_descriptor_create"] = lambda: tuple(np.asarray(x) for x in create_descriptors(list(colour)))
    kp = np.array([[12.5, 20.25, 31.0], [40.0, 33.0, 44.64], [15.0, 25.0, 64.28]], np.float32)
    cases["keypoint_hashes"] = lambda: [np.asarray(x) for x in hashing.make_keypoint_hashes([photos[0]], [kp])]

    frames = np.full((20, 96, 128), 16, np.uint8)
    for i in range(20):
        sc = np.random.default_rng(100 + i // 5).integers(60, 256, (6, 8)).astype(np.uint8)
        frames[i, 12:84] = np.kron(sc, np.ones((12, 16), np.uint8))

    def vindexer():
        vix = VideoIndexer(threshold=8)
        vix.push(frames[:9])
        vix.push(frames[9:])
        vi = vix.finish()
        return list(vi.frames), [int(x) for x in vi.hashes]

    cases["video_indexer"] = vindexer
    return cases


_CASE_NAMES = ["dcthash_256", "dcthash_general", "process_images", "idx64_mfma_full", "idx64_mfma_pre", "idx64_valu",
               "idx64_sharded", "search_index_batch", "fdct_find_batch", "video_find_batch", "idx256_knn_small",
               "idx256_knn_batch", "color_find_batch", "orb", "color_descriptor_create", "keypoint_hashes",
               "video_indexer"]


@pytest.fixture(scope="module")
def cases(gpu):
    c = _cases(gpu)
    assert sorted(c) == sorted(_CASE_NAMES)
    return c


@pytest.mark.parametrize("name", _CASE_NAMES)
def test_every_allocation_of_the_call_may_fail_once(gpu, cases, name):
    from cbird_amd import _lib

    L = _lib.lib()
    call = cases[name]
    # one-time tables of the library (DCT / area tables, the ORB pattern) are made on first use, and the runtime loads a
    # translation unit's code object (10 MB of device memory at a time) with the first launch of one of its kernels --
    # which, for a kernel only a fallback path launches, is somewhere in the walk: a dry walk first, neither is a leak
    call()
    _walk(L, call)
    gc.collect()
    live0 = _tuning(L, b"arena_live_blocks")
    free0 = _free_bytes(L)
    failed, absorbed = _walk(L, call)
    assert failed >= 1, "the call allocates nothing?"
    gc.collect()
    assert _tuning(L, b"arena_live_blocks") == live0, "an error path kept an arena block"
    free1 = _free_bytes(L)
    assert free1 >= free0 - (8 << 20), f"{(free0 - free1) >> 20} MB of device memory did not come back"
    print(f"{name}: {failed} allocation sites failed once each, {absorbed} absorbed by the call itself")


def test_driver_out_of_memory_is_retried_after_the_caches_are_given_up(gpu, cases):
    """the arena's own hipMalloc failing ("fault_driver_oom"): the calling stream's cache goes back to the driver and
    the allocation is tried again -- the caller never sees it"""
    from cbird_amd import _lib

    L = _lib.lib()
    call = cases["dcthash_general"]
    want = call()
    rel = C.c_ulonglong(0)
    assert L.cbh_trim(0, C.byref(rel)) == 0  # empty caches: the next scratch request has to go to the driver
    r0 = _tuning(L, b"arena_oom_retry_stream")
    L.cbh_set_tuning(b"fault_driver_oom", 0)
    try:
        got = call()
    finally:
        L.cbh_set_tuning(b"fault_driver_oom", -1)
    assert _same(got, want) and _tuning(L, b"arena_oom_retry_stream") == r0 + 1


def test_index_storage_refused_by_the_driver_takes_the_arenas_caches_back(gpu, cases):
    """a live stream may keep a quarter of the device cached as scratch; index storage, workspaces and
    tables come from plain hipMalloc, which knows nothing of that.  A refused persistent allocation ("fault_persist_oom")
    gives the arena's cached and pending blocks of the device back and is tried again -- the caller never sees it"""
    from cbird_amd import _lib, synth

    L = _lib.lib()
    cases["dcthash_general"]()  # leaves scratch cached on its stream
    assert _tuning(L, b"arena_cached_bytes") > 0
    r0 = _tuning(L, b"arena_oom_retry_persistent")
    h, ids = synth.make_hashes(50000, seed=2)
    idx = gpu.DctHashIndex()
    L.cbh_set_tuning(b"fault_persist_oom", 0)
    try:
        idx.load(h, ids)  # its first device allocation is "refused"
    finally:
        L.cbh_set_tuning(b"fault_persist_oom", -1)
    assert _tuning(L, b"arena_oom_retry_persistent") == r0 + 1
    assert _tuning(L, b"arena_cached_bytes") == 0  # the caches went back to the driver before the retry
    assert idx.count() == 50000 and [m.mediaId for m in idx.find(gpu.Media(id=0, dctHash=int(h[7])), gpu.SearchParams(dctThresh=1))] == [int(ids[7])]


def test_a_live_streams_cache_is_bounded_by_pool_live_keep_mb(gpu):
    """free_async keeps what a live stream has used only up to the budget: beyond it the blocks freed longest ago wait
    for the work queued behind them and go back to the driver (round 3 advice: a long-lived caller stream that once
    took a large scratch held it until cbh_trim)"""
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(1)
    _free_bytes(L)
    c0 = _tuning(L, b"arena_cached_bytes")  # (caches of streams that are still busy elsewhere)
    L.cbh_set_tuning(b"pool_live_keep_mb", 1)
    try:
        stream = torch.cuda.Stream()
        t0 = _tuning(L, b"arena_trimmed_live")
        for w, hgt in ((1920, 1080), (3840, 2160), (1280, 720), (4000, 3000)):  # scratch of very different sizes
            imgs = torch.from_numpy(rng.integers(0, 256, (4, hgt, w), dtype=np.uint8)).cuda()
            out = torch.empty(4, dtype=torch.int64, device="cuda")
            _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), 4, w, hgt, w, w * hgt, out.data_ptr(), 0,
                                               C.c_void_p(stream.cuda_stream)), "hash")
        stream.synchronize()
        assert _tuning(L, b"arena_cached_bytes") <= c0 + (1 << 20)
        assert _tuning(L, b"arena_trimmed_live") > t0
    finally:
        L.cbh_set_tuning(b"pool_live_keep_mb", 0)
        _free_bytes(L)
    assert _tuning(L, b"arena_pending_bytes") == 0  # cbh_trim reaped what was waiting


def test_a_join_that_cannot_get_its_scratch_leaves_the_call_to_the_scan(gpu):
    """"scan_mfma" 3: every allocation of the bucketed join (hamm64_join.hip) precedes its first record, so a refused one sends
    the call to the scan instead of failing it -- same results, "scan_joins" unchanged for that call"""
    import torch

    from cbird_amd import _lib, synth

    L = _lib.lib()
    n, k = 400_000, 4
    h, ids = synth.make_hashes(n, seed=11, planted_frac=0.2)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    dq = torch.from_numpy(h.view(np.int64)).cuda()

    def run():
        dout = torch.empty((n, k, 2), dtype=torch.int32, device="cuda")
        dcnt = torch.empty(n, dtype=torch.int32, device="cuda")
        tot = C.c_uint64(0)
        _lib.check(L.cbh_idx64_find_batch_dev(idx.handle, dq.data_ptr(), n, 3, k, dout.data_ptr(), dcnt.data_ptr(), C.byref(tot),
                                              None), "find_batch_dev")
        return int(tot.value), dcnt.cpu().numpy(), dout.cpu().numpy()

    want = run()
    L.cbh_set_tuning(b"scan_mfma", 3)
    try:
        j0 = _tuning(L, b"scan_joins")
        got = run()
        assert _tuning(L, b"scan_joins") == j0 + 1 and got[0] == want[0] and (got[1] == want[1]).all()
        fell_back = 0
        for site in range(12):
            j1 = _tuning(L, b"scan_joins")
            L.cbh_set_tuning(b"fault_alloc_after", site)
            try:
                got = run()
            except _lib.CbhError:
                continue  # (an allocation of the call outside the join: the walk above covers those)
            finally:
                L.cbh_set_tuning(b"fault_alloc_after", -1)
            assert got[0] == want[0] and (got[1] == want[1]).all()
            fell_back += _tuning(L, b"scan_joins") == j1
        assert fell_back >= 1
    finally:
        L.cbh_set_tuning(b"scan_mfma", 1)
        L.cbh_set_tuning(b"fault_alloc_after", -1)
