"""The cbird-side C++ bindings (cbird_amd/cpp/gpu_dcthashindex.h and gpu_indexes.h: all five Index
subclasses): compiled against a mock of the reference's headers/Qt types on CPU; built and executed
against the real library on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_adapter_compiles_against_index_interface():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_adapter"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(CPP, "test_adapter"))
    # every pure-virtual of the reference's Index (src/index.h:172-258) is overridden
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_dcthashindex.h")).read()
    for name in ("isLoaded", "memoryUsage", "count", "load", "save", "mediaIds", "add", "remove", "find",
                 "slice"):
        assert f" {name}(" in src and "override" in src


@pytest.mark.gpu
def test_adapter_runs_on_gpu(gpu):
    subprocess.check_call(["make", "-C", CPP, "test_adapter"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapter")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter ok" in out.stdout


def test_error_policy_compiles():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_errors"], stdout=subprocess.DEVNULL)
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_indexes.h")).read()
    assert "qFatal(\"%s: %s (%s)\", #call" not in src  # (round 3's abort-on-anything macro is gone)


@pytest.mark.gpu
def test_adapter_survives_device_out_of_memory_in_queries(gpu):
    """cbird_amd/cpp/gpu_errors.h: an allocation that fails once is retried after cbh_trim and answers; one that keeps
    failing makes find / findBatch / slice log with qCritical and return nothing, and the same handle answers again
    afterwards.  The reference aborts only on SQL failures (src/global.h:82) and on its own failed allocations."""
    subprocess.check_call(["make", "-C", CPP, "test_errors"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_errors")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "errors ok" in out.stdout and "no results for this call" in out.stderr


@pytest.mark.gpu
def test_adapter_aborts_when_a_mutation_cannot_allocate(gpu):
    """load() that cannot get its arrays leaves the index out of step with the database: qFatal, like the reference's
    failed allocation -- and the message names the call and the reason"""
    import signal

    subprocess.check_call(["make", "-C", CPP, "test_errors"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_errors"), "mutate"], capture_output=True, text=True, timeout=300)
    assert out.returncode == -signal.SIGABRT, (out.returncode, out.stdout + out.stderr)
    assert "GpuDctHashIndex::load: out of memory" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [("1", "8"), ("1", "3", "rccl")])
def test_adapter_over_logical_shards_equals_the_one_device_index(gpu, shape):
    """GpuDctHashIndex(GpuDeviceSet{mask, shardsPerDevice}): ONE index object over 8 logical shards (and over 3 whose
    blocks travel through ncclAllGather) -- the whole adapter test again, every find also held against the one-device
    index beside it.  src/engine.cpp:38-45 registers one object; the shards are inside it."""
    subprocess.check_call(["make", "-C", CPP, "test_adapter"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapter"), *shape], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter ok" in out.stdout and f"shards {shape[1]} devices 1" in out.stdout
    if "rccl" in shape:
        assert "collectives 0" not in out.stdout


@pytest.mark.gpu
def test_drop_in_find_under_concurrent_callers_on_a_sharded_index(gpu):
    """the 32-thread Database::similar pattern against ONE index of 4 logical shards: combining, the self-join cache
    built from the shards, results equal to the batched path's"""
    import json

    subprocess.check_call(["make", "-C", CPP, "test_coalesce"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_coalesce"), "100000", "32", "3125", "3", "1", "4"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "coalesce ok" in out.stdout
    st = json.loads(out.stdout.splitlines()[0])["coalesce"]
    assert st["finds"] == 100000 and st["cache_hits"] + st["scanned_needles"] == st["finds"]
    assert st["self_joins"] >= 1 and st["cache_hits"] > 0


@pytest.mark.gpu
def test_drop_in_find_under_concurrent_callers(gpu):
    """GpuDctHashIndex::find from 32 threads, one synchronous call per needle (the reference's Database::similar
    pattern): every result equals the batched path's; combining + the self-join cache both get exercised"""
    import json

    subprocess.check_call(["make", "-C", CPP, "test_coalesce"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_coalesce"), "200000", "32", "6250", "3"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "coalesce ok" in out.stdout
    st = json.loads(out.stdout.splitlines()[0])["coalesce"]
    assert st["finds"] == 200000 and st["cache_hits"] + st["scanned_needles"] == st["finds"]
    assert st["rounds"] < st["scanned_needles"] or st["scanned_needles"] < 64  # callers really were combined
    assert st["self_joins"] >= 1 and st["cache_hits"] > 0


def test_other_four_adapters_compile():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_adapters4"], stdout=subprocess.DEVNULL)
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_indexes.h")).read()
    for cls in ("GpuDctFeaturesIndex", "GpuCvFeaturesIndex", "GpuColorDescIndex", "GpuDctVideoIndex"):
        assert f"class {cls} : public" in src
    # every search-side virtual the reference classes override is overridden here too
    assert src.count("Index* slice(const QSet<uint32_t>& mediaIds) const override") == 4
    assert src.count("find(const Media&") == 4 and src.count(" remove(const QVector<int>&") == 4
    assert src.count("void load(QSqlDatabase& db") == 4 and src.count("void save(QSqlDatabase&") >= 3


@pytest.mark.gpu
def test_other_four_adapters_run_on_gpu(gpu, tmp_path):
    subprocess.check_call(["make", "-C", CPP, "test_adapters4"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapters4"), str(tmp_path)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapters ok" in out.stdout


@pytest.mark.gpu
def test_other_four_adapters_find_from_64_threads(gpu, tmp_path):
    """Index::find of GpuDctFeaturesIndex / GpuCvFeaturesIndex / GpuColorDescIndex / GpuDctVideoIndex from 64 threads at
    once (Database::similar's QtConcurrent pattern): every result equals the single-threaded answer, and the callers
    really shared device round trips (cbh_*_find_coalesced, combine.hip)"""
    import re

    subprocess.check_call(["make", "-C", CPP, "test_combine4"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_combine4"), "64", str(tmp_path)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "combine4 ok" in out.stdout
    per = [float(x) for x in re.findall(r"\(([0-9.]+) needles per round trip\)", out.stdout)]
    assert len(per) == 4 and all(p > 1.5 for p in per), out.stdout  # callers were combined on every index


def test_cvutil_dropins_compile():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_cvutil"], stdout=subprocess.DEVNULL)
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_cvutil.h")).read()
    assert "gpuDctHash64(const cv::Mat& cvImg, bool inPlace = false)" in src
    assert "gpuMakeKeyPointHashes(const cv::Mat& cvImg, const KeyPointList& keyPoints, KeyPointHashList& outHashes)" in src
    assert "gpuSizeLongestSide(cv::Mat& img, int size)" in src
    assert "gpuColorDescriptorCreate(const cv::Mat& cvImg, ColorDescriptor& desc)" in src
    assert "gpuMakeKeyPoints(const cv::Mat& cvImg, int numKeyPoints, KeyPointList& outKeypoints)" in src
    assert "gpuMakeKeyPointDescriptors(const cv::Mat& cvImg, KeyPointList& keyPoints, KeyPointDescriptors& outDescriptors)" in src


def _xorshift_stream(seed):
    s = seed & 0xFFFFFFFF
    while True:
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        yield s


def _checksum(img):
    v = 0
    for b in img.reshape(-1).tolist():
        v = (v * 1099511628211 + b) & 0xFFFFFFFFFFFFFFFF
    return v


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(200, 150, 7), (400, 300, 99)])
def test_cvutil_dropins_run_on_gpu(gpu, orc, w, h, seed):
    """gpuDctHash64 (whole image, view, view in place) and gpuMakeKeyPointHashes through the mock cv::Mat: the C++
    program and this test build the same image; expected values come from the oracle"""
    import numpy as np

    subprocess.check_call(["make", "-C", CPP, "test_cvutil"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_cvutil"), str(w), str(h), str(seed)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {ln.split()[0]: [int(x) for x in ln.split()[1:]] for ln in out.stdout.strip().splitlines()}
    g = _xorshift_stream(seed)
    img = np.array([next(g) >> 24 for _ in range(w * h)], np.uint8).reshape(h, w)
    assert got["whole"] == [orc.dcthash64(img)]
    vx, vy, vw, vh = w // 5, h // 4, w // 2, h // 2
    work = img.copy()
    view_hash = orc.dcthash64_rect_inplace(work, vx, vy, vw, vh)
    assert got["view"] == [view_hash] and got["view_inplace"] == [view_hash]
    assert view_hash != orc.dcthash64(img[vy:vy + vh, vx:vx + vw])  # an isolated copy reflects at its own border
    assert got["after_view_checksum"] == [_checksum(work)]
    kp = []
    for i in range(40):
        size = np.float32(31.0) * np.float32([1.0, 1.2, 1.44, 2.0736][i % 4])
        x = np.float32(next(g) % w) + np.float32(0.25)
        y = np.float32(next(g) % h) + np.float32(0.5)
        kp.append([x, y, size])
    want, after = orc.keypoint_hashes(work, np.array(kp, np.float32))
    assert len(want) >= 5
    assert got["kp"] == [42] + want.tolist()
    assert got["after_kp_checksum"] == [_checksum(after)]
    small = orc.size_longest_side(img, 128)
    assert got["resized"] == [small.shape[1], small.shape[0], _checksum(small)]
    # ORB drop-ins: gpuMakeKeyPoints then gpuMakeKeyPointDescriptors == the oracle's detect / compute
    from oracle import OrbOracle

    o = OrbOracle()
    pat = np.array([(i * 7 + (i // 4) * 3 + 3) % 27 - 13 for i in range(1024)], np.int8)
    o.set_pattern(pat)
    g3 = _xorshift_stream((seed + 17) & 0xFFFFFFFF)
    yy, xx = np.indices((h, w))
    noise = np.array([next(g3) % 9 for _ in range(w * h)], np.int64).reshape(h, w)
    scene = (np.where((xx // 9 + yy // 7) % 2 == 1, 200, 40) + noise).astype(np.uint8)
    kp = o.detect(scene, 100)
    assert got["orb_n"] == [len(kp)] and len(kp) > 20
    kp2, desc = o.compute(scene, kp)
    v = 0
    for k in kp2:
        for f in ("x", "y", "size", "angle", "response"):
            v = (v * 1099511628211 + int(np.float32(k[f]).view(np.uint32))) & 0xFFFFFFFFFFFFFFFF
        v = (v * 1099511628211 + int(k["octave"])) & 0xFFFFFFFFFFFFFFFF
    assert got["orb_kp"] == [len(kp2), v]
    assert got["orb_desc"] == [len(desc), _checksum(desc)]
    # ColorDescriptor::create drop-in
    from oracle import ColorCreateOracle

    g4 = _xorshift_stream((seed + 99) & 0xFFFFFFFF)
    bgr = np.zeros((h, w, 3), np.uint8)
    for y in range(h):
        for x in range(w):
            for c in range(3):
                bgr[y, x, c] = ((x // 23 + 2 * (y // 17) + c) % 5) * 50 + next(g4) % 7
    want_cd, _ = ColorCreateOracle().create(bgr)
    assert got["color"] == [int(want_cd[256]), _checksum(want_cd[:257])]
    assert got["color_gray"] == [77]
    # TemplateMatcher::match's scoring block (gpuTemplateScore)
    from oracle import PrestageOracle

    po = PrestageOracle()
    alpha = np.array([[255 - (x + y) % 97 for x in range(w)] for y in range(h)], np.uint8)
    tmpl4 = np.dstack([bgr, alpha])
    cand = np.zeros_like(bgr)
    cand[12:h - 12, 12:w - 12] = bgr[12:h - 12, 9:w - 15]
    d3, ch3, th3, _, _ = po.template_score(cand, bgr)
    d4, ch4, th4, _, _ = po.template_score(cand, tmpl4)
    assert got["tm3"] == [d3, ch3, th3] and got["tm4"] == [d4, ch4, th4]
    assert ch3 != ch4 and th3 != th4  # the alpha reaches both images


def test_makevideoindex_dropin_compiles():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_makevideoindex"], stdout=subprocess.DEVNULL)
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_cvutil.h")).read()
    assert "gpuMakeVideoIndex(VideoContextT& video, int threshold, VideoIndex& outIndex" in src


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [1, 16, 64])
def test_makevideoindex_dropin_runs_on_gpu(gpu, tmp_path, chunk):
    """gpuMakeVideoIndex over a mock VideoContext == the oracle's Media::makeVideoIndex: whole clip, resumed from a
    saved .vdx (media.cpp:929-936), and restarted when the decoder cannot seek"""
    import numpy as np

    from oracle import PrestageOracle, VideoOracle
    from test_video_indexer import clip

    po, vo = PrestageOracle(), VideoOracle()
    w, h, n, thr, stop = 320, 240, 70, 8, 31
    frames = clip(77, n, h, w, (30, 30, 0, 0), cut_every=13)
    raw = tmp_path / "frames.raw"
    frames.tofile(raw)
    subprocess.check_call(["make", "-C", CPP, "test_makevideoindex"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_makevideoindex"), str(raw), str(w), str(h), str(n), str(thr),
                          str(chunk), str(stop)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {}
    for line in out.stdout.splitlines():
        tag, *rest = line.split()
        got[tag] = rest
    hashes = np.array([po.process_image(f, autocrop=20)[0] for f in frames], np.uint64)

    def fmt(fh):
        f, hh = fh
        return [str(len(f))] + [f"{int(a)}:{int(b)}" for a, b in zip(f, hh)]

    full = vo.make_video_index(hashes, thr)
    part = vo.make_video_index(hashes[:stop], thr)
    assert got["full"] == fmt(full)
    assert got["progress"] == ["100", "1"]
    assert got["part"] == fmt(part) and int(part[0][-1]) == stop - 1
    assert got["resumed"] == fmt(vo.make_video_index(hashes[stop:], thr, resume=part))
    assert got["noseek"] == fmt(full)
