"""Secondary measurements for BASELINE configs[3] (ORB 256-bit knn) and configs[4] (video), 1 GPU.
Not the bench.py contract line -- numbers for DESIGN.md / NOTES.md."""
import ctypes as C, json, sys, time
import numpy as np
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
L = _lib.lib()
out = {}

# ---- configs[3]: 100k images x 500 descriptors x 256 bit, needle = 500 descriptors, k = 4 / 10, odt = 25
from cbird_amd.cvfeatures import CvFeaturesIndex
n_img, per = (int(sys.argv[1]) if len(sys.argv) > 1 else 100_000), 500
rng = np.random.default_rng(1234)
idx = CvFeaturesIndex()
class M: pass
t0 = time.time()
chunk = 2000
for c0 in range(0, n_img, chunk):
    rows = rng.integers(0, 256, (chunk * per, 32), dtype=np.uint8)
    for i in range(chunk):
        _lib.check(L.cbh_idx256_add(idx.handle, c0 + i + 1, rows[i * per:(i + 1) * per].ctypes.data, per), "add")
print("built idx256", idx.count(), "rows in", round(time.time() - t0, 1), "s", flush=True)
needle = idx.descriptorsForMediaId(77).copy()
needle[::3, 5] ^= 0x11  # perturb a third of the needle's descriptors by 2 bits
st = _lib.cbh_stats()
for k in (4, 10):
    idx.knn(needle, k, 25)
    L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0, l0 = st.scan_ms, st.scan_launches
    t0 = time.time(); reps = 5
    for _ in range(reps):
        r, d, c = idx.knn(needle, k, 25)
    wall = (time.time() - t0) / reps
    L.cbh_idx256_get_stats(idx.handle, C.byref(st))
    kms = (st.scan_ms - ms0) / (st.scan_launches - l0)
    cmp_ = idx.count() * len(needle)
    out[f"orb_knn_k{k}"] = {"rows": idx.count(), "needle_desc": len(needle), "kernel_ms": kms, "wall_ms": wall * 1e3,
                            "cmp256_per_s_kernel": cmp_ / kms * 1e3, "algorithmic_GBps": cmp_ * 32 / kms * 1e3 / 1e9,
                            "self_found": int((c >= 1).sum()),
                            # 500 needle descriptors x 5e7 rows: 2.5e10 pairs per launch against 1.6 GB of rows -- 15.6 pairs
                            # per row byte, far above the machine balance, so the matrix cores bound it, not HBM
                            # (the rows stream once: 1.6 GB / 6.3 TB/s = 0.25 ms of the launch)
                            "roofline": {"kernel": "k_hamm256_mfma (128-bit prefilter)", "bound": "mfma", "unit": "TFLOP/s",
                                         "achieved": cmp_ * 256 / kms * 1e3 / 1e12, "peak": 10000.0,
                                         "frac": cmp_ * 256 / kms * 1e3 / 1e12 / 10000.0, "traffic": None,
                                         "hbm_GBps_rows_once": idx.count() * 32 / kms * 1e3 / 1e9,
                                         "note": "256 FLOP per pair = the 128-bit sign dot product the prefilter issues "
                                                 "(two K=64 FP4 MFMAs); candidates get their second half from the raw rows"}}
    print(k, out[f"orb_knn_k{k}"], flush=True)
# a batch of 64 needle images (32k needle descriptors) in one launch
needles = np.concatenate([idx.descriptorsForMediaId(i) for i in range(1, 65)])
idx.knn(needles[:500], 10, 25)
L.cbh_idx256_get_stats(idx.handle, C.byref(st)); ms0, l0 = st.scan_ms, st.scan_launches
t0 = time.time(); r, d, c = idx.knn(needles, 10, 25); wall = time.time() - t0
L.cbh_idx256_get_stats(idx.handle, C.byref(st)); kms = st.scan_ms - ms0
out["orb_knn_batch64"] = {"needle_desc": len(needles), "kernel_ms": kms, "wall_ms": wall * 1e3,
                          "cmp256_per_s_kernel": idx.count() * len(needles) / kms * 1e3,
                          # 32k needle descriptors: matrix-core bound; the 128-bit prefilter issues 256 FLOP per pair
                          "roofline": {"kernel": "k_hamm256_mfma (128-bit prefilter)", "bound": "mfma", "unit": "TFLOP/s",
                                       "achieved": idx.count() * len(needles) * 256 / kms * 1e3 / 1e12, "peak": 10000.0,
                                       "frac": idx.count() * len(needles) * 256 / kms * 1e3 / 1e12 / 10000.0, "traffic": None}}
print(out["orb_knn_batch64"], flush=True)
del idx

# ---- configs[4]: 10k clips x 300 frame hashes, dht 5, vtrim 0, vfm 30, vfn 60, exact (vradix 0)
from cbird_amd import synth_video
from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
t0 = time.time()
clips = synth_video.make_clips_fast(n_clips, 300, seed=1234, subclip_frac=0.01, max_gap=8)
vidx = DctVideoIndex()
media = []
for i, (f, h) in enumerate(clips):
    m = M(); m.id, m.path, m.videoIndex = i + 1, f"c{i}", VideoIndex(f, h); media.append(m)
vidx.add(media)
print("built video index", n_clips, "clips in", round(time.time() - t0, 1), "s", flush=True)
p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=30, minFramesNear=60)
vidx.findVideo(media[0], p)
t0 = time.time(); hits = 0
for m in media[-200:]:
    hits += len(vidx.findVideo(m, p))
one = (time.time() - t0) / 200
t0 = time.time(); res = vidx.find_videos_batch(media[-2000:], p); batch = time.time() - t0
out["video"] = {"clips": n_clips, "entries": vidx.entries(0), "single_needle_ms": one * 1e3, "hits_200": hits,
                "batch2000_s": batch, "needle_clips_per_s_batched": 2000 / batch,
                "cmp_per_s_batched": 2000 * 300 * vidx.entries(0) / batch,
                "batch_hits": sum(len(r) for r in res)}
print(out["video"], flush=True)

# ---- ColorDescIndex: 1M synthetic descriptors (the reference marks this search "dnf" at 500k, readme.md:293)
from cbird_amd.colordesc import COLOR_DTYPE
n_col = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rngc = np.random.default_rng(77)
descs = np.zeros(n_col, COLOR_DTYPE)
descs["colors"] = rngc.integers(0, 65536, (n_col, 32, 4), dtype=np.uint16)
descs["numColors"] = rngc.integers(28, 32, n_col, dtype=np.uint8)
ids_c = np.arange(1, n_col + 1, dtype=np.uint32)
from cbird_amd.colordesc import ColorDescIndex
ci = ColorDescIndex()
_lib.check(L.cbh_color_add(ci._h, ids_c.ctypes.data, descs.ctypes.data, n_col), "color add")
ci.find_batch(descs[:1], 8)
t0 = time.time(); reps = 5
for r_ in range(reps):
    ci.find_batch(descs[r_:r_ + 1], 8)
one_c = (time.time() - t0) / reps
t0 = time.time(); gi, gs, gc = ci.find_batch(descs[:64], 8); b64 = time.time() - t0
# 258 B and ~9.2e3 flop per descriptor comparison (32x32 colour pairs x 9 flop)
out["color"] = {"descriptors": n_col, "single_needle_ms": one_c * 1e3, "batch64_s": b64,
                "desc_cmp_per_s_batched": 64 * n_col / b64, "algorithmic_TFLOPs_batched": 64 * n_col * 9216 / b64 / 1e12,
                "algorithmic_GBps_batched": 64 * n_col * 258 / b64 / 1e9, "self_first": bool((gi[:, 0] == ids_c[:64]).all()),
                # the reference's rounding order forbids FMA contraction: one flop per VALU lane-op, so the honest
                # ceiling is the non-fused f32 rate = half the 157 TF FMA peak (MI355X_MICROARCH.md); both are given
                "roofline": {"kernel": "k_color_dist2", "bound": "valu-f32 (no FMA)", "unit": "TFLOP/s",
                             "achieved": 64 * n_col * 8192 / b64 / 1e12, "peak": 78.6, "frac": 64 * n_col * 8192 / b64 / 1e12 / 78.6,
                             "frac_of_fma_peak_157": 64 * n_col * 8192 / b64 / 1e12 / 157.3, "traffic": None,
                             "note": "8 flop per colour pair (3 sub, 3 mul, 2 add) x 1024 pairs; batch64_s is wall time of "
                                     "find_batch (distance kernel + top-k + download)"}}
print(out["color"], flush=True)
print(json.dumps(out))
