"""The two result-changing knobs, priced: what they buy and what they move.
  "hash_area" 1   fractional-ratio INTER_AREA: interior pixels of a cell summed as integers (k_blur_area_regs)
  "color_fma" 1   colour distance with fused squares (k_color_dist3<.., FMA>)
One JSON line: per geometry GB/s exact / fast and how many of the hashes differ (smooth + noisy images); for the colour
index ms per 64 needles x N descriptors for the three exact kernels and the fused one, the number of int scores that
move and the largest relative difference of the float distances.
    python tools/ab/knobs_area_color.py"""
import ctypes as C, json, sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cbird_amd import _lib
from cbird_amd.colordesc import ColorDescIndex
L = _lib.lib()
dev = torch.device("cuda", 0)
ms = C.c_float(0)
res = {"hash_area": [], "color": {}}
g = torch.Generator(device=dev).manual_seed(9)
for (w, h) in ((400, 300), (533, 400), (600, 400), (1366, 768), (1920, 1080), (3000, 2000)):
    n = max(256, min(16384, int(1.5e9 // (w * h))))
    yy = torch.arange(h, device=dev, dtype=torch.float32).view(1, h, 1)
    xx = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, w)
    k = torch.rand((n, 1, 1), device=dev, generator=g) * 40 + 4
    base = 127 + 90 * torch.sin(xx / k) * torch.cos(yy / (k * 0.7 + 3))
    noise = torch.randint(-128, 128, (n, h, w), device=dev, generator=g, dtype=torch.int16).float()
    amp = (torch.arange(n, device=dev) % 4).view(n, 1, 1).float() / 3.0  # from smooth to half noise
    imgs = (base + noise * amp * 0.5).clamp(0, 255).to(torch.uint8)
    del base, noise
    row = {"geometry": [w, h], "images": n}
    outs = {}
    for name, v in (("exact", 0), ("fast", 1)):
        L.cbh_set_tuning(b"hash_area", v)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        best = 1e9
        for _ in range(3):
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, out.data_ptr(), 0, 3, C.byref(ms)), "h")
            best = min(best, ms.value)
        row[name + "_GBps"] = round(n * w * h / best * 1e-6, 1)
        outs[name] = out.clone()
    L.cbh_set_tuning(b"hash_area", 0)
    x = (outs["exact"] ^ outs["fast"])
    row["hashes_differing"] = int((x != 0).sum().item())
    row["bits_differing"] = int(sum(bin(int(v) & (2 ** 64 - 1)).count("1") for v in x[x != 0].cpu().tolist()))
    res["hash_area"].append(row)
    del imgs
# ---- colour
from test_color import synth_descriptors
n = 200_000
cd, cids = synth_descriptors(n, 8)
idx = ColorDescIndex()
class M: pass
media = []
for d, i in zip(cd, cids):
    m = M(); m.id, m.colorDescriptor = int(i), d; media.append(m)
idx.add(media)
needles = cd[:64]
import time
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r
ref = None
for name, knobs in (("k_color_dist2 (default)", {b"color_pk": 1, b"color_fma": 0}), ("k_color_dist3 exact", {b"color_pk": 2, b"color_fma": 0}),
                    ("k_color_dist3 fused", {b"color_pk": 2, b"color_fma": 1})):
    for k_, v in knobs.items():
        L.cbh_set_tuning(k_, v)
    t, r = timed(lambda: idx.find_batch(needles, 10))
    raw = idx.distances(needles[:8])
    if ref is None:
        ref, ref_raw = r, raw
    ent = {"find_batch_ms_64_needles": round(t, 2), "descriptors": n}
    if "fused" in name:
        fin = (ref_raw < 1e30) & (raw < 1e30)
        rel = np.abs(raw[fin] - ref_raw[fin]) / np.maximum(ref_raw[fin], 1e-9)
        ent["max_relative_difference_of_the_float_distance"] = float(rel.max())
        ent["int_scores_that_move"] = int((raw[fin].astype(np.int32) != ref_raw[fin].astype(np.int32)).sum())
        ent["of"] = int(fin.sum())
        ent["top10_lists_equal"] = bool(all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(r, ref)))
    else:
        assert all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(r, ref)) and (raw == ref_raw).all()
    res["color"][name] = ent
L.cbh_set_tuning(b"color_pk", 1); L.cbh_set_tuning(b"color_fma", 0)
print(json.dumps(res))
