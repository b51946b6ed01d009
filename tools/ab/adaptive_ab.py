"""Prefilter kernel (PRE) against the three-field 64-bit kernel (FULL3) per threshold on data sets whose candidate rates
differ, with the fold-candidate rate of each set beside the times -- what the launch-time choice between the two has to
reproduce: never slower than min(PRE, FULL3) x 1.05.

  image     the bench's 10^6 image-derived hashes (10 % near-duplicates of earlier images)
  dup10     the same with 10 % of the slots overwritten by exact copies of other slots, in clusters (sizes geometric,
            mean 4, up to 64), scattered
  dup50     50 % of the slots
  flat      2 % of the slots share ONE hash up to 0-2 flipped bits (blank pages / black frames): a large dense cluster
  video     BASELINE configs[4]: 10^4 clips x 300 random-walk frame hashes as the haystack, the frames of the first
            2000 clips as needles

Per (set, threshold): ms of `rounds` alternating launches of each kernel (min), the record total (must agree), the
library's own choice (knobs at their defaults) and its time, and rate(t) = P[popc(fold(a) ^ fold(b)) < t] over a
4096 x 4096 sample (torch).
    python tools/ab/adaptive_ab.py [rounds=3] [thresholds=3,4,5,6,7,8] [sets=image,dup10,dup50,flat,video]"""
import ctypes as C, json, sys
import numpy as np
import torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth_video
import bench
L = _lib.lib()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,4,5,6,7,8").split(",")]
SETS = (sys.argv[3] if len(sys.argv) > 3 else "image,dup10,dup50,flat,video").split(",")
dev = torch.device("cuda", 0)
N = 1_000_000


def image_hashes():
    out = torch.empty(N, dtype=torch.int64, device=dev)
    for c0 in range(0, N, 100000):
        c1 = min(N, c0 + 100000)
        imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
        _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
        del imgs
    return out.cpu().numpy().view(np.uint64)


def with_dups(h, frac, seed):
    rng = np.random.default_rng(seed)
    h = h.copy()
    slots = rng.permutation(len(h))
    k, end = 0, int(len(h) * frac)
    while k < end:
        c = int(min(64, rng.geometric(0.25) + 1, end - k + 1))
        src = h[slots[len(h) - 1 - k]]  # a slot from the other end of the permutation: never overwritten itself
        h[slots[k:k + c - 1]] = src
        k += max(1, c - 1)
    return h


def with_flat(h, frac, seed):
    rng = np.random.default_rng(seed)
    h = h.copy()
    m = int(len(h) * frac)
    v = np.full(m, 0x00000000FFFF0000, np.uint64)
    for _ in range(2):
        on = rng.random(m) < 0.5
        v ^= np.where(on, np.uint64(1) << rng.integers(1, 64, m).astype(np.uint64), np.uint64(0))
    h[rng.choice(len(h), m, replace=False)] = v
    return h


def fold_rates(hay, q):
    a = torch.from_numpy(hay[:: max(1, len(hay) // 4096)][:4096].view(np.int64).copy()).to(dev)
    st = max(1, len(q) // 4096)
    b = torch.from_numpy(q[st // 2:: st][:4096].view(np.int64).copy()).to(dev)
    fa = ((a ^ (a >> 32)) & 0xFFFFFFFF).view(-1, 1)
    fb = ((b ^ (b >> 32)) & 0xFFFFFFFF).view(1, -1)
    x = fa ^ fb
    d = torch.zeros_like(x)
    for s in range(32):
        d += (x >> s) & 1
    tot = float(d.numel())
    return {t: float((d < t).sum().item()) / tot for t in range(1, 10)}


base = image_hashes() if any(s != "video" for s in SETS) else None
res = {}
for name in SETS:
    if name == "image":
        hay = q = base
    elif name == "dup10":
        hay = q = with_dups(base, 0.10, 1)
    elif name == "dup50":
        hay = q = with_dups(base, 0.50, 2)
    elif name == "flat":
        hay = q = with_flat(base, 0.02, 3)
    elif name == "video":
        clips = synth_video.make_clips_fast(10000, 300, seed=1234)
        hay = np.concatenate([h for _, h in clips])
        q = np.concatenate([h for _, h in clips[:2000]])
    else:
        raise SystemExit(f"unknown set {name}")
    n, nq = len(hay), len(q)
    dh = torch.from_numpy(hay.view(np.int64).copy()).to(dev)
    dq = torch.from_numpy(q.view(np.int64).copy()).to(dev)
    ids = torch.arange(1, n + 1, dtype=torch.int32, device=dev)
    idx = cbird_amd.DctHashIndex()
    idx.load_device(dh.data_ptr(), ids.data_ptr(), n)
    cap = 1 << 27
    drec = torch.empty(cap, dtype=torch.int64, device=dev)
    dtot = torch.zeros(1, dtype=torch.int64, device=dev)
    ms = C.c_float(0)
    rates = fold_rates(hay, q)
    out = {"n": n, "nq": nq, "fold_rate": {str(t): rates[t] for t in T}, "per_threshold": {}}
    for t in T:
        cell = {"pre": [], "full3": [], "auto": []}
        for r in range(R):
            for kind, knobs in (("pre", {b"scan_mfma_pre_max": 32}), ("full3", {b"scan_mfma_pre_max": 0}), ("auto", {b"scan_mfma_pre_max": -1})):
                for k, v in knobs.items():
                    L.cbh_set_tuning(k, v)
                _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), nq, t, drec.data_ptr(), cap, dtot.data_ptr(), 2, C.byref(ms)), "t")
                cell[kind].append((round(ms.value, 3), int(dtot.item()) // 2))
                if kind == "auto":
                    v = C.c_longlong(0)
                    L.cbh_get_tuning(b"scan_pre_mask", C.byref(v))
                    auto_pre = bool((v.value >> t) & 1)
                    L.cbh_get_tuning(b"scan_probe_rate_e9", C.byref(v))
                    auto_rate = v.value / 1e9
                    L.cbh_get_tuning(b"scan_probe_true_e9", C.byref(v))
                    auto_true = v.value / 1e9
        L.cbh_set_tuning(b"scan_mfma_pre_max", -1)
        mins = {k: min(x[0] for x in v) for k, v in cell.items()}
        tot = {k: sorted({x[1] for x in v}) for k, v in cell.items()}
        best = min(mins["pre"], mins["full3"])
        out["per_threshold"][str(t)] = {"ms": mins, "records": tot["full3"], "totals_agree": len({tuple(v) for v in tot.values()}) == 1,
                                        "auto_over_best": round(mins["auto"] / best, 3), "pairs_1e12": n * nq / 1e12,
                                        "auto_kernel": "pre" if auto_pre else "full3", "library_probe_rate": auto_rate, "library_probe_true_rate": auto_true}
    res[name] = out
    print(json.dumps({name: out}), flush=True)
    del idx, dh, dq, drec
worst = max(c["auto_over_best"] for s in res.values() for c in s["per_threshold"].values())
print(json.dumps({"summary": {"worst_auto_over_best": worst,
                              "all_totals_agree": all(c["totals_agree"] for s in res.values() for c in s["per_threshold"].values())}}))
