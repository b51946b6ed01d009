"""Same-box A/B of the 32-bit prefilter scan (k_hamm64_mfma<PRE>) by prefilter word and re-check path, against FULL3,
per threshold, on the bench's image-derived hashes (and on uniform ones with `uniform`):
  full3          k_hamm64_mfma3 (no prefilter)
  lo/vec         low word, LDS queue re-check      (rounds 1-4)
  fold/vec       lo ^ hi,  LDS queue re-check
  lo/lean        low word, scalar re-check of single candidates
  fold/lean      lo ^ hi,  scalar re-check          (round 5 default)
Record totals must agree across all five (the prefilter only proposes; the re-check is exact).
    python tools/ab/pre_fold_ab.py [n=1000000] [rounds=3] [thresholds=3,4,5,6,7,8] [uniform]"""
import ctypes as C, json, sys
import torch
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib
import bench
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "3,4,5,6,7,8").split(",")]
uniform = len(sys.argv) > 4 and sys.argv[4] == "uniform"
dev = torch.device("cuda", 0)
out = torch.empty(N, dtype=torch.int64, device=dev)
if uniform:
    from cbird_amd import synth
    import numpy as np
    h, _ = synth.make_hashes(N, seed=5)
    out.copy_(torch.from_numpy(h.view(np.int64)))
else:
    for c0 in range(0, N, 100000):
        c1 = min(N, c0 + 100000)
        imgs = bench.gen_images(torch, dev, c0, c1, N, 1234)
        _lib.check(L.cbh_dcthash_batch_dev(imgs.data_ptr(), c1 - c0, 256, 256, 256, 65536, out[c0:].data_ptr(), 0, None), "h")
        del imgs
idx = cbird_amd.DctHashIndex(); ids = torch.arange(1, N + 1, dtype=torch.int32, device=dev)
idx.load_device(out.data_ptr(), ids.data_ptr(), N)
cap = 1 << 25
drec = torch.empty(cap, dtype=torch.int64, device=dev); dtot = torch.zeros(1, dtype=torch.int64, device=dev)
ms = C.c_float(0)
CFG = {"full3": (0, 1, 1), "lo/vec": (2, 0, 0), "fold/vec": (2, 1, 0), "lo/lean": (2, 0, 1), "fold/lean": (2, 1, 1)}
res = {}
for r in range(R):
    for t in T:
        for name, (pre, fold, lean) in CFG.items():
            L.cbh_set_tuning(b"scan_mfma_pre", pre)
            L.cbh_set_tuning(b"scan_pre_fold", fold)
            L.cbh_set_tuning(b"scan_pre_lean", lean)
            _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, out.data_ptr(), N, t, drec.data_ptr(), cap, dtot.data_ptr(), 3, C.byref(ms)), "t")
            res.setdefault(str(t), {}).setdefault(name, []).append([round(ms.value, 3), int(dtot.item())])
L.cbh_set_tuning(b"scan_mfma_pre", 1); L.cbh_set_tuning(b"scan_pre_fold", 1); L.cbh_set_tuning(b"scan_pre_lean", 1)
summary = {}
for t, d in res.items():
    totals = {k: sorted({x[1] for x in v}) for k, v in d.items()}
    summary[t] = {"ms_min": {k: min(x[0] for x in v) for k, v in d.items()},
                  "totals_agree": len({tuple(v) for v in totals.values()}) == 1, "total": totals["full3"]}
print(json.dumps({"n": N, "data": "uniform" if uniform else "image-derived", "summary": summary, "raw": res}))
