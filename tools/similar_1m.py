"""A 1M-needle `-similar` (dht 2, BASELINE configs[0]'s flags at configs[1]'s size) behind the C-ABI: hashes resident in
a DctHashIndex -> cbh_search_index_batch -> cbh_filter_groups.  Wall time per stage, no per-needle loop in Python."""
import ctypes as C, json, sys, time
import numpy as np
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth, SearchParams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h, ids = synth.make_hashes(n, seed=1234, planted_frac=0.05, max_dist=8)
idx = cbird_amd.DctHashIndex()
idx.load(h, ids)
L = _lib.lib()
out = {}
for dht, mt in ((2, 0), (5, 0), (2, 6)):
    p = SearchParams(dctThresh=dht, maxThresh=mt)
    idx.search_index_batch(h[:4096], ids[:4096], p)
    t0 = time.perf_counter()
    mi, ms, mc = idx.search_index_batch(h, ids, p, valid_ids=ids)
    t1 = time.perf_counter()
    rank = ids.copy()  # paths "/img/%06d": rank = id order
    pairs = np.zeros((n, p.maxMatches, 2), np.uint32)
    pairs[:, :, 0], pairs[:, :, 1] = mi, ms.view(np.uint32)
    og = np.zeros(n, np.uint32); no = C.c_size_t(0)
    _lib.check(L.cbh_filter_groups(ids.ctypes.data, pairs.ctypes.data, mc.ctypes.data, n, p.maxMatches, p.minMatches, 1,
                                   ids.ctypes.data, rank.ctypes.data, n, og.ctypes.data, C.byref(no)), "fg")
    t2 = time.perf_counter()
    out[f"dht{dht}_maxthresh{mt}"] = {"needles": n, "search_index_batch_s": round(t1 - t0, 4), "filter_groups_s": round(t2 - t1, 4),
                                      "groups": int(no.value), "needles_with_matches": int((mc > 0).sum())}
    print(out[f"dht{dht}_maxthresh{mt}"], flush=True)
print(json.dumps({"similar_1m": out}))
