"""BASELINE configs[0]: 10k synthetic 256x256 grayscale images, `-p.alg dct -p.dht 2 -similar` -- the CPU reference leg
(hash port on all host cores + the reference's real VP-tree when oracle/_ref is present + the searchIndex / filter
post-processing of oracle/search_index.c) beside the same job on one MI355X (hash kernel + cbh_search_index_batch +
cbh_filter_groups); the two group lists must be identical.  Prints one JSON object."""
import ctypes as C, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, ".")
import cbird_amd, oracle
from cbird_amd import _lib, synth, SearchParams, Media
from cbird_amd.database import similar

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
cores = len(os.sched_getaffinity(0))
imgs = synth.make_images(n, seed=1234)
orc = oracle.Oracle()
ids = np.arange(1, n + 1, dtype=np.uint32)
p = SearchParams(dctThresh=2)
res = {"workload": f"configs[0]: {n} synthetic 256x256 images, -p.alg dct -p.dht 2 -similar", "host_cores": cores}
# ---- CPU leg
t0 = time.perf_counter()
parts = np.array_split(np.arange(n), cores * 4)
with ThreadPoolExecutor(cores) as ex:
    hs = list(ex.map(lambda ix: orc.dcthash64_batch(imgs[ix]) if len(ix) else np.zeros(0, np.uint64), parts))
h_cpu = np.concatenate(hs)
t_hash = time.perf_counter() - t0
cpu = {"hash_s": round(t_hash, 4), "hash_images_per_s": n / t_hash, "hash_kind": "port (oracle/cbird_oracle.c)"}
if oracle.ref_available():
    t0 = time.perf_counter()
    tree = oracle.RefTree(h_cpu, ids)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    tree.search_many(h_cpu, 2, threads=cores)
    t_find = time.perf_counter() - t0
    cpu.update({"vptree_build_s": round(t_build, 4), "vptree_find_all_s": round(t_find, 4), "find_kind": "reference (src/tree/vptree.h)"})
rank = np.arange(n, dtype=np.int32)
t0 = time.perf_counter()
want = orc.similar_dct(h_cpu, ids, rank, h_cpu, ids, 2, 0, p.minMatches, p.maxMatches, True, True)
cpu["similar_postprocess_bruteforce_1thread_s"] = round(time.perf_counter() - t0, 4)
cpu["similar_wall_s"] = round(t_hash + cpu.get("vptree_build_s", 0) + cpu.get("vptree_find_all_s", cpu["similar_postprocess_bruteforce_1thread_s"]), 4)
res["cpu"] = cpu
# ---- GPU leg
import torch
L = _lib.lib()
d_imgs = torch.from_numpy(imgs).cuda()
out = torch.empty(n, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
def gpu_job():
    _lib.check(L.cbh_dcthash_batch_dev(d_imgs.data_ptr(), n, 256, 256, 256, 65536, out.data_ptr(), 0, None), "hash")
    h = out.cpu().numpy().view(np.uint64)
    idx = cbird_amd.DctHashIndex()
    idx.load(h, ids)
    mi, ms, mc = idx.search_index_batch(h, ids, p, valid_ids=ids)
    pairs = np.zeros((n, p.maxMatches, 2), np.uint32)
    pairs[:, :, 0], pairs[:, :, 1] = mi, ms.view(np.uint32)
    og = np.zeros(n, np.uint32); no = C.c_size_t(0)
    r = ids.copy()
    _lib.check(L.cbh_filter_groups(ids.ctypes.data, pairs.ctypes.data, mc.ctypes.data, n, p.maxMatches, p.minMatches, 1,
                                   ids.ctypes.data, r.ctypes.data, n, og.ctypes.data, C.byref(no)), "fg")
    return h, [(int(j), list(zip(mi[j, :mc[j]].tolist(), ms[j, :mc[j]].tolist()))) for j in og[: no.value]]
gpu_job()
t0 = time.perf_counter()
h_gpu, got = gpu_job()
t_gpu = time.perf_counter() - t0
res["gpu"] = {"similar_wall_s": round(t_gpu, 5), "includes": "hash kernel, index load, searchIndex batch (scan + cut), group filter, result download"}
res["hashes_identical"] = bool((h_gpu == h_cpu).all())
res["groups"] = len(want)
res["similar_results_identical"] = got == want
print(json.dumps(res))
