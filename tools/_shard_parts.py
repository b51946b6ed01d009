"""Development aid: hash / load / sweep times of one rank's share of an 8-way run, separately."""
import os, sys, time
sys.path.insert(0, ".")
import torch, bench
from cbird_amd.dist import HipOps, ShardedDctHashIndex
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ops = HipOps(0); n = 1_000_000; dhts = [1,2,3,4,5,6,7,8]
parts = []
for i0 in range(0, n, 131072):
    i1 = min(n, i0 + 131072)
    parts.append(ops.hash_images(bench.gen_images(torch, dev, i0, i1, n, 1234)).clone())
allh = torch.cat(parts); torch.cuda.synchronize()
R = 8
sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)
a, b = sh.shard_range(n, 0, R)
imgs = bench.gen_images(torch, dev, a, b, n, 1234)
ids = torch.arange(a + 1, b + 1, device=dev, dtype=torch.int32)
def T(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
h = ops.hash_images(imgs)
print("hash", T(lambda: ops.hash_images(imgs)))
print("load", T(lambda: sh.load_shard(h, ids)))
ev = []
print("sweep", T(lambda: sh.similar_sweep(allh, dhts, 8, scan_events=ev)))
torch.cuda.synchronize()
print("scan kernels per sweep", sum(e0.elapsed_time(e1) for _, e0, e1 in ev) / (len(ev) / 8))
print("per-threshold similar()", [round(T(lambda d=d: sh.similar(allh, d, 8)), 2) for d in (2, 8)])
def step():
    hh = ops.hash_images(imgs); sh.load_shard(hh, ids); sh.similar_sweep(allh, dhts, 8)
print("step", T(step))
