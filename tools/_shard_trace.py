"""Development aid: three steps of one rank's share of an R-way run (argv[1] = R, default 8) on the work stream,
to be run under `rocprofv3 --kernel-trace` (see tools/_trace_tail.py for the timeline print)."""
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)
a, b = sh.shard_range(n, 0, R)
imgs = bench.gen_images(torch, dev, a, b, n, 1234)
ids = torch.arange(a + 1, b + 1, device=dev, dtype=torch.int32)
def step():
    hh = ops.hash_images(imgs); sh.load_shard(hh, ids); sh.similar_sweep(allh, dhts, 8)
with ops.stream_ctx(ops.work_stream()):
    for _ in range(3): step()
torch.cuda.synchronize()
