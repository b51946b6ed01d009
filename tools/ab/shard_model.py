"""Development aid: what ONE rank of an R-way run of bench.py computes, timed on one GPU.

The driver's 2/4/8-GPU runs cannot be launched from a 1-GPU box; this models a rank's share instead:
hash n/R images, load them as the shard, sweep all n needles over it (bench.py's step without the RCCL
exchange), and separately the post-processing every rank performs on the union of all ranks' records
(sort + cut of the full record set).  Prints per-R: step ms (no exchange), the ideal n=1 time / R, and the
post-processing time per threshold, so the exchange-free scaling ceiling is known before the driver measures.

    python tools/shard_model.py [--images 1000000] [--ranks 1,2,4,8]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1_000_000)
    ap.add_argument("--ranks", type=str, default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch

    import bench
    from cbird_amd.dist import HipOps, ShardedDctHashIndex

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ops = HipOps(0)
    n = args.images
    dhts = [1, 2, 3, 4, 5, 6, 7, 8]
    # all needles: hash every image once, in slices (not timed)
    parts = []
    for i0 in range(0, n, 131072):
        i1 = min(n, i0 + 131072)
        parts.append(ops.hash_images(bench.gen_images(torch, dev, i0, i1, n, 1234)).clone())
    allh = torch.cat(parts)
    torch.cuda.synchronize()
    base = None
    for R in [int(x) for x in args.ranks.split(",")]:
        sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)
        a, b = sh.shard_range(n, 0, R)
        imgs = bench.gen_images(torch, dev, a, b, n, 1234)
        ids = torch.arange(a + 1, b + 1, device=dev, dtype=torch.int32)

        scan_ev = []

        def step():
            with ops.stream_ctx(ops.work_stream()):  # the NULL stream would make every call synchronous
                h = ops.hash_images(imgs)
                sh.load_shard(h, ids)
                return sh.similar_sweep(allh, dhts, 8, scan_events=scan_ev)

        step()
        torch.cuda.synchronize()
        scan_ev.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        scan_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in scan_ev) / args.steps
        if base is None:
            base = ms * R
        print(f"R={R}: rank step {ms:8.2f} ms   ideal {base / R:8.2f} ms   efficiency ceiling {base / R / ms:5.2f}"
              f"   scan kernels {scan_ms:7.2f} ms   records/rank at dht 8: {sh.last_exchange_records}")
        del imgs
    # the replicated post-processing on the union of all records (what each rank runs after the all-gather)
    sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)
    sh.load_shard(allh, torch.arange(1, n + 1, device=dev, dtype=torch.int32))
    blk, _ = sh._buffers(0)
    rec, total = blk[1:], blk[:1]
    for thr in (2, 8):
        total.zero_()
        ops.scan(allh, thr, rec, total)
        nrec = int(total.item())
        keep = rec[:nrec].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            work = keep.clone()
            ops.sort_records(work, nrec, n + 1)
            ops.select(work, nrec, n, 8)
        torch.cuda.synchronize()
        print(f"post-processing of the union (dht {thr}, {nrec} records): "
              f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per threshold")


if __name__ == "__main__":
    main()
