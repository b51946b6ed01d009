#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = configs[1] (configs[2] when N > 1, same job sharded): 1M synthetic pre-decoded 256x256
8-bit images resident in HBM.  One STEP = one pass of the hot path over that batch:
    build : dctHash64 of every image (k_dcthash_256) -> u64[1M]; all-gather of the hashes when
            N > 1; (re)load of this rank's DctHashIndex shard
    find  : all-pairs DctHashIndex find (1M needles x 1M slots) for every dht in 1..8
            (k_hamm64_mfma + record exchange/sort/select, maxMatches-style cut at k=8; the
            post-processing of one threshold overlaps the scan of the next)
value = 64-bit Hamming comparisons/s over the whole step (8 x 10^12 comparisons per step; the
build time is inside the denominator); images hashed/s is reported next to it.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# k_hamm64_mfma: one 64-bit comparison = one K=64 sign dot product = 64 multiply-adds on the FP4
# matrix path; dense FP4 MFMA peak from the same guide (~10 PF; its own micro-benchmark reaches 9.1)
FP4_PEAK_TFLOPS = 10000.0
FLOP_PER_CMP = 128.0
W = H = 256
# The job is a pure function of (--images, --seed): every image is drawn from (seed, global index), so the match count
# per threshold does not depend on the number of GPUs or on how the index is sharded.  Counts of the default job and of
# the 80 k job tools/first_contact.sh runs at N = 2/4/8 (records before the maxMatches cut, self matches included),
# taken from N = 1 runs whose full result lists equal the real VP-tree's (profiles/r04_bench_1gpu.json: full_identity).
EXPECTED_MATCHES = {
    (1_000_000, 1234): {1: 1173680, 2: 1197740, 3: 1199826, 4: 1199976, 5: 1200020, 6: 1200182, 7: 1200936, 8: 1204352},
    (80_000, 1234): {1: 93880, 2: 95842, 3: 95984, 4: 96000, 5: 96000, 6: 96000, 7: 96000, 8: 96010},
}
# (images, seed) -> sha256 of the u64[images] hashes (little-endian), from the same runs: all of them equal to the CPU
# port's (hash_identity).  The images come from torch's float cos on the device, so this pins the torch build as well.
EXPECTED_HASH_SHA256 = {
    (1_000_000, 1234): "ae7b89309a8421bdcab29137646136129fcb4526432740ff084bbdcb5eb0942e",
    (80_000, 1234): "2dd2076739283ba0b8b8c86577845060bfdd32172b07217f5896cf8059a44532",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=1_000_000, help="index size (BASELINE: 1M)")
    ap.add_argument("--dht", type=str, default="1,2,3,4,5,6,7,8", help="thresholds swept per step")
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline time budget")
    ap.add_argument("--no-video", action="store_true", help="skip the configs[4] leg (DctVideoIndex sharded by video)")
    ap.add_argument("--video-clips", type=int, default=10_000)
    ap.add_argument("--no-orb", action="store_true", help="skip the configs[3] leg (CvFeaturesIndex sharded by image)")
    ap.add_argument("--orb-images", type=int, default=100_000, help="configs[3]: images x 500 descriptors of 256 bits")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="skip the single-process leg (ONE DctHashIndex handle over all GPUs: cbh_idx64_create_sharded)")
    ap.add_argument("--no-features", action="store_true",
                    help="skip the indexer-stage leg (ORB, ColorDescriptor::create; reported beside the contract line)")
    return ap.parse_args()


# ---- synthetic pre-decoded images, generated on the device ----------------------------------
def _mix(x):
    """32-bit integer hash on int64 tensors (values kept in [0, 2^32))"""
    m = 0xFFFFFFFF
    x = ((x >> 16) ^ x) * 0x45D9F3B & m
    x = ((x >> 16) ^ x) * 0x45D9F3B & m
    return (x >> 16) ^ x


def _unit(x):
    return _mix(x).to(__import__("torch").float32) * (1.0 / 4294967296.0)


def gen_images(torch, dev, i0, i1, n_total, seed):
    """u8 [i1-i0, 256, 256]: SURVEY.md 8(d) recipe -- smooth field of 8 low-frequency cosines with
    random phase/amplitude + uniform noise in [-8, 8]; the last 10 % of the index are near-duplicates
    (extra noise, sigma ~ 2, and a brightness shift in [-5, 5]) of earlier images.  Every image is a
    pure function of (seed, global index), so any sharding produces the same data set."""
    n_base = max(1, int(n_total * 0.9))
    out = torch.empty((i1 - i0, H, W), dtype=torch.uint8, device=dev)
    yy = torch.arange(H, device=dev, dtype=torch.float32).view(1, H, 1) * (6.283185307179586 / H)
    xx = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, W) * (6.283185307179586 / W)
    pix = torch.arange(H * W, device=dev, dtype=torch.int64).view(1, H, W)
    chunk = 2048
    for c0 in range(i0, i1, chunk):
        c1 = min(i1, c0 + chunk)
        idx = torch.arange(c0, c1, device=dev, dtype=torch.int64)
        src = torch.where(idx < n_base, idx, (idx * 2654435761 + seed) % n_base)
        key = (src * 1000003 + seed * 7919) & 0xFFFFFFFF
        f = torch.full((c1 - c0, H, W), 128.0, device=dev, dtype=torch.float32)
        for k in range(8):
            fx = (_unit(key + 4 * k + 0) * 8.0 - 4.0).view(-1, 1, 1)
            fy = (_unit(key + 4 * k + 1) * 8.0 - 4.0).view(-1, 1, 1)
            amp = (_unit(key + 4 * k + 2) * 36.0 + 4.0).view(-1, 1, 1)
            ph = (_unit(key + 4 * k + 3) * 6.283185307179586).view(-1, 1, 1)
            f += amp * torch.cos(fx * xx + fy * yy + ph)
        f += _unit(key.view(-1, 1, 1) * 65537 + pix) * 16.0 - 8.0
        dup = (idx >= n_base).view(-1, 1, 1)
        if bool(dup.any()):
            k2 = (idx * 40503 + 17).view(-1, 1, 1)
            tri = (_unit(k2 * 65537 + pix) + _unit(k2 * 65537 + pix + 0x5BD1E995) - 1.0) * 4.9
            shift = torch.floor(_unit(idx * 31 + 5) * 11.0).view(-1, 1, 1) - 5.0
            f = torch.where(dup, f + tri + shift, f)
        out[c0 - i0: c1 - i0] = f.round_().clamp_(0, 255).to(torch.uint8)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime

    import torch
    import torch.distributed as dist

    from cbird_amd.dist import HipOps, ShardedDctHashIndex

    # CBH_BENCH_SHARE_GPU=1 (development aid): all ranks on cuda:0 with gloo, to exercise the N>1 code path on
    # a single-GPU box.  Never set by the driver; RCCL ("nccl") is the real transport.
    share = os.environ.get("CBH_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CBH_BENCH_FORCE_DIST=1 (development aid): initialise RCCL and run the per-threshold all-gather at world size 1
    # too, so that a one-GPU box exercises the transport of the N > 1 path.  Never set by the driver.
    force = os.environ.get("CBH_BENCH_FORCE_DIST") == "1"
    if force:
        os.environ["CBH_DIST_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # 120 s: a rank that never arrives makes the others exit non-zero instead of holding the node
        tmo = datetime.timedelta(seconds=int(os.environ.get("CBH_BENCH_INIT_TIMEOUT", "120")))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    done_group = None
    if dist.is_initialized() and world > 1:
        # At the end rank 0 runs a child process beside the line (up to 240 s) while the others wait: that wait happens on
        # a gloo group with its own long timeout, not inside an RCCL collective under the 120 s watchdog.  Made HERE, before
        # anything is measured or printed: gloo announces every rank's connections on stdout when a group is made, and
        # those fragments must not meet rank 0's JSON line.
        done_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
        dist.barrier(group=done_group)
        sys.stdout.flush()
    ops = HipOps(local_rank)  # raises without libcbird_hip.so / a gfx950 device: no fallback
    sh = ShardedDctHashIndex(ops, record_capacity=1 << 22)

    n = args.images
    dhts = [int(x) for x in args.dht.split(",") if x]
    a, b = sh.shard_range(n, rank, world)
    imgs = gen_images(torch, dev, a, b, n, args.seed)
    ids = torch.arange(a + 1, b + 1, device=dev, dtype=torch.int32)  # mediaIds = SQLite rowids
    torch.cuda.synchronize()

    ev = lambda: torch.cuda.Event(enable_timing=True)
    hash_ev, scan_ev, find_ev = [], [], []
    state = {}

    work = ops.work_stream()  # not torch's default (NULL) stream: the C-ABI calls are synchronous on that one

    def step(record: bool):
        with ops.stream_ctx(work):
            _step(record)

    def _step(record: bool):
        e0, e1 = ev(), ev()
        e0.record()
        h_local = ops.hash_images(imgs)
        e1.record()
        allh = sh.gather_hashes(h_local, n)
        sh.load_shard(h_local, ids)
        if record:
            hash_ev.append((e0, e1))
        # the whole sweep, software-pipelined: the scan of threshold i+1 runs while the records of threshold i
        # are exchanged (N > 1), sorted and cut (cbird_amd.dist.ShardedDctHashIndex.similar_sweep)
        res = sh.similar_sweep(allh, dhts, args.topk, scan_events=scan_ev if record else None,
                               find_events=find_ev if record else None)
        state.update(res)
        state["hashes"] = allh

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    if args.warmup:  # exchange blocks sized from what the warm-up produced (1.5 x the fullest rank), then one more
        sh.fit_capacity()  # untimed pass so that the timed steps allocate nothing
        step(False)
        fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel figures from the events recorded inside the timed region (this rank) --------
    torch.cuda.synchronize()
    hash_ms = sum(e0.elapsed_time(e1) for e0, e1 in hash_ev) / max(1, len(hash_ev))
    scans = {}
    for dht, s0, s1 in scan_ev:
        scans.setdefault(dht, []).append(s0.elapsed_time(s1))
    finds = {}
    for dht, f0, f1, nrec in find_ev:
        finds.setdefault(dht, []).append((f0.elapsed_time(f1), int(nrec)))
    shard_n = b - a
    sweep = []
    for dht in dhts:
        sm = sum(scans[dht]) / len(scans[dht])
        fm = sum(x for x, _ in finds[dht]) / len(finds[dht])
        # find_latency_ms: scan start -> cut results ready; the sweep is pipelined, so it spans the next
        # threshold's scan as well and is not additive
        sweep.append({"dht": dht, "scan_kernel_ms": round(sm, 3), "find_latency_ms": round(fm, 3),
                      "scan_cmp_per_s": shard_n * n / sm * 1e3, "matches": finds[dht][0][1]})
    # the matrix-core scan has two shapes: k_hamm64_mfma3 (64-bit dot products) and PRE (32-bit prefilter on lo ^ hi + exact
    # re-check of the candidates).  The library picks per launch, from the candidate rate of that launch's data
    # (hamm64_mfma.hip: pick_pre); "scan_pre_mask" reads back which thresholds took the prefilter in the timed steps,
    # which executes half the multiply-adds per comparison.  `roofline` prices whichever of the two takes more of the
    # step's time; both are reported (roofline_full3, roofline_pre) with the flops they really issue.
    import ctypes as _ct

    from cbird_amd import _lib as _cl

    _v = _ct.c_longlong(0)
    pre_mask = int(_v.value) if _cl.lib().cbh_get_tuning(b"scan_pre_mask", _ct.byref(_v)) == 0 else 0
    pre = [d for d in dhts if d < 64 and (pre_mask >> d) & 1]
    full = [d for d in dhts if d not in pre] or dhts
    pre = [d for d in pre if d not in full]
    probes = int(_v.value) if _cl.lib().cbh_get_tuning(b"scan_probes", _ct.byref(_v)) == 0 else None
    scan_ms_avg = sum(sum(scans[d]) for d in full) / sum(len(scans[d]) for d in full)
    scan_bytes = 8.0 * shard_n * n  # 8 algorithmic bytes per 64-bit comparison
    scan_gbs = scan_bytes / (scan_ms_avg * 1e-3) / 1e9
    scan_tflops = FLOP_PER_CMP * shard_n * n / (scan_ms_avg * 1e-3) / 1e12
    hash_bytes = (W * H + 8.0) * shard_n  # 65 536 B read + 8 B written per image
    hash_gbs = hash_bytes / (hash_ms * 1e-3) / 1e9

    steps = args.steps
    cmp_per_step = float(n) * float(n) * len(dhts)
    value = cmp_per_step * steps / dt
    result = {
        "metric": "Hamming comparisons/sec + images hashed/sec, 1M-image index, 1/2/4/8 MI355X",
        "value": value,
        "unit": ("pair-equivalents/s, exact results, whole step (build + dht sweep): every (needle, slot) pair of every threshold "
                 "counts once; thresholds the library routes to the 32-bit fold prefilter (see kernel_choice) compare "
                 "lo ^ hi and re-check the candidates on all 64 bits, the others compare all 64 bits of every pair "
                 "(full_64bit_compare_rate_per_s)"),
        "kernel_choice": {"prefilter_dht": pre, "full_64bit_dht": full, "chosen_by": "candidate rate of each launch "
                          "(k_fold_probe: 2048 x 2048 sampled pairs) against scan_pre_rate_e9",
                          "probes_run_by_the_library_so_far": probes},
        "images_hashed_per_s": shard_n * world / (hash_ms * 1e-3),
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64 hashes as 64 x FP4 signs, f32 accumulate (find, exact) / u8+f32 (hash)",
        "data": "synthetic",
        "config": {
            "workload": ("configs[1]: 1M pre-decoded 256x256 images, DctHashIndex build+find on 1xMI355X, "
                         "dht=1..8 sweep" if world == 1 else
                         "configs[2]: 1M-image DctHashIndex sharded across the GPUs, RCCL all-gather of "
                         "candidates (same step as configs[1]: build + dht=1..8 sweep)"),
            "images": n, "image_size": [W, H], "needles": n, "dht": dhts, "max_per_query": args.topk,
            "parallelism": f"haystack row-sharded x{world}, needles replicated",
        },
        "dht_sweep": sweep,
        # the headline counts len(dht) x N^2 pair-equivalents per step; the thresholds of kernel_choice.prefilter_dht run the
        # 32-bit prefilter (dot products of lo ^ hi + exact 64-bit re-checks of the candidates), the others compare all 64
        # bits of every pair
        "full_64bit_compare_rate_per_s": shard_n * n / (scan_ms_avg * 1e-3) * world,
        "roofline_full3": {
            "kernel": "k_hamm64_mfma3 (64-bit sign dot products, 3 needle tiles per accumulator: dht %s)" % ",".join(map(str, full)), "bound": "mfma", "achieved": scan_tflops, "peak": FP4_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": scan_tflops / FP4_PEAK_TFLOPS, "traffic": None,
            "avg_launch_ms": scan_ms_avg, "launches_per_step": len(full),
            "algorithmic_flop_per_launch": FLOP_PER_CMP * shard_n * n,
            "algorithmic_bytes_per_launch": scan_bytes, "hbm_equivalent_GBps": scan_gbs,
            "note": ("each comparison is a 64-term +-1 dot product (64 - 2*hamm64) on v_mfma_scale_f32_32x32x64_"
                     "f8f6f4 with FP4 operands: 128 FLOP.  SURVEY 8(d)'s 8 algorithmic bytes per comparison "
                     "are kept as hbm_equivalent_GBps; the kernel is not memory-bound (traffic = PMC HBM bytes "
                     "per launch), the binding unit is the matrix core.  avg_launch_ms brackets the needle "
                     "expansion kernel + the scan kernel of one launch."),
        },
        "roofline_pre": None if not pre else (lambda pre_ms: {
            "kernel": "k_hamm64_mfma<true> (PRE: dht %s)" % ",".join(map(str, pre)), "bound": "mfma",
            "binding_unit": "the SIMD issue port, shared by the MFMAs (~24 of 32 cycles each) and the VALU flag reduction",
            "avg_launch_ms": pre_ms, "launches_per_step": len(pre),
            "algorithmic_flop_per_launch": 64.0 * shard_n * n,
            "achieved": 64.0 * shard_n * n / (pre_ms * 1e-3) / 1e12,
            "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 64.0 * shard_n * n / (pre_ms * 1e-3) / 1e12 / FP4_PEAK_TFLOPS,
            "traffic": None,
            "per_dht_ms": {str(d): round(sum(scans[d]) / len(scans[d]), 3) for d in pre},
            "note": ("32-bit prefilter on lo ^ hi (a lower bound on the 64-bit distance): one K=64 MFMA covers 2 x 1024 "
                     "32-term dot products = 64 FLOP per comparison, the work this kernel's algorithm needs (the exact "
                     "re-check touches 1e-6 .. 6e-5 of the pairs); priced at 128 FLOP per comparison it would read "
                     "2x this fraction.  Bound by the VALU flag reduction that shares the issue port with the MFMAs "
                     "(4.25 VALU instructions per MFMA: one v_or3_b32 per two result registers)."),
        })(sum(sum(scans[d]) for d in pre) / sum(len(scans[d]) for d in pre)),
        "roofline_hash": {
            "kernel": "k_dcthash_256_band (horizontal 7-tap sums as i8 MFMAs, one add + half an fma per pixel)",
            "bound": "hbm", "achieved": hash_gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": hash_gbs / HBM_PEAK_GBS, "traffic": None,
            "avg_launch_ms": hash_ms, "algorithmic_bytes_per_launch": hash_bytes,
        },
    }
    # `roofline` = the scan kernel that takes more of the step (launches x average duration)
    rf3, rpre = result["roofline_full3"], result["roofline_pre"]
    dominant = rpre if rpre and rpre["avg_launch_ms"] * rpre["launches_per_step"] > rf3["avg_launch_ms"] * rf3["launches_per_step"] else rf3
    result["roofline"] = dict(dominant)
    result["roofline"]["dominant_by"] = "launches_per_step x avg_launch_ms"
    # HBM traffic is NOT measured by this run (PMC counters need their own rocprofv3 passes: tools/profile_bench.sh).
    # The per-launch figures of the committed PMC passes are attached only to the configuration they were taken on
    # (1M images, one GPU) and are labelled as such; any other --images / --gpus reports traffic = null.
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and n == 1_000_000 and world == 1:
        try:
            t = json.load(open(pmc))
            src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at 1M images, N=1; not measured in this run)"
            for key_, rf_ in (("k_hamm64_mfma", rf3), ("k_hamm64_mfma_pre", rpre)):
                if rf_ is not None and t.get(key_) is not None:
                    rf_["traffic"] = t.get(key_)
                    rf_["traffic_source"] = src
            result["roofline"]["traffic"] = dominant.get("traffic")
            if dominant.get("traffic") is not None:
                result["roofline"]["traffic_source"] = src
            result["roofline_hash"]["traffic"] = t.get("k_dcthash_256")
            result["roofline_hash"]["traffic_source"] = src
        except Exception:
            pass

    if not args.no_video:
        result["configs4_video"] = video_leg(args, torch, dist, dev, local_rank, rank, world, share)
    if not args.no_orb:
        result["configs3_cvfeatures"] = cvfeatures_leg(args, torch, dist, dev, local_rank, rank, world, share)
    if rank == 0 and world == 1:
        # outside the timed region, for the record: the same launches on the popcount (VALU) kernel
        # k_hamm64_scan that the matrix-core kernel replaced (identical records; tests/test_gpu_hamm.py)
        import ctypes as C

        blk, _ = sh._buffers(0)
        rec, total = blk[1:], blk[:1]
        ms = C.c_float(0)
        pop = {}
        ops.L.cbh_set_tuning(b"scan_mfma", 0)
        try:
            for d in sorted({dhts[0], dhts[len(dhts) // 2], dhts[-1]}):
                total.zero_()
                rc = ops.L.cbh_idx64_time_scan_dev(ops.index.handle, state["hashes"].data_ptr(), n, d,
                                                   rec.data_ptr(), rec.numel(), total.data_ptr(), 1, C.byref(ms))
                if rc == 0:
                    pop[str(d)] = round(ms.value, 3)
        finally:
            ops.L.cbh_set_tuning(b"scan_mfma", 1)
        result["popcount_kernel_scan_ms"] = pop
    if world == 1 and all(1 <= d <= 8 for d in dhts) and float(n) * float(n) * 8.8e-12 >= 1.0:  # (the join is only asked where
        # a scan takes >= 1 ms)
        # Beside the contract line, never part of `value`: the same sweep with "scan_mfma" 3 -- thresholds <= 8 answered by the
        # bucketed join (hamm64_join.hip: multi-index hashing; only pairs that share one of max(4, dht) chunk values are
        # compared, exact results) wherever its candidate count beats the exhaustive scan.  The headline metric counts
        # comparisons, so the line above is measured with every pair compared (the library's default); this is what the same
        # answers cost when they are not.
        torch.cuda.synchronize()
        jl = {"what": "dht sweep with the bucketed join allowed (scan_mfma 3): identical results, comparisons avoided, not made"}
        ops.L.cbh_set_tuning(b"scan_mfma", 3)
        try:
            v = C.c_longlong(0)
            ops.L.cbh_get_tuning(b"scan_joins", C.byref(v))
            j0 = v.value
            with ops.stream_ctx(work):
                res = sh.similar_sweep(state["hashes"], dhts, args.topk)
                torch.cuda.synchronize()
                times = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    sev = []
                    res = sh.similar_sweep(state["hashes"], dhts, args.topk, scan_events=sev)
                    torch.cuda.synchronize()
                    times.append(((time.perf_counter() - t0) * 1e3, {d: s0.elapsed_time(s1) for d, s0, s1 in sev}))
            ops.L.cbh_get_tuning(b"scan_joins", C.byref(v))
            best = min(times, key=lambda x: x[0])
            jl.update({"sweep_ms": round(best[0], 3), "scan_ms_per_dht": {str(d): round(ms, 3) for d, ms in best[1].items()},
                       "calls_answered_by_the_join": int(v.value - j0), "calls": 4 * len(dhts),
                       "sweep_ms_exhaustive_in_the_timed_steps": round(sum(s_["scan_kernel_ms"] for s_ in sweep), 3),
                       "results_equal_the_exhaustive_sweep": all(
                           bool(torch.equal(res[d][0], state[d][0]) and torch.equal(res[d][1], state[d][1])
                                and torch.equal(res[d][2], state[d][2])) for d in dhts)})
        except Exception as e:  # a leg beside the contract line must never take the line down
            jl["error"] = repr(e)[:300]
        finally:
            ops.L.cbh_set_tuning(b"scan_mfma", 1)
        result["bucketed_join"] = jl
    exp = EXPECTED_MATCHES.get((n, args.seed))
    got_matches = {s_["dht"]: s_["matches"] for s_ in sweep}
    result["matches_expected"] = None if exp is None or any(d not in exp for d in dhts) else \
        all(got_matches[d] == exp[d] for d in dhts)
    if exp is not None and result["matches_expected"] is False:
        result["matches_expected_table"] = {str(d): exp.get(d) for d in dhts}
    if rank == 0 and (n, args.seed) in EXPECTED_HASH_SHA256:  # every rank holds all hashes (gather_hashes)
        import hashlib

        result["hashes_expected"] = hashlib.sha256(
            state["hashes"].cpu().numpy().tobytes()).hexdigest() == EXPECTED_HASH_SHA256[(n, args.seed)]
    if dist.is_initialized():
        # the transport's own view of the job: how many ranks the communicator spans (a sum of ones through it) and
        # which library carried it
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(ones)
        backend = dist.get_backend()
        result["collective"] = {"backend": backend, "communicator_ranks": int(ones.item()),
                                "world_size": dist.get_world_size()}
        if backend == "nccl":
            try:
                result["collective"]["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:  # the version query is informative only
                result["collective"]["rccl_version"] = repr(e)[:80]
    if world > 1:
        # what every rank actually holds: its share of the images / index slots / descriptor rows, never another's
        mine = {"rank": rank, "device": local_rank, "images": b - a, "image_bytes": int(imgs.numel()),
                "index_slots": int(ops.index.count()),
                "orb_rows": result.get("configs3_cvfeatures", {}).get("rows_this_rank"),
                "device_bytes_allocated_by_torch": int(torch.cuda.memory_allocated(dev))}
        parts = [None] * world
        dist.all_gather_object(parts, mine)
        result["per_rank_residency"] = parts
    if rank == 0 and not args.no_sharded_leg:
        result["single_process_sharded"] = sharded_leg(args, world, local_rank, share)
    if rank == 0 and world == 1 and not args.no_features:
        result["indexer_stages"] = features_leg(torch, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, torch, imgs, state, n, dhts, sh, ops)
        # north_star's acceptance line, as fields of the contract line: every needle's full (mediaId, distance) list
        # equal to the real VP-tree's, every hash equal to the CPU port's
        result["full_identity"] = result["cpu_baseline"].get("full_identity")
        result["hash_identity"] = result["cpu_baseline"].get("hash_identity")
    def _sync():
        if dist.is_initialized():
            if done_group is not None:
                dist.barrier(group=done_group)
            else:
                dist.barrier()

    # The JSON line goes out when every rank has emptied its C stdio buffer (librccl announces its version there when the
    # first communicator is made; on a pipe that text would otherwise appear when the processes exit, after the line),
    # and nothing is printed after it.
    _flush_c_stdio()
    _sync()
    if rank == 0:
        # (N > 1: at a line start whatever a communication library left on the line)
        sys.stdout.write(("\n" if world > 1 else "") + json.dumps(result) + "\n")
        sys.stdout.flush()
    _sync()
    if dist.is_initialized():
        dist.destroy_process_group()
    _flush_c_stdio()


def _flush_c_stdio():
    import ctypes

    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def sharded_leg(args, world, local_rank, share):
    """The drop-in's own multi-GPU form, beside the torchrun harness above: ONE process, ONE DctHashIndex handle over
    the GPUs (cbh_idx64_create_sharded: per-shard scans, ncclAllGather of the per-device blocks through librccl called
    directly, merge and cut on the first device) -- cbird registers each index once and fans find() out from its own
    threads (src/engine.cpp:38-45, src/database.cpp:1400-1432).  Runs tools/sharded_leg.py as a child process of rank
    0 with its own timeout after the timed region (the other ranks idle at the final barrier): with N > 1 over the N
    GPUs of the job, with N = 1 over 8 logical shards on the one GPU -- once with the library's default exchange (copies:
    "sharded", what `speedup_vs_one_device` is quoted on) and once with their block sent through ncclAllGather
    ("sharded_rccl": the transport, on a box that has one GPU).  The same all-pairs dht sweep, needles and results
    resident; never part of `value`."""
    import subprocess

    if world > 1 and not share:
        cmd = ["--mask", hex((1 << world) - 1), "--per-device", "1", "--exchange", "both"]
    else:
        cmd = ["--mask", hex(1 << local_rank), "--per-device", "8", "--force-rccl", "--exchange", "both"]
    cmd = [sys.executable, os.path.join(ROOT, "tools", "sharded_leg.py"), "--images", str(args.images), "--dht", args.dht,
           "--topk", str(args.topk)] + cmd
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK",
              "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE"):
        env.pop(k, None)
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"rc {out.returncode}: " + (out.stderr or out.stdout)[-400:]}
    except Exception as e:  # a leg beside the contract line must never take the line down
        return {"error": repr(e)[:400]}


def features_leg(torch, dev):
    """The indexer's feature stages, reported beside the contract line and never part
    of `value`: ORB detect + describe on 2048 resident 400x300 grey images, ColorDescriptor::create on 4096 resident
    256x192 BGR images (its clustering runs one lane per image: the rate grows with the batch), Media::makeVideoIndex on
    1024 resident letterboxed 1080p frames (row a15).  Throughput only -- the
    parity of both lives in the -m gpu tests and in smoke()."""
    import ctypes as C

    import numpy as np

    from cbird_amd import _lib, orb

    L = _lib.lib()
    rng = np.random.default_rng(1)
    out = {}

    def scene(w, h, ch):
        shape = (h, w) if ch == 1 else (h, w, ch)
        img = np.full(shape, 128, np.int32)
        for _ in range((w * h) // 1000):
            x, y = int(rng.integers(0, w - 4)), int(rng.integers(0, h - 4))
            img[y: y + int(rng.integers(4, h // 4)), x: x + int(rng.integers(4, w // 4))] = \
                rng.integers(0, 256, None if ch == 1 else ch)
        return (img + rng.integers(-4, 5, shape)).clip(0, 255).astype(np.uint8)

    stream = torch.cuda.Stream()
    # ---- ORB
    n, w, h, cap = 2048, 400, 300, 512
    orb.set_pattern(orb.synthetic_pattern())  # a stand-in for OpenCV's learned test pairs (cbird_amd/orb.py)
    base = np.stack([scene(w, h, 1) for _ in range(32)])
    d = torch.from_numpy(np.concatenate([base] * (n // 32))).to(dev)
    off = np.arange(n, dtype=np.uint64) * np.uint64(w * h)
    ww, hh = np.full(n, w, np.uint32), np.full(n, h, np.uint32)
    d_kp = torch.zeros((n, cap, 6), dtype=torch.float32, device=dev)
    d_after = torch.zeros((n, cap, 2), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(n, dtype=torch.int32, device=dev)

    def run_orb():
        _lib.check(L.cbh_orb_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data, ww.ctypes.data, 400, cap,
                                 d_kp.data_ptr(), d_after.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), 0,
                                 C.c_void_p(stream.cuda_stream)), "orb")

    with torch.cuda.stream(stream):
        run_orb()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(3):
            run_orb()
        e1.record(stream)
        stream.synchronize()
    ms = e0.elapsed_time(e1) / 3
    nk = int(d_cnt.sum().item())
    out["orb_detect_describe"] = {"workload": f"{n} grey images {w}x{h} resident, 400 keypoints asked", "ms": round(ms, 3),
                                  "images_per_s": n / ms * 1e3, "keypoints_per_s": nk / ms * 1e3}
    del d, d_kp, d_after, d_desc
    # ---- ColorDescriptor::create
    n, w, h = 4096, 256, 192
    base = np.stack([scene(w, h, 3) for _ in range(32)])
    d = torch.from_numpy(np.concatenate([base] * (n // 32))).to(dev)
    off = np.arange(n, dtype=np.uint64) * np.uint64(w * h * 3)
    ww, hh, ss = np.full(n, w, np.uint32), np.full(n, h, np.uint32), np.full(n, 3 * w, np.uint32)
    d_cd = torch.zeros((n, 258), dtype=torch.uint8, device=dev)
    d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)

    def run_cd():
        _lib.check(L.cbh_color_descriptors_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data,
                                               ss.ctypes.data, 3, d_cd.data_ptr(), d_ok.data_ptr(), 0,
                                               C.c_void_p(stream.cuda_stream)), "color_descriptors")

    run_cd()
    t0 = time.perf_counter()
    run_cd()  # returns when the descriptors are complete
    dt = time.perf_counter() - t0
    out["color_descriptor_create"] = {"workload": f"{n} BGR images {w}x{h} resident", "s": round(dt, 4),
                                      "images_per_s": n / dt, "descriptors": int(d_ok.sum().item())}
    del d, d_cd, d_ok
    # ---- Media::makeVideoIndex: letterboxed 1080p frames resident (a hardware decoder's output), pushed 256 at a time
    # (the first chunk of a letterboxed video is hashed twice: the kept region is only known after it)
    from cbird_amd.video import VideoIndexer

    n, w, h, bar = 1024, 1920, 1080, 140
    g = torch.Generator(device=dev).manual_seed(7)
    frames = torch.full((n, h, w), 16, dtype=torch.uint8, device=dev)
    frames += torch.randint(0, 3, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
    body = torch.randint(40, 256, (1, h - 2 * bar, w), dtype=torch.uint8, device=dev, generator=g)
    frames[:, bar:h - bar, :] = body
    for k in range(0, n, 64):  # a scene cut every 64 frames
        frames[k:, bar:h - bar, :] = torch.roll(frames[k:, bar:h - bar, :], shifts=k * 131 + 7, dims=2)
    torch.cuda.synchronize()
    best, stored = None, 0
    for _ in range(3):
        ix = VideoIndexer(threshold=8)
        t0 = time.perf_counter()
        for i in range(0, n, 256):
            ix.push(frames[i:i + 256])
        stored = len(ix.finish().frames)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["video_index"] = {"workload": f"{n} grey frames {w}x{h} with {bar}-row bars resident, chunks of 256, "
                                      "autocrop(20) + dctHash64 + near-frame filter (threshold 8)",
                          "s": round(best, 5), "frames_per_s": n / best, "GBps": n * w * h / best / 1e9,
                          "frames_stored": stored}
    del frames
    # ---- dctHash64 at the sizes the indexer feeds (not 256 x 256): resident batches of 4 GB, the general-geometry kernels
    # (k_blur_area_regs + k_tiles_hash2 / k_tile_hash); kernel time from the library's own events (cbh_time_dcthash_dev)
    geo = []
    msv = C.c_float(0)
    for (w, h) in ((400, 300), (533, 400), (640, 480), (1024, 768), (1920, 1080), (4000, 3000)):
        n = max(64, int(4e9 // (w * h)))
        imgs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
        hashes = torch.empty(n, dtype=torch.int64, device=dev)
        best = None
        for _ in range(2):
            _lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n, w, h, w, w * h, hashes.data_ptr(), 0, 3, C.byref(msv)), "hash")
            best = msv.value if best is None else min(best, msv.value)
        geo.append({"w": w, "h": h, "images": n, "ms": round(best, 3), "images_per_s": n / best * 1e3,
                    "GBps": n * w * h / best * 1e-6, "hbm_frac": n * w * h / best * 1e-6 / 8000.0})
        del imgs, hashes
    out["dct_hash_by_geometry"] = geo
    return out


def video_leg(args, torch, dist, dev, local_rank, rank, world, share):
    """BASELINE configs[4], reported beside the contract line (never part of `value`): 10k synthetic clips x 300 frame
    hashes in a DctVideoIndex sharded BY VIDEO over the ranks (cbird_amd.dist.ShardedDctVideoIndex);
    2000 needle clips (the 1 % planted sub-clips among them) replicated, one batched findVideo per rank -- scan,
    closest frame per video and adjacency scoring on the device -- and ONE all-gather of the final matches.
    dht 5, vtrim 0, vfm 30, vfn 60, exact search (vradix 0).  Time = max over ranks, barrier on both sides."""
    from cbird_amd import synth_video
    from cbird_amd.dist import ShardedDctVideoIndex
    from cbird_amd.video import DctVideoIndex, VideoIndex, VideoSearchParams

    clips = synth_video.make_clips_fast(args.video_clips, 300, seed=args.seed, subclip_frac=0.01, max_gap=8)

    class M:
        pass

    media = []
    for i, (f, h) in enumerate(clips):
        m = M()
        m.id, m.path, m.videoIndex, m.dctHash = i + 1, "", VideoIndex(f, h), 0
        media.append(m)
    sv = ShardedDctVideoIndex(lambda: DctVideoIndex(local_rank),
                              device=dev if (dist.is_initialized() and not share) else None)
    t0 = time.perf_counter()
    sv.add(media)
    p = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=30, minFramesNear=60)
    needles = media[-min(2000, len(media)):]
    sv.find_videos_batch(needles[:64], p)  # builds the shard's search structure, warms the kernels
    t_build = time.perf_counter() - t0

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    best, hits = None, 0
    for _ in range(3):
        fence()
        t0 = time.perf_counter()
        res = sv.find_videos_batch(needles, p)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        best = dt if best is None else min(best, dt)
        hits = sum(len(r) for r in res)
    entries = sum(len(f) for f, _ in clips)  # (a few low-detail hashes are filtered at insert; within 0.1 %)
    nframes = sum(len(m.videoIndex.frames) for m in needles)
    return {"workload": "configs[4]: DctVideoIndex, %d clips x 300 frame hashes, %d needle clips batched, dht 5, "
                        "vfm 30, vfn 60, vradix 0" % (len(clips), len(needles)),
            "parallelism": f"sharded by video x{world}, needles replicated, one all-gather of final matches",
            "seconds": best, "needle_clips_per_s": len(needles) / best,
            "cmp_per_s": float(entries) * nframes / best, "matches": hits, "build_seconds_this_rank": t_build}


def cvfeatures_leg(args, torch, dist, dev, local_rank, rank, world, share):
    """BASELINE configs[3], reported beside the contract line (never part of `value`): 100k images x 500 synthetic
    256-bit descriptors in a CvFeaturesIndex sharded BY IMAGE over the ranks (cbird_amd.dist.ShardedCvFeaturesIndex,
    SURVEY.md 8e); 64 needle images (their own descriptors, a third of them perturbed by two bits) replicated: per
    rank the exact k = 10 nearest rows under cvThresh 25 of every needle descriptor on the matrix cores, ONE
    all-gather of the fixed-size candidate tables, k-way merge and the reference's scoring on every rank.
    Time = max over ranks, barrier on both sides.  Every rank draws the descriptors of image i from the same
    counter-based stream, so no rank materialises another rank's rows."""
    import numpy as np

    from cbird_amd.cvfeatures import CvFeaturesIndex
    from cbird_amd.dist import ShardedCvFeaturesIndex, ShardedDctHashIndex
    from cbird_amd.index import SearchParams

    n_img, per, chunk = int(args.orb_images), 500, 1000

    def rows_of_chunk(c):  # images [c * chunk, (c + 1) * chunk)
        return np.random.default_rng([args.seed, 3, c]).integers(0, 256, (chunk * per, 32), dtype=np.uint8)

    class Rows:  # descriptors of one image, drawn when (and only where) they are needed
        cache = {}

        def __init__(self, i):
            self.i = i

        def __len__(self):
            return per

        def __array__(self, dtype=None, copy=None):
            c = self.i // chunk
            if c not in Rows.cache:
                Rows.cache.clear()
                Rows.cache[c] = rows_of_chunk(c)
            k = self.i - c * chunk
            return Rows.cache[c][k * per:(k + 1) * per]

    class M:
        pass

    media = []
    for i in range(n_img):
        m = M()
        m.id, m.path, m.keyPointDescriptors = i + 1, "", Rows(i)
        media.append(m)
    sc = ShardedCvFeaturesIndex(lambda: CvFeaturesIndex(local_rank),
                                device=dev if (dist.is_initialized() and not share) else None)
    t0 = time.perf_counter()
    sc.add(media)
    n_needles = min(64, n_img)
    first = rows_of_chunk(0)
    needles = []
    for i in range(n_needles):
        m = M()
        d = first[i * per:(i + 1) * per].copy()
        d[::3, 5] ^= 0x11
        m.id, m.path, m.keyPointDescriptors = i + 1, "", d
        needles.append(m)
    Rows.cache.clear()
    p = SearchParams(algo=SearchParams.AlgoCVFeatures, cvThresh=25, minMatches=1, maxMatches=10)
    sc.find_batch(needles[:2], p, knn=10)
    t_build = time.perf_counter() - t0

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    best, hits, self_first = None, 0, 0
    for _ in range(2):
        fence()
        t0 = time.perf_counter()
        res = sc.find_batch(needles, p, knn=10)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        best = dt if best is None else min(best, dt)
        hits = sum(len(r) for r in res)
        self_first = sum(1 for i, r in enumerate(res) if r and r[0].mediaId == needles[i].id)
    a, b = ShardedDctHashIndex.shard_range(n_img, rank, world)
    pairs = float(n_img) * per * n_needles * per
    # ---- kernel figures on this rank's rows, HIP events around the scan launches (cbh_idx256_get_stats): the batched
    # search (64 needle images = 32k descriptors) and the reference's own query shape, ONE needle image of 500
    # descriptors (cvfeaturesindex.cpp:497).  Both bounds are stated: the 128-bit prefilter issues two K = 64 FP4 MFMAs
    # per 32 x 32 pairs (256 FLOP per pair against the 10 PFLOP/s dense FP4 peak), and the rows stream once (32 B each
    # against 8 TB/s).  At 500 needle descriptors the MFMA bound is 0.82 ms for 5*10^7 rows and the HBM bound 0.2 ms, so
    # the matrix cores bind there too.
    import ctypes as C

    from cbird_amd import _lib

    L = _lib.lib()
    st = _lib.cbh_stats()
    h256 = sc.index.handle
    rows_here = (b - a) * per

    def kernel_ms(call, reps):
        call()
        L.cbh_idx256_get_stats(h256, C.byref(st))
        ms0, l0 = st.scan_ms, st.scan_launches
        for _ in range(reps):
            call()
        L.cbh_idx256_get_stats(h256, C.byref(st))
        return (st.scan_ms - ms0) / max(1, st.scan_launches - l0)

    one = np.ascontiguousarray(needles[0].keyPointDescriptors, np.uint8)
    allq = np.ascontiguousarray(np.concatenate([m.keyPointDescriptors for m in needles]), np.uint8)

    def roof(nq, ms, kernel):
        fl = float(rows_here) * nq * 256.0
        by = float(rows_here) * 32.0
        return {"kernel": kernel, "needle_descriptors": nq, "rows": rows_here, "avg_launch_ms": ms,
                "cmp256_per_s": float(rows_here) * nq / (ms * 1e-3),
                "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": FP4_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / (ms * 1e-3) / 1e12 / FP4_PEAK_TFLOPS, "traffic": None,
                "hbm": {"achieved_GBps": by / (ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS,
                        "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by},
                "algorithmic_flop_per_launch": fl}

    roofline = {"one_needle_image": roof(len(one), kernel_ms(lambda: sc.index.knn(one, 10, 25), 5),
                                         "k_hamm256_small<16> (needle tiles stationary, rows streamed)"),
                "batch_of_64": roof(len(allq), kernel_ms(lambda: sc.index.knn(allq, 10, 25), 2),
                                    "k_hamm256_mfma3<12,2> (row tiles stationary; needle tiles streamed, three per accumulator)")}
    return {"workload": "configs[3]: CvFeaturesIndex, %d images x %d descriptors x 256 bit, %d needle images batched, "
                        "knn k=10, cvThresh 25" % (n_img, per, n_needles),
            "roofline": roofline,
            "parallelism": f"sharded by image x{world}, needles replicated, one all-gather of the candidate tables",
            "seconds": best, "needle_images_per_s": n_needles / best, "cmp256_per_s": pairs / best, "matches": hits,
            "needles_ranked_first_themselves": self_first, "rows_this_rank": (b - a) * per,
            "build_seconds_this_rank": t_build}


def _csr_of_gpu(torch, res, np):
    """(ids[nq,k], scores[nq,k], counts[nq]) with k >= max(counts) -> CSR (offsets u64, ids u32, dists i32); the rows
    are already in (score, mediaId) ascending order (topk.hip)"""
    gi, gs, gc = res
    k = gi.shape[1]
    keep = torch.arange(k, device=gi.device).view(1, -1) < gc.view(-1, 1)
    ids = gi[keep].cpu().numpy().view(np.uint32)
    dists = gs[keep].cpu().numpy().astype(np.int32)
    off = np.zeros(gc.numel() + 1, np.uint64)
    np.cumsum(gc.cpu().numpy().astype(np.uint64), out=off[1:])
    return off, ids, dists


def _digest(np, *arrays):
    import hashlib

    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def full_identity(args, torch, tree, hashes, state, n, sh, cores, thresholds):
    """Every needle of the job against the whole index through the real VP-tree, full (mediaId, distance) lists in
    canonical (distance, mediaId) order, against the GPU's full lists (the sweep's lists are cut at --topk: where some
    needle has more matches than that, the threshold is searched again uncut, outside the timed region)."""
    import numpy as np

    out = []
    for dht in thresholds:
        t0 = time.perf_counter()
        res = state[dht]
        kmax = int(res[2].max().item())
        if kmax > res[0].shape[1]:
            res = sh.similar(state["hashes"], dht, kmax)
            torch.cuda.synchronize()
        g_off, g_ids, g_d = _csr_of_gpu(torch, res, np)
        t_gpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        c_off, c_ids, c_d = tree.search_lists(hashes, dht, threads=cores)
        t_cpu = time.perf_counter() - t0
        equal = bool(len(g_ids) == len(c_ids) and (g_off == c_off).all() and (g_ids == c_ids).all()
                     and (g_d == c_d).all())
        out.append({"dht": dht, "needles": int(n), "index": int(n), "pairs": int(c_off[-1]),
                    "longest_list": kmax, "equal": equal,
                    "sha256_gpu": _digest(np, g_off, g_ids, g_d), "sha256_reference": _digest(np, c_off, c_ids, c_d),
                    "reference": "src/tree/vptree.h compiled in place (oracle/_ref), %d threads" % cores,
                    "reference_seconds": round(t_cpu, 2), "gpu_lists_seconds": round(t_gpu, 2)})
    return out[0] if len(out) == 1 else out


def hash_identity(torch, orc, imgs, gpu_hashes, cores):
    """All images of this rank through the CPU port (oracle/fast_hash.c, the vectorised form; itself checked against the
    scalar restatement on a sample in cpu_baseline) -- 512 MB chunks through two pinned buffers, the copy of one chunk
    under the hashing of the previous."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    n = len(imgs)
    chunk = 8192
    bufs = [torch.empty((chunk, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
    copy = torch.cuda.Stream()
    evs = [None, None]

    def start(c):
        a, b = c * chunk, min(n, (c + 1) * chunk)
        with torch.cuda.stream(copy):
            bufs[c % 2][: b - a].copy_(imgs[a:b], non_blocking=True)
            evs[c % 2] = torch.cuda.Event()
            evs[c % 2].record(copy)

    t0 = time.perf_counter()
    nchunks = -(-n // chunk)
    bad = 0
    cpu_all = np.empty(n, np.uint64)
    start(0)
    with ThreadPoolExecutor(cores) as ex:
        for c in range(nchunks):
            evs[c % 2].synchronize()
            a, b = c * chunk, min(n, (c + 1) * chunk)
            host = bufs[c % 2].numpy()[: b - a]
            if c + 1 < nchunks:
                start(c + 1)
            parts = [p for p in np.array_split(np.arange(b - a), min(b - a, cores * 2)) if len(p)]
            hs = list(ex.map(lambda ix: orc.dcthash64_fast256_batch(host[ix[0]: ix[-1] + 1]), parts))
            cpu_all[a:b] = np.concatenate(hs)
    dt = time.perf_counter() - t0
    bad = int((cpu_all != gpu_hashes[:n]).sum())
    return {"images": int(n), "equal": bad == 0, "differing": bad, "sha256_gpu": _digest(np, gpu_hashes[:n]),
            "sha256_port": _digest(np, cpu_all), "port": "oracle/fast_hash.c, %d threads" % cores,
            "seconds": round(dt, 2), "images_per_s_incl_copy": n / dt}


def cpu_baseline(args, torch, imgs, state, n, dhts, sh=None, ops=None):
    """The reference's own search structure (real VP-tree, oracle/_ref) -- or the C port when the
    prebuilt reference object is absent -- on this box's host cores, over a bounded sample of the
    same workload; also the checker for the GPU results of the sampled needles."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    import oracle

    cores = len(os.sched_getaffinity(0))
    orc = oracle.Oracle()
    hashes = state["hashes"].cpu().numpy().view(np.uint64)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    budget = args.cpu_seconds
    out = {"cores": cores}
    # -- hash leg: bounded sample of the images, one slice per task over all cores.  oracle/fast_hash.c is the port
    # written as a tuned CPU library does stages 1-2 (running column sums, compiler-vectorised, AVX2 clone) -- not OpenCV
    # itself, which cannot be built here, but no longer the scalar per-pixel restatement either; that one
    # (oracle/cbird_oracle.c) is timed on a small sample beside it and must give the same hashes.
    m_img = min(len(imgs), 32768)
    sample = imgs[:m_img].cpu().numpy()
    t0 = time.perf_counter()
    parts = np.array_split(np.arange(m_img), cores * 4)
    with ThreadPoolExecutor(cores) as ex:
        hs = list(ex.map(lambda ix: orc.dcthash64_fast256_batch(sample[ix]) if len(ix) else np.zeros(0, np.uint64), parts))
    t_hash = time.perf_counter() - t0
    cpu_h = np.concatenate(hs)
    m_sc = min(m_img, 16 * cores)
    t0 = time.perf_counter()
    parts_sc = np.array_split(np.arange(m_sc), cores)
    with ThreadPoolExecutor(cores) as ex:
        hs_sc = list(ex.map(lambda ix: orc.dcthash64_batch(sample[ix]) if len(ix) else np.zeros(0, np.uint64), parts_sc))
    t_sc = time.perf_counter() - t0
    out["hash_images_per_s"] = m_img / t_hash
    out["hash_kind"] = "port, vectorised (oracle/fast_hash.c: running column sums, gcc -O3 + AVX2 clone); not OpenCV"
    out["hash_images_per_s_scalar_port"] = m_sc / t_sc
    out["hash_agrees_with_gpu"] = bool((cpu_h == hashes[:m_img]).all() and
                                       (np.concatenate(hs_sc) == hashes[:m_sc]).all())
    if sh is not None:
        out["hash_identity"] = hash_identity(torch, orc, imgs, hashes, cores)
    # -- find leg
    use_ref = oracle.ref_available()
    rng = np.random.default_rng(args.seed)
    if use_ref:
        tree = oracle.RefTree(hashes, ids)
        m = 256 * cores
        needles_ix = rng.choice(n, min(n, m), replace=False)
        t0 = time.perf_counter()
        for dht in dhts:
            tree.search_many(hashes[needles_ix], dht, threads=cores)
        t_cal = time.perf_counter() - t0
        m = int(min(n, max(m, m * budget * 0.7 / max(t_cal, 1e-3))))
        needles_ix = rng.choice(n, m, replace=False)
        t0 = time.perf_counter()
        counts = {}
        for dht in dhts:
            _, counts[dht] = tree.search_many(hashes[needles_ix], dht, threads=cores, want_counts=True)
        t_find = time.perf_counter() - t0
        kind = "reference"
        what = "VP-tree (src/tree/vptree.h compiled in place)"
        if sh is not None and hasattr(tree, "search_lists") and hasattr(tree.lib(), "ref_dcttree_search_lists"):
            ident = [d for d in (2,) if d in dhts] or dhts[:1]  # BASELINE configs[0]/[1]: -p.dht 2
            out["full_identity"] = full_identity(args, torch, tree, hashes, state, n, sh, cores, ident)
    else:
        m = 64
        needles_ix = rng.choice(n, m, replace=False)
        t0 = time.perf_counter()
        counts = {}
        with ThreadPoolExecutor(cores) as ex:
            for dht in dhts:
                parts = np.array_split(needles_ix, cores)
                tot = list(ex.map(lambda ix: orc.find64_batch(hashes, ids, hashes[ix], dht, 1)[2], parts))
                counts[dht] = np.concatenate(tot)
        t_find = time.perf_counter() - t0
        kind = "port"
        what = "brute-force popcount scan (oracle/cbird_oracle.c)"
    agrees = True
    differing = {}
    for dht in dhts:
        gpu_cnt = state[dht][2][torch.from_numpy(needles_ix).to(state[dht][2].device)].cpu().numpy()
        # the real tree also returns removed (id 0) slots; the synthetic index has none
        bad = int((gpu_cnt.astype(np.int64) != counts[dht].astype(np.int64)).sum())
        if bad:
            differing[str(dht)] = bad
        agrees &= bad == 0
    out.update({
        "value": float(m) * n * len(dhts) / t_find,
        "unit": "needle x index pair-equivalents/s (tree-pruned)" if use_ref else "64-bit Hamming comparisons/s",
        "kind": kind,
        "sample": (f"{m} needles of the 1M x 1M job x dht {dhts} against the full {n}-entry index with {what}, "
                   f"{cores} threads, {t_find:.1f} s; hash leg: {m_img} images, {t_hash:.1f} s"),
        "find_agrees_with_gpu": agrees,
        "find_needles_differing_per_dht": differing,
    })
    return out


if __name__ == "__main__":
    main()
