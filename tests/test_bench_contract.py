"""bench.py prints ONE JSON line with the contract's keys (small run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--images", "40000", "--steps", "2",
                          "--warmup", "1", "--cpu-seconds", "2", "--orb-images", "4000"], capture_output=True, text=True,
                         timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 40000.0 ** 2 * 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert c["find_agrees_with_gpu"] is True and c["hash_agrees_with_gpu"] is True
    # north_star's acceptance line as fields: every needle's full (mediaId, distance) list equal to the real VP-tree's
    # at dht 2, every hash equal to the CPU port's -- digests of both sides in the line
    fi, hi = d["full_identity"], d["hash_identity"]
    assert fi["dht"] == 2 and fi["needles"] == 40000 and fi["equal"] is True and fi["sha256_gpu"] == fi["sha256_reference"]
    assert fi["pairs"] == [s_ for s_ in d["dht_sweep"] if s_["dht"] == 2][0]["matches"]
    assert hi["images"] == 40000 and hi["equal"] is True and hi["sha256_gpu"] == hi["sha256_port"]
    assert d["matches_expected"] is None  # (the table holds the default job and first_contact's 80 k job)
    o = d["configs3_cvfeatures"]
    assert o["needles_ranked_first_themselves"] == 64 and o["rows_this_rank"] == 4000 * 500


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_gpu(gpu):
    """the N > 1 code path of bench.py (sharded index, record exchange, pipelined sweep on the work stream) with both
    ranks on cuda:0 over gloo (CBH_BENCH_SHARE_GPU=1, a development aid; the driver's runs use RCCL): same match
    counts as the single-rank run of the same job"""
    import json

    env = dict(os.environ, CBH_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--images", "60000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--dht", "2,5,8", "--orb-images",
            "3000", "--video-clips", "2000"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + args, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    r2 = json.loads([ln for ln in two.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "strong"
    assert r2["collective"]["communicator_ranks"] == 2 and r2["collective"]["world_size"] == 2
    assert [s["matches"] for s in r2["dht_sweep"]] == [s["matches"] for s in r1["dht_sweep"]]
    # the sharded ORB and video legs give the unsharded results
    assert r2["configs3_cvfeatures"]["matches"] == r1["configs3_cvfeatures"]["matches"] > 0
    assert r2["configs3_cvfeatures"]["needles_ranked_first_themselves"] == 64
    assert r2["configs4_video"]["matches"] == r1["configs4_video"]["matches"]


@pytest.mark.gpu
def test_bench_eight_ranks_preflight(gpu):
    """Pre-flight of the driver's 8-GPU run, which this pool cannot offer: `torchrun --nproc-per-node 8 bench.py --gpus
    8` with all ranks on cuda:0 over gloo (CBH_BENCH_SHARE_GPU=1).  All three legs finish, the contract line is printed
    once, the match counts of every threshold and of the video / ORB legs equal the N = 1 run of the same job, and no
    rank holds another rank's data (its images, index slots and descriptor rows are exactly its shard_range share)."""
    env = dict(os.environ, CBH_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    n, orb = 80001, 4001  # ragged on purpose: 80001 = 8 * 10000 + 1
    args = ["--images", str(n), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-features", "--no-sharded-leg",
            "--dht", "2,5,8", "--orb-images", str(orb), "--video-clips", "2000"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    eight = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                            "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                            "--gpus", "8"] + args, capture_output=True, text=True, timeout=1200, env=env)
    assert eight.returncode == 0, eight.stderr[-3000:]
    lines = [ln for ln in eight.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    r1 = json.loads([ln for ln in one.stdout.strip().splitlines() if ln.startswith("{")][-1])
    r8 = json.loads(lines[0])
    assert r8["n_gpus"] == 8 and r8["scaling"] == "strong" and "configs[2]" in r8["config"]["workload"]
    assert [s["matches"] for s in r8["dht_sweep"]] == [s["matches"] for s in r1["dht_sweep"]]
    assert r8["configs3_cvfeatures"]["matches"] == r1["configs3_cvfeatures"]["matches"] > 0
    assert r8["configs3_cvfeatures"]["needles_ranked_first_themselves"] == 64
    assert r8["configs4_video"]["matches"] == r1["configs4_video"]["matches"]
    res = r8["per_rank_residency"]
    assert [p["rank"] for p in res] == list(range(8))
    for p in res:
        a, b = p["rank"] * n // 8, (p["rank"] + 1) * n // 8
        assert p["images"] == b - a == p["index_slots"] and p["image_bytes"] == (b - a) * 256 * 256
        oa, ob = p["rank"] * orb // 8, (p["rank"] + 1) * orb // 8
        assert p["orb_rows"] == (ob - oa) * 500
        # images + hashes + exchange blocks + leg scratch: nowhere near a second rank's 655 MB of images
        assert p["device_bytes_allocated_by_torch"] < 1.6 * p["image_bytes"] + (256 << 20)
    assert sum(p["images"] for p in res) == n and sum(p["orb_rows"] for p in res) == orb * 500
