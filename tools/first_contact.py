#!/usr/bin/env python3
"""First contact with a multi-GPU node: every code path that needs two physical devices, in one command, < 3 min.

    tools/first_contact.sh [--out first_contact.json]

This pool hands out one GPU per box, so `sharded.hip`'s device-to-device branches (peer copies, cross-device events,
ncclCommInitAll over several devices) and `cbird_amd/dist.py` over RCCL at world > 1 have never executed.  The legs,
each a child process with its own timeout (a wedged leg costs its timeout, not the node):

  probe         usable devices, peer-access matrix
  sharded_leg   tools/sharded_leg.py --mask <all> --exchange both at 1M hashes: ONE DctHashIndex handle over all GPUs,
                records to the root by peer copies and, second leg, through one grouped ncclAllGather; match counts of
                both against the one-device index
  sharded_capi  tests/test_sharded_capi.py with the multi-device shapes it adds when it sees > 1 GPU (every device x 1
                shard with both exchanges, two devices x 2 shards): the oracle / golden suites on a real multi-GPU handle
  bench_N       bench.py --gpus N (N = 2, 4, 8 up to what is there) at 80 k images through torch.distributed.run: the
                line must say matches_expected: true (equal to the N = 1 job), hashes_expected: true, and the
                communicator must span N ranks

Prints ONE JSON object: {"devices": D, "legs": {name: {"ok": bool, "seconds": s, "error": first error text, ...}}}.
On a one-GPU box the same legs run in their one-device forms (logical shards, ranks sharing the GPU over gloo) --
a rehearsal of the script, not of the transport; "devices": 1 says so."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout, env=None):
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        return p.returncode, p.stdout, p.stderr, time.perf_counter() - t0
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        return -9, out, err + f"\n[timeout after {timeout} s]", time.perf_counter() - t0


def first_error(text):
    lines = [l for l in text.splitlines() if l.strip()]
    for l in lines:
        if any(k in l for k in ("Error", "error", "FAILED", "failed", "Traceback", "assert", "rc ")):
            return l.strip()[:300]
    return (lines[-1].strip()[:300] if lines else "")


def last_json(text):
    for l in reversed(text.splitlines()):
        if l.startswith("{"):
            try:
                return json.loads(l)
            except Exception:
                pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--images", type=int, default=1_000_000, help="hashes of the sharded leg")
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import torch

    ndev = torch.cuda.device_count()
    legs = {}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    py = sys.executable

    # ---- probe (a child: this process never initialises the GPU)
    code = ("import json, torch; from cbird_amd import _lib; L = _lib.lib(); n = torch.cuda.device_count(); "
            "print(json.dumps({'usable': L.cbh_device_count(), 'mask': hex(L.cbh_usable_device_mask()), "
            "'peer': [[int(a == b or torch.cuda.can_device_access_peer(a, b)) for b in range(n)] for a in range(n)], "
            "'names': [torch.cuda.get_device_name(d) for d in range(n)]}))")
    rc, out, err, dt = run([py, "-c", code], 120, env)
    info = last_json(out) or {}
    legs["probe"] = {"ok": rc == 0 and info.get("usable", 0) >= 1, "seconds": round(dt, 1), **info,
                     "error": None if rc == 0 else first_error(err or out)}
    mask = int(info.get("mask", "0x1"), 16) or 1
    multi = bin(mask).count("1") > 1

    # ---- one handle over all GPUs, both exchanges
    cmd = [py, "tools/sharded_leg.py", "--images", str(args.images), "--repeats", "2"]
    cmd += ["--mask", hex(mask), "--per-device", "1", "--exchange", "both"] if multi else \
        ["--mask", "0x1", "--per-device", "8", "--force-rccl"]
    rc, out, err, dt = run(cmd, 300, env)
    j = last_json(out) or {}
    ok = rc == 0 and j.get("matches_equal") is True
    if ok and multi:
        ok = j["sharded"]["devices"] == bin(mask).count("1") and j["sharded"]["peer_copies"] > 0 and \
            j.get("sharded_rccl", {}).get("collectives", 0) > 0 and j["sharded_rccl"]["collective_fallbacks"] == 0
    legs["sharded_leg"] = {"ok": ok, "seconds": round(dt, 1), "error": None if ok else first_error(err or out),
                           "one_device_ms": j.get("one_device", {}).get("sweep_ms"),
                           "peer_copies_ms": j.get("sharded", {}).get("sweep_ms"),
                           "rccl_ms": j.get("sharded_rccl", {}).get("sweep_ms"),
                           "rccl_fallbacks": j.get("sharded_rccl", {}).get("collective_fallbacks"),
                           "matches_equal": j.get("matches_equal")}

    # ---- the sharded C-ABI suites on the shapes a multi-GPU box adds
    sel = "alldev or dev2x2" if multi else "shards2x"
    rc, out, err, dt = run([py, "-m", "pytest", "tests/test_sharded_capi.py", "-m", "gpu", "-x", "-q", "-k", sel,
                            "-p", "no:cacheprovider"], 600, env)
    tail = [l for l in out.splitlines() if " passed" in l or " failed" in l or " error" in l]
    legs["sharded_capi"] = {"ok": rc == 0, "seconds": round(dt, 1), "selection": sel, "summary": tail[-1] if tail else "",
                            "error": None if rc == 0 else first_error(out + err)}

    # ---- bench.py through torch.distributed.run
    sizes = [n for n in (2, 4, 8) if n <= max(ndev, 1)] if multi else [2]
    for n in sizes:
        e = dict(env)
        if not multi:
            e["CBH_BENCH_SHARE_GPU"] = "1"  # ranks share the one GPU over gloo: a rehearsal of the launch only
        cmd = [py, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + n), "bench.py", "--gpus", str(n), "--images", "80000", "--steps", "2",
               "--warmup", "1", "--no-video", "--no-orb", "--no-sharded-leg"]
        rc, out, err, dt = run(cmd, 420, e)
        j = last_json(out) or {}
        col = j.get("collective", {})
        ok = rc == 0 and j.get("matches_expected") is True and j.get("hashes_expected") is True and \
            col.get("communicator_ranks") == n and j.get("n_gpus") == n
        legs[f"bench_{n}"] = {"ok": ok, "seconds": round(dt, 1), "value": j.get("value"), "ms_per_step": j.get("ms_per_step"),
                              "matches_expected": j.get("matches_expected"), "hashes_expected": j.get("hashes_expected"),
                              "collective": col, "error": None if ok else first_error(err or out)}

    res = {"devices": bin(mask).count("1"), "rehearsal_on_one_device": not multi,
           "ok": all(l["ok"] for l in legs.values()), "legs": legs}
    text = json.dumps(res)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
