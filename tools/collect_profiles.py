#!/usr/bin/env python3
"""Turn gpurun_out/prof_summary/ (written by tools/profile_bench.sh on the GPU box) into the
tracked files under profiles/: trimmed kernel stats, the PMC rows of our kernels, the
per-launch HBM traffic (pmc_traffic.json, read by bench.py) and an agreement table.

    python tools/collect_profiles.py [round_tag]        # default r01
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_summary")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"

FORMULA = ("(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH_SIZE half-count "
           "correction, MI355X_MICROARCH.md section HBM)")


def short(name):
    """'void cbh::(anonymous namespace)::k_foo<1, 2>(args...)' -> 'k_foo<1, 2>'"""
    s = name
    if "::k_" in s:
        s = "k_" + s.split("::k_", 1)[1]
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


def main():
    os.makedirs(DST, exist_ok=True)
    # 1. kernel stats: our kernels + the ten largest others for context
    with open(os.path.join(SRC, "kernel_stats.csv")) as f:
        rows = list(csv.DictReader(f))
    ours = [r for r in rows if any(k in r["Name"] for k in ("cbh::", "rocprim", "hipcub"))]
    others = [r for r in rows if r not in ours][:10]
    with open(os.path.join(DST, f"{TAG}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        for r in ours + others:
            w.writerow(r)
    stats_ms = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in ours if "cbh::" in r["Name"]}
    calls = {short(r["Name"]): int(r["Calls"]) for r in ours if "cbh::" in r["Name"]}

    # 2. PMC rows
    pmc = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        src = os.path.join(SRC, f"pmc_{c}_cbh.csv")
        shutil.copyfile(src, os.path.join(DST, f"{TAG}_pmc_{c}.csv"))
        acc = defaultdict(list)
        with open(src) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == c:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[c] = acc

    def traffic(prefix):
        fv = [v for k, vs in pmc["FETCH_SIZE"].items() if k.startswith(prefix) for v in vs]
        wv = [v for k, vs in pmc["WRITE_SIZE"].items() if k.startswith(prefix) for v in vs]
        if not fv or not wv:
            return None, None, None
        fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
        return (2.0 * fm + wm) * 1024.0, fm, wm

    out = {}
    # (name prefix of the kernel -> key in pmc_traffic.json; "k_hamm64_mfma<" = the prefilter kernel, "k_hamm64_mfma3" = FULL3)
    for prefix, key in (("k_hamm64_mfma3", "k_hamm64_mfma"), ("k_hamm64_mfma<", "k_hamm64_mfma_pre"),
                        ("k_hamm64_scan", "k_hamm64_scan"), ("k_dcthash_256", "k_dcthash_256")):
        t, fm, wm = traffic(prefix)
        out[key] = t
        out[key + "_detail"] = {"FETCH_SIZE_KiB_raw": fm, "WRITE_SIZE_KiB_raw": wm, "formula": FORMULA}
    out["source"] = (f"profiles/{TAG}_pmc_FETCH_SIZE.csv, profiles/{TAG}_pmc_WRITE_SIZE.csv (separate rocprofv3 "
                     "--pmc passes of `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sharded-leg`)")
    with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)

    # 3. bench lines
    shutil.copyfile(os.path.join(SRC, "bench_under_kernel_trace.json"),
                    os.path.join(DST, f"{TAG}_bench_under_kernel_trace.json"))
    with open(os.path.join(SRC, "bench_under_kernel_trace.json")) as f:
        line = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])

    # 4. agreement table (printed; paste into profiles/README.md)
    sweep = {d["dht"]: d["scan_kernel_ms"] for d in line["dht_sweep"]}
    print("kernel | rocprofv3 --stats avg ms (calls) | HIP events in bench.py")
    for k in sorted(stats_ms):
        print(f"{k} | {stats_ms[k]:.3f} ({calls[k]})")
    print("bench.py per-dht scan_kernel_ms:", sweep)
    print("bench.py hash avg_launch_ms:", line["roofline_hash"]["avg_launch_ms"])
    print("bench.py scan avg_launch_ms:", line["roofline"]["avg_launch_ms"])
    print("traffic:", {k: v for k, v in out.items() if not k.endswith("detail") and k != "source"})


if __name__ == "__main__":
    main()
