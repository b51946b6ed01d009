"""Where a sharded handle's dht sweep spends what the plain handle's does not: one sweep
(cbh_idx64_find_batch_dev at dht 1..8, 10^6 needles against 10^6 slots) through a plain handle or a handle over 8 logical
shards of device 0, run under `rocprofv3 --kernel-trace`, then the trace's last sweep taken apart per threshold: wall time,
the union of the kernels' busy time, idle gaps, and GPU time by kernel.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st_one -- python3 tools/ab/sharded_trace.py run one
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st_sh8 -- python3 tools/ab/sharded_trace.py run sharded8
    python3 tools/ab/sharded_trace.py report gpurun_out/st_one gpurun_out/st_sh8
"""
import csv, glob, json, os, sys, time

sys.path.insert(0, ".")


def run(kind):
    import ctypes as C
    import numpy as np
    import torch
    from cbird_amd import _lib, synth
    L = _lib.lib()
    n, topk = 1000000, 8
    dev = torch.device("cuda", 0)
    h, ids = synth.make_hashes(n, seed=1234)
    dq = torch.from_numpy(h.view(np.int64)).to(dev)
    out = torch.empty((n, topk, 2), dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    hnd = L.cbh_idx64_create(0) if kind == "one" else L.cbh_idx64_create_sharded(1, 8)
    assert hnd and L.cbh_idx64_load(hnd, h.ctypes.data, ids.ctypes.data, n) == 0
    tot = C.c_uint64(0)

    def sweep():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d in range(1, 9):
            assert L.cbh_idx64_find_batch_dev(hnd, dq.data_ptr(), n, d, topk, out.data_ptr(), cnt.data_ptr(), C.byref(tot), None) == 0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    sweep()
    sweep()
    time.sleep(0.5)  # the report finds the last sweep behind this gap
    print(json.dumps({"kind": kind, "sweep_ms": round(sweep(), 3)}))
    L.cbh_idx64_destroy(hnd)


def kname(n):
    import re
    m = re.search(r"(k_\w+(?:<[^>(]*>)?)", n)
    return m.group(1) if m else n.split("(")[0][-40:]


def last_sweep(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    k = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kname(r["Kernel_Name"])) for r in rows))
    cut = max(range(1, len(k)), key=lambda i: k[i][0] - k[i - 1][1])
    return k[cut:]


def report(dirs):
    for d in dirs:
        k = last_sweep(d)
        # thresholds: split at the scan kernels' first launches -- a new threshold starts with k_fold_probe or k_expand_needles
        # after a reduction kernel; simpler: the 8 largest gaps between "cut"-side kernels and the next scan are not needed --
        # report the whole sweep and the per-kernel sums
        t0, t1 = k[0][0], max(e for _, e, _ in k)
        busy, cur_s, cur_e = 0, k[0][0], k[0][1]
        gaps = []
        for s, e, _ in k[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, cur_e - t0))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        by = {}
        for s, e, nme in k:
            a = by.setdefault(nme, [0, 0])
            a[0] += 1
            a[1] += e - s
        print(f"== {d}: last sweep {len(k)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, busy union {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
        gaps.sort(reverse=True)
        print("   largest gaps (us @ ms into the sweep):", ", ".join(f"{g / 1e3:.0f}@{at / 1e6:.1f}" for g, at in gaps[:24]))
        for nme, (c, ns) in sorted(by.items(), key=lambda x: -x[1][1])[:14]:
            print(f"   {nme:40s} x{c:4d}  {ns / 1e6:9.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2:])
