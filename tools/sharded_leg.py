"""ONE process, one DctHashIndex handle over several GPUs (cbh_idx64_create_sharded) beside the one-device index:
the all-pairs threshold sweep of BASELINE configs[1]/[2] through cbh_idx64_find_batch_dev on both, match counts
compared, one JSON line.  bench.py starts this as a child process (own timeout) and attaches the line as
`single_process_sharded`; it also runs on its own:

    python tools/sharded_leg.py --mask 0xff                  # the 8 GPUs of a node, records to the root by peer copies
    python tools/sharded_leg.py --mask 0xff --exchange both  # ... and once more through one grouped ncclAllGather
    python tools/sharded_leg.py --mask 1 --per-device 8      # eight logical shards on device 0 (a one-GPU box)
    python tools/sharded_leg.py --mask 1 --per-device 8 --force-rccl   # ... their block through ncclAllGather too

The hashes are the bench's uniform-random set with planted neighbours (cbird_amd.synth.make_hashes); resident needles
and results (the *_dev entry point), so what is timed is scan + exchange + cut, not PCIe.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mask", type=lambda x: int(x, 0), default=1)
    ap.add_argument("--per-device", type=int, default=1)
    ap.add_argument("--images", type=int, default=1_000_000)
    ap.add_argument("--dht", type=str, default="1,2,3,4,5,6,7,8")
    ap.add_argument("--topk", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--force-rccl", action="store_true")
    ap.add_argument("--exchange", default=None,
                    help="1 = copies into the root block (the library's default), 0 = ncclAllGather between devices, "
                         "both = one leg each; default: 1, or 0 with --force-rccl")
    args = ap.parse_args()
    import torch

    from cbird_amd import _lib, synth

    L = _lib.lib()
    _lib.require_device()
    root = (args.mask & -args.mask).bit_length() - 1
    torch.cuda.set_device(root)
    dev = torch.device("cuda", root)
    n = args.images
    dhts = [int(x) for x in args.dht.split(",") if x]
    h, ids = synth.make_hashes(n, seed=1234)
    L.cbh_set_tuning(b"shard_force_rccl", 1 if args.force_rccl else 0)
    ex = args.exchange if args.exchange is not None else ("0" if args.force_rccl else "1")
    exchanges = [1, 0] if ex == "both" else [int(ex)]
    dq = torch.from_numpy(h.view(np.int64)).to(dev)
    out = torch.empty((n, args.topk, 2), dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)

    def sweep(handle):
        tot = C.c_uint64(0)
        totals = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d in dhts:
            _lib.check(L.cbh_idx64_find_batch_dev(handle, dq.data_ptr(), n, d, args.topk, out.data_ptr(), cnt.data_ptr(),
                                                  C.byref(tot), None), "find_batch_dev")
            totals[d] = int(tot.value)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, totals

    res = {"device_mask": hex(args.mask), "shards_per_device": args.per_device, "images": n, "dht": dhts,
           "exchange": ["copies into the root block" if e else "ncclAllGather between devices, copies inside one"
                        for e in exchanges],
           "force_rccl": bool(args.force_rccl)}
    legs = {}
    plan = [("one_device", lambda: L.cbh_idx64_create(root), 1)]
    for e in exchanges:
        plan.append(("sharded" if e == exchanges[0] else "sharded_rccl",
                     lambda: L.cbh_idx64_create_sharded(args.mask, args.per_device), e))
    for name, make, e in plan:
        L.cbh_set_tuning(b"shard_exchange", e)
        hnd = make()
        if not hnd:
            raise SystemExit(f"{name}: cannot create the index (mask {args.mask:#x})")
        _lib.check(L.cbh_idx64_load(hnd, h.ctypes.data, ids.ctypes.data, n), "load")
        sweep(hnd)  # warm-up: buffers, needle scratch, communicators
        times = []
        for _ in range(args.repeats):
            ms, totals = sweep(hnd)
            times.append(ms)
        st = _lib.cbh_shard_stats()
        _lib.check(L.cbh_idx64_shard_stats(hnd, C.byref(st)), "shard_stats")
        cs = _lib.cbh_stats()
        _lib.check(L.cbh_idx64_get_stats(hnd, C.byref(cs)), "get_stats")
        legs[name] = {"sweep_ms": round(min(times), 3), "sweep_ms_all": [round(t, 3) for t in times],
                      "cmp_per_s": float(n) * n * len(dhts) / (min(times) * 1e-3), "matches": totals,
                      "shards": st.shards, "devices": st.devices, "collectives": st.collectives,
                      "peer_copies": st.peer_copies, "local_copies": st.local_copies, "rescans": st.rescans,
                      "collective_fallbacks": st.collective_fallbacks,
                      "scan_kernel_ms_per_sweep": round(cs.scan_ms / (args.repeats + 1), 3)}
        L.cbh_idx64_destroy(hnd)
    res.update(legs)
    res["matches_equal"] = all(l["matches"] == legs["one_device"]["matches"] for l in legs.values())
    res["speedup_vs_one_device"] = legs["one_device"]["sweep_ms"] / legs["sharded"]["sweep_ms"]
    if "sharded_rccl" in legs:
        res["speedup_vs_one_device_rccl"] = legs["one_device"]["sweep_ms"] / legs["sharded_rccl"]["sweep_ms"]
    print(json.dumps(res))
    if not res["matches_equal"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
