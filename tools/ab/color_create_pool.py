"""ColorDescriptor::create at large batches against the arena's live-stream budget ("pool_live_keep_mb"):
seconds per call over consecutive calls on one stream, with the arena's counters after each.

    python tools/ab/color_create_pool.py [--images 32768,65536,100000] [--keep 0,16384]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", default="32768,65536,100000")
    ap.add_argument("--keep", default="0,16384")
    ap.add_argument("--w", type=int, default=256)
    ap.add_argument("--h", type=int, default=192)
    args = ap.parse_args()
    import torch

    from cbird_amd import _lib

    L = _lib.lib()

    def tun(k):
        v = C.c_longlong(0)
        L.cbh_get_tuning(k, C.byref(v))
        return v.value

    rng = np.random.default_rng(1)
    w, h = args.w, args.h
    base = rng.integers(0, 256, (32, h, w, 3), dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d_base = torch.from_numpy(base).to(dev)
    stream = torch.cuda.Stream()
    for keep in [int(x) for x in args.keep.split(",")]:
        L.cbh_set_tuning(b"pool_live_keep_mb", keep)
        for n in [int(x) for x in args.images.split(",")]:
            d = d_base.repeat((n + 31) // 32, 1, 1, 1)[:n].contiguous()
            off = np.arange(n, dtype=np.uint64) * np.uint64(w * h * 3)
            ww, hh, ss = np.full(n, w, np.uint32), np.full(n, h, np.uint32), np.full(n, 3 * w, np.uint32)
            d_desc = torch.zeros((n, 258), dtype=torch.uint8, device=dev)
            d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            times = []
            for _ in range(4):
                t0 = time.time()
                _lib.check(L.cbh_color_descriptors_dev(d.data_ptr(), n, off.ctypes.data, ww.ctypes.data, hh.ctypes.data,
                                                       ss.ctypes.data, 3, d_desc.data_ptr(), d_ok.data_ptr(), 0,
                                                       C.c_void_p(stream.cuda_stream)), "color_descriptors")
                stream.synchronize()
                times.append(round(time.time() - t0, 3))
            print(json.dumps({"pool_live_keep_mb": keep, "images": n, "s_per_call": times,
                              "images_per_s_last": round(n / times[-1]),
                              "cached_GB": round(tun(b"arena_cached_bytes") / 2**30, 2),
                              "pending_GB": round(tun(b"arena_pending_bytes") / 2**30, 2),
                              "trimmed_live": tun(b"arena_trimmed_live"), "released": tun(b"arena_released")}), flush=True)
            del d, d_desc, d_ok
            L.cbh_trim(0, None)


if __name__ == "__main__":
    main()
