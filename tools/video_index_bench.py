"""Frames/s of the streaming video indexer (cbh_vindexer_*: Media::makeVideoIndex, src/media.cpp:925-1037) for frames
that are already in device memory (a hardware decoder's output) and for frames handed over in host memory.
One JSON line per geometry.

    python tools/video_index_bench.py [--frames 2048] [--chunk 256]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--only", default="", help="w,h,bar_rows,side_cols: just that geometry")
    args = ap.parse_args()
    import torch

    from cbird_amd.video import VideoIndexer

    dev = torch.device("cuda", 0)
    geos = ((1920, 1080, 0, 0), (1920, 1080, 140, 0), (1920, 1080, 0, 240), (1280, 720, 0, 0),
                              (1280, 720, 90, 0), (1280, 720, 0, 160), (640, 360, 0, 0), (640, 360, 45, 0), (256, 256, 0, 0),
            (128, 72, 0, 0), (128, 72, 9, 0))  # (cbird's own decode size: maxW = maxH = 128, src/scanner.cpp:1043-1048)
    if args.only:
        geos = (tuple(int(x) for x in args.only.split(",")),)
    for (w, h, bar, side) in geos:
        n = args.frames if w * h <= 1280 * 720 else min(args.frames, 4096) if args.frames > 2048 else min(args.frames, 1024)
        g = torch.Generator(device=dev).manual_seed(w + bar + side)
        # slowly varying content (so the near-frame filter has something to drop) + bars with a little noise
        base = torch.randint(40, 256, (1, h - 2 * bar, w), dtype=torch.uint8, device=dev, generator=g)
        frames = torch.full((n, h, w), 16, dtype=torch.uint8, device=dev)
        frames += torch.randint(0, 3, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
        noise = torch.randint(0, 4, (n, h - 2 * bar, w), dtype=torch.uint8, device=dev, generator=g)
        frames[:, bar:h - bar, :] = torch.clamp(base.to(torch.int16) + noise.to(torch.int16), 40, 255).to(torch.uint8)
        for k in range(0, n, 97):  # scene cuts
            frames[k:, bar:h - bar, :] = torch.roll(frames[k:, bar:h - bar, :], shifts=k * 131 + 7, dims=2)
        if side:  # pillarbox: 4:3 content in a 16:9 frame
            frames[:, :, :side] = torch.randint(16, 19, (n, h, side), dtype=torch.uint8, device=dev, generator=g)
            frames[:, :, w - side:] = torch.randint(16, 19, (n, h, side), dtype=torch.uint8, device=dev, generator=g)
        torch.cuda.synchronize()
        rec = {"w": w, "h": h, "bar_rows": bar, "side_cols": side, "frames": n, "chunk": args.chunk}
        for crop in (20, -1):
            best = None
            for rep in range(3):
                ix = VideoIndexer(threshold=8, autocrop_range=crop)
                t0 = time.perf_counter()
                for i in range(0, n, args.chunk):
                    ix.push(frames[i:i + args.chunk])
                vi = ix.finish()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            key = "autocrop20" if crop == 20 else "no_autocrop"
            rec[key] = {"device_frames_per_s": round(n / best, 1), "GBps": round(n * w * h / best / 1e9, 1),
                        "stored": len(vi.frames)}
        host = frames[: min(n, 512)].cpu().numpy()
        ix = VideoIndexer(threshold=8)
        ix.push(host[:8])
        t0 = time.perf_counter()
        ix.push(host)
        dt = time.perf_counter() - t0
        rec["host_frames_per_s"] = round(len(host) / dt, 1)
        rec["host_GBps"] = round(host.nbytes / dt / 1e9, 1)
        print(json.dumps(rec), flush=True)
        del frames, noise, base
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
