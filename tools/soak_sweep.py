"""Soak of the pipelined threshold sweep (cbird_amd.dist.ShardedDctHashIndex.similar_sweep: scan of threshold i+1
overlapping the cut of threshold i on a side stream) against the oracle's brute-force counts, on random index sizes
where the scans are short and the overlap is tight.  Prints one JSON line.

    python tools/soak_sweep.py [--configs 24] [--repeats 20] [--seed 1]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=24)
    ap.add_argument("--repeats", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch

    from cbird_amd import synth
    from cbird_amd.dist import HipOps, ShardedDctHashIndex
    from oracle import Oracle

    orc = Oracle()
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda", 0)
    ops = HipOps(0)
    bad = []
    sweeps = 0
    for c in range(args.configs):
        n = int(rng.integers(300, 20000))
        h, ids = synth.make_hashes(n, seed=int(rng.integers(1, 1 << 30)))
        dhts = sorted(rng.choice(np.arange(1, 12), int(rng.integers(2, 8)), replace=False).tolist())
        k = int(rng.choice([1, 4, 8]))
        want = {d: orc.find64_batch(h, ids, h, d, k) for d in dhts}
        sh = ShardedDctHashIndex(ops, record_capacity=1 << int(rng.integers(12, 22)))
        dh = torch.from_numpy(h.view(np.int64)).to(dev)
        di = torch.from_numpy(ids.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        with ops.stream_ctx(ops.work_stream()):
            for r in range(args.repeats):
                sh.load_shard(dh, di)
                res = sh.similar_sweep(dh, dhts, k)
                torch.cuda.synchronize()
                sweeps += 1
                for d in dhts:
                    gi, gs, gc = (t.cpu().numpy() for t in res[d])
                    wi, ws, wc = want[d]
                    live = np.arange(k)[None, :] < np.minimum(wc, k)[:, None]
                    if not ((gc == wc.astype(np.int32)).all() and (gi.view(np.uint32)[live] == wi[live]).all()
                            and (gs[live] == ws[live]).all()):
                        bad.append((c, n, r, d))
                if r == 2:
                    sh.fit_capacity()
    print(json.dumps({"configs": args.configs, "sweeps": sweeps, "mismatches": bad[:20], "n_mismatches": len(bad)}))


if __name__ == "__main__":
    main()
