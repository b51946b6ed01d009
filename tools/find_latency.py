"""Single-needle DctHashIndex::find latency (the -similar-to path) on a 1M-entry index, per threshold;
the reference's VP-tree needs 43 us (dht 2), 390 us (dht 5), 1.24 ms (dht 8) per query on one core."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import synth
h, ids = synth.make_hashes(1_000_000, seed=1234)
idx = cbird_amd.DctHashIndex(); idx.load(h, ids)
for dht in (2, 5, 8):
    p = cbird_amd.SearchParams(dctThresh=dht)
    ms = [cbird_amd.Media(id=0, dctHash=int(x)) for x in h[:2000]]
    for m in ms[:50]: idx.find(m, p)
    t0 = time.perf_counter()
    n = 0
    for m in ms: n += len(idx.find(m, p))
    dt = time.perf_counter() - t0
    print(f"dht {dht}: {dt / len(ms) * 1e6:7.1f} us per find ({n} matches)")
