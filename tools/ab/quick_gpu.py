"""Quick on-GPU timing of the raw kernels (development aid, not the bench contract)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import cbird_amd
from cbird_amd import _lib, synth

L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
h, ids = synth.make_hashes(N, seed=1234)
idx = cbird_amd.DctHashIndex()
idx.load(h, ids)
dq = torch.from_numpy(h.view(np.int64)).cuda()
cap = 1 << 24
drec = torch.empty(cap, dtype=torch.int64, device="cuda")
dtot = torch.zeros(1, dtype=torch.int64, device="cuda")
ms = C.c_float(0)
for thr in (1, 2, 3, 5, 6, 8, 12):
    _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap,
                                         dtot.data_ptr(), 1, C.byref(ms)), "warm")
    _lib.check(L.cbh_idx64_time_scan_dev(idx.handle, dq.data_ptr(), N, thr, drec.data_ptr(), cap,
                                         dtot.data_ptr(), 3, C.byref(ms)), "time")
    tot = int(dtot.item()) // 3
    print(f"scan N={N} nq={N} dht={thr}: {ms.value:.2f} ms  {N*N/ms.value*1e3:.3e} cmp/s  matches={tot}")

# full find_batch (scan+sort+select), device resident
k = 8
dout = torch.empty((N, k, 2), dtype=torch.int32, device="cuda")
dcnt = torch.empty(N, dtype=torch.int32, device="cuda")
tot = C.c_uint64(0)
for thr in (2, 5):
    torch.cuda.synchronize()
    t0 = time.time()
    _lib.check(L.cbh_idx64_find_batch_dev(idx.handle, dq.data_ptr(), N, thr, k, dout.data_ptr(),
                                          dcnt.data_ptr(), C.byref(tot), None), "fb")
    t1 = time.time()
    print(f"find_batch_dev dht={thr}: {1e3*(t1-t0):.2f} ms total={tot.value}")

# hashing
n_img = 16384
imgs = torch.randint(0, 256, (n_img, 256, 256), dtype=torch.uint8, device="cuda")
dout_h = torch.empty(n_img, dtype=torch.int64, device="cuda")
_lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n_img, 256, 256, 256, 65536, dout_h.data_ptr(), 0, 1,
                                  C.byref(ms)), "h")
_lib.check(L.cbh_time_dcthash_dev(imgs.data_ptr(), n_img, 256, 256, 256, 65536, dout_h.data_ptr(), 0, 3,
                                  C.byref(ms)), "h")
print(f"dcthash {n_img} imgs: {ms.value:.2f} ms  {n_img/ms.value*1e3:.3e} img/s  {n_img*65544/ms.value*1e-6:.1f} GB/s")
