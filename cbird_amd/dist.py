"""Sharded DctHashIndex across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The reference has no distributed mode (its only fan-out is QtConcurrent::map over needles,
src/database.cpp:1400-1432); this is the MI355X-native counterpart designed from scratch:

  * the haystack is row-sharded: rank r owns a contiguous slice of (hash, mediaId) slots;
    needles are replicated (an all-pairs job gathers every rank's freshly built hashes once);
  * each rank scans ITS shard for ALL needles with the same HIP kernel as the single-GPU path and
    produces an unordered list of cbh_record (needle<<39 | distance<<32 | mediaId);
  * ONE exchange step: all-gather of the per-rank record lists (sizes first, then the lists padded
    to the longest one).  A threshold search over a union of shards is the union of the per-shard
    results, so the merged list sorted ascending is exactly the single-GPU list;
  * every rank then orders the merged records and cuts each needle's list at max_per_query.

`ops` supplies the device work.  `HipOps` (the product) drives the C-ABI and fails loudly without
the library / a gfx950 device; the CPU test-suite injects its own ops object to exercise the
sharding and exchange logic over gloo.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib
from .index import DctHashIndex


class HipOps:
    """Device work through libcbird_hip.so on torch CUDA(HIP) tensors, current torch stream."""

    def __init__(self, device: int) -> None:
        self.L = _lib.lib()
        _lib.require_device()
        self.device = device
        self.torch_device = torch.device("cuda", device)
        self.index = DctHashIndex(device)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.torch_device).cuda_stream

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype, device=self.torch_device)

    # -- streams (similar_sweep overlaps the scan of one threshold with the post-processing of the previous) ----
    def side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.torch_device)
        return self._side

    def work_stream(self):
        """A non-default stream for the whole step.  The C-ABI treats stream NULL as "synchronous call"
        (include/cbird_hip.h), and torch's default stream IS the NULL stream: on it every scan / sort / select would
        block the host and nothing could be queued ahead.  Callers wrap their step in stream_ctx(work_stream())."""
        if getattr(self, "_work", None) is None:
            self._work = torch.cuda.Stream(device=self.torch_device)
        return self._work

    def current_stream(self):
        return torch.cuda.current_stream(self.torch_device)

    def stream_ctx(self, stream):
        return torch.cuda.stream(stream)

    def new_event(self):
        return torch.cuda.Event(enable_timing=True)

    def hash_images(self, imgs: torch.Tensor) -> torch.Tensor:
        """imgs: u8 [n, h, w] on this device -> int64 [n] (u64 bit patterns)"""
        n, h, w = imgs.shape
        out = torch.empty(n, dtype=torch.int64, device=self.torch_device)
        _lib.check(self.L.cbh_dcthash_batch_dev(imgs.data_ptr(), n, w, h, imgs.stride(1), imgs.stride(0),
                                                out.data_ptr(), self.device, self._stream()),
                   "dcthash_batch_dev")
        return out

    def load_shard(self, hashes: torch.Tensor, ids: torch.Tensor) -> None:
        """(re)load this rank's shard into the same index object (no reallocation when it fits)"""
        _lib.check(self.L.cbh_idx64_load_dev(self.index.handle, hashes.data_ptr(), ids.data_ptr(),
                                             hashes.numel(), self._stream()), "load_dev")

    def scan(self, queries: torch.Tensor, thresh: int, rec: torch.Tensor, total: torch.Tensor) -> None:
        _lib.check(self.L.cbh_idx64_scan_dev(self.index.handle, queries.data_ptr(), queries.numel(),
                                             int(thresh), rec.data_ptr(), rec.numel(), total.data_ptr(),
                                             self._stream()), "scan_dev")

    def sort_records(self, rec: torch.Tensor, n: int, nq: int) -> None:
        _lib.check(self.L.cbh_sort_records_dev(rec.data_ptr(), n, nq, self.device, self._stream()),
                   "sort_records_dev")

    def select(self, rec: torch.Tensor, n: int, nq: int, k: int):
        out = torch.empty((nq, max(k, 1), 2), dtype=torch.int32, device=self.torch_device)
        counts = torch.empty(nq, dtype=torch.int32, device=self.torch_device)
        _lib.check(self.L.cbh_select_records_dev(rec.data_ptr(), n, nq, k, out.data_ptr(),
                                                 counts.data_ptr(), self.device, self._stream()),
                   "select_records_dev")
        return out[:, :k, 0], out[:, :k, 1], counts


class ShardedDctHashIndex:
    def __init__(self, ops, group=None, record_capacity: int = 1 << 22) -> None:
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.record_capacity = record_capacity
        self._rec = None
        self._total = None
        self.last_exchange_records = 0

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        """all_gather_into_tensor; RCCL ("nccl") on device tensors.  With the gloo backend (CPU test-suite,
        or several ranks sharing one GPU in tests) device tensors are staged through the host."""
        if inp.is_cuda and dist.get_backend(self.group) == "gloo":
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    # -- build ---------------------------------------------------------------------------------
    @staticmethod
    def shard_range(n: int, rank: int, world: int):
        """rank r owns [r*n/R, (r+1)*n/R)"""
        return (rank * n) // world, ((rank + 1) * n) // world

    def gather_hashes(self, local_hashes: torch.Tensor, n_total: int) -> torch.Tensor:
        """all-gather the per-rank hash slices (needles are replicated on every rank)"""
        if self.world == 1:
            return local_hashes
        sizes = [self.shard_range(n_total, r, self.world) for r in range(self.world)]
        m = max(b - a for a, b in sizes)
        pad = self.ops.empty(m, torch.int64)
        pad[: local_hashes.numel()] = local_hashes
        if local_hashes.numel() < m:
            pad[local_hashes.numel():] = 0
        out = self.ops.empty(m * self.world, torch.int64)
        self._all_gather(out, pad)
        if all(b - a == m for a, b in sizes):
            return out
        return torch.cat([out[r * m: r * m + (b - a)] for r, (a, b) in enumerate(sizes)])

    def load_shard(self, hashes: torch.Tensor, ids: torch.Tensor) -> None:
        self.ops.load_shard(hashes, ids)

    # -- find ----------------------------------------------------------------------------------
    def _buffers(self):
        if self._rec is None or self._rec.numel() < self.record_capacity:
            self._rec = self.ops.empty(self.record_capacity, torch.int64)
            self._total = self.ops.empty(1, torch.int64)
        return self._rec, self._total

    def _exchange_and_cut(self, rec, total, nq: int, max_per_query: int):
        """counts -> (all-gather of the record lists) -> sort -> first max_per_query per needle.  Returns None when a
        rank's records did not fit its buffer (after growing self.record_capacity for the rescan)."""
        if self.world > 1:  # sizes first: one small all-gather gives max and sum
            counts = self.ops.empty(self.world, torch.int64)
            self._all_gather(counts, total)
            counts_h = counts.tolist()
            n_local, n_max, n_total = counts_h[self.rank], max(counts_h), sum(counts_h)
        else:
            n_local = n_max = n_total = int(total.item())
        if n_max > rec.numel():
            self.record_capacity = int(n_max * 1.25) + 1024  # same decision on every rank
            return None
        if self.world == 1:
            merged = rec
        else:
            rec[n_local:n_max] = nq << 39  # pad = a record of needle index nq: sorts last
            merged = self.ops.empty(n_max * self.world, torch.int64)
            self._all_gather(merged, rec[:n_max])
        self.last_exchange_records = n_total
        # pads sort to the end; only the first n_total records are real
        self.ops.sort_records(merged, merged.numel() if self.world > 1 else n_total, nq + 1)
        return self.ops.select(merged, n_total, nq, max_per_query)

    def similar(self, queries: torch.Tensor, thresh: int, max_per_query: int, scan_events=None):
        """All needles against the union of all shards.  Returns (ids[nq,k] i32 view of u32,
        scores[nq,k] i32, counts[nq] i32) -- identical on every rank."""
        nq = queries.numel()
        if nq >= (1 << 25):
            raise ValueError("at most 2^25-1 needles per call")
        while True:
            rec, total = self._buffers()
            total.zero_()
            if scan_events is not None:  # HIP events on the stream the scan kernel is launched on
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            self.ops.scan(queries, thresh, rec, total)
            if scan_events is not None:
                e1.record()
                scan_events.append((thresh, e0, e1))
            res = self._exchange_and_cut(rec, total, nq, max_per_query)
            if res is not None:
                return res

    def similar_sweep(self, queries: torch.Tensor, thresholds, max_per_query: int, scan_events=None,
                      find_events=None):
        """similar() for several thresholds, software-pipelined: while the records of threshold i are exchanged,
        sorted and cut on a side stream, the scan of threshold i+1 already runs on the main stream (two record
        buffers).  Returns {thresh: (ids, scores, counts)}, identical to calling similar() per threshold."""
        nq = queries.numel()
        if nq >= (1 << 25):
            raise ValueError("at most 2^25-1 needles per call")
        ops = self.ops
        if not hasattr(ops, "side_stream"):  # device work injected by a test: no streams, plain loop
            return {t: self.similar(queries, t, max_per_query, scan_events) for t in thresholds}
        main, side = ops.current_stream(), ops.side_stream()
        if getattr(self, "_sweep_bufs", None) is None or self._sweep_bufs[0][0].numel() < self.record_capacity:
            self._sweep_bufs = [(ops.empty(self.record_capacity, torch.int64), ops.empty(1, torch.int64))
                                for _ in range(2)]
        side.wait_stream(main)  # queries (and the index) are ready
        results, pending, reuse = {}, None, [None, None]

        def finish(p):
            thr, rec, total, ev_scan, f0 = p
            with ops.stream_ctx(side):
                side.wait_event(ev_scan)
                res = self._exchange_and_cut(rec, total, nq, max_per_query)
                done = ops.new_event()
                done.record(side)
            if res is None:  # did not fit: drain, grow (record_capacity was raised) and redo this one plainly
                main.wait_stream(side)
                self._sweep_bufs = None
                res = self.similar(queries, thr, max_per_query)
                done = ops.new_event()
                done.record(main)
            elif find_events is not None:
                find_events.append((thr, f0, done, int(self.last_exchange_records)))
            return res, done

        for i, thr in enumerate(thresholds):
            if self._sweep_bufs is None:  # a rescan replaced the buffers
                self._sweep_bufs = [(ops.empty(self.record_capacity, torch.int64), ops.empty(1, torch.int64))
                                    for _ in range(2)]
                reuse = [None, None]
            rec, total = self._sweep_bufs[i % 2]
            if reuse[i % 2] is not None:
                main.wait_event(reuse[i % 2])  # the post-processing that read this buffer has finished
            f0 = ops.new_event()
            f0.record(main)
            total.zero_()
            if scan_events is not None:
                e0, e1 = ops.new_event(), ops.new_event()
                e0.record(main)
            ops.scan(queries, thr, rec, total)
            if scan_events is not None:
                e1.record(main)
                scan_events.append((thr, e0, e1))
            ev_scan = ops.new_event()
            ev_scan.record(main)
            if pending is not None:
                j = pending[0]
                results[pending[1][0]], reuse[j] = finish(pending[1])
            pending = (i % 2, (thr, rec, total, ev_scan, f0))
        if pending is not None:
            results[pending[1][0]], _ = finish(pending[1])
        main.wait_stream(side)  # results were produced on the side stream
        return results


class NeedleParallel:
    """Data parallelism over needles, the reference's own strategy (QtConcurrent::map over the haystack items,
    src/database.cpp:1400-1432) across processes instead of threads: every rank holds the whole index (the
    fdct / video / ORB / colour indexes of BASELINE's configs are a few GB, a fraction of one GPU's 288 GB),
    takes a contiguous slice of the needle list, runs the index's batched find on it and the per-needle
    results are gathered.  No data-path collective: the only exchange is the final result gather."""

    def __init__(self, group=None) -> None:
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def my_slice(self, n: int):
        return ShardedDctHashIndex.shard_range(n, self.rank, self.world)

    def run(self, needles, find_batch, gather: bool = True):
        """find_batch(list_of_needles) -> list of per-needle results (any picklable objects).  Returns the
        results for ALL needles in order (gather=True) or just this rank's slice."""
        needles = list(needles)
        a, b = self.my_slice(len(needles))
        mine = find_batch(needles[a:b]) if b > a else []
        if self.world == 1 or not gather:
            return mine
        parts = [None] * self.world
        dist.all_gather_object(parts, mine, group=self.group)
        out = []
        for p in parts:
            out += p
        return out
