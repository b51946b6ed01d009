"""Sharded DctHashIndex across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

The reference has no distributed mode (its only fan-out is QtConcurrent::map over needles,
src/database.cpp:1400-1432); this is the MI355X-native counterpart designed from scratch:

  * the haystack is row-sharded: rank r owns a contiguous slice of (hash, mediaId) slots;
    needles are replicated (an all-pairs job gathers every rank's freshly built hashes once);
  * each rank scans ITS shard for ALL needles with the same HIP kernel as the single-GPU path and
    produces an unordered list of cbh_record (needle<<39 | distance<<32 | mediaId);
  * ONE exchange step: a single all_gather_into_tensor of fixed-size blocks { count, records[cap] } (the
    count travels in word 0 -- no sizes-first round, no host synchronisation).  A threshold search over a
    union of shards is the union of the per-shard results;
  * every rank then cuts each needle's list at max_per_query with the counting select (topk.hip), which
    reads the R blocks where the all-gather put them.

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

    def topk(self, blocks: torch.Tensor, nb: int, stride: int, cap: int, nq: int, k: int, status: torch.Tensor):
        """K4 counting select over nb blocks { count, records[cap] } (cbh_records_topk_dev); status: int32[1]"""
        out = torch.empty((nq, max(k, 1), 2), dtype=torch.int32, device=self.torch_device)
        counts = torch.empty(nq, dtype=torch.int32, device=self.torch_device)
        _lib.check(self.L.cbh_records_topk_dev(blocks.data_ptr(), nb, stride, cap, nq, k, out.data_ptr(),
                                               counts.data_ptr(), status.data_ptr(), self.device, self._stream()),
                   "records_topk_dev")
        return out[:, :k, 0], out[:, :k, 1], counts

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
    """One exchange step per threshold, and nothing on it synchronises with the host:

        scan     each rank writes ONE block { u64 count; u64 records[cap] } (count = word 0, written by the scan kernel)
        gather   one all_gather_into_tensor of the R fixed-size blocks                       (R > 1 only)
        cut      counting select over the R blocks (cbh_records_topk_dev): per needle the first max_per_query
                 matches in ascending (score, mediaId) order + the match count -- identical on every rank

    A rank whose matches exceed cap is noticed by the cut itself (status word on the device, derived from the gathered
    counts, hence the same on every rank); the caller looks at it once per call / per sweep and redoes the thresholds
    concerned with a larger cap."""

    def __init__(self, ops, group=None, record_capacity: int = 1 << 22) -> None:
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # record_capacity = what the whole job may produce per threshold; a rank's block holds its share (x2 slack)
        self.record_capacity = record_capacity
        self._bufs = {}
        self._cap = None          # explicit block capacity (fit_capacity), else derived from record_capacity
        self._max_count = None    # 0-dim tensor: largest per-rank count seen since the last fit_capacity()
        self.last_exchange_records = None  # 0-dim tensor (device): records of the last cut, all ranks together

    def _block_cap(self) -> int:
        if self._cap is not None:
            return self._cap
        return max(64, -(-2 * self.record_capacity // self.world) if self.world > 1 else self.record_capacity)

    def fit_capacity(self, slack: float = 1.5, granule: int = 1 << 14) -> int:
        """Size the exchange blocks to `slack` x the largest per-rank match count seen so far (call after a warm-up
        pass; one host read).  Every rank computes the same value: the counts come from the gathered blocks.  Smaller
        blocks = a smaller all-gather per threshold; a later overflow still just doubles them."""
        if self._max_count is None:
            return self._block_cap()
        m = int(self._max_count.item())
        self._cap = max(granule, -(-int(m * slack) // granule) * granule)
        self._bufs = {}
        self._max_count = None
        return self._cap

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
    def _buffers(self, slot: int = 0):
        """(block, gathered blocks, status) of pipeline slot `slot`; block = int64[1 + cap], word 0 = count"""
        cap = self._block_cap()
        b = self._bufs.get(slot)
        if b is None or b[0].numel() != 1 + cap:
            blk = self.ops.empty(1 + cap, torch.int64)
            allb = self.ops.empty((1 + cap) * self.world, torch.int64) if self.world > 1 else blk
            b = self._bufs[slot] = (blk, allb)
        return b

    def _scan_into(self, blk, queries, thresh):
        blk[:1].zero_()
        self.ops.scan(queries, thresh, blk[1:], blk[:1])

    def _exchange_and_cut(self, blk, allb, nq: int, max_per_query: int, status):
        """(all-gather of the blocks) -> counting select.  No host synchronisation; `status` (int32[1] on the device)
        becomes non-zero when some rank's block overflowed."""
        if self.world > 1:
            self._all_gather(allb, blk)
        cap = blk.numel() - 1
        cnts = allb[:: 1 + cap]
        self.last_exchange_records = cnts.clamp(max=cap).sum()
        self._max_count = cnts.max() if self._max_count is None else torch.maximum(self._max_count, cnts.max())
        return self.ops.topk(allb, self.world, 1 + cap, cap, nq, max_per_query, status)

    def _grow(self):
        # same decision on every rank (the status word is derived from all ranks' counts)
        if self._cap is not None:
            self._cap *= 2
        else:
            self.record_capacity *= 2
        self._bufs = {}

    def similar(self, queries: torch.Tensor, thresh: int, max_per_query: int, scan_events=None):
        """All needles against the union of all shards.  Returns (ids[nq,k] i32 view of u32,
        scores[nq,k] i32, counts[nq] i32) -- identical on every rank."""
        nq = queries.numel()
        if nq >= (1 << 25):
            raise ValueError("at most 2^25-1 needles per call")
        while True:
            blk, allb = self._buffers(0)
            status = self.ops.empty(1, torch.int32)
            status.zero_()
            if scan_events is not None:  # HIP events on the stream the scan kernel is launched on
                e0, e1 = self.ops.new_event(), self.ops.new_event()
                e0.record()
            self._scan_into(blk, queries, thresh)
            if scan_events is not None:
                e1.record()
            res = self._exchange_and_cut(blk, allb, nq, max_per_query, status)
            if int(status.item()) == 0:
                if scan_events is not None:
                    scan_events.append((thresh, e0, e1))
                return res
            self._grow()

    def similar_sweep(self, queries: torch.Tensor, thresholds, max_per_query: int, scan_events=None,
                      find_events=None):
        """similar() for several thresholds, software-pipelined: while the records of threshold i are exchanged
        and cut on a side stream, the scan of threshold i+1 already runs on the main stream (two blocks).  The host
        never waits inside the loop; the overflow words of all thresholds are read once at the end and the
        thresholds concerned are redone.  Returns {thresh: (ids, scores, counts)}, identical to calling similar()
        per threshold."""
        nq = queries.numel()
        if nq >= (1 << 25):
            raise ValueError("at most 2^25-1 needles per call")
        ops = self.ops
        thresholds = list(thresholds)
        if not hasattr(ops, "side_stream"):  # device work injected by a test: no streams, plain loop
            return {t: self.similar(queries, t, max_per_query, scan_events) for t in thresholds}
        main, side = ops.current_stream(), ops.side_stream()
        side.wait_stream(main)  # queries (and the index) are ready
        status = ops.empty(len(thresholds), torch.int32)
        status.zero_()
        results, pending, reuse = {}, None, [None, None]
        scan_tmp, find_tmp = [], []

        def finish(p):
            i, thr, blk, allb, ev_scan, f0 = p
            with ops.stream_ctx(side):
                side.wait_event(ev_scan)
                res = self._exchange_and_cut(blk, allb, nq, max_per_query, status[i: i + 1])
                nrec = self.last_exchange_records
                done = ops.new_event()
                done.record(side)
            find_tmp.append((thr, f0, done, nrec))
            return res, done

        for i, thr in enumerate(thresholds):
            blk, allb = self._buffers(i % 2)
            if reuse[i % 2] is not None:
                main.wait_event(reuse[i % 2])  # the cut that read this block has finished
            f0 = ops.new_event()
            f0.record(main)
            e0, e1 = ops.new_event(), ops.new_event()
            e0.record(main)
            self._scan_into(blk, queries, thr)
            e1.record(main)
            scan_tmp.append((thr, e0, e1))
            ev_scan = ops.new_event()
            ev_scan.record(main)
            if pending is not None:
                results[pending[1]], reuse[pending[0] % 2] = finish(pending)
            pending = (i, thr, blk, allb, ev_scan, f0)
        if pending is not None:
            results[pending[1]], _ = finish(pending)
        main.wait_stream(side)  # results were produced on the side stream
        st = status.cpu().tolist()  # the one synchronisation of the sweep
        redo = [t for t, s in zip(thresholds, st) if s]
        if scan_events is not None:
            scan_events += [e for e in scan_tmp if e[0] not in redo]
        if find_events is not None:
            find_events += [e for e in find_tmp if e[0] not in redo]
        for thr in redo:  # a block overflowed: larger blocks, this threshold alone (its own events)
            self._grow()
            f0 = ops.new_event()
            f0.record(main)
            results[thr] = self.similar(queries, thr, max_per_query, scan_events)
            if find_events is not None:
                done = ops.new_event()
                done.record(main)
                find_events.append((thr, f0, done, self.last_exchange_records))
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
