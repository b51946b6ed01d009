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

import os

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
        # similar_sweep calls this inside stream_ctx(side): the two tensors belong to the side stream's allocator pool
        # but are read by main-stream code afterwards.  Without this the caching allocator may hand the block to the
        # next sweep's side-stream cut while a main-stream reader of the old result is still queued.
        work = getattr(self, "_work", None)
        for t in (out, counts):
            if work is not None:
                t.record_stream(work)
            t.record_stream(torch.cuda.default_stream(self.torch_device))
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


def _force_collectives() -> bool:
    """CBH_DIST_FORCE_COLLECTIVES=1 with an initialised process group: run the collectives at world size 1 too, so a
    one-GPU box exercises the RCCL transport of every sharded index (same code path as R > 1)"""
    return dist.is_initialized() and os.environ.get("CBH_DIST_FORCE_COLLECTIVES") == "1"


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
        # CBH_DIST_FORCE_COLLECTIVES=1: run the collectives at world size 1 too (a one-GPU box can then exercise the
        # RCCL transport itself -- same code path as R > 1, the gathered buffer is just one block long)
        self.collective = self.world > 1 or _force_collectives()
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
        if not self.collective:
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
            allb = self.ops.empty((1 + cap) * self.world, torch.int64) if self.collective else blk
            b = self._bufs[slot] = (blk, allb)
        return b

    def _scan_into(self, blk, queries, thresh):
        blk[:1].zero_()
        self.ops.scan(queries, thresh, blk[1:], blk[:1])

    def _exchange_and_cut(self, blk, allb, nq: int, max_per_query: int, status):
        """(all-gather of the blocks) -> counting select.  No host synchronisation; `status` (int32[1] on the device)
        becomes non-zero when some rank's block overflowed."""
        if self.collective:
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

    def _grow_to_fit(self, slack: float = 1.5, granule: int = 1 << 14):
        """after an overflow: blocks of `slack` x the largest per-rank count seen (every rank computes the same value from
        the gathered counts), at least double the current size"""
        cur = self._block_cap()
        want = 2 * cur
        if self._max_count is not None:
            want = max(want, -(-int(int(self._max_count.item()) * slack) // granule) * granule)
        self._cap = want
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
        if os.environ.get("CBH_SWEEP_SERIAL") == "1":  # diagnostic knob: no side stream, one threshold after the other
            res = {}
            for t in thresholds:
                f0 = ops.new_event()
                f0.record(ops.current_stream())
                res[t] = self.similar(queries, t, max_per_query, scan_events)
                if find_events is not None:
                    done = ops.new_event()
                    done.record(ops.current_stream())
                    find_events.append((t, f0, done, self.last_exchange_records))
            return res
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
        if redo:  # blocks overflowed: grow ONCE for the sweep -- to 1.5 x the fullest rank's count (tracked on the
            # device, the same value on every rank) -- not once per overflowed threshold (4 overflows = blocks x16)
            self._grow_to_fit()
        for thr in redo:  # ... then these thresholds alone (their own events)
            f0 = ops.new_event()
            f0.record(main)
            results[thr] = self.similar(queries, thr, max_per_query, scan_events)
            if find_events is not None:
                done = ops.new_event()
                done.record(main)
                find_events.append((thr, f0, done, self.last_exchange_records))
        return results


def _gather_rows(rows, group=None, device=None, cap0: int = 4096, order_cols=None):
    """All ranks' int32 row lists [n_r, w] -> one array, ranks in order (or ordered by the unsigned columns
    `order_cols`, most significant first).  ONE fixed-size all_gather_into_tensor with the row count in word 0 of every
    block (the exchange pattern of ShardedDctHashIndex); a block that would not fit is noticed by every rank from the
    gathered counts and the gather is redone with larger blocks.  `device`: torch device the collective runs on (RCCL
    needs device tensors; gloo takes host tensors).  With a device the block is assembled, gathered, compacted and
    ordered THERE -- one upload of this rank's rows, one download of the final list; the host neither pads nor merges."""
    import numpy as np

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rows = np.ascontiguousarray(rows, np.int32)
    n, w = rows.shape
    if world == 1 and not _force_collectives():
        if order_cols and n:
            u = rows.view(np.uint32)
            rows = rows[np.lexsort(tuple(u[:, c] for c in reversed(order_cols)))]
        return rows
    tdev = torch.device("cpu") if device is None else device
    mine = torch.from_numpy(rows).to(tdev, non_blocking=True).reshape(-1)
    cap = cap0  # the same on every rank; grows to the largest count seen (also the same on every rank)
    while True:
        blk = torch.zeros(1 + cap * w, dtype=torch.int32, device=tdev)
        blk[0] = n
        m = min(n, cap)
        blk[1: 1 + m * w] = mine[: m * w]
        out = torch.empty((1 + cap * w) * world, dtype=torch.int32, device=tdev)
        dist.all_gather_into_tensor(out, blk, group=group)
        o = out.view(world, 1 + cap * w)
        counts = o[:, 0]
        cmax = int(counts.max().item())  # the one host read: every rank takes the same decision
        if cmax <= cap:
            body = o[:, 1:].reshape(world, cap, w)
            live = torch.arange(cap, device=tdev)[None, :] < counts[:, None]
            allr = body[live]  # ranks in order, rows in order
            if order_cols and allr.shape[0]:
                key = torch.zeros(allr.shape[0], dtype=torch.int64, device=tdev)
                for c in order_cols:  # unsigned 32-bit columns folded into one 63-bit key (two columns at most)
                    key = (key << 32) | (allr[:, c].to(torch.int64) & 0xFFFFFFFF)
                allr = allr[torch.argsort(key, stable=True)]
            return allr.cpu().numpy()
        cap = cmax  # same value on every rank


class ShardedDctVideoIndex:
    """DctVideoIndex sharded BY VIDEO: rank r holds the videos [r*V/R, (r+1)*V/R) of the load order, so
    the closest-frame-per-video reduce and the adjacency scoring of findVideo are local to the shard that owns the
    video.  Needles are replicated; every rank searches all of them against its videos (one batched launch), and
    ONE all-gather of the final matches (needle, mediaId, score, range: 24 bytes each) gives every rank the complete
    result.  A union of disjoint video sets ordered by mediaId is exactly the single-index result
    (src/dctvideoindex.cpp:475-509,595-654 emit per video, in std::map<mediaId> order)."""

    def __init__(self, make_index, group=None, device=None) -> None:
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.index = make_index()  # a cbird_amd.video.DctVideoIndex (or a stand-in with add / find_videos_batch)
        self.device = device

    def add(self, media) -> None:
        """every rank is handed the same list; it keeps its contiguous share"""
        media = list(media)
        a, b = ShardedDctHashIndex.shard_range(len(media), self.rank, self.world)
        self.index.add(media[a:b])

    def find_videos_batch(self, needles, p):
        import numpy as np

        from .index import Match, MatchRange

        needles = list(needles)
        local = self.index.find_videos_batch(needles, p)
        rows = np.array([(k, m.mediaId, m.score, m.range.srcIn, m.range.dstIn, m.range.len)
                         for k, r in enumerate(local) for m in r], np.int64).reshape(-1, 6)
        rows = rows.astype(np.uint32).view(np.int32) if len(rows) else np.zeros((0, 6), np.int32)
        # gathered, compacted and ordered by (needle, mediaId) on the collective's device
        allr = _gather_rows(rows, self.group, self.device, order_cols=(0, 1))
        out = [[] for _ in needles]
        ids = allr[:, 1].view(np.uint32) if len(allr) else np.zeros(0, np.uint32)
        for i in range(len(allr)):
            r = allr[i]
            out[int(r[0])].append(Match(int(ids[i]), int(r[2]), MatchRange(int(r[3]), int(r[4]), int(r[5]))))
        return out


class ShardedCvFeaturesIndex:
    """CvFeaturesIndex sharded BY IMAGE: rank r holds the descriptor rows of its images, so the
    first-row -> mediaId map is local.  Per needle descriptor every rank computes its local k nearest rows
    (distance, global row, mediaId); one all-gather of those fixed-size tables, a k-way merge of the R sorted lists per
    descriptor by (distance, global row) -- the order of the unsharded knn -- and the reference's scoring
    (src/cvfeaturesindex.cpp:499-596, cbh_cvfeatures_score) on the merged table, identically on every rank."""

    def __init__(self, make_index, group=None, device=None) -> None:
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.index = make_index()  # a cbird_amd.cvfeatures.CvFeaturesIndex (or a stand-in with add / knn_media / count)
        self.device = device
        self.row_offset = 0

    def add(self, media) -> None:
        """every rank is handed the same list (ascending media id, as load() does); it keeps its contiguous share and
        remembers how many rows the ranks before it hold (global row = local row + offset: the tie-break of the knn)"""
        media = list(media)
        a, b = ShardedDctHashIndex.shard_range(len(media), self.rank, self.world)
        self.row_offset = sum(len(m.keyPointDescriptors) for m in media[:a])
        self.index.add(media[a:b])

    def find_batch(self, needles, p, knn: int = 10):
        import numpy as np

        from .index import Match

        needles = list(needles)
        rows = [np.ascontiguousarray(m.keyPointDescriptors, np.uint8).reshape(-1, 32) for m in needles]
        offs = np.zeros(len(needles) + 1, np.uint64)
        np.cumsum([len(r) for r in rows], out=offs[1:])
        allq = np.concatenate(rows) if rows else np.zeros((0, 32), np.uint8)
        nq = len(allq)
        row, dst, media, cnt = self.index.knn_media(allq, knn, int(p.cvThresh))
        # sortable key per candidate: distance << 40 | global row; empty places sort last
        key = (dst.astype(np.int64) << 40) | (row.astype(np.int64) + self.row_offset)
        key[np.arange(knn)[None, :] >= np.minimum(cnt, knn)[:, None]] = np.iinfo(np.int64).max
        if self.world > 1 or _force_collectives():
            tab = np.concatenate([key.view(np.int32).reshape(nq, -1), media.view(np.int32),
                                  cnt.view(np.int32)[:, None]], 1)  # [nq, 2k + k + 1] int32, fixed size
            t = torch.from_numpy(np.ascontiguousarray(tab)).reshape(-1)
            src = t if self.device is None else t.to(self.device, non_blocking=True)
            out = torch.empty(self.world * t.numel(), dtype=torch.int32, device=src.device)
            dist.all_gather_into_tensor(out, src, group=self.group)
            # k-way merge of the R sorted candidate lists of every needle descriptor where the all-gather left them
            # (the collective's device): one sort of [nq, R*k] keys, the first k kept; only the merged table
            # ([nq, k] keys + mediaIds + counts) comes back to the host, which scores it
            o = out.view(self.world, nq, 3 * knn + 1)
            keys = o[:, :, : 2 * knn].contiguous().view(torch.int64).permute(1, 0, 2).reshape(nq, -1)  # [nq, R*k]
            medias = o[:, :, 2 * knn: 3 * knn].permute(1, 0, 2).reshape(nq, -1)
            cnts = (o[:, :, 3 * knn].to(torch.int64) & 0xFFFFFFFF).sum(0)
            skey, order = torch.sort(keys, dim=1, stable=True)
            key = skey[:, :knn].cpu().numpy()
            media = torch.gather(medias, 1, order[:, :knn]).cpu().numpy().view(np.uint32)
            cnt = np.minimum(cnts.cpu().numpy(), np.iinfo(np.uint32).max).astype(np.uint32)
        dst = np.where(key == np.iinfo(np.int64).max, 0, key >> 40).astype(np.uint16)
        media = np.ascontiguousarray(media, np.uint32)
        cnt = np.ascontiguousarray(np.minimum(cnt, knn), np.uint32)  # places filled in the merged table
        L = _lib.lib()
        cap = nq * knn + 1
        buf = (_lib.cbh_match * cap)()
        out_offs = np.zeros(len(needles) + 1, np.uint64)
        _lib.check(L.cbh_cvfeatures_score(media.ctypes.data, np.ascontiguousarray(dst).ctypes.data, cnt.ctypes.data,
                                          offs.ctypes.data, len(needles), knn, buf, cap, out_offs.ctypes.data), "score")
        return [[Match(buf[j].id, buf[j].score) for j in range(int(out_offs[i]), int(out_offs[i + 1]))]
                for i in range(len(needles))]


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
