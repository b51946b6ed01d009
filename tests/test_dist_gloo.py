"""CPU suite, world_size 2 over gloo: the sharding + single fixed-size all-gather exchange (count in word 0 of every
block, overflow noticed by the cut and redone with larger blocks) of cbird_amd.dist.ShardedDctHashIndex.  The device work is injected (numpy + oracle stand-ins living
in this test file); the product's HipOps is covered by the gpu-marked tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401


class FakeOps:
    """CPU stand-in for cbird_amd.dist.HipOps (test infrastructure: uses the oracle)."""

    def __init__(self):
        from oracle import Oracle

        self.orc = Oracle()
        self.h = self.ids = None

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype)

    def hash_images(self, imgs):
        return torch.from_numpy(self.orc.dcthash64_batch(imgs.numpy()).view(np.int64))

    def load_shard(self, hashes, ids):
        self.h = hashes.numpy().view(np.uint64).copy()
        self.ids = ids.numpy().view(np.uint32).copy()

    def scan(self, queries, thresh, rec, total):
        q = queries.numpy().view(np.uint64)
        out = []
        for j, t in enumerate(q.tolist()):
            i, d = self.orc.scan64(self.h, self.ids, t, thresh)
            out += [(j << 39) | (int(dd) << 32) | int(ii) for ii, dd in zip(i, d)]
        n = len(out)
        m = min(n, rec.numel())
        rec[:m] = torch.tensor(out[:m][::-1], dtype=torch.int64)  # deliberately unordered
        total += n

    def topk(self, blocks, nb, stride, cap, nq, k, status):
        """numpy statement of cbh_records_topk_dev: blocks of { count, records[cap] } -> per-needle cut"""
        b = blocks.numpy().view(np.uint64)
        recs = []
        for i in range(nb):
            c = int(b[i * stride])
            if c > cap:
                status |= 1
            recs.append(b[i * stride + 1: i * stride + 1 + min(c, cap)])
        r = np.sort(np.concatenate(recs)) if recs else np.zeros(0, np.uint64)
        qi = (r >> np.uint64(39)).astype(np.int64)
        ids = np.zeros((nq, k), np.int32)
        sc = np.zeros((nq, k), np.int32)
        cnt = np.bincount(qi[qi < nq], minlength=nq).astype(np.int32)
        start = np.searchsorted(qi, np.arange(nq))
        for j in range(nq):
            m = min(k, cnt[j])
            seg = r[start[j]: start[j] + m]
            ids[j, :m] = (seg & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
            sc[j, :m] = ((seg >> np.uint64(32)) & np.uint64(0x7F)).astype(np.int32)
        return torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt)

    def new_event(self):
        raise AssertionError("no events on the CPU stand-in")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ragged, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd import synth
        from cbird_amd.dist import ShardedDctHashIndex

        h, ids = synth.make_hashes(n, seed=77, planted_frac=0.3)
        h[5] = 0  # a null needle / uncomputed hash
        sh = ShardedDctHashIndex(FakeOps(), record_capacity=64 if ragged else 1 << 16)
        a, b = sh.shard_range(n, rank, world)
        local = torch.from_numpy(h[a:b].view(np.int64).copy())
        allq = sh.gather_hashes(local, n)
        assert (allq.numpy().view(np.uint64) == h).all()
        sh.load_shard(local, torch.from_numpy(ids[a:b].view(np.int32).copy()))
        res = {}
        for dht in (2, 7):
            i, s, c = sh.similar(allq, dht, 4)
            res[dht] = (i.numpy().copy(), s.numpy().copy(), c.numpy().copy())
        q_out.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,ragged,world", [(601, False, 2), (1000, True, 2), (457, True, 3)])
def test_sharded_similar_world2_equals_single(n, ragged, world):
    from cbird_amd import synth
    from oracle import Oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ragged, q)) for r in range(world)]
    [p.start() for p in procs]
    results = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    h, ids = synth.make_hashes(n, seed=77, planted_frac=0.3)
    h[5] = 0
    orc = Oracle()
    for dht in (2, 7):
        wi, ws, wc = orc.find64_batch(h, ids, h, dht, 4)
        for r in range(world):
            gi, gs, gc = results[r][dht]
            assert (gc == wc.astype(np.int32)).all(), (dht, r)
            assert (gi.view(np.uint32) == wi).all() and (gs == ws).all(), (dht, r)


def test_shard_ranges_cover_everything():
    from cbird_amd.dist import ShardedDctHashIndex

    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [ShardedDctHashIndex.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def _np_worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbird_amd.dist import NeedleParallel

        npar = NeedleParallel()
        calls = []

        def fb(chunk):
            calls.append(len(chunk))
            return [(x, x * x) for x in chunk]

        res = npar.run(list(range(11)), fb)
        q_out.put((rank, res, calls, npar.my_slice(11)))
    finally:
        dist.destroy_process_group()


def test_needle_parallel_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_np_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    slices = sorted(g[3] for g in got)
    assert slices == [(0, 5), (5, 11)]
    for rank, res, calls, sl in got:
        assert res == [(x, x * x) for x in range(11)]
        assert calls == [sl[1] - sl[0]]
