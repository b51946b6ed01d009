"""K4, the counting select of cbird_amd/csrc/topk.hip (cbh_records_topk_dev): per needle the first k records in
ascending (score, mediaId) order + the count, from UNORDERED records in { count, records[cap] } blocks -- against a
numpy sort of the same records (what Database::searchIndex's std::sort + maxMatches cut produces,
src/database.cpp:1729-1737, with ties fixed to ascending mediaId)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_records(rng, nq, n, long_needles=(), dup_frac=0.0):
    q = rng.integers(0, nq, n).astype(np.uint64)
    for j, m in long_needles:  # needles with long result lists (duplicates / videos)
        q[rng.choice(n, m, replace=False)] = j
    d = rng.integers(0, 65, n).astype(np.uint64)
    ids = rng.integers(1, 2 ** 32, n, dtype=np.uint64)
    r = (q << np.uint64(39)) | (d << np.uint64(32)) | ids
    nd = int(n * dup_frac)
    if nd:
        r[rng.choice(n, nd, replace=False)] = r[rng.choice(n, nd)]  # identical (needle, score, id) records
    return r


def reference_cut(r, nq, k):
    r = np.sort(r)
    qi = (r >> np.uint64(39)).astype(np.int64)
    cnt = np.bincount(qi[qi < nq], minlength=nq).astype(np.uint32)
    start = np.searchsorted(qi, np.arange(nq))
    ids = np.zeros((nq, max(k, 1)), np.uint32)
    sc = np.zeros((nq, max(k, 1)), np.int32)
    for j in np.nonzero(cnt)[0]:
        m = min(k, int(cnt[j]))
        seg = r[start[j]: start[j] + m]
        ids[j, :m] = (seg & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        sc[j, :m] = ((seg >> np.uint64(32)) & np.uint64(0x7F)).astype(np.int32)
    return ids[:, :k], sc[:, :k], cnt


def run_topk(blocks_np, nb, stride, cap, nq, k):
    import torch

    from cbird_amd import _lib

    L = _lib.lib()
    blocks = torch.from_numpy(blocks_np.view(np.int64)).cuda()
    out = torch.full((nq, max(k, 1), 2), -7, dtype=torch.int32, device="cuda")
    counts = torch.full((nq,), -7, dtype=torch.int32, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(L.cbh_records_topk_dev(blocks.data_ptr(), nb, stride, cap, nq, k, out.data_ptr(), counts.data_ptr(),
                                      status.data_ptr(), 0, None), "topk")
    o = out.cpu().numpy()
    return o[:, :k, 0].view(np.uint32), o[:, :k, 1], counts.cpu().numpy().view(np.uint32), int(status.item())


@pytest.mark.parametrize("nq,n,k", [(1000, 3000, 8), (1, 500, 5), (70000, 100000, 4), (513, 20000, 64), (300, 0, 3),
                                    (4097, 9000, 1), (100, 5000, 0)])
def test_single_block_equals_sorted_cut(gpu, nq, n, k):
    rng = np.random.default_rng(nq * 31 + n + k)
    r = make_records(rng, nq, n, dup_frac=0.05)
    blk = np.concatenate([[np.uint64(n)], r, rng.integers(0, 2 ** 63, 17, dtype=np.uint64)])  # junk past the count
    gi, gs, gc, st = run_topk(blk, 1, 0, n + 17, nq, k)
    wi, ws, wc = reference_cut(r, nq, k)
    assert st == 0 and (gc == wc).all()
    assert (gi == wi).all() and (gs == ws).all()


def test_long_segments_and_duplicates(gpu):
    """needles with thousands of matches take the workgroup-per-needle path; equal records keep their multiplicity"""
    rng = np.random.default_rng(5)
    nq, n = 2000, 60000
    r = make_records(rng, nq, n, long_needles=[(7, 9000), (1999, 20000), (0, 65), (1000, 64)], dup_frac=0.3)
    r[:40] = (np.uint64(7) << np.uint64(39)) | np.uint64(5)  # forty identical best records of needle 7
    for k in (3, 10, 64):
        blk = np.concatenate([[np.uint64(n)], r])
        gi, gs, gc, st = run_topk(blk, 1, 0, n, nq, k)
        wi, ws, wc = reference_cut(r, nq, k)
        assert st == 0 and (gc == wc).all() and (gi == wi).all() and (gs == ws).all(), k


def test_multi_block_exchange_layout_and_overflow(gpu):
    """R blocks as one all_gather_into_tensor delivers them: ragged counts, an empty block, pads ignored; a block
    whose count exceeds cap raises the status bit (and its surviving records still take part)"""
    rng = np.random.default_rng(9)
    nq, cap, nb = 5000, 4096, 4
    stride = cap + 1
    counts = [4096, 0, 1234, 17]
    blocks = rng.integers(0, 2 ** 63, nb * stride, dtype=np.uint64)  # junk everywhere first
    parts = []
    for b, c in enumerate(counts):
        r = make_records(rng, nq, c)
        blocks[b * stride] = c
        blocks[b * stride + 1: b * stride + 1 + c] = r
        parts.append(r)
    allr = np.concatenate(parts)
    gi, gs, gc, st = run_topk(blocks, nb, stride, cap, nq, 6)
    wi, ws, wc = reference_cut(allr, nq, 6)
    assert st == 0 and (gc == wc).all() and (gi == wi).all() and (gs == ws).all()
    blocks[2 * stride] = cap + 5  # claims more than fits: records 1234..4095 of that block are junk-but-read
    blocks[2 * stride + 1: 3 * stride] = make_records(rng, nq, cap)
    allr = np.concatenate([parts[0], blocks[2 * stride + 1: 3 * stride], parts[3]])
    gi, gs, gc, st = run_topk(blocks, nb, stride, cap, nq, 6)
    wi, ws, wc = reference_cut(allr, nq, 6)
    assert st == 1 and (gc == wc).all() and (gi == wi).all()


def test_find_batch_uses_the_counting_select_and_matches_the_sort_path(gpu, orc):
    """k <= 64 runs topk.hip, k > 64 the radix sort: same answers, both equal to the oracle"""
    from cbird_amd import synth

    h, ids = synth.make_hashes(30000, seed=3, planted_frac=0.2)
    idx = gpu.DctHashIndex()
    idx.load(h, ids)
    q = h[:5000]
    a = idx.find_batch(q, 9, 64)
    b = idx.find_batch(q, 9, 65)
    w = orc.find64_batch(h, ids, q, 9, 65)
    assert (a[2] == b[2]).all() and (b[2] == w[2]).all()
    assert (a[0] == b[0][:, :64]).all() and (a[1] == b[1][:, :64]).all()
    assert (b[0] == w[0]).all() and (b[1] == w[1]).all()
