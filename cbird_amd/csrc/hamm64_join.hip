// hamm64_join.hip -- the 64-bit Hamming threshold search as a bucketed join (multi-index hashing) for small thresholds.
//
// The same predicate as hamm64_scan.hip / hamm64_mfma.hip (DctHashIndex::find, src/dcthashindex.cpp:193-220:
// hamm64(q, hash[i]) < thresh && id[i] != 0, q != 0), the same records -- but not every pair is looked at.  Two 64-bit
// words that differ in at most d = thresh - 1 bits agree on at least one of any m > d disjoint chunks of their bits
// (pigeonhole), so with m = max(4, thresh) chunks of 64 / m bits a needle has to be compared only with the slots that share
// one of its m chunk values.  On hashes whose bits are close to independent (the bench's 10^6 image hashes, NOTES 15) that is
// 1 / 10 000 of the all-pairs scan at thresholds <= 4 and 1 / 28 at threshold 8; on a library of near-identical hashes it is
// MORE work than the scan -- the launcher counts the candidate pairs exactly (sum over chunk values of slots x needles, from
// the two histograms) before it commits, and hands the call back (CBH_E_UNSUPPORTED) when the matrix-core scan is cheaper.
//
//   k_join_hist        a histogram of every chunk's values, slots and needles alike (one atomic per item and chunk; every
//                      n / 16384-th item only for the sampled pre-check, whose pairs k_join_pairs_only adds up)
//   k_join_scan        per chunk: exclusive scans -> where each value's slots / needles start; the jobs of the wide join (one
//                      per 512 slots x 2048 needles of a value) and their prefix; the chunk's candidate pairs
//   k_join_scatter     slots (hash, id) and needles (hash, needle index) in chunk-value order, one copy per chunk
//   k_join_pairs       chunks of <= 11 bits: a workgroup per job, two slots per lane, the value's needles through the scalar
//                      cache eight at a time, one min-test per block and the exact look only behind it
//   k_join_narrow      chunks of 12-13 bits (threshold 5): a lane per slot walks its value's needles
//   k_join_by_needle   four 16-bit chunks (thresholds <= 4): only the slots' side is prepared; a lane per NEEDLE walks the
//                      ~15 slots of each of its four values
// A pair that also agrees on an EARLIER chunk is that chunk's to report; records are parked per wave in LDS and appended with
// one atomic per flush.  Scratch from the stream-ordered arena; nothing is cached on the index (a call at 10^6 x 10^6 costs
// 0.8 ms of bookkeeping at thresholds <= 4, 1.7 ms above, besides the join proper).
#include "cbh_internal.h"

#include <atomic>

namespace cbh {
namespace {

constexpr int kJT = 256;        // threads
constexpr uint32_t kJH = 2 * kJT;  // slots per job of the wide join (two per lane)
constexpr uint32_t kJQ = 2048;  // needles per job (a value's needles beyond that make further jobs)
constexpr int kMaxChunks = 8;
constexpr uint32_t kOutCap = 96;  // records a wave parks before it appends them

struct JoinPlan {
  int m;                       // chunks
  int lo[kMaxChunks + 1];      // chunk j = bits [lo[j], lo[j + 1])
  uint32_t voff[kMaxChunks + 1];  // chunk j's values start at voff[j] in the per-value arrays (+ j for the "+1" slots)
};

__device__ __forceinline__ uint32_t chunk_of(uint64_t h, int lo, int hi) {
  return (uint32_t)((h >> lo) & ((1ull << (hi - lo)) - 1ull));
}

// hist[voff[j] + j + value] += 1 for every item and chunk (the arrays carry one extra entry per chunk for the scans' ends)
__global__ __launch_bounds__(256) void k_join_hist(const uint64_t* __restrict__ x, uint32_t n, uint32_t stride, JoinPlan P,
                                                   uint32_t* __restrict__ hist) {
  const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * stride;  // (stride > 1: the sampled pre-check)
  if (i >= n) return;
  const uint64_t h = x[i];
#pragma unroll 1
  for (int j = 0; j < P.m; ++j) atomicAdd(&hist[P.voff[j] + (uint32_t)j + chunk_of(h, P.lo[j], P.lo[j + 1])], 1u);
}

// One workgroup per chunk.  In: the two histograms.  Out: start_h / start_q (exclusive scans, nv + 1 entries per chunk),
// jobstart (exclusive scan of ceil(nh / 256) * ceil(nq / kJQ) per value, nv + 1 entries), stats[j] = {jobs, 0, pairs lo, hi}.
__global__ __launch_bounds__(1024) void k_join_scan(JoinPlan P, const uint32_t* __restrict__ hist_h,
                                                    const uint32_t* __restrict__ hist_q, uint32_t* __restrict__ start_h,
                                                    uint32_t* __restrict__ start_q, uint32_t* __restrict__ jobstart,
                                                    unsigned long long* __restrict__ stats) {
  const int j = blockIdx.x;
  const uint32_t nv = 1u << (P.lo[j + 1] - P.lo[j]);
  const uint32_t off = P.voff[j] + (uint32_t)j;
  __shared__ uint32_t sa[1024], sb[1024], sc[1024];
  __shared__ unsigned long long sp[1024];
  const uint32_t per = (nv + 1023u) / 1024u, v0 = threadIdx.x * per, v1 = min(nv, v0 + per);
  uint32_t a = 0, b = 0, c = 0;
  unsigned long long p = 0;
  for (uint32_t v = v0; v < v1; ++v) {
    const uint32_t nh = hist_h[off + v], nq = hist_q[off + v];
    a += nh;
    b += nq;
    c += (nh && nq) ? ((nh + kJH - 1u) / kJH) * ((nq + kJQ - 1u) / kJQ) : 0u;
    p += (unsigned long long)nh * nq;
  }
  sa[threadIdx.x] = a, sb[threadIdx.x] = b, sc[threadIdx.x] = c, sp[threadIdx.x] = p;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {  // inclusive scans of the per-thread sums
    const uint32_t ta = threadIdx.x >= d ? sa[threadIdx.x - d] : 0u, tb = threadIdx.x >= d ? sb[threadIdx.x - d] : 0u,
                   tc = threadIdx.x >= d ? sc[threadIdx.x - d] : 0u;
    const unsigned long long tp_ = threadIdx.x >= d ? sp[threadIdx.x - d] : 0ull;
    __syncthreads();
    sa[threadIdx.x] += ta, sb[threadIdx.x] += tb, sc[threadIdx.x] += tc, sp[threadIdx.x] += tp_;
    __syncthreads();
  }
  uint32_t ea = sa[threadIdx.x] - a, eb = sb[threadIdx.x] - b, ec = sc[threadIdx.x] - c;
  for (uint32_t v = v0; v < v1; ++v) {
    const uint32_t nh = hist_h[off + v], nq = hist_q[off + v];
    start_h[off + v] = ea, start_q[off + v] = eb, jobstart[off + v] = ec;
    ea += nh;
    eb += nq;
    ec += (nh && nq) ? ((nh + kJH - 1u) / kJH) * ((nq + kJQ - 1u) / kJQ) : 0u;
  }
  if (threadIdx.x == 1023) {
    start_h[off + nv] = sa[1023], start_q[off + nv] = sb[1023], jobstart[off + nv] = sc[1023];
    stats[2 * j] = sc[1023];
    stats[2 * j + 1] = sp[1023];
  }
}

// the sampled pre-check needs the candidate pairs only: stats[2 j + 1] = sum over chunk j's values of slots x needles
__global__ __launch_bounds__(1024) void k_join_pairs_only(JoinPlan P, const uint32_t* __restrict__ hist_h,
                                                          const uint32_t* __restrict__ hist_q,
                                                          unsigned long long* __restrict__ stats) {
  const int j = blockIdx.x;
  const uint32_t nv = 1u << (P.lo[j + 1] - P.lo[j]), off = P.voff[j] + (uint32_t)j;
  __shared__ unsigned long long sp[1024];
  unsigned long long p = 0;
  for (uint32_t v = threadIdx.x; v < nv; v += 1024u) p += (unsigned long long)hist_h[off + v] * hist_q[off + v];
  sp[threadIdx.x] = p;
  __syncthreads();
  for (uint32_t d = 512; d > 0; d >>= 1) {
    if (threadIdx.x < d) sp[threadIdx.x] += sp[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) stats[2 * j] = 0ull, stats[2 * j + 1] = sp[0];
}

// item i of x goes to position start[value] + (its turn among the value's items) of chunk j's copy; aux = ids (slots) or
// nullptr (needles: the item's own index)
__global__ __launch_bounds__(256) void k_join_scatter(const uint64_t* __restrict__ x, const uint32_t* __restrict__ aux,
                                                      uint32_t n, JoinPlan P, const uint32_t* __restrict__ start,
                                                      uint32_t* __restrict__ cursor, uint64_t* __restrict__ out_x,
                                                      uint32_t* __restrict__ out_aux) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = x[i];
  const uint32_t a = aux ? aux[i] : i;
#pragma unroll 1
  for (int j = 0; j < P.m; ++j) {
    const uint32_t v = P.voff[j] + (uint32_t)j + chunk_of(h, P.lo[j], P.lo[j + 1]);
    const uint32_t pos = start[v] + atomicAdd(&cursor[v], 1u);
    out_x[(size_t)j * n + pos] = h;
    out_aux[(size_t)j * n + pos] = a;
  }
}

// per-wave record buffer in LDS: `has` lanes append {needle, dist, id}; one atomic per flush
__device__ __forceinline__ void join_flush(uint64_t* __restrict__ buf, uint32_t& cnt, cbh_record* __restrict__ rec,
                                           unsigned long long cap, unsigned long long* __restrict__ total) {
  if (cnt == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(total, (unsigned long long)cnt);
  base = __shfl(base, 0);
  for (uint32_t k = lane; k < cnt; k += 64u)
    if (base + k < cap) rec[base + k] = buf[k];
  cnt = 0;
}

// one candidate pair that passed the distance test: is it this chunk's to report, and what is the slot's id?
__device__ __forceinline__ bool join_mine(const JoinPlan& P, int j, uint32_t x0, uint32_t x1, uint64_t qq,
                                          const uint32_t* __restrict__ hay_id, size_t at, uint32_t keep0, uint32_t* id) {
  const uint64_t x = ((uint64_t)x1 << 32) | x0;
  bool mine = qq != 0;  // null needles never match
  for (int e = 0; e < j; ++e) mine = mine && chunk_of(x, P.lo[e], P.lo[e + 1]) != 0u;  // an earlier chunk's pair
  if (mine) {
    *id = hay_id[at];
    mine = *id != 0 || keep0;  // removed slots only where the caller asked for them
  }
  return mine;
}

// `hit` lanes park {needle, dist, id}; wave-uniform count, flushed with one atomic
__device__ __forceinline__ void join_push(uint64_t* __restrict__ buf, uint32_t& cnt, bool hit, uint32_t qidx, uint32_t d,
                                          uint32_t id, cbh_record* __restrict__ rec, unsigned long long cap,
                                          unsigned long long* __restrict__ total) {
  const unsigned long long bm = __builtin_amdgcn_ballot_w64(hit);
  if (bm == 0) return;
  if (hit) {
    const uint32_t k = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
    buf[k] = ((cbh_record)qidx << 39) | ((cbh_record)d << 32) | id;
  }
  cnt += (uint32_t)__popcll(bm);
  if (cnt >= kOutCap) join_flush(buf, cnt, rec, cap, total);
}

// chunk j's join, wide buckets (chunks of <= 11 bits: hundreds to thousands of slots and needles per value): blockIdx.x = a
// job = (value, tile of 2 x 256 of its slots, block of kJQ of its needles).  A lane holds two slots, the needles stream
// through the scalar cache eight at a time; one test per eight needles and two slots, the exact look only behind it.
__global__ __launch_bounds__(kJT) void k_join_pairs(int j, JoinPlan P, uint32_t n, uint32_t nq,
                                                    const uint64_t* __restrict__ hay_x, const uint32_t* __restrict__ hay_id,
                                                    const uint64_t* __restrict__ q_x, const uint32_t* __restrict__ q_idx,
                                                    const uint32_t* __restrict__ start_h, const uint32_t* __restrict__ start_q,
                                                    const uint32_t* __restrict__ jobstart, uint32_t thresh,
                                                    cbh_record* __restrict__ rec, unsigned long long cap,
                                                    unsigned long long* __restrict__ total, uint32_t keep0) {
  __shared__ uint64_t s_out[kJT / 64][kOutCap + 64];
  const uint32_t off = P.voff[j] + (uint32_t)j, nv = 1u << (P.lo[j + 1] - P.lo[j]);
  // the value whose jobs hold this one: the last v with jobstart[v] <= job (uniform: scalar loads)
  const uint32_t job = blockIdx.x;
  uint32_t lo = 0, hi = nv;  // jobstart[lo] <= job < jobstart[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (jobstart[off + mid] <= job) lo = mid; else hi = mid;
  }
  const uint32_t v = lo, rel = job - jobstart[off + v];
  const uint32_t hs = start_h[off + v], he = start_h[off + v + 1], qs = start_q[off + v], qe = start_q[off + v + 1];
  const uint32_t ht = (he - hs + kJH - 1u) / kJH;  // slot tiles of the value
  const uint32_t it = rel % ht, iq = rel / ht;
  const uint32_t i0 = hs + it * kJH + threadIdx.x, i1 = i0 + (uint32_t)kJT;
  const bool live0 = i0 < he, live1 = i1 < he;
  const uint64_t* __restrict__ hxp = hay_x + (size_t)j * n;
  const uint64_t a0 = hxp[live0 ? i0 : hs], a1 = hxp[live1 ? i1 : hs];
  const uint32_t a0l = (uint32_t)a0, a0h = (uint32_t)(a0 >> 32), a1l = (uint32_t)a1, a1h = (uint32_t)(a1 >> 32);
  const uint32_t q0 = qs + iq * kJQ, q1 = min(qe, q0 + kJQ);
  uint64_t* buf = s_out[threadIdx.x >> 6];
  uint32_t cnt = 0;  // (wave-uniform)
  const uint64_t* __restrict__ qx = q_x + (size_t)j * nq;
  const uint32_t* __restrict__ qix = q_idx + (size_t)j * nq;
  auto exact = [&](uint32_t qi) {  // one needle against the lane's two slots
    const uint64_t qq = qx[qi];
    const uint32_t ql = (uint32_t)qq, qh = (uint32_t)(qq >> 32);
    {
      const uint32_t x0 = a0l ^ ql, x1 = a0h ^ qh, d = __popc(x0) + __popc(x1);
      uint32_t id = 0;
      const bool hit = live0 && d < thresh && join_mine(P, j, x0, x1, qq, hay_id, (size_t)j * n + i0, keep0, &id);
      join_push(buf, cnt, hit, hit ? qix[qi] : 0u, d, id, rec, cap, total);
    }
    {
      const uint32_t x0 = a1l ^ ql, x1 = a1h ^ qh, d = __popc(x0) + __popc(x1);
      uint32_t id = 0;
      const bool hit = live1 && d < thresh && join_mine(P, j, x0, x1, qq, hay_id, (size_t)j * n + i1, keep0, &id);
      join_push(buf, cnt, hit, hit ? qix[qi] : 0u, d, id, rec, cap, total);
    }
  };
  uint32_t qi = q0;
  // (a 64-bit pointer bump keeps the eight loads of a block at constant offsets from one SGPR base -- wave-uniform, so they
  // are scalar loads; the next block is fetched while this one is compared: hamm64_scan.hip's loop)
  const uint2* __restrict__ qp = reinterpret_cast<const uint2*>(qx) + q0;
  uint2 cur[8];
  if (qi + 8u <= q1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = qp[k];
  }
  for (; qi + 8u <= q1; qi += 8u) {
    uint2 nxt[8];
    qp += 8;
    if (qi + 16u <= q1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) nxt[k] = qp[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) nxt[k] = make_uint2(0u, 0u);
    }
    uint32_t m0 = 64u, m1 = 64u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      m0 = min(m0, (uint32_t)__popc(a0l ^ cur[k].x) + (uint32_t)__popc(a0h ^ cur[k].y));
      m1 = min(m1, (uint32_t)__popc(a1l ^ cur[k].x) + (uint32_t)__popc(a1h ^ cur[k].y));
    }
    if (__builtin_amdgcn_ballot_w64((live0 && m0 < thresh) || (live1 && m1 < thresh)) != 0) {  // (one block in 10^2 .. 10^4)
#pragma unroll 1
      for (uint32_t k = 0; k < 8u; ++k) exact(qi + k);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
  }
#pragma unroll 1
  for (; qi < q1; ++qi) exact(qi);
  join_flush(buf, cnt, rec, cap, total);
}

// chunk j's join, narrow buckets (chunks of >= 12 bits: a handful of slots and needles per value): a lane takes one slot of
// the chunk's order and walks ITS value's needles (the lanes of a wave walk different lists: per-lane loads, a uniform loop
// until the longest is done) -- 10^8 .. 10^9 pairs in all, where a workgroup per value would be a million launches of
// fifteen lanes.
__global__ __launch_bounds__(kJT) void k_join_narrow(int j, JoinPlan P, uint32_t n, uint32_t nq,
                                                     const uint64_t* __restrict__ hay_x, const uint32_t* __restrict__ hay_id,
                                                     const uint64_t* __restrict__ q_x, const uint32_t* __restrict__ q_idx,
                                                     const uint32_t* __restrict__ start_q, uint32_t thresh,
                                                     cbh_record* __restrict__ rec, unsigned long long cap,
                                                     unsigned long long* __restrict__ total, uint32_t keep0) {
  __shared__ uint64_t s_out[kJT / 64][kOutCap + 64];
  const uint32_t off = P.voff[j] + (uint32_t)j;
  const uint32_t i = blockIdx.x * (uint32_t)kJT + threadIdx.x;
  const bool live = i < n;
  const uint64_t a = hay_x[(size_t)j * n + (live ? i : 0u)];
  const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
  const uint32_t v = chunk_of(a, P.lo[j], P.lo[j + 1]);
  uint32_t qi = live ? start_q[off + v] : 0u;
  const uint32_t qe = live ? start_q[off + v + 1] : 0u;
  uint64_t* buf = s_out[threadIdx.x >> 6];
  uint32_t cnt = 0;
  const uint64_t* __restrict__ qx = q_x + (size_t)j * nq;
  while (__builtin_amdgcn_ballot_w64(qi < qe) != 0) {
    const bool act = qi < qe;
    const uint64_t qq = act ? qx[qi] : 0ull;
    const uint32_t x0 = al ^ (uint32_t)qq, x1 = ah ^ (uint32_t)(qq >> 32), d = __popc(x0) + __popc(x1);
    bool hit = act && d < thresh;
    if (__builtin_amdgcn_ballot_w64(hit) != 0) {
      uint32_t id = 0;
      hit = hit && join_mine(P, j, x0, x1, qq, hay_id, (size_t)j * n + i, keep0, &id);
      join_push(buf, cnt, hit, hit ? q_idx[(size_t)j * nq + qi] : 0u, d, id, rec, cap, total);
    }
    ++qi;
  }
  join_flush(buf, cnt, rec, cap, total);
}

// four chunks of 16 bits (thresholds <= 4): the needles need no order at all -- a lane takes a NEEDLE
// and, chunk after chunk, walks the slots that share its value (the slots' side alone is histogrammed, scanned and
// scattered: half the bookkeeping of a call whose join proper is a tenth of a millisecond)
__global__ __launch_bounds__(kJT) void k_join_by_needle(JoinPlan P, uint32_t n, uint32_t nq, const uint64_t* __restrict__ hay_x,
                                                        const uint32_t* __restrict__ hay_id, const uint64_t* __restrict__ q,
                                                        const uint32_t* __restrict__ start_h, uint32_t thresh,
                                                        cbh_record* __restrict__ rec, unsigned long long cap,
                                                        unsigned long long* __restrict__ total, uint32_t keep0) {
  __shared__ uint64_t s_out[kJT / 64][kOutCap + 64];
  const uint32_t i = blockIdx.x * (uint32_t)kJT + threadIdx.x;
  const uint64_t qq = i < nq ? q[i] : 0ull;
  const bool live = qq != 0;  // (null needles never match)
  const uint32_t ql = (uint32_t)qq, qh = (uint32_t)(qq >> 32);
  uint64_t* buf = s_out[threadIdx.x >> 6];
  uint32_t cnt = 0;
#pragma unroll 1
  for (int j = 0; j < P.m; ++j) {
    const uint32_t off = P.voff[j] + (uint32_t)j + chunk_of(qq, P.lo[j], P.lo[j + 1]);
    uint32_t hi_ = live ? start_h[off] : 0u;
    const uint32_t he = live ? start_h[off + 1] : 0u;
    const uint64_t* __restrict__ hx = hay_x + (size_t)j * n;
    while (__builtin_amdgcn_ballot_w64(hi_ < he) != 0) {
      const bool act = hi_ < he;
      const uint64_t a = act ? hx[hi_] : 0ull;
      const uint32_t x0 = (uint32_t)a ^ ql, x1 = (uint32_t)(a >> 32) ^ qh, d = __popc(x0) + __popc(x1);
      bool hit = act && d < thresh;
      if (__builtin_amdgcn_ballot_w64(hit) != 0) {
        uint32_t id = 0;
        hit = hit && join_mine(P, j, x0, x1, qq, hay_id, (size_t)j * n + hi_, keep0, &id);
        join_push(buf, cnt, hit, i, d, id, rec, cap, total);
      }
      ++hi_;
    }
  }
  join_flush(buf, cnt, rec, cap, total);
}

constexpr int g_join_model_ps_e3 = 250;  // the launcher's cost model: 0.25 ns of ONE SIMD lane... i.e. 2.5e-13 s of the
                                         // machine per candidate pair (measured 2.8e-13 at threshold 8)

std::atomic<long long> g_n_join{0};

}  // namespace

long long get_scan_joins() { return g_n_join.load(); }

bool scan_join_possible(size_t n, size_t nq, int thresh, unsigned flags, const uint64_t* d_qmask) {
  (void)flags;
  return thresh >= 1 && thresh <= kMaxChunks && d_qmask == nullptr && n >= 1 && nq >= 1 && n < 0xfffffff0ull &&
         nq <= CBH_MAX_QUERIES_PER_CALL;
}

// CBH_OK: done (records appended, *d_total advanced like the scans do); CBH_E_UNSUPPORTED: the caller's scan is cheaper
// (or `force` is false and the call is too small to be worth the bookkeeping) -- nothing has been written.
int launch_hamm64_join(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n, const uint64_t* d_q, size_t nq,
                       int thresh, cbh_record* d_rec, size_t cap, unsigned long long* d_total, hipStream_t stream,
                       unsigned flags, bool force, double scan_ms_estimate) {
  JoinPlan P;
  memset(&P, 0, sizeof P);
  P.m = std::max(4, thresh);
  uint32_t nvals = 0;
  for (int j = 0; j <= P.m; ++j) P.lo[j] = (64 * j + P.m / 2) / P.m;
  for (int j = 0; j < P.m; ++j) {
    P.voff[j] = nvals;
    nvals += 1u << (P.lo[j + 1] - P.lo[j]);
  }
  P.voff[P.m] = nvals;
  const size_t nslots = (size_t)nvals + (size_t)P.m;  // one extra entry per chunk
  Scratch scratch(stream);
  uint32_t *hist_h = nullptr, *hist_q = nullptr, *start_h = nullptr, *start_q = nullptr, *jobstart = nullptr;
  unsigned long long* stats = nullptr;
  CBH_HIP(scratch.get(&hist_h, nslots * 4));
  CBH_HIP(scratch.get(&hist_q, nslots * 4));
  CBH_HIP(scratch.get(&start_h, nslots * 4));
  CBH_HIP(scratch.get(&start_q, nslots * 4));
  CBH_HIP(scratch.get(&jobstart, nslots * 4));
  CBH_HIP(scratch.get(&stats, 2 * kMaxChunks * 8));
  unsigned long long h_stats[2 * kMaxChunks];
  auto count = [&](uint32_t sh, uint32_t sq, bool pairs_only) -> int {  // histograms of every sh-th slot / sq-th needle,
                                                                        // scans (or just the pair count), read-back
    CBH_HIP(hipMemsetAsync(hist_h, 0, nslots * 4, stream));
    CBH_HIP(hipMemsetAsync(hist_q, 0, nslots * 4, stream));
    const size_t nh_ = (n + sh - 1) / sh, nq_ = (nq + sq - 1) / sq;
    hipLaunchKernelGGL(k_join_hist, dim3((unsigned)((nh_ + 255) / 256)), dim3(256), 0, stream, d_hashes, (uint32_t)n, sh, P,
                       hist_h);
    hipLaunchKernelGGL(k_join_hist, dim3((unsigned)((nq_ + 255) / 256)), dim3(256), 0, stream, d_q, (uint32_t)nq, sq, P, hist_q);
    if (pairs_only)
      hipLaunchKernelGGL(k_join_pairs_only, dim3((unsigned)P.m), dim3(1024), 0, stream, P, hist_h, hist_q, stats);
    else
      hipLaunchKernelGGL(k_join_scan, dim3((unsigned)P.m), dim3(1024), 0, stream, P, hist_h, hist_q, start_h, start_q,
                         jobstart, stats);
    CBH_HIP(hipGetLastError());
    CBH_HIP(hipMemcpyAsync(h_stats, stats, (size_t)2 * P.m * 8, hipMemcpyDeviceToHost, stream));
    CBH_HIP(hipStreamSynchronize(stream));
    return CBH_OK;
  };
  int rc;
  // (16-bit chunks only: ~15 slots a value.  At five chunks of 12-13 bits a needle walks 5 x 122 slots by per-lane loads: 5.8 ms
  // against the 2.9 of sorting both sides)
  const bool by_needle = P.m == 4;
  if (!force) {
    // a library of near-identical hashes makes the full histogram itself expensive (10^6 atomics on one counter): look at
    // 16 384 of each side first and leave if THEIR candidate pairs, scaled up, already say the scan is cheaper
    const uint32_t sh = (uint32_t)std::max<size_t>(1, n / 16384), sq = (uint32_t)std::max<size_t>(1, nq / 16384);
    if (sh > 1 || sq > 1 || by_needle) {
      if ((rc = count(sh, sq, true))) return rc;
      double sp = 0;
      for (int j = 0; j < P.m; ++j) sp += (double)h_stats[2 * j + 1];
      // (sampling thins the occupied values' pairs by sh x sq on average; a generous factor keeps borderline calls in --
      // the by-needle form has no exact count behind this one: there the estimate decides, with less slack)
      const double est_ms = sp * (double)sh * (double)sq * (double)g_join_model_ps_e3 * 1e-12;
      if (est_ms > (by_needle ? 0.6 : 4.0) * scan_ms_estimate) return CBH_E_UNSUPPORTED;
    }
  }
  if (by_needle) {
    uint64_t* hx = nullptr;
    uint32_t* hid = nullptr;
    CBH_HIP(scratch.get(&hx, (size_t)P.m * n * 8));
    CBH_HIP(scratch.get(&hid, (size_t)P.m * n * 4));
    CBH_HIP(hipMemsetAsync(hist_h, 0, nslots * 4, stream));
    CBH_HIP(hipMemsetAsync(hist_q, 0, nslots * 4, stream));  // (no needles' side: the scans see empty needle buckets)
    hipLaunchKernelGGL(k_join_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_hashes, (uint32_t)n, 1u, P,
                       hist_h);
    hipLaunchKernelGGL(k_join_scan, dim3((unsigned)P.m), dim3(1024), 0, stream, P, hist_h, hist_q, start_h, start_q, jobstart,
                       stats);
    CBH_HIP(hipMemsetAsync(hist_h, 0, nslots * 4, stream));
    hipLaunchKernelGGL(k_join_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_hashes, d_ids, (uint32_t)n, P,
                       start_h, hist_h, hx, hid);
    hipLaunchKernelGGL(k_join_by_needle, dim3((unsigned)((nq + kJT - 1) / kJT)), dim3(kJT), 0, stream, P, (uint32_t)n,
                       (uint32_t)nq, hx, hid, d_q, start_h, (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total,
                       (uint32_t)(flags & 1u));
    CBH_HIP(hipGetLastError());
    g_n_join++;
    return CBH_OK;
  }
  if ((rc = count(1, 1, false))) return rc;
  double pairs = 0;
  unsigned long long jobs_max = 0;
  for (int j = 0; j < P.m; ++j) {
    pairs += (double)h_stats[2 * j + 1];
    jobs_max = std::max(jobs_max, h_stats[2 * j]);
  }
  if (jobs_max > 0x7fffffffull) return CBH_E_UNSUPPORTED;
  if (!force) {
    const double join_ms = pairs * (double)g_join_model_ps_e3 * 1e-12 + 0.1 * P.m + 0.2;
    if (join_ms > 0.9 * scan_ms_estimate) return CBH_E_UNSUPPORTED;
  }
  uint64_t *hx = nullptr, *qx = nullptr;
  uint32_t *hid = nullptr, *qidx = nullptr;
  CBH_HIP(scratch.get(&hx, (size_t)P.m * n * 8));
  CBH_HIP(scratch.get(&hid, (size_t)P.m * n * 4));
  CBH_HIP(scratch.get(&qx, (size_t)P.m * nq * 8));
  CBH_HIP(scratch.get(&qidx, (size_t)P.m * nq * 4));
  // (the histograms have served the scans: they become the scatters' cursors)
  CBH_HIP(hipMemsetAsync(hist_h, 0, nslots * 4, stream));
  CBH_HIP(hipMemsetAsync(hist_q, 0, nslots * 4, stream));
  hipLaunchKernelGGL(k_join_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_hashes, d_ids, (uint32_t)n, P,
                     start_h, hist_h, hx, hid);
  hipLaunchKernelGGL(k_join_scatter, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, d_q,
                     (const uint32_t*)nullptr, (uint32_t)nq, P, start_q, hist_q, qx, qidx);
  for (int j = 0; j < P.m; ++j) {
    if (h_stats[2 * j] == 0) continue;
    if (P.lo[j + 1] - P.lo[j] >= 12)
      hipLaunchKernelGGL(k_join_narrow, dim3((unsigned)((n + kJT - 1) / kJT)), dim3(kJT), 0, stream, j, P, (uint32_t)n,
                         (uint32_t)nq, hx, hid, qx, qidx, start_q, (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total,
                         (uint32_t)(flags & 1u));
    else
      hipLaunchKernelGGL(k_join_pairs, dim3((unsigned)h_stats[2 * j]), dim3(kJT), 0, stream, j, P, (uint32_t)n, (uint32_t)nq,
                         hx, hid, qx, qidx, start_h, start_q, jobstart, (uint32_t)thresh, d_rec, (unsigned long long)cap,
                         d_total, (uint32_t)(flags & 1u));
  }
  CBH_HIP(hipGetLastError());
  g_n_join++;
  return CBH_OK;
}

}  // namespace cbh
