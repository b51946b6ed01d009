// reduce.hip -- K5 and K8: the per-needle reductions of DctFeaturesIndex::find and DctVideoIndex::findVideo on the
// device, so that only final matches cross PCIe (round 1 copied every candidate record back and reduced them in
// std::map loops on the host).
//
// K5  k_fdct_pairs / k_fdct_runs / k_fdct_score      src/dctfeaturesindex.cpp:291-358
//     input: per needle hash its first <= 10 candidates (the counting select of topk.hip, removed entries still
//     occupying places, :301-308).  Every surviving candidate becomes ONE 64-bit key
//         needle image (25) | mediaId (32) | distance (7)
//     an ascending sort groups the votes of a (needle, media) pair into a run; the run head counts the votes and sums
//     the distances (:314-323), atomicMax gives the needle's maxMatches over media != needle (:320), and the score
//     rule (:334-355) runs per run: needle itself -1, maxMatches == 1 -> int(10 * avg distance), else maxMatches - votes.
//
// K8  k_video_winners / k_video_score                 src/dctvideoindex.cpp:475-509, 595-654
//     input: unordered scan records (needle frame, distance, entry position) grouped per needle frame by the
//     count/scan/scatter of topk.hip.  A record WINS when no other record of its needle frame points into the same
//     video with a smaller (distance, position): the closest frame per video (:499-502; position order = the
//     reference's scan order among equals), minus the needle's own video when filterSelf (:494).  Winners become
//     (key = video (24) | needle frame (25), value = entry position); sorted, a run of equal (video, needle) is the
//     candidate list of that pair in source-frame order, and one lane walks it: numAdjacent with frameMargin 15
//     (:593-613), the vfm / vfn gates (:619-641), score 100 - percentNear, range = first pair .. max(src, dst) extent
//     (:645-651).
// Both end in an atomic cursor: the result list is compact and unordered, the host orders it (it is final matches
// only) -- by (needle, mediaId), the QMap / std::map order of the reference.
#include <cstring>


#include "cbh_internal.h"

namespace cbh {
namespace {

constexpr unsigned long long kPad = ~0ull;

// ---- K5 ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fdct_pairs(const cbh_match* __restrict__ top, const uint32_t* __restrict__ counts,
                                                    const uint32_t* __restrict__ qneedle, unsigned nq, int k,
                                                    unsigned long long* __restrict__ keys) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nq * (size_t)k) return;
  const unsigned j = (unsigned)(t / (unsigned)k), r = (unsigned)(t % (unsigned)k);
  unsigned long long key = kPad;
  if (r < min((unsigned)k, counts[j])) {
    const cbh_match m = top[t];
    if (m.id != 0)  // "zero index means deleted" (:308)
      key = ((unsigned long long)qneedle[j] << 39) | ((unsigned long long)m.id << 7) | (unsigned long long)(m.score & 0x7f);
  }
  keys[t] = key;
}

struct FdctRun {
  uint32_t needle, id, votes, sum;
};

__global__ __launch_bounds__(256) void k_fdct_runs(const unsigned long long* __restrict__ keys, size_t n,
                                                   const uint32_t* __restrict__ needle_id,
                                                   FdctRun* __restrict__ runs, unsigned* __restrict__ n_runs,
                                                   unsigned* __restrict__ maxm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = keys[i];
  if (key == kPad) return;
  const unsigned long long grp = key >> 7;
  if (i > 0 && (keys[i - 1] >> 7) == grp) return;  // not a run head
  unsigned votes = 0, sum = 0;
  for (size_t j = i; j < n && (keys[j] >> 7) == grp; ++j) {
    ++votes;
    sum += (unsigned)(keys[j] & 0x7f);
  }
  FdctRun r;
  r.needle = (uint32_t)(key >> 39);
  r.id = (uint32_t)(grp & 0xffffffffull);
  r.votes = votes;
  r.sum = sum;
  runs[atomicAdd(n_runs, 1u)] = r;
  if (r.id != needle_id[r.needle]) atomicMax(&maxm[r.needle], votes);
}

__global__ __launch_bounds__(256) void k_fdct_score(const FdctRun* __restrict__ runs, const unsigned* __restrict__ n_runs,
                                                    const uint32_t* __restrict__ needle_id,
                                                    const unsigned* __restrict__ maxm, cbh_nmatch* __restrict__ out) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= *n_runs) return;
  const FdctRun r = runs[i];
  const unsigned mm = maxm[r.needle];
  cbh_nmatch o;
  o.needle = r.needle;
  o.id = r.id;
  if (r.id == needle_id[r.needle]) {
    o.score = -1;
  } else if (mm == 1) {
    const float avg = (float)r.sum / (float)r.votes;  // IEEE division, as the host expression
    o.score = (int32_t)(10 * avg);
  } else {
    o.score = (int32_t)(mm - r.votes);
  }
  out[i] = o;
}

// ---- K8 ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_video_winners(const unsigned long long* __restrict__ seg,
                                                       const unsigned* __restrict__ off, unsigned nq,
                                                       const uint32_t* __restrict__ evidx,
                                                       const uint32_t* __restrict__ vmedia,
                                                       const uint32_t* __restrict__ qneedle,
                                                       const uint32_t* __restrict__ needle_id, int filter_self,
                                                       unsigned long long* __restrict__ keys,
                                                       uint32_t* __restrict__ vals, size_t total) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  // the needle frame this element belongs to: last q with off[q] <= g
  unsigned lo = 0, hi = nq;
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (off[mid] <= g) lo = mid; else hi = mid;
  }
  const unsigned qi = lo;
  const unsigned long long mine = seg[g];
  const uint32_t pos = (uint32_t)(mine & 0xffffffffull);
  const uint32_t vi = evidx[pos - 1];
  bool win = !(filter_self && vmedia[vi] == needle_id[qneedle[qi]]);
  if (win) {
    const unsigned a = off[qi], b = off[qi + 1];
    for (unsigned e = a; e < b; ++e) {
      const unsigned long long o = seg[e];
      if (o < mine && evidx[(uint32_t)(o & 0xffffffffull) - 1] == vi) {
        win = false;
        break;
      }
    }
  }
  keys[g] = win ? (((unsigned long long)vi << 25) | (unsigned long long)qi) : kPad;
  vals[g] = pos;
}

__global__ __launch_bounds__(256) void k_video_score(const unsigned long long* __restrict__ keys,
                                                     const uint32_t* __restrict__ vals, size_t total,
                                                     const uint32_t* __restrict__ qneedle,
                                                     const int32_t* __restrict__ qframe,
                                                     const int32_t* __restrict__ eframe,
                                                     const uint32_t* __restrict__ vmedia, int min_matched,
                                                     int min_near, cbh_nvmatch* __restrict__ out,
                                                     unsigned* __restrict__ n_out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const unsigned long long key = keys[i];
  if (key == kPad) return;
  const unsigned long long qmask = (1ull << 25) - 1;
  const uint32_t vi = (uint32_t)(key >> 25);
  const uint32_t k = qneedle[(uint32_t)(key & qmask)];
  if (i > 0) {
    const unsigned long long p = keys[i - 1];
    if ((uint32_t)(p >> 25) == vi && qneedle[(uint32_t)(p & qmask)] == k) return;  // not the head of its group
  }
  const int frameMargin = 15;
  int num = 0, numAdjacent = 0, lastFrame = 0;
  int first_src = 0, first_dst = 0, last_src = 0, last_dst = 0;
  for (size_t j = i; j < total; ++j) {
    const unsigned long long kj = keys[j];
    if (kj == kPad || (uint32_t)(kj >> 25) != vi) break;
    const uint32_t qj = (uint32_t)(kj & qmask);
    if (qneedle[qj] != k) break;
    const int src = qframe[qj], dst = eframe[vals[j] - 1];
    if (abs(dst - lastFrame) < frameMargin) numAdjacent++;
    lastFrame = dst;
    if (num == 0) first_src = src, first_dst = dst;
    last_src = src, last_dst = dst;
    ++num;
  }
  const int percentNear = numAdjacent * 100 / num;
  if (num < min_matched || percentNear < min_near) return;
  cbh_nvmatch o;
  o.needle = k;
  o.m.id = vmedia[vi];
  o.m.score = 100 - percentNear;
  o.m.src_in = first_src;
  o.m.dst_in = first_dst;
  o.m.len = max(last_src - first_src, last_dst - first_dst);
  out[atomicAdd(n_out, 1u)] = o;
}

}  // namespace

// K5.  d_top/d_counts: the per-needle-hash cut (k places each); d_qneedle[nq]: needle image of every needle hash;
// d_needle_id[n_needles].  On return *h_n = number of results, h_out filled (unordered).  Synchronises `s`.
int launch_fdct_vote(const cbh_match* d_top, const uint32_t* d_counts, const uint32_t* d_qneedle, size_t nq, int k,
                     const uint32_t* d_needle_id, size_t n_needles, std::vector<cbh_nmatch>* h_out, hipStream_t s) {
  h_out->clear();
  const size_t n = nq * (size_t)k;
  if (n == 0) return CBH_OK;
  unsigned long long* keys = nullptr;
  FdctRun* runs = nullptr;
  unsigned* misc = nullptr;  // [0] = n_runs, [1..] = maxm[n_needles]
  cbh_nmatch* out = nullptr;
  hipError_t e = cbh::malloc_async((void**)&keys, n * 8, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&runs, n * sizeof(FdctRun), s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&misc, (1 + n_needles) * 4, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&out, n * sizeof(cbh_nmatch), s);
  int rc = CBH_OK;
  unsigned n_runs = 0;
  if (e == hipSuccess) e = hipMemsetAsync(misc, 0, (1 + n_needles) * 4, s);
  if (e == hipSuccess) {
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_fdct_pairs, dim3(g), dim3(256), 0, s, d_top, d_counts, d_qneedle, (unsigned)nq, k, keys);
    rc = sort_keys_u64(keys, n, 64, s);
    if (!rc) {
      hipLaunchKernelGGL(k_fdct_runs, dim3(g), dim3(256), 0, s, keys, n, d_needle_id, runs, misc, misc + 1);
      hipLaunchKernelGGL(k_fdct_score, dim3(g), dim3(256), 0, s, runs, misc, d_needle_id, misc + 1, out);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(&n_runs, misc, 4, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e == hipSuccess && n_runs) {
        h_out->resize(n_runs);
        e = hipMemcpyAsync(h_out->data(), out, (size_t)n_runs * sizeof(cbh_nmatch), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
      }
    }
  }
  for (void* p : {(void*)keys, (void*)runs, (void*)misc, (void*)out})
    if (p) (void)cbh::free_async(p, s);
  if (rc) return rc;
  CBH_HIP(e);
  return CBH_OK;
}

// K8.  d_off/d_seg: records grouped per needle frame (launch_records_group), `total` of them.
int launch_video_reduce(const unsigned* d_off, const unsigned long long* d_seg, size_t total, size_t nq,
                        const uint32_t* d_evidx, const int32_t* d_eframe, const uint32_t* d_vmedia,
                        const uint32_t* d_qneedle, const int32_t* d_qframe, const uint32_t* d_needle_id, int filter_self,
                        int min_matched, int min_near, std::vector<cbh_nvmatch>* h_out, hipStream_t s) {
  h_out->clear();
  if (total == 0 || nq == 0) return CBH_OK;
  unsigned long long* keys = nullptr;
  uint32_t* vals = nullptr;
  cbh_nvmatch* out = nullptr;
  unsigned* n_out = nullptr;
  hipError_t e = cbh::malloc_async((void**)&keys, total * 8, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&vals, total * 4, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&out, total * sizeof(cbh_nvmatch), s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&n_out, 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(n_out, 0, 4, s);
  int rc = CBH_OK;
  unsigned n = 0;
  if (e == hipSuccess) {
    const unsigned g = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_video_winners, dim3(g), dim3(256), 0, s, d_seg, d_off, (unsigned)nq, d_evidx, d_vmedia, d_qneedle,
                       d_needle_id, filter_self, keys, vals, total);
    rc = sort_pairs_u64_u32(keys, vals, total, 64, s);
    if (!rc) {
      hipLaunchKernelGGL(k_video_score, dim3(g), dim3(256), 0, s, keys, vals, total, d_qneedle, d_qframe, d_eframe,
                         d_vmedia, min_matched, min_near, out, n_out);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipMemcpyAsync(&n, n_out, 4, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e == hipSuccess && n) {
        h_out->resize(n);
        e = hipMemcpyAsync(h_out->data(), out, (size_t)n * sizeof(cbh_nvmatch), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
      }
    }
  }
  for (void* p : {(void*)keys, (void*)vals, (void*)out, (void*)n_out})
    if (p) (void)cbh::free_async(p, s);
  if (rc) return rc;
  CBH_HIP(e);
  return CBH_OK;
}

}  // namespace cbh
