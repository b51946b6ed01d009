// fdct.hip -- DctFeaturesIndex::find on top of the 64-bit scan; top-10 cut (topk.hip) and vote reduction (reduce.hip)
// on the device.
//
// Reference: src/dctfeaturesindex.cpp:260-358.  The index is a flat multiset of
// (mediaId, keypoint hash) entries (HammingTree values, src/tree/hammingtree.h:66-74): here a
// cbh_idx64 whose ids repeat.  For every needle hash the reference takes the tree's candidates
// under dctThresh sorted by distance and keeps the first 10 (:301-303) -- removed entries
// (index 0) still occupy places in that cut and are skipped only afterwards (:308) -- then votes
// per mediaId (:314-323) and scores (:334-355).
// The reference tree is approximate (it only descends the needle's own branch,
// hammingtree.h:248-252); the scan is exact, so its candidate set is a superset and equals the
// tree's while the tree is a single leaf (<= 8192 entries).
#include <algorithm>
#include <map>
#include <unordered_set>

#include "cbh_index.h"

namespace cbh {
// tuning knobs "fdct_host_vote" / "video_host_reduce": where the per-needle reduction runs.  0 (default) = on the device
// for batches, on the host for a single needle (its candidates are a few KB; the device route costs more launches
// and synchronisations than it saves: 0.34 vs 0.18 ms per video needle); 1 = always host (the round-1 path);
// 2 = always device.
int g_fdct_host_vote = 0;
int g_video_host_reduce = 0;
}  // namespace cbh

namespace {

struct Needle {
  size_t begin, end;  // range of needle hashes
  uint32_t id;
};

// votes + scores for one needle from its per-hash top-10 table
void vote(const cbh_match* top, const uint32_t* counts, const Needle& nd, int k,
          std::vector<cbh_match>* out) {
  std::map<uint32_t, uint32_t> matches;  // QMap: ascending mediaId
  std::map<uint32_t, int> scores;
  uint32_t maxMatches = 0;
  for (size_t j = nd.begin; j < nd.end; ++j) {
    const uint32_t len = std::min<uint32_t>((uint32_t)k, counts[j]);
    for (uint32_t t = 0; t < len; ++t) {
      const cbh_match& m = top[j * (size_t)k + t];
      if (m.id == 0) continue;  // "zero index means deleted" (:308); ids are unsigned here
      uint32_t& c = matches[m.id];
      c += 1;
      scores[m.id] += m.score;
      if (nd.id != m.id) maxMatches = std::max(c, maxMatches);
    }
  }
  for (auto& kv : matches) {
    cbh_match r;
    r.id = kv.first;
    const float avgScore = (float)scores[kv.first] / (float)kv.second;
    if (kv.first == nd.id)
      r.score = -1;
    else if (maxMatches == 1)
      r.score = (int32_t)(10 * avgScore);
    else
      r.score = (int32_t)(maxMatches - kv.second);
    out->push_back(r);
  }
}

// ---- HammingTree-compatible candidate sets -----------------------------------------------------------
// The reference tree (src/tree/hammingtree.h) is a binary trie on hash bits 0,1,2,...: a node splits on
// bit = depth (getBit, :243) as soon as more than CLUSTER_SIZE / 8 = 8192 values have been routed to it
// (:384-414) and never merges again (remove() only zeroes indices, :347-364).  Routed counts only grow, so
// the final shape does not depend on the insertion order: node (depth d, prefix p) is internal iff more
// than 8192 stored hashes have low d bits == p.  search() (:244-252) descends the needle's own bits to
// ONE leaf and scans only that leaf, i.e. it returns the entries under the threshold that share the
// needle's low `depth(leaf)` bits.  tree_masks() reproduces exactly that as an equal-bits mask per needle
// hash, which the scan kernels apply to their (rare) hits.
constexpr size_t kLeafCap = 64 * 1024 / sizeof(uint64_t);

uint64_t bitrev64(uint64_t x) {
  x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
  return __builtin_bswap64(x);
}

void split_node(const std::vector<uint64_t>& rev, size_t lo, size_t hi, int depth, uint64_t prefix,
                std::unordered_set<uint64_t>* internal) {
  if (hi - lo <= kLeafCap || depth >= 58) return;  // (the reference allows depth < 63; 2^58 * 8192 entries do not exist)
  internal->insert(((uint64_t)depth << 58) | prefix);
  // in bit-reversed order the values whose bit `depth` is 0 come first
  const uint64_t bit = 1ull << (63 - depth);
  const uint64_t base = rev[lo] & ~((bit << 1) - 1);  // the bits above `bit` are the (reversed) prefix
  const size_t mid = std::lower_bound(rev.begin() + (long)lo, rev.begin() + (long)hi, base | bit) - rev.begin();
  split_node(rev, lo, mid, depth + 1, prefix, internal);
  split_node(rev, mid, hi, depth + 1, prefix | (1ull << depth), internal);
}

int ensure_tree(cbh_idx64* idx) {
  std::lock_guard<std::mutex> lk(idx->tree_mu);
  if (idx->tree_valid) return CBH_OK;
  std::vector<uint64_t> h(idx->n);
  std::vector<uint32_t> ids(idx->n);
  int rc = idx->n ? cbh_idx64_download(idx, h.data(), ids.data(), idx->n) : CBH_OK;
  if (rc) return rc;
  for (auto& x : h) x = bitrev64(x);
  std::sort(h.begin(), h.end());
  idx->tree_internal.clear();
  split_node(h, 0, h.size(), 0, 0, &idx->tree_internal);
  idx->tree_valid = true;
  return CBH_OK;
}

int fdct_core(cbh_idx64* idx, const uint64_t* hashes, const std::vector<Needle>& needles, size_t nq,
              int thresh, std::vector<std::vector<cbh_match>>* results, int tree_compat = 0) {
  const int k = 10;
  results->assign(needles.size(), {});
  if (nq == 0 || idx->n == 0 || thresh <= 0) return CBH_OK;
  if (nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  Workspace* ws = L.ws;
  if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, nq))) return rc;
  if ((rc = Workspace::grow(&ws->d_out, &ws->out_cap, nq * (size_t)k))) return rc;
  if ((rc = Workspace::grow(&ws->d_counts, &ws->counts_cap, nq))) return rc;
  CBH_HIP(hipMemcpyAsync(ws->d_q, hashes, nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  std::vector<uint64_t> masks;
  if (tree_compat) {
    masks.resize(nq);
    if ((rc = cbh_idx64_tree_masks(idx, hashes, nq, masks.data()))) return rc;
    if ((rc = Workspace::grow(&ws->d_qmask, &ws->qmask_cap, nq))) return rc;
    CBH_HIP(hipMemcpyAsync(ws->d_qmask, masks.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  }
  unsigned long long total = 0;
  rc = scan_all(idx, ws, ws->d_q, nq, thresh, ws->stream, &total, SCAN_KEEP_ID0,
                tree_compat ? ws->d_qmask : nullptr);
  if (rc) return rc;
  hipStream_t s = ws->stream;
  // per needle hash: the first 10 candidates by (distance, id) -- K4 counting select on the workspace block
  {
    void* scratch = nullptr;
    const size_t ncap = std::min<size_t>(ws->rec_cap, (size_t)total + 1);
    CBH_HIP(cbh::malloc_async(&scratch, topk_scratch_bytes(nq, ncap) + 16, s));
    unsigned* d_status = (unsigned*)((char*)scratch + topk_scratch_bytes(nq, ncap));
    rc = topk_scratch_init(scratch, nq, s);
    if (!rc) rc = launch_records_topk(ws->d_total, 1, 0, ncap, nq, k, ws->d_out, ws->d_counts, d_status, scratch, s);
    (void)cbh::free_async(scratch, s);
    if (rc) return rc;
  }
  if (g_fdct_host_vote == 1 || (g_fdct_host_vote == 0 && needles.size() == 1)) {  // host reduction
    std::vector<cbh_match> top(nq * (size_t)k);
    std::vector<uint32_t> counts(nq);
    CBH_HIP(hipMemcpyAsync(top.data(), ws->d_out, top.size() * sizeof(cbh_match), hipMemcpyDeviceToHost, s));
    CBH_HIP(hipMemcpyAsync(counts.data(), ws->d_counts, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CBH_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < needles.size(); ++i) vote(top.data(), counts.data(), needles[i], k, &(*results)[i]);
    return CBH_OK;
  }
  // K5 on the device (reduce.hip): votes per (needle image, media), maxMatches, score rule; only results come back
  std::vector<uint32_t> qneedle(nq), nid(needles.size());
  for (size_t i = 0; i < needles.size(); ++i) {
    nid[i] = needles[i].id;
    for (size_t j = needles[i].begin; j < needles[i].end; ++j) qneedle[j] = (uint32_t)i;
  }
  uint32_t *d_qneedle = nullptr, *d_nid = nullptr;
  hipError_t e = cbh::malloc_async((void**)&d_qneedle, nq * 4, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_nid, std::max<size_t>(1, nid.size()) * 4, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_qneedle, qneedle.data(), nq * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !nid.empty()) e = hipMemcpyAsync(d_nid, nid.data(), nid.size() * 4, hipMemcpyHostToDevice, s);
  std::vector<cbh_nmatch> flat;
  if (e == hipSuccess) rc = launch_fdct_vote(ws->d_out, ws->d_counts, d_qneedle, nq, k, d_nid, nid.size(), &flat, s);
  if (d_qneedle) (void)cbh::free_async(d_qneedle, s);
  if (d_nid) (void)cbh::free_async(d_nid, s);
  CBH_HIP(e);
  if (rc) return rc;
  std::sort(flat.begin(), flat.end(), [](const cbh_nmatch& a, const cbh_nmatch& b) {
    return a.needle != b.needle ? a.needle < b.needle : a.id < b.id;  // QMap order: ascending mediaId
  });
  for (const cbh_nmatch& m : flat) (*results)[m.needle].push_back(cbh_match{m.id, m.score});
  return CBH_OK;
}

}  // namespace

extern "C" {

int cbh_idx64_tree_masks(cbh_idx64* idx, const uint64_t* q, size_t nq, uint64_t* out_masks) {
  if (!idx || (nq && (!q || !out_masks))) return CBH_E_INVAL;
  int rc = ensure_tree(idx);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(idx->tree_mu);
  for (size_t i = 0; i < nq; ++i) {
    int d = 0;
    while (d < 58 && idx->tree_internal.count(((uint64_t)d << 58) | (q[i] & ((1ull << d) - 1)))) ++d;
    out_masks[i] = (1ull << d) - 1;  // the leaf holds the entries sharing the needle's low d bits
  }
  return CBH_OK;
}

int cbh_fdct_find(cbh_idx64* idx, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                  cbh_match* out, size_t cap, size_t* n_out) {
  return cbh_fdct_find_ex(idx, hashes, n, needle_id, thresh, 0, out, cap, n_out);
}

int cbh_fdct_find_ex(cbh_idx64* idx, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                     int tree_compat, cbh_match* out, size_t cap, size_t* n_out) {
  if (!idx || !n_out || (cap && !out) || (n && !hashes)) return CBH_E_INVAL;
  *n_out = 0;
  std::vector<Needle> nd{{0, n, needle_id}};
  std::vector<std::vector<cbh_match>> res;
  int rc = fdct_core(idx, hashes, nd, n, thresh, &res, tree_compat);
  if (rc) return rc;
  *n_out = res[0].size();
  for (size_t i = 0; i < res[0].size() && i < cap; ++i) out[i] = res[0][i];
  return CBH_OK;
}

int cbh_fdct_find_batch(cbh_idx64* idx, const uint64_t* hashes, const uint64_t* offsets,
                        const uint32_t* needle_ids, size_t n_needles, int thresh, cbh_match* out,
                        size_t cap, uint64_t* out_offsets) {
  return cbh_fdct_find_batch_ex(idx, hashes, offsets, needle_ids, n_needles, thresh, 0, out, cap, out_offsets);
}

int cbh_fdct_find_batch_ex(cbh_idx64* idx, const uint64_t* hashes, const uint64_t* offsets,
                           const uint32_t* needle_ids, size_t n_needles, int thresh, int tree_compat,
                           cbh_match* out, size_t cap, uint64_t* out_offsets) {
  if (!idx || !offsets || !needle_ids || !out_offsets || (cap && !out)) return CBH_E_INVAL;
  std::vector<Needle> nd(n_needles);
  for (size_t i = 0; i < n_needles; ++i) {
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
    nd[i] = Needle{(size_t)offsets[i], (size_t)offsets[i + 1], needle_ids[i]};
  }
  const size_t nq = n_needles ? (size_t)offsets[n_needles] : 0;
  if (nq && !hashes) return CBH_E_INVAL;
  std::vector<std::vector<cbh_match>> res;
  int rc = fdct_core(idx, hashes, nd, nq, thresh, &res, tree_compat);
  if (rc) return rc;
  uint64_t pos = 0;
  for (size_t i = 0; i < n_needles; ++i) {
    out_offsets[i] = pos;
    for (auto& m : res[i]) {
      if (pos < cap) out[pos] = m;
      ++pos;
    }
  }
  out_offsets[n_needles] = pos;
  return pos > cap ? CBH_E_OVERFLOW : CBH_OK;
}

// HammingTree::findIndex (hammingtree.h:110-112): the hashes stored for one mediaId, used by
// DctFeaturesIndex::find when the needle carries no hashes (dctfeaturesindex.cpp:270-276)
int cbh_idx64_hashes_for_id(const cbh_idx64* idx, uint32_t id, uint64_t* out, size_t cap,
                            size_t* n_out) {
  if (!idx || !n_out) return CBH_E_INVAL;
  std::vector<uint64_t> h(idx->n);
  std::vector<uint32_t> ids(idx->n);
  int rc = cbh_idx64_download(idx, h.data(), ids.data(), idx->n);
  if (rc) return rc;
  size_t m = 0;
  for (size_t i = 0; i < idx->n; ++i)
    if (ids[i] == id) {
      if (out && m < cap) out[m] = h[i];
      ++m;
    }
  *n_out = m;
  return CBH_OK;
}

}  // extern "C"
