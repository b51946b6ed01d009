// fdct.hip -- DctFeaturesIndex::find on top of the 64-bit scan (host-side aggregation).
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
#include <map>

#include "cbh_index.h"

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

int fdct_core(cbh_idx64* idx, const uint64_t* hashes, const std::vector<Needle>& needles, size_t nq,
              int thresh, std::vector<std::vector<cbh_match>>* results) {
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
  unsigned long long total = 0;
  rc = scan_all(idx, ws, ws->d_q, nq, thresh, ws->stream, &total, SCAN_KEEP_ID0);
  if (rc) return rc;
  rc = launch_sort_records(ws->d_rec, ws->d_alt, (size_t)total, nq, ws->d_tmp, ws->tmp_bytes, ws->stream);
  if (rc) return rc;
  rc = launch_select_records(ws->d_rec, (size_t)total, nq, k, ws->d_out, ws->d_counts, ws->stream);
  if (rc) return rc;
  std::vector<cbh_match> top(nq * (size_t)k);
  std::vector<uint32_t> counts(nq);
  CBH_HIP(hipMemcpyAsync(top.data(), ws->d_out, top.size() * sizeof(cbh_match), hipMemcpyDeviceToHost,
                         ws->stream));
  CBH_HIP(hipMemcpyAsync(counts.data(), ws->d_counts, nq * sizeof(uint32_t), hipMemcpyDeviceToHost,
                         ws->stream));
  CBH_HIP(hipStreamSynchronize(ws->stream));
  for (size_t i = 0; i < needles.size(); ++i) vote(top.data(), counts.data(), needles[i], k, &(*results)[i]);
  return CBH_OK;
}

}  // namespace

extern "C" {

int cbh_fdct_find(cbh_idx64* idx, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                  cbh_match* out, size_t cap, size_t* n_out) {
  if (!idx || !n_out || (cap && !out) || (n && !hashes)) return CBH_E_INVAL;
  *n_out = 0;
  std::vector<Needle> nd{{0, n, needle_id}};
  std::vector<std::vector<cbh_match>> res;
  int rc = fdct_core(idx, hashes, nd, n, thresh, &res);
  if (rc) return rc;
  *n_out = res[0].size();
  for (size_t i = 0; i < res[0].size() && i < cap; ++i) out[i] = res[0][i];
  return CBH_OK;
}

int cbh_fdct_find_batch(cbh_idx64* idx, const uint64_t* hashes, const uint64_t* offsets,
                        const uint32_t* needle_ids, size_t n_needles, int thresh, cbh_match* out,
                        size_t cap, uint64_t* out_offsets) {
  if (!idx || !offsets || !needle_ids || !out_offsets || (cap && !out)) return CBH_E_INVAL;
  std::vector<Needle> nd(n_needles);
  for (size_t i = 0; i < n_needles; ++i) {
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
    nd[i] = Needle{(size_t)offsets[i], (size_t)offsets[i + 1], needle_ids[i]};
  }
  const size_t nq = n_needles ? (size_t)offsets[n_needles] : 0;
  if (nq && !hashes) return CBH_E_INVAL;
  std::vector<std::vector<cbh_match>> res;
  int rc = fdct_core(idx, hashes, nd, nq, thresh, &res);
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
