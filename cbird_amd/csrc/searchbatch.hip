// searchbatch.hip -- Database::searchIndex (src/database.cpp:1691-1757) for a whole needle batch over the other four
// indexes: DctFeaturesIndex, DctVideoIndex, CvFeaturesIndex, ColorDescIndex (DctHashIndex: search.hip).
//
// The reference runs, per needle: find(); while the result has <= minMatches entries raise the threshold -- dctThresh++
// for dct / dct features / video, cvThresh += 5 for ORB, nothing for colour (:1703-1725) -- and find() again, keeping the
// last result when maxThresh is passed; std::sort by score (:1729; equal scores: ascending mediaId here); skip the
// needle itself when filterSelf (:1735); stop at maxMatches (:1736); skip ids the caller's idMap does not hold without
// consuming a place (:1755).  Here the find() of one threshold level is ONE batched call of the index for every needle
// still pending at that level (cbh_fdct_find_batch_ex / cbh_vidx_find_videos_batch / cbh_idx256_find_batch /
// cbh_color_find_batch: scans, per-needle reductions and scoring on the device), and only the needles that still have
// too few matches go on to the next level.  Host code around those entry points; no kernels of its own.
#include <algorithm>
#include <vector>

#include "cbh_index.h"

namespace {

bool id_known(const uint32_t* valid, size_t n_valid, uint32_t id) {
  return !valid || std::binary_search(valid, valid + n_valid, id);
}

// M has .id and .score.  find(pending needle indices, threshold, results[pending.size()]) -> rc
template <class M, class Find>
int search_index_levels(size_t n, const uint32_t* needle_ids, int thresh, int max_thresh, int step, int min_matches,
                        int max_matches, int filter_self, const uint32_t* valid, size_t n_valid, Find find, M* out,
                        uint32_t* out_counts) {
  std::vector<uint32_t> pending(n);
  for (size_t j = 0; j < n; ++j) pending[j] = (uint32_t)j, out_counts[j] = 0;
  std::vector<std::vector<M>> res;
  for (int t = thresh; !pending.empty(); t += step) {
    res.assign(pending.size(), {});
    int rc = find(pending, t, res);
    if (rc) return rc;
    std::vector<uint32_t> next;
    for (size_t i = 0; i < pending.size(); ++i) {
      const uint32_t j = pending[i];
      std::vector<M>& m = res[i];
      // `while (matches.count() <= params.minMatches)`: one more level unless that would pass maxThresh (:1705-1724)
      if (max_thresh > 0 && step > 0 && (long long)m.size() <= (long long)min_matches && t + step <= max_thresh) {
        next.push_back(j);
        continue;
      }
      std::stable_sort(m.begin(), m.end(), [](const M& a, const M& b) { return a.score != b.score ? a.score < b.score : a.id < b.id; });
      uint32_t g = 0;
      for (const M& x : m) {
        if (filter_self && x.id == needle_ids[j]) continue;
        if ((int)g >= max_matches) break;
        if (!id_known(valid, n_valid, x.id)) continue;
        out[(size_t)j * (size_t)max_matches + g++] = x;
      }
      out_counts[j] = g;
    }
    pending.swap(next);
    if (step <= 0) break;
  }
  return CBH_OK;
}

}  // namespace

extern "C" {

int cbh_fdct_search_index_batch(cbh_idx64* idx, const uint64_t* hashes, const uint64_t* offsets, const uint32_t* needle_ids,
                                size_t n_needles, int thresh, int max_thresh, int tree_compat, int min_matches,
                                int max_matches, int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid,
                                cbh_match* out, uint32_t* out_counts) {
  if (!idx || max_matches < 0 || (n_needles && (!offsets || !needle_ids || !out_counts)) ||
      (n_needles && max_matches && !out))
    return CBH_E_INVAL;
  for (size_t i = 0; i < n_needles; ++i)
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
  if (n_needles && offsets[n_needles] && !hashes) return CBH_E_INVAL;
  auto find = [&](const std::vector<uint32_t>& pend, int t, std::vector<std::vector<cbh_match>>& res) {
    std::vector<uint64_t> h, offs(1, 0);
    std::vector<uint32_t> ids;
    for (uint32_t j : pend) {
      h.insert(h.end(), hashes + offsets[j], hashes + offsets[j + 1]);
      offs.push_back(h.size());
      ids.push_back(needle_ids[j]);
    }
    std::vector<cbh_match> buf(h.size() * 10 + 1);  // at most 10 candidates per needle hash vote (:301-303)
    std::vector<uint64_t> oo(pend.size() + 1);
    int rc = cbh_fdct_find_batch_ex(idx, h.data(), offs.data(), ids.data(), pend.size(), t, tree_compat, buf.data(),
                                    buf.size(), oo.data());
    if (rc) return rc;
    for (size_t i = 0; i < pend.size(); ++i) res[i].assign(buf.begin() + (long)oo[i], buf.begin() + (long)oo[i + 1]);
    return (int)CBH_OK;
  };
  return search_index_levels<cbh_match>(n_needles, needle_ids, thresh, max_thresh, 1, min_matches, max_matches, filter_self,
                                        valid_ids_sorted, n_valid, find, out, out_counts);
}

int cbh_vidx_search_index_batch(cbh_vidx* v, const int32_t* frames, const uint64_t* hashes, const uint64_t* offsets,
                                const uint32_t* needle_ids, size_t n_needles, int thresh, int max_thresh, int skip_frames,
                                int min_frames_matched, int min_frames_near, int min_matches, int max_matches,
                                int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid, cbh_vmatch* out,
                                uint32_t* out_counts) {
  if (!v || max_matches < 0 || (n_needles && (!offsets || !needle_ids || !out_counts)) ||
      (n_needles && max_matches && !out))
    return CBH_E_INVAL;
  for (size_t i = 0; i < n_needles; ++i)
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
  if (n_needles && offsets[n_needles] && (!hashes || !frames)) return CBH_E_INVAL;
  const size_t n_videos = cbh_vidx_count(v);
  auto find = [&](const std::vector<uint32_t>& pend, int t, std::vector<std::vector<cbh_vmatch>>& res) {
    std::vector<int32_t> f;
    std::vector<uint64_t> h, offs(1, 0);
    std::vector<uint32_t> ids;
    for (uint32_t j : pend) {
      f.insert(f.end(), frames + offsets[j], frames + offsets[j + 1]);
      h.insert(h.end(), hashes + offsets[j], hashes + offsets[j + 1]);
      offs.push_back(h.size());
      ids.push_back(needle_ids[j]);
    }
    // findVideo drops the needle's own video itself when filterSelf (src/dctvideoindex.cpp:494), so the count that
    // decides about another level excludes it, as in the reference
    std::vector<cbh_vmatch> buf(std::max<size_t>(1, std::min<size_t>(pend.size() * n_videos, (size_t)1 << 22)));
    std::vector<uint64_t> oo(pend.size() + 1);
    int rc = cbh_vidx_find_videos_batch(v, f.data(), h.data(), offs.data(), ids.data(), pend.size(), t, skip_frames,
                                        min_frames_matched, min_frames_near, filter_self, buf.data(), buf.size(), oo.data());
    if (rc == CBH_E_OVERFLOW) {
      buf.resize((size_t)oo[pend.size()]);
      rc = cbh_vidx_find_videos_batch(v, f.data(), h.data(), offs.data(), ids.data(), pend.size(), t, skip_frames,
                                      min_frames_matched, min_frames_near, filter_self, buf.data(), buf.size(),
                                      oo.data());
    }
    if (rc) return rc;
    for (size_t i = 0; i < pend.size(); ++i) res[i].assign(buf.begin() + (long)oo[i], buf.begin() + (long)oo[i + 1]);
    return (int)CBH_OK;
  };
  return search_index_levels<cbh_vmatch>(n_needles, needle_ids, thresh, max_thresh, 1, min_matches, max_matches,
                                         filter_self, valid_ids_sorted, n_valid, find, out, out_counts);
}

int cbh_idx256_search_index_batch(cbh_idx256* ix, const uint8_t* rows, const uint64_t* offsets, const uint32_t* needle_ids,
                                  size_t n_needles, int thresh, int max_thresh, int k, int min_matches, int max_matches,
                                  int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid, cbh_match* out,
                                  uint32_t* out_counts) {
  if (!ix || max_matches < 0 || k <= 0 || (n_needles && (!offsets || !needle_ids || !out_counts)) ||
      (n_needles && max_matches && !out))
    return CBH_E_INVAL;
  for (size_t i = 0; i < n_needles; ++i)
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
  if (n_needles && offsets[n_needles] && !rows) return CBH_E_INVAL;
  auto find = [&](const std::vector<uint32_t>& pend, int t, std::vector<std::vector<cbh_match>>& res) {
    std::vector<uint8_t> r;
    std::vector<uint64_t> offs(1, 0);
    for (uint32_t j : pend) {
      r.insert(r.end(), rows + offsets[j] * 32, rows + offsets[j + 1] * 32);
      offs.push_back(r.size() / 32);
    }
    std::vector<cbh_match> buf(r.size() / 32 * (size_t)k + 1);  // a needle descriptor votes for at most k media
    std::vector<uint64_t> oo(pend.size() + 1);
    int rc = cbh_idx256_find_batch(ix, r.data(), offs.data(), pend.size(), t, k, buf.data(), buf.size(), oo.data());
    if (rc) return rc;
    for (size_t i = 0; i < pend.size(); ++i) res[i].assign(buf.begin() + (long)oo[i], buf.begin() + (long)oo[i + 1]);
    return (int)CBH_OK;
  };
  return search_index_levels<cbh_match>(n_needles, needle_ids, thresh, max_thresh, 5, min_matches, max_matches, filter_self,
                                        valid_ids_sorted, n_valid, find, out, out_counts);  // cvThresh += 5 (:1714)
}

int cbh_color_search_index_batch(cbh_color* c, const void* needle_descs, const uint32_t* needle_ids, size_t n_needles,
                                 int max_matches, int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid,
                                 cbh_match* out, uint32_t* out_counts) {
  if (!c || max_matches < 0 || (n_needles && (!needle_descs || !needle_ids || !out_counts)) ||
      (n_needles && max_matches && !out))
    return CBH_E_INVAL;
  // no thresholding for colour (:1717-1718): one level.  The cut needs the first maxMatches known, non-self entries in
  // (score, id) order: fetch some more than that, and take the complete list for the needles whose fetched places ran out
  const int slack = 8, kk = max_matches + 1 + slack;
  auto find = [&](const std::vector<uint32_t>& pend, int, std::vector<std::vector<cbh_match>>& res) {
    std::vector<cbh_match> top(pend.size() * (size_t)kk);
    std::vector<uint32_t> cnt(pend.size());
    int rc = cbh_color_find_batch(c, needle_descs, pend.size(), kk, top.data(), cnt.data());  // (pend is 0..n-1 here)
    if (rc) return rc;
    for (size_t i = 0; i < pend.size(); ++i) {
      const size_t have = std::min<size_t>(cnt[i], (size_t)kk);
      res[i].assign(top.begin() + (long)(i * (size_t)kk), top.begin() + (long)(i * (size_t)kk + have));
      if (cnt[i] > (uint32_t)kk) {  // were enough of the fetched places usable?
        int usable = 0;
        for (const cbh_match& m : res[i])
          if (!(filter_self && m.id == needle_ids[pend[i]]) && id_known(valid_ids_sorted, n_valid, m.id)) ++usable;
        if (usable < max_matches) {  // no: the whole list of this needle
          std::vector<cbh_match> all(cnt[i]);
          size_t nn = 0;
          rc = cbh_color_find(c, (const uint8_t*)needle_descs + (size_t)pend[i] * CBH_COLOR_DESC_BYTES, all.data(),
                              all.size(), &nn);
          if (rc) return rc;
          all.resize(std::min(nn, all.size()));
          res[i].swap(all);
        }
      }
    }
    return (int)CBH_OK;
  };
  return search_index_levels<cbh_match>(n_needles, needle_ids, 0, 0, 0, 0, max_matches, filter_self, valid_ids_sorted,
                                        n_valid, find, out, out_counts);
}

}  // extern "C"
