// search.hip -- Database::searchIndex and the group filtering of Database::similar for a whole needle batch, behind
// the C-ABI: a 1M-needle `-similar` runs without a per-needle loop in the caller's language.
//
//   cbh_search_index_batch   src/database.cpp:1691-1757 for DctHashIndex: find; the maxThresh escalation (only the
//                            needles whose match count is still <= minMatches are rescanned at the next threshold);
//                            order by (score, mediaId); filterSelf; stop at maxMatches; ids that are not in the
//                            caller's idMap are skipped without consuming a place.  Scans and the per-needle cut run
//                            on the device (hamm64 kernels + the counting select of topk.hip); the few needles whose
//                            cut cannot be decided from the fetched places (more skipped entries than slack) take the
//                            exact single-needle find.
//   cbh_filter_groups        src/database.cpp:1245 (a group needs more than minMatches members, needle included),
//                            :1252-1272 (filterGroups: in needle-path order, a group whose SET of paths was seen
//                            before is dropped -- the reference compares qHash of the concatenated sorted paths, this
//                            compares the sets), :1463 (result order by path).  Host code; paths enter as ranks.
#include <algorithm>
#include <set>

#include "cbh_index.h"

namespace {

constexpr int kSlack = 8;  // places fetched beyond maxMatches + 1 (self) for entries the idMap does not know

bool id_valid(const uint32_t* valid, size_t n_valid, uint32_t id) {
  if (!valid) return true;
  return std::binary_search(valid, valid + n_valid, id);
}

// the filter loop of searchIndex (:1733-1755) over an ordered candidate list; returns false when the list ran out
// before the cut was decided although the index has more matches (`have_all` false)
bool cut_group(const cbh_match* cand, size_t n_cand, bool have_all, uint32_t needle_id, int max_matches, int filter_self,
               const uint32_t* valid, size_t n_valid, cbh_match* out, uint32_t* n_out) {
  uint32_t g = 0;
  for (size_t k = 0; k < n_cand; ++k) {
    if (filter_self && cand[k].id == needle_id) continue;
    if ((int)g >= max_matches) {
      *n_out = g;
      return true;
    }
    if (!id_valid(valid, n_valid, cand[k].id)) continue;
    out[g++] = cand[k];
  }
  *n_out = g;
  return have_all || (int)g >= max_matches;
}

}  // namespace

extern "C" {

int cbh_search_index_batch(cbh_idx64* idx, const uint64_t* q, const uint32_t* needle_ids, size_t nq, int thresh,
                           int max_thresh, int min_matches, int max_matches, int filter_self,
                           const uint32_t* valid_ids_sorted, size_t n_valid, cbh_match* out, uint32_t* out_counts) {
  if (!idx || max_matches < 0 || max_matches > kTopkMaxK - 1 - kSlack || (nq && (!q || !needle_ids || !out_counts)) ||
      (nq && max_matches && !out) || nq > CBH_MAX_QUERIES_PER_CALL)
    return CBH_E_INVAL;
  if (nq == 0) return CBH_OK;
  memset(out_counts, 0, nq * sizeof(uint32_t));
  if (idx->n == 0) return CBH_OK;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  const int k = max_matches + 1 + kSlack;
  std::vector<uint32_t> pending;
  pending.reserve(nq);
  for (size_t j = 0; j < nq; ++j)
    if (q[j] != 0) pending.push_back((uint32_t)j);  // a null needle finds nothing at any threshold (:196-200)
  std::vector<uint64_t> hq;
  std::vector<cbh_match> top;
  std::vector<uint32_t> counts, slow;
  int rc = CBH_OK;
  {
    WsLease L(idx, &rc);
    if (!L.ws) return rc;
    Workspace* ws = L.ws;
    hipStream_t s = ws->stream;
    for (int t = thresh; !pending.empty(); ++t) {
      const size_t np = pending.size();
      hq.resize(np);
      for (size_t i = 0; i < np; ++i) hq[i] = q[pending[i]];
      if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, np))) return rc;
      if ((rc = Workspace::grow(&ws->d_out, &ws->out_cap, np * (size_t)k))) return rc;
      if ((rc = Workspace::grow(&ws->d_counts, &ws->counts_cap, np))) return rc;
      CBH_HIP(hipMemcpyAsync(ws->d_q, hq.data(), np * sizeof(uint64_t), hipMemcpyHostToDevice, s));
      unsigned long long total = 0;
      if (t > 0) {
        if ((rc = scan_all(idx, ws, ws->d_q, np, t, s, &total))) return rc;
      } else {
        if ((rc = ws->ensure_records(1024))) return rc;
        CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), s));
      }
      if (total >= (1ull << 32)) return CBH_E_OVERFLOW;
      {
        void* scratch = nullptr;
        const size_t ncap = std::min<size_t>(ws->rec_cap, (size_t)total + 1);
        CBH_HIP(cbh::malloc_async(&scratch, topk_scratch_bytes(np, ncap) + 16, s));
        unsigned* d_status = (unsigned*)((char*)scratch + topk_scratch_bytes(np, ncap));
        rc = topk_scratch_init(scratch, np, s);
        if (!rc) rc = launch_records_topk(ws->d_total, 1, 0, ncap, np, k, ws->d_out, ws->d_counts, d_status, scratch, s);
        (void)cbh::free_async(scratch, s);
        if (rc) return rc;
      }
      top.resize(np * (size_t)k);
      counts.resize(np);
      CBH_HIP(hipMemcpyAsync(top.data(), ws->d_out, top.size() * sizeof(cbh_match), hipMemcpyDeviceToHost, s));
      CBH_HIP(hipMemcpyAsync(counts.data(), ws->d_counts, np * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      CBH_HIP(hipStreamSynchronize(s));
      std::vector<uint32_t> next;
      for (size_t i = 0; i < np; ++i) {
        const uint32_t j = pending[i];
        // `while (matches.count() <= params.minMatches)`: raise the threshold unless that would pass maxThresh
        if (max_thresh > 0 && (long long)counts[i] <= (long long)min_matches && t + 1 <= max_thresh) {
          next.push_back(j);
          continue;
        }
        const size_t have = std::min<size_t>(counts[i], (size_t)k);
        if (!cut_group(top.data() + i * (size_t)k, have, counts[i] <= (uint32_t)k, needle_ids[j], max_matches,
                       filter_self, valid_ids_sorted, n_valid, out + (size_t)j * (size_t)max_matches, &out_counts[j])) {
          slow.push_back(j);
          slow.push_back((uint32_t)t);
        }
      }
      pending.swap(next);
    }
  }
  // the undecided few: every match of the needle, exactly
  std::vector<cbh_match> all;
  for (size_t i = 0; i + 1 < slow.size(); i += 2) {
    const uint32_t j = slow[i];
    const int t = (int)slow[i + 1];
    size_t n = 0;
    all.resize(1024);
    for (;;) {
      rc = cbh_idx64_find(idx, q[j], t, all.data(), all.size(), &n);
      if (rc) return rc;
      if (n <= all.size()) break;
      all.resize(n);
    }
    (void)cut_group(all.data(), n, true, needle_ids[j], max_matches, filter_self, valid_ids_sorted, n_valid,
                    out + (size_t)j * (size_t)max_matches, &out_counts[j]);
  }
  return CBH_OK;
}

int cbh_filter_groups(const uint32_t* needle_ids, const cbh_match* matches, const uint32_t* counts, size_t nq,
                      int max_matches, int min_matches, int filter_groups, const uint32_t* ids_sorted,
                      const uint32_t* path_rank, size_t n_ids, uint32_t* out_group, size_t* n_out) {
  if (!n_out || (nq && (!needle_ids || !counts || !out_group)) || (nq && max_matches && !matches) ||
      (n_ids && (!ids_sorted || !path_rank)) || max_matches < 0)
    return CBH_E_INVAL;
  *n_out = 0;
  auto rank_of = [&](uint32_t id, uint32_t* r) {
    const uint32_t* p = std::lower_bound(ids_sorted, ids_sorted + n_ids, id);
    if (p == ids_sorted + n_ids || *p != id) return false;
    *r = path_rank[p - ids_sorted];
    return true;
  };
  struct G {
    uint32_t rank, j;
  };
  std::vector<G> order;
  for (size_t j = 0; j < nq; ++j) {
    if (counts[j] == 0) continue;                                          // empty result: not a group (:1409)
    if ((long long)counts[j] + 1 <= (long long)min_matches) continue;     // filterMatch (:1245), needle included
    uint32_t r;
    if (!rank_of(needle_ids[j], &r)) return CBH_E_INVAL;
    order.push_back(G{r, (uint32_t)j});
  }
  std::sort(order.begin(), order.end(), [](const G& a, const G& b) { return a.rank != b.rank ? a.rank < b.rank : a.j < b.j; });
  std::set<std::vector<uint32_t>> seen;
  size_t m = 0;
  for (const G& gp : order) {
    if (filter_groups) {
      std::vector<uint32_t> key;
      key.push_back(gp.rank);
      for (uint32_t t = 0; t < counts[gp.j]; ++t) {
        uint32_t r;
        if (!rank_of(matches[(size_t)gp.j * (size_t)max_matches + t].id, &r)) return CBH_E_INVAL;
        key.push_back(r);
      }
      std::sort(key.begin(), key.end());
      if (!seen.insert(std::move(key)).second) continue;
    }
    out_group[m++] = gp.j;
  }
  *n_out = m;
  return CBH_OK;
}

/* The whole of filterMatch (src/database.cpp:1209-1248) and filterMatches (:1250-1278) on per-needle results, host code:
 * path / inPath (:1217-1229), filterParent (:1231-1242), the count rule (:1245), filterGroups (:1253-1272), then
 * mergeGroups (Media::mergeGroupList, src/media.cpp:300-324) or expandGroups (:326-331), and the final order by the
 * first member's path (:1463).  Media enter as ids with three attributes the caller derives from their paths. */
int cbh_filter_groups_ex(const uint32_t* needle_ids, const cbh_match* matches, const uint32_t* counts, size_t nq,
                         int max_matches, const cbh_filter_params* p, const uint32_t* ids_sorted,
                         const uint32_t* path_rank, const uint32_t* dir_id, const uint8_t* under_prefix, size_t n_ids,
                         uint64_t* out_first, size_t cap_groups, cbh_match* out_members, size_t cap_members,
                         size_t* n_groups, size_t* n_members) {
  if (!p || !n_groups || !n_members || (nq && (!needle_ids || !counts)) || (nq && max_matches && !matches) ||
      (n_ids && (!ids_sorted || !path_rank)) || max_matches < 0 || (p->filter_parent && n_ids && !dir_id) ||
      (p->path_mode && n_ids && !under_prefix) || p->path_mode < 0 || p->path_mode > 2 || !out_first ||
      (cap_members && !out_members))
    return CBH_E_INVAL;
  *n_groups = *n_members = 0;
  auto slot_of = [&](uint32_t id) -> long {
    const uint32_t* q = std::lower_bound(ids_sorted, ids_sorted + n_ids, id);
    return (q == ids_sorted + n_ids || *q != id) ? -1 : (long)(q - ids_sorted);
  };
  struct M {
    uint32_t id, rank;
    int32_t score;
  };
  typedef std::vector<M> Group;
  std::vector<Group> list;
  for (size_t j = 0; j < nq; ++j) {
    if (counts[j] == 0) continue;  // a needle without a result is no group (:1409)
    const long ns = slot_of(needle_ids[j]);
    if (ns < 0) return CBH_E_INVAL;
    Group g;
    g.push_back(M{needle_ids[j], path_rank[ns], -1});  // the needle is a haystack Media: score -1 (media.cpp:112)
    const bool path_on = p->path_mode != 0 && counts[j] + 1 > 1;
    for (uint32_t t = 0; t < counts[j]; ++t) {
      const cbh_match& m = matches[j * (size_t)max_matches + t];
      const long ms = slot_of(m.id);
      if (ms < 0) return CBH_E_INVAL;
      // only results under path / not under path: `(!inPath) ^ startsWith(prefix)` keeps the match (:1226-1227)
      if (path_on && ((p->path_mode == 2) ^ (under_prefix[ms] != 0)) == 0) continue;
      g.push_back(M{m.id, path_rank[ms], m.score});
    }
    if (p->filter_parent && g.size() > 1) {  // remove a match in the needle's directory / zip (:1232-1241)
      const uint32_t parent = dir_id[ns];
      Group kept;
      kept.push_back(g[0]);
      for (size_t i = 1; i < g.size(); ++i)
        if (dir_id[slot_of(g[i].id)] != parent) kept.push_back(g[i]);
      g.swap(kept);
    }
    if ((long long)g.size() > (long long)p->min_matches) list.push_back(std::move(g));  // (:1245)
  }
  auto by_first_path = [](const Group& a, const Group& b) {
    if (a.empty()) return true;  // Media::sortGroupList's comparator (media.cpp:336-341)
    if (b.empty()) return false;
    return a[0].rank < b[0].rank;
  };
  if (p->filter_groups) {
    std::stable_sort(list.begin(), list.end(), by_first_path);
    std::set<std::vector<uint32_t>> seen;
    std::vector<Group> filtered;
    for (Group& g : list) {
      std::vector<uint32_t> key;
      for (const M& m : g) key.push_back(m.rank);
      std::sort(key.begin(), key.end());
      if (seen.insert(std::move(key)).second) filtered.push_back(std::move(g));
    }
    list.swap(filtered);
  }
  if (p->merge_groups) {
    // merge 1-connected matches: if a contains b's first member, b's other members join a and b is emptied; a is then
    // ordered by score (std::sort on Media::operator<; equal scores: by path here)
    auto contains = [](const Group& g, uint32_t id) {
      for (const M& m : g)
        if (m.id == id) return true;
      return false;
    };
    for (size_t i = 0; i < list.size(); ++i)
      for (size_t j = 0; j < list.size(); ++j) {
        if (i == j) continue;
        Group& a = list[i];
        Group& b = list[j];
        if (!b.empty() && contains(a, b[0].id)) {
          for (size_t k = 1; k < b.size(); ++k)
            if (!contains(a, b[k].id)) a.push_back(b[k]);
          b.clear();
          std::sort(a.begin(), a.end(), [](const M& x, const M& y) { return x.score != y.score ? x.score < y.score : x.rank < y.rank; });
        }
      }
    std::vector<Group> fin;
    for (Group& g : list)
      if (!g.empty()) fin.push_back(std::move(g));
    list.swap(fin);
  } else if (p->expand_groups) {
    std::vector<Group> ex;
    for (const Group& g : list)
      for (size_t i = 1; i < g.size(); ++i) ex.push_back(Group{g[0], g[i]});
    list.swap(ex);
  }
  std::stable_sort(list.begin(), list.end(), by_first_path);  // Media::sortGroupList(list, {"path"}) (:1463)
  size_t members = 0;
  for (const Group& g : list) members += g.size();
  *n_groups = list.size();
  *n_members = members;
  if (list.size() > cap_groups || members > cap_members) return CBH_E_OVERFLOW;
  size_t pos = 0;
  for (size_t gi = 0; gi < list.size(); ++gi) {
    out_first[gi] = pos;
    for (const M& m : list[gi]) out_members[pos++] = cbh_match{m.id, m.score};
  }
  out_first[list.size()] = pos;  // (out_first has cap_groups + 1 entries)
  return CBH_OK;
}

}  // extern "C"
