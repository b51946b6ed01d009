/* oracle/search_index.c -- TEST INFRASTRUCTURE ONLY (rules: see cbird_oracle.c's header).
 *
 * Independent CPU restatement of the caller contract around DctHashIndex::find, line by line from fully visible
 * reference code (nothing here comes from OpenCV/Qt internals, so there is nothing to pin beyond the code itself):
 *
 *   Database::searchIndex   src/database.cpp:1691-1757   find; maxThresh escalation (+1 dht while matches <= minMatches,
 *                                                        :1703-1725); std::sort by score (:1729; ties: the reference's
 *                                                        order is unspecified -- fixed to ascending mediaId, SURVEY.md
 *                                                        section 7 hard part 2); filterSelf (:1735); stop at maxMatches
 *                                                        (:1736); ids missing from idMap are warned about and skipped
 *                                                        WITHOUT consuming a place (:1739-1755)
 *   Database::similar       src/database.cpp:1400-1463   one searchIndex per haystack item; a result is kept only when
 *                                                        it is non-empty (:1409), the needle is prepended (:1422);
 *                                                        filterMatch keeps groups with count > minMatches (:1245);
 *                                                        filterMatches (filterGroups, :1252-1272): groups ordered by
 *                                                        the needle's path, a group whose SET of paths was seen before
 *                                                        is dropped (the reference compares qHash of the concatenated
 *                                                        sorted paths; this compares the sets themselves); final order
 *                                                        by path (:1463)
 *   DctHashIndex::find      src/dcthashindex.cpp:193-220  via orc_scan64 semantics (restated locally: strict <, id != 0,
 *                                                        null needle -> nothing)
 *
 * Media are given by parallel arrays: id (unique, != 0), dct hash, path rank (the position of the media's path in the
 * sorted order of all paths: everything the reference does with paths here is ordering and equality).  The index may
 * hold entries whose id is not in the haystack (stale index) -- they are skipped like :1755.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int score;
  uint32_t id;
} sm_match;

static int cmp_match(const void* a, const void* b) {
  const sm_match *x = (const sm_match*)a, *y = (const sm_match*)b;
  if (x->score != y->score) return x->score < y->score ? -1 : 1;
  return x->id < y->id ? -1 : x->id > y->id ? 1 : 0;
}

static size_t find_all(const uint64_t* ih, const uint32_t* ii, size_t n_idx, uint64_t target, int thresh, sm_match* out) {
  size_t m = 0;
  if (target == 0) return 0;
  for (size_t i = 0; i < n_idx; ++i) {
    const int d = __builtin_popcountll(target ^ ih[i]);
    if (d < thresh && ii[i] != 0) {
      out[m].score = d;
      out[m].id = ii[i];
      ++m;
    }
  }
  return m;
}

/* id -> haystack position by binary search over (sorted ids, positions) */
static long lookup(const uint32_t* sid, const uint32_t* spos, size_t n, uint32_t id) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (sid[mid] < id) lo = mid + 1; else hi = mid;
  }
  return (lo < n && sid[lo] == id) ? (long)spos[lo] : -1;
}

typedef struct {
  uint32_t id;
  uint32_t pos;
} idpos;
static int cmp_idpos(const void* a, const void* b) {
  const idpos *x = (const idpos*)a, *y = (const idpos*)b;
  return x->id < y->id ? -1 : x->id > y->id ? 1 : 0;
}

/* searchIndex for needle `j` of the haystack.  Writes at most max_matches (id, score); returns their number. */
static int search_index(const uint64_t* ih, const uint32_t* ii, size_t n_idx, uint64_t needle_hash, uint32_t needle_id,
                        int thresh, int max_thresh, int min_matches, int max_matches, int filter_self,
                        const uint32_t* sid, const uint32_t* spos, size_t n_hay, sm_match* scratch, sm_match* out) {
  size_t m = find_all(ih, ii, n_idx, needle_hash, thresh, scratch);
  if (max_thresh > 0) {
    int t = thresh;
    while ((long long)m <= (long long)min_matches) {
      ++t;
      if (t > max_thresh) break;
      m = find_all(ih, ii, n_idx, needle_hash, t, scratch);
    }
  }
  qsort(scratch, m, sizeof(sm_match), cmp_match);
  int g = 0;
  for (size_t k = 0; k < m; ++k) {
    if (filter_self && scratch[k].id == needle_id) continue;
    if (g >= max_matches) break;
    if (lookup(sid, spos, n_hay, scratch[k].id) < 0) continue; /* "no media with id": skipped, no place consumed */
    out[g++] = scratch[k];
  }
  return g;
}

static const int32_t* g_rank;
static int cmp_by_rank(const void* a, const void* b) {
  const int32_t x = g_rank[*(const uint32_t*)a], y = g_rank[*(const uint32_t*)b];
  return x < y ? -1 : x > y ? 1 : 0;
}
static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : x > y ? 1 : 0;
}

/* Database::similar over an in-memory haystack (= the needles) against an index.
 * out_needle[g] = haystack position of group g's needle; out_first[g]..out_first[g+1] index out_ids/out_scores (the
 * matches, needle not included).  Returns the number of groups, or -1 when a capacity is too small. */
long long orc_similar_dct(const uint64_t* hay_hash, const uint32_t* hay_id, const int32_t* hay_rank, size_t n_hay,
                          const uint64_t* idx_hash, const uint32_t* idx_id, size_t n_idx, int thresh, int max_thresh,
                          int min_matches, int max_matches, int filter_self, int filter_groups, uint32_t* out_needle,
                          uint64_t* out_first, uint32_t* out_ids, int32_t* out_scores, size_t cap_groups,
                          size_t cap_items) {
  idpos* ip = (idpos*)malloc((n_hay ? n_hay : 1) * sizeof(idpos));
  uint32_t* sid = (uint32_t*)malloc((n_hay ? n_hay : 1) * 4);
  uint32_t* spos = (uint32_t*)malloc((n_hay ? n_hay : 1) * 4);
  for (size_t i = 0; i < n_hay; ++i) ip[i].id = hay_id[i], ip[i].pos = (uint32_t)i;
  qsort(ip, n_hay, sizeof(idpos), cmp_idpos);
  for (size_t i = 0; i < n_hay; ++i) sid[i] = ip[i].id, spos[i] = ip[i].pos;
  free(ip);
  sm_match* scratch = (sm_match*)malloc((n_idx ? n_idx : 1) * sizeof(sm_match));
  sm_match* res = (sm_match*)malloc((n_hay ? n_hay : 1) * (size_t)(max_matches > 0 ? max_matches : 1) * sizeof(sm_match));
  int* cnt = (int*)calloc(n_hay ? n_hay : 1, sizeof(int));
  const size_t mm = (size_t)(max_matches > 0 ? max_matches : 1);
  /* 1. one searchIndex per haystack item; empty results are dropped, the needle is prepended (count + 1) */
  uint32_t* order = (uint32_t*)malloc((n_hay ? n_hay : 1) * 4);
  size_t n_groups = 0;
  for (size_t j = 0; j < n_hay; ++j) {
    cnt[j] = search_index(idx_hash, idx_id, n_idx, hay_hash[j], hay_id[j], thresh, max_thresh, min_matches, max_matches,
                          filter_self, sid, spos, n_hay, scratch, res + j * mm);
    if (cnt[j] <= 0) continue;                     /* :1409 */
    if (cnt[j] + 1 > min_matches) order[n_groups++] = (uint32_t)j; /* filterMatch :1245 */
  }
  /* 2. filterMatches: by needle path, drop groups whose set of paths was seen before */
  g_rank = hay_rank;
  qsort(order, n_groups, 4, cmp_by_rank);
  long long out_g = 0;
  size_t items = 0;
  int32_t* keys = (int32_t*)malloc((n_groups ? n_groups : 1) * (mm + 1) * sizeof(int32_t));
  size_t n_keys = 0;
  for (size_t gi = 0; gi < n_groups; ++gi) {
    const uint32_t j = order[gi];
    int32_t key[65];
    const int c = cnt[j];
    key[0] = hay_rank[j];
    for (int k = 0; k < c; ++k) key[1 + k] = hay_rank[lookup(sid, spos, n_hay, res[j * mm + (size_t)k].id)];
    qsort(key, (size_t)c + 1, sizeof(int32_t), cmp_i32);
    int dup = 0;
    if (filter_groups)
      for (size_t q = 0; q < n_keys && !dup; ++q) {
        const int32_t* kq = keys + q * (mm + 1);
        int same = 1;
        for (size_t t = 0; t <= mm && same; ++t) {
          const int32_t a = t <= (size_t)c ? key[t] : -1;
          same = kq[t] == a;
        }
        dup = same;
      }
    if (dup) continue;
    if (filter_groups) {
      int32_t* kq = keys + n_keys * (mm + 1);
      for (size_t t = 0; t <= mm; ++t) kq[t] = t <= (size_t)c ? key[t] : -1;
      ++n_keys;
    }
    if ((size_t)out_g >= cap_groups || items + (size_t)c > cap_items) {
      out_g = -1;
      break;
    }
    out_needle[out_g] = j;
    out_first[out_g] = items;
    for (int k = 0; k < c; ++k) {
      out_ids[items] = res[j * mm + (size_t)k].id;
      out_scores[items] = res[j * mm + (size_t)k].score;
      ++items;
    }
    ++out_g;
  }
  if (out_g >= 0) out_first[out_g] = items;
  free(keys);
  free(order);
  free(cnt);
  free(res);
  free(scratch);
  free(sid);
  free(spos);
  return out_g;
}
