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
#include <stdio.h>
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

/* ---- filterMatch / filterMatches on path STRINGS (src/database.cpp:1209-1278, src/media.cpp:198-208, 300-331,
 * 1039-1043, 1083-1099): a third statement beside the C-ABI's id/attribute form (cbh_filter_groups_ex) and
 * cbird_amd/database.py, working on what the reference works on.  Test infrastructure.
 *
 * Groups enter flattened: group g = members[first[g] .. first[g+1]), member = index into paths[] (the needle first),
 * scores beside them (the needle's is -1).  Groups leave the same way.  Returns the number of groups, -1 = capacity. */
static const char* const kZipMarkers[20] = {".zip:", ".ZIP:", ".cbz:", ".CBZ:", ".epub:", ".EPUB:", ".odt:", ".ODT:",
                                           ".ods:", ".ODS:", ".odp:", ".ODP:", ".docx:", ".DOCX:", ".pptx:", ".PPTX:",
                                           ".xlsx:", ".XLSX:", ".xps:", ".XPS"};

/* Media::parseArchivePath: length of the parent (archive) path, or -1 */
static long archive_parent_len(const char* path) {
  const long len = (long)strlen(path);
  long end = -1;
  for (long i = len - 1; i >= 0; --i)
    if (path[i] == ':') {
      end = i;
      break;
    }
  while (end > 1) {
    for (int m = 0; m < 20; ++m) {
      const long ml = (long)strlen(kZipMarkers[m]);
      const long start = end - ml + 1;
      if (start < 0) continue;
      if (start + ml <= len && strncmp(path + start, kZipMarkers[m], (size_t)ml) == 0) return start + ml - 1;
    }
    long prev = -1;
    for (long i = end - 1; i >= 0; --i)
      if (path[i] == ':') {
        prev = i;
        break;
      }
    end = prev;
  }
  return -1;
}

/* Media::dirPath into buf */
static void dir_path(const char* path, char* buf, size_t cap) {
  long n = archive_parent_len(path);
  if (n < 0) {
    n = -1;
    for (long i = (long)strlen(path) - 1; i >= 0; --i)
      if (path[i] == '/') {
        n = i;
        break;
      }
    if (n < 0) n = 0;
  }
  if ((size_t)n >= cap) n = (long)cap - 1;
  memcpy(buf, path, (size_t)n);
  buf[n] = 0;
}

typedef struct {
  int* m;     /* member = index into paths */
  int* s;     /* score */
  int n, cap;
} fgroup;

static void fg_push(fgroup* g, int m, int s) {
  if (g->n == g->cap) {
    g->cap = g->cap ? g->cap * 2 : 8;
    g->m = (int*)realloc(g->m, (size_t)g->cap * sizeof(int));
    g->s = (int*)realloc(g->s, (size_t)g->cap * sizeof(int));
  }
  g->m[g->n] = m, g->s[g->n] = s, g->n++;
}
static int fg_contains(const fgroup* g, const char* const* paths, int m) {
  for (int i = 0; i < g->n; ++i)
    if (strcmp(paths[g->m[i]], paths[m]) == 0) return 1; /* Media::operator== compares paths */
  return 0;
}
static const char* const* g_paths;
static int cmp_str_idx(const void* a, const void* b) { return strcmp(g_paths[*(const int*)a], g_paths[*(const int*)b]); }

/* stable insertion sort of groups by the first member's path (Media::sortGroupList) */
static void sort_groups(fgroup* list, int n, const char* const* paths) {
  for (int i = 1; i < n; ++i) {
    fgroup t = list[i];
    int j = i - 1;
    while (j >= 0) {
      /* cmp(t, list[j]): "a.count() < 1 -> true; b.count() < 1 -> false; else path(a0) < path(b0)" */
      int less;
      if (t.n < 1) less = 1;
      else if (list[j].n < 1) less = 0;
      else less = strcmp(paths[t.m[0]], paths[list[j].m[0]]) < 0;
      if (!less) break;
      list[j + 1] = list[j];
      --j;
    }
    list[j + 1] = t;
  }
}

long long orc_filter_groups_paths(const char* const* paths, const uint64_t* first, const int32_t* members,
                                  const int32_t* scores, size_t n_groups_in, const char* db_path, const char* param_path,
                                  int in_path, int filter_parent, int min_matches, int filter_groups, int merge_groups,
                                  int expand_groups, uint64_t* out_first, int32_t* out_members, int32_t* out_scores,
                                  size_t cap_groups, size_t cap_members) {
  fgroup* list = (fgroup*)calloc(n_groups_in ? n_groups_in : 1, sizeof(fgroup));
  int n = 0;
  char prefix[4096];
  prefix[0] = 0;
  if (param_path[0]) { /* :1221-1223 */
    if (strncmp(param_path, db_path, strlen(db_path)) == 0)
      snprintf(prefix, sizeof prefix, "%s", param_path);
    else
      snprintf(prefix, sizeof prefix, "%s/%s", db_path, param_path);
  }
  for (size_t g = 0; g < n_groups_in; ++g) {
    const size_t a = (size_t)first[g], b = (size_t)first[g + 1];
    if (b - a <= 1) continue; /* results without a match never become a group (:1409; the needle was prepended) */
    fgroup grp = {0, 0, 0, 0};
    for (size_t t = a; t < b; ++t) fg_push(&grp, members[t], scores[t]);
    /* filterMatch */
    if (param_path[0] && grp.n > 1) {
      fgroup tmp = {0, 0, 0, 0};
      fg_push(&tmp, grp.m[0], grp.s[0]);
      for (int i = 1; i < grp.n; ++i) {
        const int starts = strncmp(paths[grp.m[i]], prefix, strlen(prefix)) == 0;
        if ((!in_path) ^ starts) fg_push(&tmp, grp.m[i], grp.s[i]);
      }
      free(grp.m), free(grp.s);
      grp = tmp;
    }
    if (filter_parent && grp.n > 1) {
      char parent[4096], d[4096];
      dir_path(paths[grp.m[0]], parent, sizeof parent);
      for (int i = 1; i < grp.n; ++i) {
        dir_path(paths[grp.m[i]], d, sizeof d);
        if (strcmp(d, parent) == 0) { /* match.remove(i); --i; */
          memmove(grp.m + i, grp.m + i + 1, (size_t)(grp.n - i - 1) * sizeof(int));
          memmove(grp.s + i, grp.s + i + 1, (size_t)(grp.n - i - 1) * sizeof(int));
          grp.n--;
          --i;
        }
      }
    }
    if (grp.n > min_matches)
      list[n++] = grp;
    else
      free(grp.m), free(grp.s);
  }
  /* filterMatches */
  if (filter_groups) {
    sort_groups(list, n, paths);
    char** keys = (char**)calloc((size_t)(n ? n : 1), sizeof(char*));
    int n_keys = 0, kept = 0;
    for (int g = 0; g < n; ++g) {
      int* idx = (int*)malloc((size_t)list[g].n * sizeof(int));
      memcpy(idx, list[g].m, (size_t)list[g].n * sizeof(int));
      g_paths = paths;
      qsort(idx, (size_t)list[g].n, sizeof(int), cmp_str_idx);
      size_t len = 1;
      for (int i = 0; i < list[g].n; ++i) len += strlen(paths[idx[i]]) + 1;
      char* str = (char*)malloc(len);
      str[0] = 0;
      for (int i = 0; i < list[g].n; ++i) strcat(str, paths[idx[i]]), strcat(str, "\n"); /* (the reference hashes the
                                                       bare concatenation; a separator keeps distinct sets distinct) */
      free(idx);
      int dup = 0;
      for (int q = 0; q < n_keys && !dup; ++q) dup = strcmp(keys[q], str) == 0;
      if (dup) {
        free(str);
        free(list[g].m), free(list[g].s);
      } else {
        keys[n_keys++] = str;
        list[kept++] = list[g];
      }
    }
    for (int q = 0; q < n_keys; ++q) free(keys[q]);
    free(keys);
    n = kept;
  }
  if (merge_groups) { /* Media::mergeGroupList */
    for (int i = 0; i < n; i++) {
      fgroup* a = &list[i];
      for (int j = 0; j < n; j++)
        if (i != j) {
          fgroup* b = &list[j];
          if (b->n > 0 && fg_contains(a, paths, b->m[0])) {
            for (int k = 1; k < b->n; k++)
              if (!fg_contains(a, paths, b->m[k])) fg_push(a, b->m[k], b->s[k]);
            b->n = 0;
            /* std::sort(a) on Media::operator< (score); equal scores by path */
            for (int x = 1; x < a->n; ++x) {
              const int mm = a->m[x], ss = a->s[x];
              int y = x - 1;
              while (y >= 0 && (a->s[y] > ss || (a->s[y] == ss && strcmp(paths[a->m[y]], paths[mm]) > 0))) {
                a->m[y + 1] = a->m[y], a->s[y + 1] = a->s[y];
                --y;
              }
              a->m[y + 1] = mm, a->s[y + 1] = ss;
            }
          }
        }
    }
    int kept = 0;
    for (int g = 0; g < n; ++g) {
      if (list[g].n > 0)
        list[kept++] = list[g];
      else
        free(list[g].m), free(list[g].s);
    }
    n = kept;
  } else if (expand_groups) { /* Media::expandGroupList */
    int total = 0;
    for (int g = 0; g < n; ++g) total += list[g].n - 1;
    fgroup* ex = (fgroup*)calloc((size_t)(total ? total : 1), sizeof(fgroup));
    int e = 0;
    for (int g = 0; g < n; ++g) {
      for (int i = 1; i < list[g].n; ++i) {
        fg_push(&ex[e], list[g].m[0], list[g].s[0]);
        fg_push(&ex[e], list[g].m[i], list[g].s[i]);
        ++e;
      }
      free(list[g].m), free(list[g].s);
    }
    free(list);
    list = ex;
    n = total;
  }
  sort_groups(list, n, paths); /* :1463 */
  long long rc = n;
  size_t pos = 0;
  for (int g = 0; g < n; ++g) {
    if ((size_t)g >= cap_groups || pos + (size_t)list[g].n > cap_members) {
      rc = -1;
      break;
    }
    out_first[g] = pos;
    for (int i = 0; i < list[g].n; ++i) out_members[pos] = list[g].m[i], out_scores[pos] = list[g].s[i], ++pos;
  }
  if (rc >= 0) out_first[n] = pos;
  for (int g = 0; g < n; ++g) free(list[g].m), free(list[g].s);
  free(list);
  return rc;
}
