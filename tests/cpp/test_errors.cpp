// tests/cpp/test_errors.cpp -- what the adapters do when the device runs out of memory (cbird_amd/cpp/gpu_errors.h):
// a query retries once after releasing cached scratch, then logs with qCritical and returns nothing; a mutation that
// still fails aborts like the reference's own failed allocation (qFatal).  Allocation failures are injected with
// cbh_set_tuning("fault_alloc_after" / "fault_alloc_sticky") (include/cbird_hip.h).
//   test_errors            the query legs; exits 0
//   test_errors mutate     arms a lasting failure and calls load(): must die in qFatal (SIGABRT)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "gpu_dcthashindex.h"

#define CHECK(x)                                                  \
  do {                                                            \
    if (!(x)) {                                                   \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); \
      return 1;                                                   \
    }                                                             \
  } while (0)

static long long tuning(const char* key) {
  long long v = -2;
  cbh_get_tuning(key, &v);
  return v;
}

static bool same(const QVector<Index::Match>& a, const QVector<Index::Match>& b) {
  if (a.count() != b.count()) return false;
  for (int i = 0; i < a.count(); ++i)
    if (a[i].mediaId != b[i].mediaId || a[i].score != b[i].score) return false;
  return true;
}

int main(int argc, char** argv) {
  QSqlDatabase db;
  std::mt19937_64 rng(99);
  const int n = 30000;
  for (int i = 0; i < n; ++i) {
    uint64_t h = rng() & ~1ull;
    if (i % 5 == 4) h = uint64_t(db.media[size_t(i - 2)].phash_dct) ^ (1ull << (1 + i % 63));
    db.media.push_back({uint32_t(i + 1), 1, int64_t(h)});
  }
  if (argc > 1 && !strcmp(argv[1], "mutate")) {
    GpuDctHashIndex idx;
    cbh_set_tuning("fault_alloc_sticky", 1);
    cbh_set_tuning("fault_alloc_after", 0);
    idx.load(db, "", "");  // the index cannot hold the database's rows: there is nothing sensible to go on with
    fprintf(stderr, "load() returned although every allocation failed\n");
    return 1;
  }
  SearchParams p;
  p.dctThresh = 4;
  Media needle("needle", 5, uint64_t(db.media[4].phash_dct));
  QVector<Index::Match> want;
  {
    GpuDctHashIndex ref;
    ref.load(db, "", "");
    want = ref.find(needle, p);
    CHECK(want.count() >= 2);
  }
  // 1. a failure that lasts: find() logs and returns nothing, the process and the handle live on
  {
    GpuDctHashIndex idx;
    idx.load(db, "", "");
    cbh_set_tuning("fault_alloc_sticky", 1);
    cbh_set_tuning("fault_alloc_after", 0);
    const int before = g_mockCriticals;
    const long long fired0 = tuning("fault_fired");
    QVector<Index::Match> got = idx.find(needle, p);
    cbh_set_tuning("fault_alloc_after", -1);
    cbh_set_tuning("fault_alloc_sticky", 0);
    CHECK(tuning("fault_fired") >= fired0 + 2);  // the first attempt and the one after cbh_trim
    CHECK(got.count() == 0 && g_mockCriticals == before + 1);
    CHECK(same(idx.find(needle, p), want));  // the same handle, the same call, now that memory is back
    MediaGroup needles;
    for (int i = 0; i < 400; ++i) needles.append(Media("n", i + 1, uint64_t(db.media[size_t(i)].phash_dct)));
    auto full = idx.findBatch(needles, p);
    cbh_set_tuning("fault_alloc_sticky", 1);
    cbh_set_tuning("fault_alloc_after", 0);
    auto none = idx.findBatch(needles, p);
    QSet<uint32_t> some;
    for (uint32_t id = 10; id < 60; ++id) some.insert(id);
    Index* sub = idx.slice(some);
    cbh_set_tuning("fault_alloc_after", -1);
    cbh_set_tuning("fault_alloc_sticky", 0);
    CHECK(none.count() == 400 && full.count() == 400);
    bool any = false;
    for (auto& r : none) any |= r.count() != 0;
    CHECK(!any && g_mockCriticals >= before + 2);
    CHECK(sub && sub->count() == 0);  // a slice that could not be made searches nothing
    delete sub;
    auto again = idx.findBatch(needles, p);
    for (int i = 0; i < 400; ++i) CHECK(same(again[i], full[i]));
  }
  // 2. a failure that passes: whichever allocation of a fresh index's first find() fails once, the retry answers
  int fired_at = 0;
  for (int k = 0; k < 64; ++k) {
    GpuDctHashIndex idx;
    idx.load(db, "", "");
    const int before = g_mockCriticals;
    const long long fired0 = tuning("fault_fired");
    cbh_set_tuning("fault_alloc_after", k);
    QVector<Index::Match> got = idx.find(needle, p);
    const bool fired = tuning("fault_fired") != fired0;
    cbh_set_tuning("fault_alloc_after", -1);
    CHECK(same(got, want) && g_mockCriticals == before);
    if (!fired) break;  // find() makes fewer than k + 1 allocations: every one of them has had its turn
    fired_at = k + 1;
  }
  CHECK(fired_at >= 1);
  printf("errors ok: %d allocation sites of find() failed once each and were retried; lasting failures logged %d times\n",
         fired_at, g_mockCriticals);
  return 0;
}
