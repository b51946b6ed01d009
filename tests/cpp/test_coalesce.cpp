// The drop-in find() under the reference's calling pattern: Database::similar issues ONE synchronous find() per
// needle from a pool of threads (QtConcurrent::map, src/database.cpp:1400-1432).  T threads x M finds each against
// an N-entry GpuDctHashIndex; every result must equal the batched path's, and the wall time is reported next to it.
//   test_coalesce [N=1000000] [threads=64] [finds_per_thread=16384] [dht=2]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "gpu_dcthashindex.h"

#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                    \
    }                                                              \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000000;
  const int T = argc > 2 ? atoi(argv[2]) : 64;
  const int M = argc > 3 ? atoi(argv[3]) : 16384;
  const int dht = argc > 4 ? atoi(argv[4]) : 2;
  GpuDeviceSet devs;  // optional: device mask and shards per device of the index under test
  if (argc > 6) devs = GpuDeviceSet{uint32_t(strtoul(argv[5], nullptr, 0)), atoi(argv[6])};
  QSqlDatabase db;
  std::mt19937_64 rng(99);
  for (int i = 0; i < n; ++i) {
    uint64_t h = rng() & ~1ull;
    if (i % 10 == 9) h = uint64_t(db.media[size_t(i - 1 - int(rng() % 8))].phash_dct) ^ (1ull << (1 + i % 63));
    if (!h) h = 2;
    db.media.push_back({uint32_t(i + 1), 1, int64_t(h)});
  }
  GpuDctHashIndex idx(devs);
  idx.load(db, "", "");
  CHECK(idx.count() == n);
  SearchParams p;
  p.dctThresh = dht;
  p.maxMatches = 15;
  const size_t total = std::min<size_t>(size_t(T) * size_t(M), size_t(n));
  // reference: the batched extension over the same needles
  MediaGroup needles;
  for (size_t i = 0; i < total; ++i) needles.append(Media("n", int(i + 1), uint64_t(db.media[i].phash_dct)));
  (void)idx.findBatch(needles, p);  // warm
  double t0 = now();
  QVector<QVector<Index::Match>> want = idx.findBatch(needles, p);
  const double t_batch = now() - t0;
  // T threads, one synchronous find() per needle
  std::vector<QVector<Index::Match>> got(total);
  t0 = now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {  // needle i goes to thread i mod T, as a work-sharing pool would interleave them
      for (size_t i = size_t(t); i < total; i += size_t(T)) got[i] = idx.find(needles[int(i)], p);
    });
  for (auto& x : th) x.join();
  const double t_find = now() - t0;
  size_t matches = 0;
  for (size_t i = 0; i < total; ++i) {
    const QVector<Index::Match>& g = got[i];
    const QVector<Index::Match>& w = want[int(i)];
    matches += size_t(g.count());
    CHECK(g.count() >= w.count());  // the batch result is cut at maxMatches + 1
    CHECK(w.count() == std::min(g.count(), p.maxMatches + 1));
    for (int j = 0; j < w.count(); ++j) CHECK(g[j].mediaId == w[j].mediaId && g[j].score == w[j].score);
    for (int j = 1; j < g.count(); ++j)
      CHECK(g[j - 1].score < g[j].score || (g[j - 1].score == g[j].score && g[j - 1].mediaId <= g[j].mediaId));
  }
  cbh_coalesce_stats st;
  CHECK(cbh_idx64_coalesce_stats(idx.handle(), &st) == CBH_OK);
  CHECK(st.finds == total);
  // a needle that is NOT an index entry still gets the exact answer (combined scan, not the cache)
  Media foreign("f", 0, (uint64_t(db.media[5].phash_dct) ^ 0x8000000000000000ull));
  QVector<Index::Match> fr = idx.find(foreign, p);
  bool has5 = false;
  for (auto& r : fr) has5 |= (r.mediaId == 6 && r.score == 1);
  CHECK(has5);
  // add() invalidates the cache: the new entry is found at once
  MediaGroup g2;
  g2.append(Media("new", n + 1, uint64_t(db.media[5].phash_dct)));
  idx.add(g2);
  bool has_new = false;
  for (auto& r : idx.find(needles[5], p)) has_new |= (r.mediaId == uint32_t(n + 1) && r.score == 0);
  CHECK(has_new);
  printf("{\"coalesce\": {\"index\": %d, \"threads\": %d, \"finds\": %zu, \"dht\": %d, \"matches\": %zu, "
         "\"threaded_find_s\": %.4f, \"finds_per_s\": %.0f, \"batch_path_s\": %.4f, \"ratio\": %.2f, "
         "\"rounds\": %llu, \"scanned_needles\": %llu, \"cache_hits\": %llu, \"self_joins\": %llu}}\n",
         n, T, total, dht, matches, t_find, double(total) / t_find, t_batch, t_find / t_batch,
         (unsigned long long)st.rounds, (unsigned long long)st.scanned_needles, (unsigned long long)st.cache_hits,
         (unsigned long long)st.self_joins);
  printf("coalesce ok\n");
  return 0;
}
