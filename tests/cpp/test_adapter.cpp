// Exercises cbird_amd/cpp/gpu_dcthashindex.h (the cbird-side binding) against the mock of the
// reference headers, on a real MI355X.  Mirrors unit/testdcthashindex.cpp + unit/testindexbase.cpp:
// defaults, empty, load, memoryUsage == 12 B * count, find == brute force, add/remove, slice.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>

#include "gpu_dcthashindex.h"

#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                   \
    }                                                             \
  } while (0)

static int hamm64(uint64_t a, uint64_t b) { return __builtin_popcountll(a ^ b); }

// usage: test_adapter [device_mask shards_per_device]  -- with a shape, the index under test is ONE GpuDctHashIndex over
// that many shards (GpuDeviceSet), and every find additionally has to equal the one-device index's answer
int main(int argc, char** argv) {
  GpuDeviceSet devs;
  if (argc > 1 && !strcmp(argv[1], "all"))  // every usable device of the node, as Engine::Engine would ask for it
    devs = GpuDeviceSet::all();
  else if (argc > 2)
    devs = GpuDeviceSet{uint32_t(strtoul(argv[1], nullptr, 0)), atoi(argv[2])};
  if (argc > 3 && !strcmp(argv[3], "rccl")) {  // blocks through ncclAllGather
    cbh_set_tuning("shard_force_rccl", 1);
    cbh_set_tuning("shard_exchange", 0);
  }
  QSqlDatabase db;
  std::mt19937_64 rng(1234);
  const int n = 20000;
  for (int i = 0; i < n; ++i) {
    uint64_t h = rng() & ~1ull;
    if (i % 10 == 9) h = uint64_t(db.media[size_t(i - 3)].phash_dct) ^ (1ull << (1 + i % 63));
    db.media.push_back({uint32_t(i + 1), (i % 50 == 49) ? 2 : 1, int64_t(h)});  // some videos (type 2)
  }
  GpuDctHashIndex idx(devs);
  GpuDctHashIndex plain;  // the one-device index beside it
  CHECK(cbh_idx64_shard_count(idx.handle()) == (devs.single() ? 1 : __builtin_popcount(devs.mask) * devs.shardsPerDevice));
  CHECK(!idx.isLoaded() && idx.count() == 0 && idx.memoryUsage() == 0);
  CHECK(idx.id() == SearchParams::AlgoDCT && idx.databaseId() == 0);
  idx.load(db, "", "");
  plain.load(db, "", "");
  CHECK(idx.isLoaded());
  CHECK(idx.count() == n - n / 50);
  CHECK(idx.memoryUsage() == size_t(12) * size_t(idx.count()));
  SearchParams p;
  p.dctThresh = 3;
  int checked = 0;
  for (int i = 0; i < n; i += 97) {
    if (db.media[size_t(i)].type != 1) continue;
    Media needle("needle", i + 1, uint64_t(db.media[size_t(i)].phash_dct));
    QVector<Index::Match> got = idx.find(needle, p);
    std::vector<std::pair<int, uint32_t>> want;
    for (auto& r : db.media)
      if (r.type == 1 && hamm64(uint64_t(r.phash_dct), needle.dctHash()) < p.dctThresh)
        want.push_back({hamm64(uint64_t(r.phash_dct), needle.dctHash()), r.id});
    std::sort(want.begin(), want.end());
    CHECK(size_t(got.count()) == want.size());
    for (size_t j = 0; j < want.size(); ++j)
      CHECK(got[j].mediaId == want[j].second && got[j].score == want[j].first);
    QVector<Index::Match> one = plain.find(needle, p);
    CHECK(one.count() == got.count());
    for (int j = 0; j < one.count(); ++j) CHECK(one[j].mediaId == got[j].mediaId && one[j].score == got[j].score);
    ++checked;
  }
  CHECK(checked > 100);
  // remove / add (testindexbase.cpp:148-218)
  QVector<int> rm;
  rm.append(10);
  rm.append(7);
  idx.remove(rm);
  Media m10("x", 10, uint64_t(db.media[9].phash_dct));
  for (auto& r : idx.find(m10, p)) CHECK(r.mediaId != 10 && r.mediaId != 7);
  CHECK(idx.count() == n - n / 50);
  MediaGroup g;
  g.append(m10);
  idx.add(g);
  bool self = false;
  for (auto& r : idx.find(m10, p)) self |= (r.mediaId == 10 && r.score == 0);
  CHECK(self);
  // slice
  QSet<uint32_t> want;
  for (uint32_t id = 100; id < 200; ++id) want.insert(id);
  Index* sub = idx.slice(want);
  CHECK(sub && sub->isLoaded() && sub->count() == 98);  // ids 100 and 150 are videos
  CHECK(cbh_idx64_shard_count(static_cast<GpuDctHashIndex*>(sub)->handle()) == cbh_idx64_shard_count(idx.handle()));
  delete sub;
  // batched extension
  MediaGroup needles;
  for (int i = 0; i < 300; ++i) needles.append(Media("n", i + 1, uint64_t(db.media[size_t(i)].phash_dct)));
  auto res = idx.findBatch(needles, p);
  CHECK(res.count() == 300);
  {  // ... and equal to the one-device index's (which saw the same remove / add)
    plain.remove(rm);
    plain.add(g);
    auto res1 = plain.findBatch(needles, p);
    for (int i = 0; i < 300; ++i) {
      CHECK(res1[i].count() == res[i].count());
      for (int j = 0; j < res[i].count(); ++j)
        CHECK(res1[i][j].mediaId == res[i][j].mediaId && res1[i][j].score == res[i][j].score);
    }
  }
  for (int i = 0; i < 300; ++i) {
    if (db.media[size_t(i)].type != 1 || i + 1 == 7) continue;
    CHECK(res[i].count() >= 1 && res[i][0].score == 0);
  }
  // searchIndex for the whole batch: self filtered, cut at maxMatches, unknown ids skipped without taking a place
  {
    SearchParams sp;
    sp.dctThresh = 3;
    sp.maxMatches = 2;
    std::vector<uint32_t> known;
    for (auto& r : db.media)
      if (r.id % 7 != 0) known.push_back(r.id);  // every 7th media is "not in the database" for this caller
    auto sb = idx.searchIndexBatch(needles, sp, &known);
    CHECK(sb.count() == 300);
    for (int i = 0; i < 300; ++i) {
      CHECK(sb[i].count() <= 2);
      int prev = -1;
      for (auto& mt : sb[i]) {
        CHECK(mt.mediaId != uint32_t(i + 1) && mt.mediaId % 7 != 0 && mt.score < 3 && mt.score >= prev);
        prev = mt.score;
      }
    }
  }
  QSet<mediaid_t> ids = idx.mediaIds(db, "", "");
  CHECK(ids.size() == size_t(n - n / 50 - 2 + 1));
  if (!devs.single()) {
    cbh_shard_stats st;
    CHECK(cbh_idx64_shard_stats(idx.handle(), &st) == CBH_OK);
    printf("shards %u devices %u scans %llu rescans %llu collectives %llu local copies %llu peer copies %llu\n", st.shards,
           st.devices, (unsigned long long)st.scans, (unsigned long long)st.rescans, (unsigned long long)st.collectives,
           (unsigned long long)st.local_copies, (unsigned long long)st.peer_copies);
    CHECK(st.scans >= st.shards);
  }
  printf("adapter ok: %d needles checked\n", checked);
  return 0;
}
