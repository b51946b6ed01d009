// oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C-ABI wrapper that compiles the REAL reference search structure in place from
// /root/reference (nothing is copied into this repo):
//   src/hamm.h:24-26         hamm64
//   src/tree/vptree.h:36-273 VpTree (exact threshold search used by DctHashIndex::find)
// The 20-line DctTree adaptor (src/tree/dcttree.h:103-138) cannot be included because
// it drags index.h -> media.h -> OpenCV/Qt6; its vpValue/vpDistance glue is restated
// below with the same field layout and the same min()/max() sentinels.
//
// Built by oracle/Makefile into oracle/_ref/libcbird_ref.so (git-ignored, travels to
// the GPU box with the snapshot).  Used (a) to pin the C restatement in
// oracle/cbird_oracle.c, (b) to generate tests/golden/*.json, (c) as the
// cpu_baseline{"kind":"reference"} leg of bench.py.
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <thread>
#include <utility>
#include <vector>

// Q_ASSERT / qInfo (vptree.h:74,121-234) are Qt's own when Qt headers exist (oracle/Makefile passes -DCBIRD_REF_QT and
// links Qt5Core: this image's conda Qt 5.9.7); only a build without any Qt falls back to the two stand-ins below, which
// touch no arithmetic.
#ifdef CBIRD_REF_QT
#include <QtCore/QtGlobal>
#else
#define Q_ASSERT(x) assert(x)
#define qInfo printf
#endif

#include "hamm.h"         // -I/root/reference/src
#include "tree/vptree.h"  // -I/root/reference/src

namespace {

// glue equivalent to DctTree::vpValue / vpDistance (dcttree.h:104-112)
struct vpValue {
  uint64_t hash;
  uint32_t id;
  vpValue() : hash(0), id(0) {}
  vpValue(uint64_t h, uint32_t i) : hash(h), id(i) {}
  static vpValue min() { return vpValue(0, 0); }
  static vpValue max() { return vpValue(UINT64_MAX, 0); }
};
inline int vpDistance(vpValue a, vpValue b) { return hamm64(a.hash, b.hash); }

struct RefTree {
  VpTree<vpValue, int, vpDistance> tree;
  size_t n = 0;
};

}  // namespace

extern "C" {

int ref_hamm64(uint64_t a, uint64_t b) { return hamm64(a, b); }

// DctTree::create (dcttree.h:117-122)
void* ref_dcttree_create(const uint64_t* hashes, const uint32_t* ids, int n) {
  if (n <= 0) return nullptr;  // DctHashIndex::buildTree only builds when _numHashes > 0
  auto* t = new RefTree;
  std::vector<vpValue> values;
  values.reserve(size_t(n));
  for (int i = 0; i < n; ++i) values.push_back(vpValue(hashes[i], ids[i]));
  t->tree.create(values);
  t->n = size_t(n);
  return t;
}

void ref_dcttree_destroy(void* t) { delete static_cast<RefTree*>(t); }

// DctTree::search (dcttree.h:124-137): results ascending by distance, ties in heap order.
// Returns the full match count; writes at most `cap` entries.
int ref_dcttree_search(void* tp, uint64_t target, int threshold, uint32_t* out_ids,
                       int32_t* out_dist, int cap) {
  auto* t = static_cast<RefTree*>(tp);
  std::vector<int> distances;
  std::vector<vpValue> results;
  t->tree.search(vpValue{target, 0}, threshold, &results, &distances);
  int n = int(results.size());
  for (int i = 0; i < n && i < cap; ++i) {
    out_ids[i] = results[size_t(i)].id;
    out_dist[i] = distances[size_t(i)];
  }
  return n;
}

// All-needles driver shaped like Database::similar's QtConcurrent::map fan-out
// (database.cpp:1400-1432): one needle per task over `threads` workers, each calling
// the tree search.  Returns the total match count (sum over needles); per-needle
// counts go to `counts` when non-null.  Used for CPU-baseline timing.
long long ref_dcttree_search_many(void* tp, const uint64_t* needles, int nq, int threshold,
                                  int threads, uint32_t* counts) {
  auto* t = static_cast<RefTree*>(tp);
  if (threads < 1) threads = 1;
  std::atomic<int> next(0);
  std::atomic<long long> total(0);
  auto work = [&]() {
    std::vector<int> distances;
    std::vector<vpValue> results;
    long long local = 0;
    for (;;) {
      int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nq) break;
      if (needles[i] == 0) {  // DctHashIndex::find: null needle hash -> empty (dcthashindex.cpp:196-200)
        if (counts) counts[i] = 0;
        continue;
      }
      t->tree.search(vpValue{needles[i], 0}, threshold, &results, &distances);
      if (counts) counts[i] = uint32_t(results.size());
      local += (long long)results.size();
    }
    total += local;
  };
  std::vector<std::thread> pool;
  for (int k = 1; k < threads; ++k) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  return total.load();
}

// Full result lists of every needle, for identity checks at full size (bench.py cpu_baseline, tests/test_full_size.py):
// the same fan-out as ref_dcttree_search_many, but every needle's (mediaId, distance) list is kept, put in the
// canonical order (distance ascending, then mediaId ascending -- the tree's own order among equal distances is its heap
// order, which no other structure reproduces) and handed back in CSR form.
struct RefLists {
  std::vector<uint64_t> offsets;  // nq + 1
  std::vector<uint32_t> ids;
  std::vector<int32_t> dists;
};

void* ref_dcttree_search_lists(void* tp, const uint64_t* needles, int nq, int threshold, int threads) {
  auto* t = static_cast<RefTree*>(tp);
  if (threads < 1) threads = 1;
  struct Part {
    std::vector<uint64_t> keys;                   // dist << 32 | id, per needle sorted
    std::vector<std::pair<int, uint32_t>> spans;  // (needle, count), in the order this worker met them
  };
  std::vector<Part> parts{size_t(threads)};
  std::atomic<int> next(0);
  auto work = [&](int w) {
    std::vector<int> distances;
    std::vector<vpValue> results;
    Part& p = parts[size_t(w)];
    for (;;) {
      int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nq) break;
      if (needles[i] == 0 || !t) {  // dcthashindex.cpp:196-200
        p.spans.emplace_back(i, 0u);
        continue;
      }
      t->tree.search(vpValue{needles[i], 0}, threshold, &results, &distances);
      size_t at = p.keys.size();
      for (size_t k = 0; k < results.size(); ++k)
        p.keys.push_back(uint64_t(uint32_t(distances[k])) << 32 | results[k].id);
      std::sort(p.keys.begin() + long(at), p.keys.end());
      p.spans.emplace_back(i, uint32_t(results.size()));
    }
  };
  std::vector<std::thread> pool;
  for (int k = 1; k < threads; ++k) pool.emplace_back(work, k);
  work(0);
  for (auto& th : pool) th.join();
  auto* out = new RefLists;
  out->offsets.assign(size_t(nq) + 1, 0);
  for (auto& p : parts)
    for (auto& s : p.spans) out->offsets[size_t(s.first) + 1] = s.second;
  for (int i = 0; i < nq; ++i) out->offsets[size_t(i) + 1] += out->offsets[size_t(i)];
  out->ids.resize(out->offsets[size_t(nq)]);
  out->dists.resize(out->offsets[size_t(nq)]);
  for (auto& p : parts) {
    size_t at = 0;
    for (auto& s : p.spans) {
      size_t o = out->offsets[size_t(s.first)];
      for (uint32_t k = 0; k < s.second; ++k, ++at) {
        out->ids[o + k] = uint32_t(p.keys[at]);
        out->dists[o + k] = int32_t(p.keys[at] >> 32);
      }
    }
  }
  return out;
}

unsigned long long ref_lists_total(void* lp) { return static_cast<RefLists*>(lp)->ids.size(); }

void ref_lists_copy(void* lp, uint64_t* offsets, uint32_t* ids, int32_t* dists) {
  auto* l = static_cast<RefLists*>(lp);
  std::copy(l->offsets.begin(), l->offsets.end(), offsets);
  std::copy(l->ids.begin(), l->ids.end(), ids);
  std::copy(l->dists.begin(), l->dists.end(), dists);
}

void ref_lists_free(void* lp) { delete static_cast<RefLists*>(lp); }

}  // extern "C"
