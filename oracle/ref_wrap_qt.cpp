// oracle/ref_wrap_qt.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the REAL reference structures behind DctFeaturesIndex and DctVideoIndex in place from
// /root/reference (nothing copied):
//   src/tree/hammingtree.h:51-564  HammingTree_t<uint32_t>  (DctFeaturesIndex, src/dctfeaturesindex.h:28)
//   src/tree/radix.h:36-231        RadixMap_t<index_t>      (DctVideoIndex)
// They need a handful of Qt names; the build container has a conda Qt 5.9.7 (headers under
// /opt/conda/include/qt), which lacks Q_DISABLE_COPY_MOVE (5.13+) -- the one shim here; the
// three names hammingtree.h/radix.h take from src/global.h:58-66 come from that header itself.
// The index classes themselves (QtSql, Media, OpenCV types) cannot be compiled; the voting logic of
// DctFeaturesIndex::find (src/dctfeaturesindex.cpp:285-358) is restated in ref_fdct_find below on
// top of the real tree's candidates.
#include <QtCore/QByteArray>
#include <QtCore/QDebug>
#include <QtCore/QFile>
#include <QtCore/QList>
#include <QtCore/QString>
#include <cmath>
#include <climits>
#include <cstdint>
#include <functional>
#include <map>
#include <vector>

// Qt 5.9.7 (this image's conda Qt) predates Q_DISABLE_COPY_MOVE (5.13); the only stand-in in this file.  Everything else
// the two headers need -- strict_malloc / strict_realloc / dcthash_t -- comes from the reference's own src/global.h,
// included in place (its Qt6-only `qq` macro is never expanded, so it compiles under Qt5).
#ifndef Q_DISABLE_COPY_MOVE
#define Q_DISABLE_COPY_MOVE(Class) \
  Q_DISABLE_COPY(Class)            \
  Class(Class&&) = delete;         \
  Class& operator=(Class&&) = delete;
#endif

#include "global.h"            // -I/root/reference/src (src/global.h:52-66)
#include "tree/hammingtree.h"  // -I/root/reference/src
#include "tree/radix.h"

typedef HammingTree_t<uint32_t> HammingTree;

// DctVideoIndex's payload type (src/dctvideoindex.h:37-43): 24-bit video index + 24-bit frame
struct VideoTreeIndex {
  uint32_t idx : 24;
  uint32_t frame : 24;
  VideoTreeIndex(int v = 0) : idx(uint32_t(v)), frame(0) {}
  VideoTreeIndex(uint32_t i, uint32_t f) : idx(i), frame(f) {}
} __attribute__((packed));
typedef RadixMap_t<VideoTreeIndex> RadixMap;

extern "C" {

void* ref_htree_create() { return new HammingTree; }
void ref_htree_destroy(void* t) { delete static_cast<HammingTree*>(t); }
size_t ref_htree_size(void* t) { return static_cast<HammingTree*>(t)->size(); }

// DctFeaturesIndex::add / load chunk insert (dctfeaturesindex.cpp:152-156, 229-238)
void ref_htree_insert(void* t, const uint32_t* ids, const uint64_t* hashes, size_t n) {
  std::vector<HammingTree::Value> values;
  values.reserve(n);
  for (size_t i = 0; i < n; ++i) values.push_back(HammingTree::Value(ids[i], hashes[i]));
  static_cast<HammingTree*>(t)->insert(values);
}

// DctFeaturesIndex::remove (dctfeaturesindex.cpp:240-249)
void ref_htree_remove(void* t, const uint32_t* ids, size_t n) {
  std::unordered_set<HammingTree::index_t> set(ids, ids + n);
  static_cast<HammingTree*>(t)->remove(set);
}

// HammingTree::write / read (hammingtree.h:166-200): the `dctfeatures.cache` file of DctFeaturesIndex::save/load
// (dctfeaturesindex.cpp:34,100-181).  Returns 1 on success.
int ref_htree_write(void* t, const char* path) {
  QFile f(QString::fromUtf8(path));
  if (!f.open(QFile::WriteOnly | QFile::Truncate)) return 0;
  static_cast<HammingTree*>(t)->write(f);
  return 1;
}
int ref_htree_read(void* t, const char* path) {
  QFile f(QString::fromUtf8(path));
  if (!f.open(QFile::ReadOnly)) return 0;
  return static_cast<HammingTree*>(t)->read(f) ? 1 : 0;
}

// HammingTree::search (hammingtree.h:103-108): matches sorted by distance (std::sort, unstable)
int ref_htree_search(void* t, uint64_t hash, int threshold, uint32_t* out_idx, uint64_t* out_hash,
                     int32_t* out_dist, int cap) {
  std::vector<HammingTree::Match> m;
  static_cast<HammingTree*>(t)->search(hash, threshold, m);
  for (int i = 0; i < int(m.size()) && i < cap; ++i) {
    out_idx[i] = m[size_t(i)].value.index;
    out_hash[i] = m[size_t(i)].value.hash;
    out_dist[i] = m[size_t(i)].distance;
  }
  return int(m.size());
}

// The voting of DctFeaturesIndex::find restated line by line (dctfeaturesindex.cpp:285-358) over the
// real tree.  Returns the number of results (ascending mediaId, QMap key order).
int ref_fdct_find(void* t, const uint64_t* nHash, int numNeedleHashes, int needleId, int dctThresh,
                  uint32_t* out_ids, int32_t* out_scores, int cap) {
  HammingTree* tree = static_cast<HammingTree*>(t);
  std::vector<std::vector<HammingTree::Match>> cand((size_t)numNeedleHashes);
  for (int j = 0; j < numNeedleHashes; j++) tree->search(nHash[j], dctThresh, cand[size_t(j)]);
  std::map<uint32_t, uint32_t> matches;
  std::map<uint32_t, int> scores;
  uint32_t maxMatches = 0;
  for (int j = 0; j < numNeedleHashes; j++) {
    int len = std::min(10, (int)cand[size_t(j)].size());
    for (int k = 0; k < len; k++) {
      const HammingTree::Match& match = cand[size_t(j)][size_t(k)];
      int index = int(match.value.index);
      if (index <= 0) continue;
      int mediaId = index;
      if (matches.count(uint32_t(mediaId))) {
        matches[uint32_t(mediaId)]++;
        scores[uint32_t(mediaId)] += match.distance;
      } else {
        matches[uint32_t(mediaId)] = 1;
        scores[uint32_t(mediaId)] = match.distance;
      }
      if (needleId != mediaId) maxMatches = std::max(matches[uint32_t(mediaId)], maxMatches);
    }
  }
  int n = 0;
  for (auto& kv : matches)
    if (kv.second > 0) {
      uint32_t mediaId = kv.first;
      int score = 0;
      float avgScore = (float)scores[mediaId] / kv.second;
      if (mediaId == uint32_t(needleId))
        score = -1;
      else if (maxMatches == 1)
        score = 10 * avgScore;
      else
        score = int(maxMatches - kv.second);
      if (n < cap) {
        out_ids[n] = mediaId;
        out_scores[n] = score;
      }
      ++n;
    }
  return n;
}

// ---- RadixMap --------------------------------------------------------------------------------------
void* ref_radix_create(unsigned radix) { return new RadixMap(radix); }
void ref_radix_destroy(void* r) { delete static_cast<RadixMap*>(r); }
void ref_radix_insert(void* r, const uint32_t* vidx, const uint32_t* frame, const uint64_t* hashes,
                      size_t n) {
  std::vector<RadixMap::Value> values;
  values.reserve(n);
  for (size_t i = 0; i < n; ++i) values.push_back(RadixMap::Value(VideoTreeIndex(vidx[i], frame[i]), hashes[i]));
  static_cast<RadixMap*>(r)->insert(values);
}
// RadixMap::search (radix.h:187-210): one bucket, matches in bucket order
int ref_radix_search(void* r, uint64_t hash, int threshold, uint32_t* out_vidx, uint32_t* out_frame,
                     uint64_t* out_hash, int32_t* out_dist, int cap) {
  std::vector<RadixMap::Match> m;
  static_cast<RadixMap*>(r)->search(hash, (RadixMap::distance_t)threshold, m);
  for (int i = 0; i < int(m.size()) && i < cap; ++i) {
    out_vidx[i] = m[size_t(i)].value.index.idx;
    out_frame[i] = m[size_t(i)].value.index.frame;
    out_hash[i] = m[size_t(i)].value.hash;
    out_dist[i] = m[size_t(i)].distance;
  }
  return int(m.size());
}
size_t ref_radix_index_of(void* r, uint64_t hash) { return static_cast<RadixMap*>(r)->indexOf(hash); }

}  // extern "C"
