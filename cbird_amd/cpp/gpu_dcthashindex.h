// gpu_dcthashindex.h -- cbird-side binding of the MI355X DctHashIndex (drop-in for
// src/dcthashindex.{h,cpp}).
//
// This is the file a cbird maintainer adds to src/ (see INTEGRATION.md): a subclass of the
// reference's `Index` plugin interface (src/index.h:150-281) with exactly the overrides of
// DctHashIndex (src/dcthashindex.h:29-67).  All array work -- storage, add/remove, find, slice --
// goes through the C-ABI of libcbird_hip.so (include/cbird_hip.h); the SQL of load()/mediaIds()
// stays here, verbatim in meaning (`select id,phash_dct from media where type=1`,
// src/dcthashindex.cpp:82-89).
//
// It compiles against the real cbird headers ("index.h", Qt6) and, for this repository's tests
// where Qt6/OpenCV are not installed, against tests/cpp/mock/index.h which declares the same
// names with the same signatures.
#pragma once

#include <algorithm>
#include <string>
#include <vector>

#include "cbird_hip.h"
#include "gpu_devices.h"
#include "index.h"  // cbird's src/index.h (or the test mock)
#include "gpu_errors.h"

class GpuDctHashIndex : public Index {
  Q_DISABLE_COPY_MOVE(GpuDctHashIndex)

 public:
  explicit GpuDctHashIndex(int device = 0) : _device(device) {
    _id = SearchParams::AlgoDCT;  // dcthashindex.cpp:31
    _idx = cbh_idx64_create(device);
    if (!_idx) qFatal("GpuDctHashIndex: no usable MI355X (gfx950) device %d", device);
  }
  /// one index over several GPUs: `new GpuDctHashIndex(GpuDeviceSet::all())` in Engine::Engine
  explicit GpuDctHashIndex(const GpuDeviceSet& devs) : _device(devs.first()) {
    _id = SearchParams::AlgoDCT;
    _idx = devs.single() ? cbh_idx64_create(_device) : cbh_idx64_create_sharded(devs.mask, devs.shardsPerDevice);
    if (!_idx) qFatal("GpuDctHashIndex: device mask 0x%x names a device that is not a usable MI355X", devs.mask);
  }
  ~GpuDctHashIndex() override { cbh_idx64_destroy(_idx); }

  bool isLoaded() const override { return cbh_idx64_is_loaded(_idx) != 0; }
  int count() const override { return int(cbh_idx64_count(_idx)); }
  size_t memoryUsage() const override { return cbh_idx64_memory_usage(_idx); }

  // DctHashIndex::load (dcthashindex.cpp:70-114): hashes always come from the database
  void load(QSqlDatabase& db, const QString& cachePath, const QString& dataPath) override {
    (void)cachePath;
    (void)dataPath;
    if (isLoaded()) return;
    QSqlQuery query(db);
    query.setForwardOnly(true);
    if (!query.exec("select id,phash_dct from media where type=1")) SQL_FATAL(exec);
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> ids;
    while (query.next()) {
      ids.push_back(query.value(0).toUInt());
      hashes.push_back(uint64_t(query.value(1).toLongLong()));
    }
    mutate("load", [&] { return cbh_idx64_load(_idx, hashes.data(), ids.data(), hashes.size()); });
  }

  void save(QSqlDatabase& db, const QString& cachePath) override {
    (void)db;
    (void)cachePath;  // nothing to cache: the SoA is rebuilt from SQL (dcthashindex.cpp:116-120)
  }

  QSet<mediaid_t> mediaIds(QSqlDatabase& db, const QString& cachePath,
                           const QString& dataPath) const override {
    (void)cachePath;
    (void)dataPath;
    QSet<mediaid_t> set;
    if (isLoaded()) {  // dcthashindex.cpp:129-133
      size_t n = 0;
      if (!query("mediaIds", [&] { return cbh_idx64_media_ids(_idx, nullptr, 0, &n); })) return set;
      std::vector<uint32_t> ids(n ? n : 1);
      if (!query("mediaIds", [&] { return cbh_idx64_media_ids(_idx, ids.data(), ids.size(), &n); })) return set;
      for (size_t i = 0; i < n; ++i) set.insert(ids[i]);
      return set;
    }
    QSqlQuery query(db);  // dcthashindex.cpp:136-153
    query.setForwardOnly(true);
    if (!query.exec("select id,phash_dct from media where type=1")) SQL_FATAL(exec);
    while (query.next()) {
      mediaid_t id = query.value(0).toUInt();
      uint64_t hash = uint64_t(query.value(1).toLongLong());
      if (hash != 0) set.insert(id);
    }
    return set;
  }

  // dcthashindex.cpp:158-173 (the tree rebuild disappears: the scan needs no tree)
  void add(const MediaGroup& media) override {
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> ids;
    for (const Media& m : media) {
      hashes.push_back(m.dctHash());
      ids.push_back(uint32_t(m.id()));
    }
    mutate("add", [&] { return cbh_idx64_add(_idx, hashes.data(), ids.data(), hashes.size()); });
  }

  // dcthashindex.cpp:175-191
  void remove(const QVector<int>& removed) override {
    if (!isLoaded()) return;
    std::vector<uint32_t> ids;
    for (int id : removed) ids.push_back(uint32_t(id));
    mutate("remove", [&] { return cbh_idx64_remove(_idx, ids.data(), ids.size()); });
  }

  // dcthashindex.cpp:193-220.  Thread-safe for concurrent callers (QtConcurrent workers under
  // Database's read lock, database.cpp:1400-1432,1698).
  QVector<Index::Match> find(const Media& m, const SearchParams& p) override {
    QVector<Index::Match> results;
    uint64_t target = m.dctHash();
    if (!target) {
      qWarning() << "no hash for needle:" << m.path();
      return results;
    }
    if (count() <= 0) {
      qWarning() << "empty/null tree";
      return results;
    }
    // cbh_idx64_find_coalesced: same results as cbh_idx64_find; concurrent QtConcurrent workers share one scan per
    // round trip, and an all-pairs run (Database::similar) ends up served from one whole-index self-join
    std::vector<cbh_match> buf(64);
    size_t n = 0;
    for (;;) {
      if (!query("find", [&] { return cbh_idx64_find_coalesced(_idx, target, p.dctThresh, buf.data(), buf.size(), &n); }))
        return results;
      if (n <= buf.size()) break;
      buf.resize(n);
    }
    for (size_t i = 0; i < n; ++i) results.append(Index::Match(buf[i].id, buf[i].score));
    return results;
  }

  // dcthashindex.cpp:222-250; caller deletes (database.cpp:1435,1491)
  Index* slice(const QSet<uint32_t>& mediaIds) const override {
    Q_ASSERT(isLoaded());
    std::vector<uint32_t> ids;
    for (uint32_t id : mediaIds) ids.push_back(id);
    cbh_idx64* sub = cbh_idx64_slice(_idx, ids.data(), ids.size());
    if (!sub && cbh_last_error_code() == CBH_E_NOMEM) {  // transient: give cached scratch back, once more
      gpuidx::releaseScratch(cbh_idx64_device_mask(_idx));
      sub = cbh_idx64_slice(_idx, ids.data(), ids.size());
    }
    if (!sub) {  // a slice that cannot be made is an empty one (its searches find nothing), not the end of the process
      qCritical("GpuDctHashIndex::slice: %s", cbh_last_error());
      return new GpuDctHashIndex(_device);
    }
    return new GpuDctHashIndex(_device, sub);
  }

  /// MI355X-native extension used by a batched Database::similar: every needle of `needles` in
  /// one launch; results[i] = matches of needles[i] already sorted by (score, id) and cut at
  /// maxMatches+1 (room for the self match that searchIndex filters, database.cpp:1733-1735).
  QVector<QVector<Index::Match>> findBatch(const MediaGroup& needles, const SearchParams& p) {
    const int k = p.maxMatches + 1;
    std::vector<uint64_t> q;
    for (const Media& m : needles) q.push_back(m.dctHash());
    std::vector<cbh_match> out(q.size() * size_t(k));
    std::vector<uint32_t> counts(q.size());
    QVector<QVector<Index::Match>> res;
    if (!query("find_batch",
               [&] { return cbh_idx64_find_batch(_idx, q.data(), q.size(), p.dctThresh, k, out.data(), counts.data()); })) {
      for (size_t i = 0; i < q.size(); ++i) res.append(QVector<Index::Match>());
      return res;
    }
    for (size_t i = 0; i < q.size(); ++i) {
      QVector<Index::Match> r;
      const size_t m = std::min<size_t>(counts[i], size_t(k));
      for (size_t j = 0; j < m; ++j) r.append(Index::Match(out[i * k + j].id, out[i * k + j].score));
      res.append(r);
    }
    return res;
  }

  /// Database::searchIndex (src/database.cpp:1691-1757) for every needle in one call: the maxThresh escalation
  /// (only needles still at <= minMatches are searched again), (score, mediaId) order, filterSelf, the maxMatches cut,
  /// and ids that are not in `knownIds` (the caller's idMap keys, ascending; nullptr = all known) skipped WITHOUT
  /// consuming a place -- unlike findBatch above, whose fixed cut at maxMatches + 1 is only right when every index
  /// entry is a known media.  result[i] = the matches of needles[i] as searchIndex would append them to its group.
  QVector<QVector<Index::Match>> searchIndexBatch(const MediaGroup& needles, const SearchParams& p,
                                                  const std::vector<uint32_t>* knownIds = nullptr) {
    std::vector<uint64_t> q;
    std::vector<uint32_t> ids;
    for (const Media& m : needles) {
      q.push_back(m.dctHash());
      ids.push_back(uint32_t(m.id()));
    }
    const size_t k = size_t(p.maxMatches);
    std::vector<cbh_match> out(q.size() * std::max<size_t>(k, 1));
    std::vector<uint32_t> counts(q.size());
    QVector<QVector<Index::Match>> res;
    if (!query("search_index_batch", [&] {
          return cbh_search_index_batch(_idx, q.data(), ids.data(), q.size(), p.dctThresh, p.maxThresh, p.minMatches,
                                        p.maxMatches, p.filterSelf ? 1 : 0, knownIds ? knownIds->data() : nullptr,
                                        knownIds ? knownIds->size() : 0, out.data(), counts.data());
        })) {
      for (size_t i = 0; i < q.size(); ++i) res.append(QVector<Index::Match>());
      return res;
    }
    for (size_t i = 0; i < q.size(); ++i) {
      QVector<Index::Match> r;
      for (size_t j = 0; j < counts[i]; ++j) r.append(Index::Match(out[i * k + j].id, out[i * k + j].score));
      res.append(r);
    }
    return res;
  }

  cbh_idx64* handle() const { return _idx; }  // for statistics (cbh_idx64_get_stats / cbh_idx64_coalesce_stats)

 private:
  GpuDctHashIndex(int device, cbh_idx64* adopted) : _device(device), _idx(adopted) {
    _id = SearchParams::AlgoDCT;
  }
  // the reference has no error codes on this surface (gpu_errors.h, gpuidx::run): load/add/remove retry once after
  // releasing cached scratch and then abort like the reference's failed allocation; find & co. log and return nothing
  template <class Call>
  void mutate(const char* what, Call&& call) const {
    (void)gpuidx::run(gpuidx::Mutation, (std::string("GpuDctHashIndex::") + what).c_str(), call, cbh_idx64_device_mask(_idx));
  }
  template <class Call>
  bool query(const char* what, Call&& call) const {
    return gpuidx::run(gpuidx::Query, (std::string("GpuDctHashIndex::") + what).c_str(), call, cbh_idx64_device_mask(_idx));
  }
  int _device;
  cbh_idx64* _idx;
};
