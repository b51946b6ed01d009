// gpu_indexes.h -- cbird-side bindings of the other four Index subclasses to libcbird_hip.so:
//   GpuDctFeaturesIndex (src/dctfeaturesindex.{h,cpp}), GpuCvFeaturesIndex (src/cvfeaturesindex.{h,cpp}),
//   GpuColorDescIndex (src/colordescindex.{h,cpp}), GpuDctVideoIndex (src/dctvideoindex.{h,cpp}).
// Same pattern as gpu_dcthashindex.h: the class keeps cbird's SQL/file handling (tables kphash / matrix /
// color, <id>.vdx files) and forwards storage + search to the C-ABI.  Methods that only touch SQL
// (createTables/addRecords/removeRecords) are cbird's own code and are inherited unchanged by deriving from
// the reference class when built inside cbird; this header shows the search-side overrides.
//
// Build note: needs cbird's index.h / media.h (Qt6, OpenCV types KeyPointDescriptors = cv::Mat,
// ColorDescriptor, VideoIndex).  In this repository it is compiled against tests/cpp/mock/ (declarations
// of those classes and types with the reference's signatures) and run on the GPU by
// tests/cpp/test_adapters4.cpp; the scoring rules behind the C-ABI calls are oracle-checked by
// tests/test_fdct.py, tests/test_cvfeatures.py, tests/test_color.py and tests/test_video.py.
#pragma once

#include <algorithm>
#include <cstring>
#include <set>
#include <vector>

#include "cbird_hip.h"
#include "gpu_devices.h"
#include "colordescindex.h"
#include "cvfeaturesindex.h"
#include "dctfeaturesindex.h"
#include "dctvideoindex.h"
#include "videoindex.h"
#include "gpu_errors.h"

// gpu_errors.h (gpuidx::run): a mutation that fails after one retry aborts like the reference's failed allocation; a
// query that fails logs with qCritical and the caller returns its empty result
#define CBH_MUTATE(call) (void)gpuidx::run(gpuidx::Mutation, #call, [&] { return (call); }, this->scratchMask())
#define CBH_QUERY(call) gpuidx::run(gpuidx::Query, #call, [&] { return (call); }, this->scratchMask())

// ---- DctFeaturesIndex ---------------------------------------------------------------------------------
class GpuDctFeaturesIndex : public DctFeaturesIndex {  // inherits createTables/addRecords/removeRecords/mediaIds
 public:
  // treeCompat = false: exact candidates (superset of the reference's).  true: every needle hash only sees its
  // HammingTree leaf (src/tree/hammingtree.h:244-252) -- results equal cbird's also on multi-leaf trees.
  GpuDctFeaturesIndex(int device = 0, bool treeCompat = false)
      : _idx(cbh_idx64_create(device)), _treeCompat(treeCompat) {
    if (!_idx) qFatal("no usable MI355X device");
  }
  // one index over several GPUs / logical shards (GpuDeviceSet, gpu_dcthashindex.h)
  GpuDctFeaturesIndex(const GpuDeviceSet& devs, bool treeCompat = false)
      : _idx(devs.single() ? cbh_idx64_create(devs.first()) : cbh_idx64_create_sharded(devs.mask, devs.shardsPerDevice)),
        _treeCompat(treeCompat) {
    if (!_idx) qFatal("device mask 0x%x names a device that is not a usable MI355X", devs.mask);
  }
  ~GpuDctFeaturesIndex() override { cbh_idx64_destroy(_idx); }
  bool isLoaded() const override { return cbh_idx64_is_loaded(_idx); }
  int count() const override { return int(cbh_idx64_count(_idx)); }
  size_t memoryUsage() const override { return cbh_idx64_memory_usage(_idx); }

  // `select media_id,hashes from kphash` (dctfeaturesindex.cpp:129-156); no cache file is needed
  void load(QSqlDatabase& db, const QString&, const QString&) override {
    QSqlQuery query(db);
    query.setForwardOnly(true);
    if (!query.exec("select media_id,hashes from kphash")) SQL_FATAL(exec);
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> ids;
    while (query.next()) {
      const uint32_t mediaId = query.value(0).toUInt();
      const QByteArray blob = query.value(1).toByteArray();
      if (size_t(blob.size()) % sizeof(uint64_t) != 0) continue;  // "ignoring invalid data" (:145-148)
      const uint64_t* p = reinterpret_cast<const uint64_t*>(blob.constData());
      for (size_t j = 0; j < size_t(blob.size()) / 8; ++j) {
        ids.push_back(mediaId);
        hashes.push_back(p[j]);
      }
    }
    CBH_MUTATE(cbh_idx64_load(_idx, hashes.data(), ids.data(), hashes.size()));
  }
  // slice(): HammingTree::slice keeps the values whose index is in the set (dctfeaturesindex.cpp:239-258); the
  // caller owns the result
  Index* slice(const QSet<uint32_t>& mediaIds) const override {
    std::vector<uint32_t> ids(mediaIds.begin(), mediaIds.end());
    cbh_idx64* sub = cbh_idx64_slice(_idx, ids.data(), ids.size());
    if (!sub && cbh_last_error_code() == CBH_E_NOMEM) {  // transient: give cached scratch back, once more
      gpuidx::releaseScratch(scratchMask());
      sub = cbh_idx64_slice(_idx, ids.data(), ids.size());
    }
    if (!sub) {  // a slice that cannot be made is an empty one (its searches find nothing), not the end of the process
      qCritical("GpuDctFeaturesIndex::slice: %s", cbh_last_error());
      sub = cbh_idx64_create(cbh_idx64_device_mask(_idx) ? __builtin_ctz(cbh_idx64_device_mask(_idx)) : 0);
      if (!sub) qFatal("GpuDctFeaturesIndex::slice: no usable MI355X device");
    }
    return new GpuDctFeaturesIndex(sub, _treeCompat);
  }
  void save(QSqlDatabase&, const QString&) override {}
  void add(const MediaGroup& media) override {
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> ids;
    for (const Media& m : media)
      for (uint64_t h : m.keyPointHashes()) {
        ids.push_back(uint32_t(m.id()));
        hashes.push_back(h);
      }
    if (!hashes.empty()) CBH_MUTATE(cbh_idx64_add(_idx, hashes.data(), ids.data(), hashes.size()));
  }
  void remove(const QVector<int>& ids) override {
    if (ids.count() <= 0 || !isLoaded()) return;
    std::vector<uint32_t> v(ids.begin(), ids.end());
    CBH_MUTATE(cbh_idx64_remove_ids_only(_idx, v.data(), v.size()));
  }
  QVector<Index::Match> find(const Media& needle, const SearchParams& params) override {
    KeyPointHashList hashes = needle.keyPointHashes();
    std::vector<uint64_t> h(hashes.begin(), hashes.end());
    if (h.empty() && needle.id() > 0) {  // _tree->findIndex (:270-276)
      size_t n = 0;
      if (!CBH_QUERY(cbh_idx64_hashes_for_id(_idx, uint32_t(needle.id()), nullptr, 0, &n))) return {};
      h.resize(n);
      if (!CBH_QUERY(cbh_idx64_hashes_for_id(_idx, uint32_t(needle.id()), h.data(), n, &n))) return {};
    }
    if (h.empty()) {
      qWarning() << "needle has no hashes" << needle.id() << needle.path();
      return {};
    }
    std::vector<cbh_match> out(h.size() * 10 + 1);
    size_t n = 0;
    if (!CBH_QUERY(cbh_fdct_find_coalesced(_idx, h.data(), h.size(), uint32_t(needle.id()), params.dctThresh,
                                           _treeCompat ? 1 : 0, out.data(), out.size(), &n)))
      return {};
    QVector<Index::Match> results;
    for (size_t i = 0; i < n; ++i) results.append(Index::Match(out[i].id, out[i].score));
    return results;
  }

  cbh_idx64* handle() const { return _idx; }  // for statistics (cbh_combine_stats)
  uint32_t scratchMask() const { return cbh_idx64_device_mask(_idx); }  // the devices this index lives on

 private:
  GpuDctFeaturesIndex(cbh_idx64* adopted, bool treeCompat) : _idx(adopted), _treeCompat(treeCompat) {}
  cbh_idx64* _idx;
  bool _treeCompat;
};

// ---- CvFeaturesIndex ----------------------------------------------------------------------------------
class GpuCvFeaturesIndex : public CvFeaturesIndex {
 public:
  // one index sharded by image over several GPUs / logical shards (GpuDeviceSet, gpu_devices.h)
  GpuCvFeaturesIndex(const GpuDeviceSet& devs)
      : _device(devs.first()),
        _devs(devs),
        _idx(devs.single() ? cbh_idx256_create(devs.first()) : cbh_idx256_create_sharded(devs.mask, devs.shardsPerDevice)) {
    if (!_idx) qFatal("device mask 0x%x names a device that is not a usable MI355X", devs.mask);
  }
  GpuCvFeaturesIndex(int device = 0) : _device(device), _idx(cbh_idx256_create(device)) {
    if (!_idx) qFatal("no usable MI355X device");
  }
  ~GpuCvFeaturesIndex() override { cbh_idx256_destroy(_idx); }
  bool isLoaded() const override { return cbh_idx256_is_loaded(_idx); }
  int count() const override { return int(cbh_idx256_count(_idx)); }
  size_t memoryUsage() const override { return cbh_idx256_memory_usage(_idx); }
  // load(): the reference's SQL loop over `matrix` (cvfeaturesindex.cpp:184-232) -- rows arrive in ascending
  // media_id, empty ones are skipped, inconsistent ones ignored with a message; no FLANN index, no cache file
  void load(QSqlDatabase& db, const QString&, const QString&) override {
    QSqlQuery query(db);
    query.setForwardOnly(true);
    if (!query.exec("select media_id,rows,cols,type,stride,data from matrix order by media_id")) SQL_FATAL(exec)
    uint32_t lastId = 0;
    while (query.next()) {
      const uint32_t id = query.value(0).toUInt();
      const int rows = query.value(1).toInt(), cols = query.value(2).toInt(), type = query.value(3).toInt(),
                stride = query.value(4).toInt();
      if (rows <= 0) continue;  // "skip empty descriptors as they would violate the requirements on _idMap"
      const QByteArray data = qUncompress(query.value(5).toByteArray());
      if (lastId >= id || type != CV_8UC1 || cols != 32 || stride != cols || size_t(data.size()) != size_t(rows) * 32u) {
        qWarning() << "sql: ignoring invalid data @ media_id=" << id;
        continue;
      }
      CBH_MUTATE(cbh_idx256_add(_idx, id, reinterpret_cast<const uint8_t*>(data.constData()), size_t(rows)));
      lastId = id;
    }
  }
  void save(QSqlDatabase&, const QString&) override {}  // nothing to cache: the rows are the index
  void addOne(uint32_t mediaId, const cv::Mat& desc) {  // desc: rows x 32, CV_8U, continuous
    if (desc.rows > 0) CBH_MUTATE(cbh_idx256_add(_idx, mediaId, desc.ptr<uint8_t>(0), size_t(desc.rows)));
  }
  void add(const MediaGroup& media) override {
    for (const Media& m : media) {
      const KeyPointDescriptors& desc = m.keyPointDescriptors();
      if (desc.rows <= 0) {
        qWarning() << "no descriptors for" << m.path();
        continue;
      }
      addOne(uint32_t(m.id()), desc);
    }
  }
  void remove(const QVector<int>& ids) override {
    std::vector<uint32_t> v(ids.begin(), ids.end());
    CBH_MUTATE(cbh_idx256_remove(_idx, v.data(), v.size()));
  }
  QVector<Index::Match> find(const Media& needle, const SearchParams& params) override {
    cv::Mat descriptors = needle.keyPointDescriptors();
    std::vector<uint8_t> own;
    const uint8_t* rows = descriptors.rows > 0 ? descriptors.ptr<uint8_t>(0) : nullptr;
    size_t n_desc = size_t(std::max(descriptors.rows, 0));
    if (!n_desc) {  // descriptorsForMediaId (:443)
      size_t first = 0, cnt = 0;
      if (!CBH_QUERY(cbh_idx256_rows_of(_idx, uint32_t(needle.id()), &first, &cnt))) return {};
      own.resize(cnt * 32);
      if (cnt && !CBH_QUERY(cbh_idx256_download_rows(_idx, first, cnt, own.data()))) return {};
      rows = own.data();
      n_desc = cnt;
    }
    if (!n_desc) {
      qWarning() << "needle has no descriptors" << needle.id() << needle.path();
      return {};
    }
    std::vector<cbh_match> out(n_desc * 10 + 1);
    size_t n = 0;
    if (!CBH_QUERY(cbh_idx256_find_coalesced(_idx, rows, n_desc, params.cvThresh, 10, out.data(), out.size(), &n)))
      return {};
    QVector<Index::Match> results;
    for (size_t i = 0; i < n; ++i) results.append(Index::Match(out[i].id, out[i].score));
    return results;
  }

  // slice(): the descriptors of the given media in ascending id order, like load() builds them
  // (cvfeaturesindex.cpp:285-312)
  Index* slice(const QSet<uint32_t>& mediaIds) const override {
    GpuCvFeaturesIndex* chunk = _devs.single() ? new GpuCvFeaturesIndex(_device) : new GpuCvFeaturesIndex(_devs);
    std::vector<uint32_t> ids(mediaIds.begin(), mediaIds.end());
    std::sort(ids.begin(), ids.end());
    std::vector<uint8_t> rows;
    for (uint32_t id : ids) {
      size_t first = 0, cnt = 0;
      if (!CBH_QUERY(cbh_idx256_rows_of(_idx, id, &first, &cnt)) || !cnt) continue;
      rows.resize(cnt * 32);
      if (!CBH_QUERY(cbh_idx256_download_rows(_idx, first, cnt, rows.data()))) continue;
      (void)CBH_QUERY(cbh_idx256_add(chunk->_idx, id, rows.data(), cnt));  // (a slice that lacks a media finds less)
    }
    return chunk;
  }

  cbh_idx256* handle() const { return _idx; }  // for statistics (cbh_combine_stats)
  uint32_t scratchMask() const { return _devs.single() ? 1u << _device : _devs.mask; }  // the devices this index lives on

 private:
  int _device = 0;
  GpuDeviceSet _devs;  // (declared before _idx: the sharded constructor initialises it first)
  cbh_idx256* _idx;
};

// ---- ColorDescIndex -----------------------------------------------------------------------------------
static_assert(sizeof(ColorDescriptor) == CBH_COLOR_DESC_BYTES, "ColorDescriptor layout (src/cvutil.h:96-113)");
class GpuColorDescIndex : public ColorDescIndex {
 public:
  GpuColorDescIndex(int device = 0) : _device(device), _idx(cbh_color_create(device)) {
    if (!_idx) qFatal("no usable MI355X device");
  }
  ~GpuColorDescIndex() override { cbh_color_destroy(_idx); }
  bool isLoaded() const override { return cbh_color_is_loaded(_idx); }
  int count() const override { return int(cbh_color_count(_idx)); }
  size_t memoryUsage() const override { return cbh_color_memory_usage(_idx); }
  // load(): `select media_id,color_desc from color` (colordescindex.cpp:125-150); a blob of the wrong size becomes an
  // empty descriptor ("no color desc for id ..., correct by re-indexing")
  void load(QSqlDatabase& db, const QString&, const QString&) override {
    QSqlQuery query(db);
    query.setForwardOnly(true);
    if (!query.exec("select media_id,color_desc from color")) SQL_FATAL(exec)
    std::vector<uint32_t> ids;
    std::vector<ColorDescriptor> descs;
    while (query.next()) {
      ids.push_back(query.value(0).toUInt());
      const QByteArray bytes = query.value(1).toByteArray();
      ColorDescriptor d;
      if (size_t(bytes.size()) == sizeof(ColorDescriptor))
        memcpy(&d, bytes.constData(), sizeof(ColorDescriptor));
      else
        qWarning() << "no color desc for id" << ids.back() << ", correct by re-indexing";
      descs.push_back(d);
    }
    if (!ids.empty()) addRows(ids.data(), descs.data(), ids.size());
  }
  void save(QSqlDatabase&, const QString&) override {}  // "no caching" (colordescindex.cpp:155-159)
  // mediaIds(): the loaded reference reads its private _mediaId array (colordescindex.cpp:161-190), which this class
  // leaves empty -- the ids come from the GPU index instead (removed entries appear as id 0, like there); not
  // loaded: the inherited SQL query
  QSet<mediaid_t> mediaIds(QSqlDatabase& db, const QString& cachePath, const QString& dataPath) const override {
    if (!isLoaded()) return ColorDescIndex::mediaIds(db, cachePath, dataPath);
    const size_t n = size_t(count());
    std::vector<uint32_t> ids(n);
    std::vector<ColorDescriptor> descs(n);
    QSet<mediaid_t> result;
    if (!CBH_QUERY(cbh_color_download(_idx, ids.data(), descs.data(), n))) return result;
    for (uint32_t id : ids) result.insert(id);
    return result;
  }
  void addRows(const uint32_t* ids, const ColorDescriptor* descs, size_t n) {
    CBH_MUTATE(cbh_color_add(_idx, ids, descs, n));
  }
  void add(const MediaGroup& media) override {
    std::vector<uint32_t> ids;
    std::vector<ColorDescriptor> descs;
    for (const Media& m : media) {
      ids.push_back(uint32_t(m.id()));
      descs.push_back(m.colorDescriptor());
    }
    if (!ids.empty()) addRows(ids.data(), descs.data(), ids.size());
  }
  void remove(const QVector<int>& toRemove) override {
    if (!isLoaded()) return;
    std::vector<uint32_t> v(toRemove.begin(), toRemove.end());
    CBH_MUTATE(cbh_color_remove(_idx, v.data(), v.size()));
  }
  bool findIndexData(Media& m) const override {
    ColorDescriptor d;
    if (cbh_color_find_index_data(_idx, uint32_t(m.id()), &d) != 1) return false;
    m.setColorDescriptor(d);
    return true;
  }
  QVector<Index::Match> find(const Media& m, const SearchParams&) override {
    QVector<Index::Match> results;
    ColorDescriptor target = m.colorDescriptor();
    if (target.numColors <= 0) {
      Media tmp = m;
      if (findIndexData(tmp))
        target = tmp.colorDescriptor();
      else
        qWarning() << "needle has no color descriptor" << m.id() << m.path();
      if (target.numColors <= 0) return results;
    }
    std::vector<cbh_match> out(size_t(std::max(count(), 1)));
    size_t n = 0;
    if (!CBH_QUERY(cbh_color_find_coalesced(_idx, &target, out.data(), out.size(), &n))) return results;
    for (size_t i = 0; i < n; ++i) results.append(Index::Match(out[i].id, out[i].score));
    return results;
  }

  // slice(): the descriptors whose media id is in the set, in index order (colordescindex.cpp:231-248)
  Index* slice(const QSet<uint32_t>& mediaIds) const override {
    GpuColorDescIndex* chunk = new GpuColorDescIndex(_device);
    const size_t n = size_t(count());
    std::vector<uint32_t> ids(n), keepIds;
    std::vector<ColorDescriptor> descs(n), keep;
    if (n && !CBH_QUERY(cbh_color_download(_idx, ids.data(), descs.data(), n))) return chunk;
    for (size_t i = 0; i < n; ++i)
      if (mediaIds.contains(ids[i])) {
        keepIds.push_back(ids[i]);
        keep.push_back(descs[i]);
      }
    if (!keepIds.empty()) chunk->addRows(keepIds.data(), keep.data(), keepIds.size());
    return chunk;
  }

  cbh_color* handle() const { return _idx; }  // for statistics (cbh_combine_stats)
  uint32_t scratchMask() const { return 1u << _device; }

 private:
  int _device = 0;
  cbh_color* _idx;
};

// ---- DctVideoIndex ------------------------------------------------------------------------------------
class GpuDctVideoIndex : public DctVideoIndex {
 public:
  // radixCompat = false: exact search (the reference's vradix 0).  true: honour params.videoRadix like the
  // reference's RadixMap does (a needle frame only sees its bucket, src/tree/radix.h:135-141).
  GpuDctVideoIndex(int device = 0, bool radixCompat = false)
      : _device(device), _idx(cbh_vidx_create(device)), _radixCompat(radixCompat) {
    if (!_idx) qFatal("no usable MI355X device");
  }
  // the frame index over several GPUs / logical shards: entries are in video order, so the row shares are by video
  GpuDctVideoIndex(const GpuDeviceSet& devs, bool radixCompat = false)
      : _device(devs.first()),
        _devs(devs),
        _idx(devs.single() ? cbh_vidx_create(devs.first()) : cbh_vidx_create_sharded(devs.mask, devs.shardsPerDevice)),
        _radixCompat(radixCompat) {
    if (!_idx) qFatal("device mask 0x%x names a device that is not a usable MI355X", devs.mask);
  }
  ~GpuDctVideoIndex() override { cbh_vidx_destroy(_idx); }
  bool isLoaded() const override { return _loaded; }
  int count() const override { return int(cbh_vidx_count(_idx)); }
  // memoryUsage() (dctvideoindex.cpp:57-59: `_tree ? _tree->stats().memory : 0`): the inherited one would report the
  // reference's own, never-built, tree -- 0 for ever
  size_t memoryUsage() const override { return cbh_vidx_memory_usage(_idx); }
  void save(QSqlDatabase&, const QString&) override {}  // nothing to cache, like the reference (dctvideoindex.cpp:213-216)
  // load(): `select id from media where type=video order by id` (dctvideoindex.cpp:172-211); each id's
  // <dataPath>/<id>.vdx is read with cbird's own VideoIndex::load and handed over
  void load(QSqlDatabase& db, const QString&, const QString& dataPath) override {
    _dataPath = dataPath;
    QSqlQuery query(db);
    query.setForwardOnly(true);
    if (!query.prepare("select id from media where type=:type order by id")) SQL_FATAL(prepare);
    query.bindValue(":type", Media::TypeVideo);
    if (!query.exec()) SQL_FATAL(exec);
    while (query.next()) addOne(query.value(0).toUInt());
    _loaded = true;
  }
  void add(const MediaGroup& media) override {
    for (auto& m : media) addOne(uint32_t(m.id()));
  }
  void remove(const QVector<int>& ids) override {
    std::vector<uint32_t> v(ids.begin(), ids.end());
    CBH_MUTATE(cbh_vidx_remove(_idx, v.data(), v.size()));
    for (uint32_t id : v) _ids.erase(id);
  }
  // mediaIds(): the loaded reference reads its private _mediaId list (dctvideoindex.cpp:218-231), empty here: this
  // class keeps its own; not loaded: the inherited SQL + file-exists scan
  QSet<mediaid_t> mediaIds(QSqlDatabase& db, const QString& cachePath, const QString& dataPath) const override {
    if (!isLoaded()) return DctVideoIndex::mediaIds(db, cachePath, dataPath);
    QSet<mediaid_t> result;
    for (uint32_t id : _ids) result.insert(id);
    return result;
  }
  QVector<Index::Match> find(const Media& needle, const SearchParams& p) override {
    std::vector<cbh_vmatch> out(size_t(std::max(count(), 1)));
    size_t n = 0;
    if (!CBH_QUERY(cbh_vidx_set_radix(_idx, _radixCompat ? p.videoRadix : 0))) return {};
    if (needle.type() == Media::TypeImage) {
      if (needle.dctHash() == 0) {
        qWarning() << "needle has no dct hash" << needle.id() << needle.path();
        return {};
      }
      if (!CBH_QUERY(cbh_vidx_find_frame(_idx, needle.dctHash(), p.dctThresh, p.skipFrames, needle.matchRange().dstIn,
                                         out.data(), out.size(), &n)))
        return {};
    } else if (needle.type() == Media::TypeVideo) {
      VideoIndex src;
      if (needle.id() == 0)
        src = needle.videoIndex();
      else
        src.load(QString("%1/%2.vdx").arg(_dataPath).arg(needle.id()));
      if (src.isEmpty()) {
        qWarning() << "needle video index is empty:" << needle.path();
        return {};
      }
      if (!CBH_QUERY(cbh_vidx_find_video_coalesced(_idx, src.frames.data(), src.hashes.data(), src.frames.size(),
                                                   uint32_t(needle.id()), p.dctThresh, p.skipFrames, p.minFramesMatched,
                                                   p.minFramesNear, p.filterSelf, out.data(), out.size(), &n)))
        return {};
    }
    QVector<Index::Match> results;
    for (size_t i = 0; i < n; ++i) {
      Index::Match m(out[i].id, out[i].score);
      m.range = MatchRange(out[i].src_in, out[i].dst_in, out[i].len);
      results.append(m);
    }
    return results;
  }

  // slice(): "replicate what load() does, but use the subset" (dctvideoindex.cpp:389-397)
  Index* slice(const QSet<uint32_t>& mediaIds) const override {
    GpuDctVideoIndex* copy = _devs.single() ? new GpuDctVideoIndex(_device, _radixCompat)
                                            : new GpuDctVideoIndex(_devs, _radixCompat);
    copy->_dataPath = _dataPath;
    for (uint32_t id : mediaIds) copy->addOne(id);
    copy->_loaded = true;
    return copy;
  }

  cbh_vidx* handle() const { return _idx; }  // for statistics (cbh_combine_stats)

 private:
  void addOne(uint32_t id) {
    VideoIndex vi;
    const QString path = QString("%1/%2.vdx").arg(_dataPath).arg(id);
    if (QFileInfo(path).exists())
      vi.load(path);
    else
      qWarning() << "index file missing:" << path;
    CBH_MUTATE(cbh_vidx_add_video(_idx, id, vi.frames.data(), vi.hashes.data(), vi.frames.size()));
    _ids.insert(id);
  }
  std::set<uint32_t> _ids;  // what DctVideoIndex::_mediaId holds in the reference
  uint32_t scratchMask() const { return _devs.single() ? 1u << _device : _devs.mask; }
  int _device = 0;
  GpuDeviceSet _devs;  // (declared before _idx: the sharded constructor initialises it first)
  cbh_vidx* _idx;
  bool _radixCompat;
  QString _dataPath;
  bool _loaded = false;
};
