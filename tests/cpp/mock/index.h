// mock/index.h -- TEST SCAFFOLD ONLY: the names gpu_dcthashindex.h needs from cbird's
// src/index.h, src/media.h, src/global.h and Qt6 (QtCore/QtSql), re-declared with the same
// signatures so the adapter can be compiled and exercised in a container without Qt6/OpenCV.
// Nothing here is product code; with real cbird this directory is not on the include path.
#pragma once
#include <cassert>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include <zlib.h>

#include "cbird_hip.h"

#define Q_DISABLE_COPY_MOVE(C) \
  C(const C&) = delete;        \
  C& operator=(const C&) = delete;
#define Q_ASSERT(x) assert(x)
#define Q_UNUSED(x) (void)x
typedef uint32_t mediaid_t;

[[noreturn]] inline void qFatal(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  abort();
}
struct QString : std::string {
  using std::string::string;
  QString() {}
  QString(const std::string& s) : std::string(s) {}
  QString arg(const std::string& v) const {  // replaces the lowest-numbered %N
    QString s(*this);
    for (int n = 1; n < 10; ++n) {
      const std::string key = "%" + std::to_string(n);
      const size_t p = s.find(key);
      if (p != std::string::npos) {
        s.replace(p, key.size(), v);
        break;
      }
    }
    return s;
  }
  QString arg(unsigned v) const { return arg(std::to_string(v)); }
  QString arg(int v) const { return arg(std::to_string(v)); }
};
struct QByteArray : std::string {
  using std::string::string;
  const char* constData() const { return data(); }
};
struct QFileInfo {
  std::string p;
  explicit QFileInfo(const QString& s) : p(s) {}
  bool exists() const { return std::ifstream(p).good(); }
};
struct QDebugMock {
  template <typename T>
  QDebugMock& operator<<(const T& v) {
    std::cerr << v << ' ';
    return *this;
  }
  ~QDebugMock() { std::cerr << '\n'; }
};
inline QDebugMock qWarning() { return QDebugMock(); }
// the printf-style overloads Qt also has (qWarning("..."), qDebug("..."))
template <class... A>
inline void qWarning(const char* fmt, A... a) {
  if constexpr (sizeof...(A) == 0) std::cerr << fmt << '\n';
  else { fprintf(stderr, fmt, a...); fputc('\n', stderr); }
}
inline int g_mockCriticals = 0;  // how many qCritical lines were logged (tests/cpp/test_errors.cpp)
template <class... A>
inline void qCritical(const char* fmt, A... a) {
  ++g_mockCriticals;
  if constexpr (sizeof...(A) == 0) std::cerr << fmt << '\n';
  else { fprintf(stderr, fmt, a...); fputc('\n', stderr); }
}
template <class... A>
inline void qDebug(const char* fmt, A... a) {
  if constexpr (sizeof...(A) == 0) std::cerr << fmt << '\n';
  else { fprintf(stderr, fmt, a...); fputc('\n', stderr); }
}
inline const char* qPrintable(const QString& s) { return s.c_str(); }

// Qt's qCompress / qUncompress: 4-byte big-endian uncompressed length + a zlib stream
inline QByteArray qCompress(const QByteArray& in) {
  uLongf cap = compressBound(uLong(in.size()));
  std::string out(4 + cap, '\0');
  const uint32_t n = uint32_t(in.size());
  out[0] = char(n >> 24), out[1] = char(n >> 16), out[2] = char(n >> 8), out[3] = char(n);
  compress(reinterpret_cast<Bytef*>(&out[4]), &cap, reinterpret_cast<const Bytef*>(in.data()), uLong(in.size()));
  out.resize(4 + cap);
  QByteArray r;
  r.assign(out);
  return r;
}
inline QByteArray qUncompress(const QByteArray& in) {
  QByteArray r;
  if (in.size() < 4) return r;
  const unsigned char* p = reinterpret_cast<const unsigned char*>(in.data());
  uLongf n = (uLongf(p[0]) << 24) | (uLongf(p[1]) << 16) | (uLongf(p[2]) << 8) | uLongf(p[3]);
  std::string out(n, '\0');
  if (uncompress(reinterpret_cast<Bytef*>(&out[0]), &n, p + 4, uLong(in.size() - 4)) != Z_OK) return r;
  out.resize(n);
  r.assign(out);
  return r;
}

template <typename T>
struct QVector : std::vector<T> {
  using std::vector<T>::vector;
  void append(const T& v) { this->push_back(v); }
  int count() const { return int(this->size()); }
};
template <typename T>
struct QSet : std::set<T> {
  bool contains(const T& v) const { return this->count(v) != 0; }
};

// ---- QtSql: in-memory `media` and `kphash` tables ----------------------------------------------------
struct QSqlDatabase {
  struct Row {
    uint32_t id;
    int type;
    int64_t phash_dct;
  };
  std::vector<Row> media;
  std::vector<std::pair<uint32_t, QByteArray>> kphash;  // (media_id, hashes blob)
  struct MatrixRow {                                     // table `matrix` (cvfeaturesindex.cpp:54-65)
    uint32_t media_id;
    int rows, cols, type, stride;
    QByteArray data;  // qCompress'd row bytes
  };
  std::vector<MatrixRow> matrix;                         // must be kept in ascending media_id ("order by media_id")
  std::vector<std::pair<uint32_t, QByteArray>> color;   // table `color` (media_id, color_desc blob)
};
struct QVariant {
  int64_t v = 0;
  QByteArray b;
  unsigned toUInt() const { return unsigned(v); }
  int toInt() const { return int(v); }
  long long toLongLong() const { return v; }
  unsigned long long toULongLong() const { return (unsigned long long)v; }
  QByteArray toByteArray() const { return b; }
};
struct QSqlError {
  QString text() const { return "mock"; }
};
struct QSqlQuery {
  QSqlDatabase& db;
  long pos = -1;
  int stmt = 0;  // 1: id,phash_dct of images; 2: kphash rows; 3: ids of one media type; 4: matrix rows; 5: color rows
  int bound_type = 0;
  explicit QSqlQuery(QSqlDatabase& d) : db(d) {}
  void setForwardOnly(bool) {}
  bool prepare(const char* sql) {
    stmt = std::string(sql) == "select id from media where type=:type order by id" ? 3 : 0;
    return stmt != 0;
  }
  void bindValue(const char*, int v) { bound_type = v; }
  bool exec() {
    pos = -1;
    return stmt == 3;
  }
  bool exec(const char* sql) {
    const std::string q(sql);
    stmt = q == "select id,phash_dct from media where type=1" ? 1
           : q == "select media_id,hashes from kphash"       ? 2
           : q == "select media_id,rows,cols,type,stride,data from matrix order by media_id" ? 4
           : q == "select media_id,color_desc from color"    ? 5
                                                              : 0;
    pos = -1;
    return stmt != 0;
  }
  bool next() {
    if (stmt == 2) return ++pos < long(db.kphash.size());
    if (stmt == 4) return ++pos < long(db.matrix.size());
    if (stmt == 5) return ++pos < long(db.color.size());
    const int want = stmt == 1 ? 1 : bound_type;
    while (++pos < long(db.media.size()))
      if (db.media[size_t(pos)].type == want) return true;
    return false;
  }
  QVariant value(int col) const {
    QVariant v;
    if (stmt == 2) {
      if (col == 0) v.v = db.kphash[size_t(pos)].first;
      else v.b = db.kphash[size_t(pos)].second;
      return v;
    }
    if (stmt == 4) {
      const auto& m = db.matrix[size_t(pos)];
      const int64_t f[5] = {int64_t(m.media_id), m.rows, m.cols, m.type, m.stride};
      if (col < 5) v.v = f[col];
      else v.b = m.data;
      return v;
    }
    if (stmt == 5) {
      if (col == 0) v.v = db.color[size_t(pos)].first;
      else v.b = db.color[size_t(pos)].second;
      return v;
    }
    const auto& r = db.media[size_t(pos)];
    v.v = col == 0 ? int64_t(r.id) : r.phash_dct;
    return v;
  }
  QSqlError lastError() const { return {}; }
};
#define SQL_FATAL(x) qFatal("QSqlQuery." #x ": %s", qPrintable(query.lastError().text()));

// ---- value types the other indexes carry (shapes only) ----------------------------------------------
typedef uint64_t dcthash_t;
typedef std::vector<uint64_t> KeyPointHashList;
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_8UC4 24
namespace cv {
struct Size {
  int width = 0, height = 0;
};
struct Point {
  int x = 0, y = 0;
};
struct Point2f {
  float x = 0, y = 0;
};
struct KeyPoint {  // opencv2/features2d (2.4): pt, size, angle, response, octave, class_id
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : size(s), angle(a), response(r), octave(o), class_id(c) {
    pt.x = x, pt.y = y;
  }
};
struct Mat {  // rows x cols bytes (CV_8UC1); views share the parent's storage like cv::Mat
  int rows = 0, cols = 0;
  size_t step = 0;
  uint8_t* data = nullptr;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), step(size_t(c)), _store(new std::vector<uint8_t>(size_t(r) * size_t(c))) {
    data = _store->data();
    _whole.width = c, _whole.height = r;
  }
  Mat(int r, int c, int type) : rows(r), cols(c), _cn((type >> 3) + 1) {  // CV_8UC1 / CV_8UC3 / CV_8UC4
    step = size_t(c) * size_t(_cn);
    _store.reset(new std::vector<uint8_t>(size_t(r) * step));
    data = _store->data();
    _whole.width = c, _whole.height = r;
  }
  template <typename T>
  const T* ptr(int r) const { return reinterpret_cast<const T*>(data + size_t(r) * step); }
  template <typename T>
  T* ptr(int r) { return reinterpret_cast<T*>(data + size_t(r) * step); }
  int type() const { return (_cn - 1) << 3; }
  int channels() const { return _cn; }
  int depth() const { return 0; }  // CV_8U
  Mat colRange(int x0, int x1) const {
    Mat m(*this);
    m.data += size_t(x0) * size_t(_cn), m.cols = x1 - x0, m._ofs.x += x0;
    return m;
  }
  Mat rowRange(int y0, int y1) const {
    Mat m(*this);
    m.data += size_t(y0) * step, m.rows = y1 - y0, m._ofs.y += y0;
    return m;
  }
  void locateROI(Size& wholeSize, Point& ofs) const { wholeSize = _whole, ofs = _ofs; }

 private:
  int _cn = 1;
  std::shared_ptr<std::vector<uint8_t>> _store;
  Size _whole;
  Point _ofs;
};
}  // namespace cv
typedef std::vector<cv::KeyPoint> KeyPointList;
typedef cv::Mat KeyPointDescriptors;
struct DescriptorColor {  // src/cvutil.h:75-97
  uint16_t l, u, v, w;
};
struct ColorDescriptor {  // src/cvutil.h:102-113
  enum { NUM_DESC_COLORS = 32 };
  DescriptorColor colors[NUM_DESC_COLORS] = {};
  uint8_t numColors = 0;
};  // 32 * 8 + 1, padded to the 2-byte alignment of its members: 258 bytes
class VideoIndex {  // src/videoindex.h:40-52; load/save through the library's .vdx codec
 public:
  std::vector<int> frames;
  std::vector<dcthash_t> hashes;
  bool isEmpty() const { return frames.size() == 0 || hashes.size() == 0; }
  void load(const QString& file) {
    std::ifstream f(file, std::ios::binary);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.empty()) return;
    frames.resize(buf.size());  // a file of L bytes holds fewer than L frames
    hashes.resize(buf.size());
    const long long n = cbh_vdx_decode(buf.data(), buf.size(), frames.data(), hashes.data(), buf.size());
    frames.resize(size_t(n > 0 ? n : 0));
    hashes.resize(size_t(n > 0 ? n : 0));
  }
  void save(const QString& file) const {
    const size_t need = cbh_vdx_encode(frames.data(), hashes.data(), frames.size(), "0.8.1", nullptr, 0);
    std::vector<uint8_t> buf(need);
    cbh_vdx_encode(frames.data(), hashes.data(), frames.size(), "0.8.1", buf.data(), need);
    std::ofstream(file, std::ios::binary).write(reinterpret_cast<const char*>(buf.data()), long(need));
  }
};

// ---- src/media.h (the slice the indexes touch) ------------------------------------------------------
class MatchRange {
 public:
  int srcIn = -1, dstIn = -1, len = 0;
  MatchRange() {}
  MatchRange(int s, int d, int l) : srcIn(s), dstIn(d), len(l) {}
};
class Media {
 public:
  enum { TypeImage = 1, TypeVideo = 2, TypeAudio = 3 };
  Media() {}
  Media(const QString& path, int id, uint64_t dctHash) : _path(path), _id(id), _dctHash(dctHash) {}
  static int typeFlag(int type) { return 1 << (type - 1); }
  int id() const { return _id; }
  int type() const { return _type; }
  void setType(int t) { _type = t; }
  uint64_t dctHash() const { return _dctHash; }
  const QString& path() const { return _path; }
  const KeyPointHashList& keyPointHashes() const { return _kph; }
  void setKeyPointHashes(const KeyPointHashList& h) { _kph = h; }
  const KeyPointDescriptors& keyPointDescriptors() const { return _kpd; }
  void setKeyPointDescriptors(const KeyPointDescriptors& d) { _kpd = d; }
  const ColorDescriptor& colorDescriptor() const { return _cd; }
  void setColorDescriptor(const ColorDescriptor& c) { _cd = c; }
  const VideoIndex& videoIndex() const { return _vi; }
  void setVideoIndex(const VideoIndex& v) { _vi = v; }
  const MatchRange& matchRange() const { return _range; }

 private:
  QString _path;
  int _id = 0, _type = TypeImage;
  uint64_t _dctHash = 0;
  KeyPointHashList _kph;
  KeyPointDescriptors _kpd;
  ColorDescriptor _cd;
  VideoIndex _vi;
  MatchRange _range;
};
typedef QVector<Media> MediaGroup;

// ---- src/index.h:36-148 (fields used on this path) and :150-281 ---------------------------------
class SearchParams {
 public:
  enum { AlgoDCT = 0, AlgoDCTFeatures = 1, AlgoCVFeatures = 2, AlgoColor = 3, AlgoVideo = 4, NumAlgos = 5 };
  int algo = AlgoDCT, dctThresh = 5, cvThresh = 25, minMatches = 1, maxMatches = 5, maxThresh = 0;
  int skipFrames = 300, minFramesMatched = 30, minFramesNear = 60, videoRadix = 10;
  bool filterSelf = true;
};

class Index {
  Q_DISABLE_COPY_MOVE(Index)
 public:
  virtual ~Index() {}
  struct Match {
    uint32_t mediaId;
    int score;
    MatchRange range;
    Match() : mediaId(0), score(0) {}
    Match(uint32_t mediaId_, int score_) : mediaId(mediaId_), score(score_) {}
  };
  int id() const { return _id; }
  virtual bool isLoaded() const = 0;
  virtual size_t memoryUsage() const = 0;
  virtual int count() const = 0;
  virtual int databaseId() const { return id(); }
  virtual void createTables(QSqlDatabase& db) const { (void)db; }
  virtual void addRecords(QSqlDatabase& db, const MediaGroup& media) const {
    (void)db;
    (void)media;
  }
  virtual void removeRecords(QSqlDatabase& db, const QVector<int>& mediaIds) const {
    (void)db;
    (void)mediaIds;
  }
  virtual void load(QSqlDatabase& db, const QString& cachePath, const QString& dataPath) = 0;
  virtual void save(QSqlDatabase& db, const QString& cachePath) = 0;
  virtual QSet<mediaid_t> mediaIds(QSqlDatabase& db, const QString& cachePath,
                                   const QString& dataPath) const = 0;
  virtual void add(const MediaGroup& media) = 0;
  virtual void remove(const QVector<int>& id) = 0;
  virtual QVector<Index::Match> find(const Media& m, const SearchParams& p) = 0;
  virtual bool findIndexData(Media& m) const {
    Q_UNUSED(m);
    return false;
  }
  virtual Index* slice(const QSet<uint32_t>& mediaIds) const {
    (void)mediaIds;
    return nullptr;
  }
  virtual int resultTypes() const { return Media::typeFlag(Media::TypeImage); }

 protected:
  int _id;
  Index() { _id = -1; }
};
inline bool operator<(const Index::Match& m1, const Index::Match& m2) { return m1.score < m2.score; }
