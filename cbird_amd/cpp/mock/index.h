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
#include <iostream>
#include <set>
#include <string>
#include <utility>
#include <vector>

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
inline const char* qPrintable(const QString& s) { return s.c_str(); }

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

// ---- QtSql: an in-memory `media` table -------------------------------------------------------
struct QSqlDatabase {
  struct Row {
    uint32_t id;
    int type;
    int64_t phash_dct;
  };
  std::vector<Row> media;
};
struct QVariant {
  int64_t v;
  unsigned toUInt() const { return unsigned(v); }
  long long toLongLong() const { return v; }
  unsigned long long toULongLong() const { return (unsigned long long)v; }
};
struct QSqlError {
  QString text() const { return "mock"; }
};
struct QSqlQuery {
  QSqlDatabase& db;
  long pos = -1;
  bool ok = false;
  explicit QSqlQuery(QSqlDatabase& d) : db(d) {}
  void setForwardOnly(bool) {}
  bool exec(const char* sql) {
    ok = std::string(sql) == "select id,phash_dct from media where type=1";
    pos = -1;
    return ok;
  }
  bool next() {
    while (++pos < long(db.media.size()))
      if (db.media[size_t(pos)].type == 1) return true;
    return false;
  }
  QVariant value(int col) const {
    const auto& r = db.media[size_t(pos)];
    return QVariant{col == 0 ? int64_t(r.id) : r.phash_dct};
  }
  QSqlError lastError() const { return {}; }
};
#define SQL_FATAL(x) qFatal("QSqlQuery." #x ": %s", qPrintable(query.lastError().text()));

// ---- src/media.h (the slice DctHashIndex touches) -------------------------------------------------
class MatchRange {
 public:
  int srcIn = -1, dstIn = -1, len = 0;
};
class Media {
 public:
  enum { TypeImage = 1, TypeVideo = 2, TypeAudio = 3 };
  Media() {}
  Media(const QString& path, int id, uint64_t dctHash) : _path(path), _id(id), _dctHash(dctHash) {}
  static int typeFlag(int type) { return 1 << (type - 1); }
  int id() const { return _id; }
  uint64_t dctHash() const { return _dctHash; }
  const QString& path() const { return _path; }

 private:
  QString _path;
  int _id = 0;
  uint64_t _dctHash = 0;
};
typedef QVector<Media> MediaGroup;

// ---- src/index.h:36-148 (fields used on this path) and :150-281 ---------------------------------
class SearchParams {
 public:
  enum { AlgoDCT = 0, AlgoDCTFeatures = 1, AlgoCVFeatures = 2, AlgoColor = 3, AlgoVideo = 4, NumAlgos = 5 };
  int algo = AlgoDCT, dctThresh = 5, cvThresh = 25, minMatches = 1, maxMatches = 5, maxThresh = 0;
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
