// mock/dctvideoindex.h -- TEST SCAFFOLD ONLY (see mock/index.h): declaration of the reference class
// DctVideoIndex (src/dctvideoindex.h) that gpu_indexes.h derives from; cbird's own SQL-side methods are inherited from it
// in a real build and are no-ops here.
#pragma once
#include "index.h"
class DctVideoIndex : public Index {
 public:
  DctVideoIndex() { _id = SearchParams::AlgoVideo; }
  void save(QSqlDatabase&, const QString&) override {}
  void load(QSqlDatabase&, const QString&, const QString&) override {}
  QSet<mediaid_t> mediaIds(QSqlDatabase&, const QString&, const QString&) const override { return {}; }
  int databaseId() const override { return 0; }
  size_t memoryUsage() const override { return 0; }
};
