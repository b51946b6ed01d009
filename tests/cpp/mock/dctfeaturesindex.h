// mock/dctfeaturesindex.h -- TEST SCAFFOLD ONLY (see mock/index.h): declaration of the reference class
// DctFeaturesIndex (src/dctfeaturesindex.h) that gpu_indexes.h derives from; cbird's own SQL-side methods are inherited from it
// in a real build and are no-ops here.
#pragma once
#include "index.h"
class DctFeaturesIndex : public Index {
 public:
  DctFeaturesIndex() { _id = SearchParams::AlgoDCTFeatures; }
  void save(QSqlDatabase&, const QString&) override {}
  void load(QSqlDatabase&, const QString&, const QString&) override {}
  QSet<mediaid_t> mediaIds(QSqlDatabase&, const QString&, const QString&) const override { return {}; }
};
