// combine.hip -- drop-in throughput for an UNMODIFIED caller on the other four indexes: concurrent find() calls share
// one batched search.
//
// Database::similar calls searchIndex -> index->find(needle) once per haystack item from as many pool threads as the
// host has cores (src/database.cpp:1400-1432, 1698-1700), each call synchronous.  cbh_idx64_find_coalesced
// (coalesce.hip) combines such callers for DctHashIndex; this file does the same for DctFeaturesIndex, DctVideoIndex,
// CvFeaturesIndex and ColorDescIndex with one generic leader / follower queue: callers that arrive while a search is in
// flight queue up, one of them (the leader) takes every queued request whose parameters equal those of the oldest one
// -- up to kMaxBatch -- serves them with ONE call of the index's batch entry point (cbh_fdct_find_batch_ex /
// cbh_vidx_find_videos_batch / cbh_idx256_find_batch / cbh_color_find_all_batch: one scan, one reduction), and hands
// every caller its own result.  T blocked callers = T needles per device round trip.  Results are those of the
// single-needle entry points, which the batch entry points already equal (tests/test_fdct.py, test_video.py,
// test_cvfeatures.py, test_color.py).  No self-join cache here: these needles carry hundreds of hashes each and the
// batched searches are already the efficient form.
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include "cbh_index.h"

namespace cbh {

namespace {

constexpr size_t kMaxBatch = 256;  // needles per combined search
constexpr size_t kMaxRoundWeight = (size_t)32 << 20;  // ... and result places per round (256 MB of cbh_match)

struct Stats {
  uint64_t finds = 0, rounds = 0;
};

struct CombinerBase {
  std::mutex mu;
  std::condition_variable cv;
  bool leader = false;
  Stats st;
  virtual ~CombinerBase() {}
};

template <class Req>
struct Combiner : CombinerBase {
  std::deque<Req*> pending;

  // serve(batch): fills rc / n_out / out of every request of the batch
  template <class Serve>
  int submit(Req& me, Serve serve) {
    std::unique_lock<std::mutex> lk(mu);
    st.finds++;
    pending.push_back(&me);
    while (!me.done) {
      if (leader) {
        cv.wait(lk);
        continue;
      }
      leader = true;  // serve everything that is queued, round after round, until this request is done
      while (!me.done) {
        std::vector<Req*> batch;
        Req* first = pending.front();
        size_t weight = 0;  // result places the round may need (a ColorDescIndex needle returns every entry)
        for (auto it = pending.begin(); it != pending.end() && batch.size() < kMaxBatch;) {
          if ((*it)->compatible(*first) && (batch.empty() || weight + (*it)->weight <= kMaxRoundWeight)) {
            weight += (*it)->weight;
            batch.push_back(*it);
            it = pending.erase(it);
          } else {
            ++it;
          }
        }
        lk.unlock();
        try {
          serve(batch);
        } catch (...) {  // std::bad_alloc of a staging vector: the round fails, the queue goes on
          for (Req* r : batch) r->rc = CBH_E_NOMEM, r->n_out = 0;
        }
        lk.lock();
        st.rounds++;
        for (Req* r : batch) r->done = true;
        cv.notify_all();
      }
      leader = false;
      cv.notify_all();
    }
    return me.rc;
  }
};

// one combiner per index handle, created on first use, dropped with the handle
struct Registry {
  std::mutex mu;
  std::map<const void*, CombinerBase*> by_handle;
};
Registry& registry() {
  static Registry* r = new Registry;
  return *r;
}
template <class C>
C* combiner_of(const void* handle) {
  Registry& R = registry();
  std::lock_guard<std::mutex> lk(R.mu);
  auto it = R.by_handle.find(handle);
  if (it != R.by_handle.end()) return static_cast<C*>(it->second);
  C* c = new (std::nothrow) C;
  if (!c) return nullptr;
  R.by_handle[handle] = c;
  return c;
}

struct ReqBase {
  bool done = false;
  int rc = CBH_OK;
  size_t cap = 0, n_out = 0;
  size_t weight = 1;  // result places this request may need in the round's buffer
};
struct FdctReq : ReqBase {
  const uint64_t* hashes;
  size_t n;
  uint32_t needle_id;
  int thresh, tree_compat;
  cbh_match* out;
  bool compatible(const FdctReq& o) const { return thresh == o.thresh && tree_compat == o.tree_compat; }
};
struct VideoReq : ReqBase {
  const int32_t* frames;
  const uint64_t* hashes;
  size_t n;
  uint32_t needle_id;
  int thresh, skip, vfm, vfn, filter_self;
  cbh_vmatch* out;
  bool compatible(const VideoReq& o) const {
    return thresh == o.thresh && skip == o.skip && vfm == o.vfm && vfn == o.vfn && filter_self == o.filter_self;
  }
};
struct OrbReq : ReqBase {
  const uint8_t* rows;
  size_t n_desc;
  int thresh, k;
  cbh_match* out;
  bool compatible(const OrbReq& o) const { return thresh == o.thresh && k == o.k; }
};
struct ColorReq : ReqBase {
  const uint8_t* desc;
  cbh_match* out;
  bool compatible(const ColorReq&) const { return true; }
};

template <class Req, class M>
void scatter(std::vector<Req*>& batch, int rc, const std::vector<M>& buf, const std::vector<uint64_t>& oo) {
  for (size_t i = 0; i < batch.size(); ++i) {
    Req* r = batch[i];
    r->rc = rc;
    r->n_out = 0;
    if (rc) continue;
    r->n_out = (size_t)(oo[i + 1] - oo[i]);
    const size_t m = std::min(r->n_out, r->cap);
    for (size_t t = 0; t < m; ++t) r->out[t] = buf[(size_t)oo[i] + t];
  }
}

}  // namespace

void combiner_drop(const void* handle) {
  Registry& R = registry();
  std::lock_guard<std::mutex> lk(R.mu);
  auto it = R.by_handle.find(handle);
  if (it == R.by_handle.end()) return;
  delete it->second;
  R.by_handle.erase(it);
}

}  // namespace cbh

using namespace cbh;

extern "C" {

int cbh_fdct_find_coalesced(cbh_idx64* idx, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                            int tree_compat, cbh_match* out, size_t cap, size_t* n_out) {
  if (!idx || !n_out || (cap && !out) || (n && !hashes)) return CBH_E_INVAL;
  *n_out = 0;
  auto* co = combiner_of<Combiner<FdctReq>>(idx);
  if (!co) return CBH_E_NOMEM;
  FdctReq me;
  me.hashes = hashes, me.n = n, me.needle_id = needle_id, me.thresh = thresh, me.tree_compat = tree_compat;
  me.out = out, me.cap = cap;
  int rc = co->submit(me, [&](std::vector<FdctReq*>& batch) {
    std::vector<uint64_t> h, offs(1, 0), oo(batch.size() + 1);
    std::vector<uint32_t> ids;
    for (FdctReq* r : batch) {
      h.insert(h.end(), r->hashes, r->hashes + r->n);
      offs.push_back(h.size());
      ids.push_back(r->needle_id);
    }
    std::vector<cbh_match> buf(h.size() * 10 + 1);
    int rc2 = cbh_fdct_find_batch_ex(idx, h.data(), offs.data(), ids.data(), batch.size(), batch[0]->thresh,
                                     batch[0]->tree_compat, buf.data(), buf.size(), oo.data());
    scatter(batch, rc2, buf, oo);
  });
  *n_out = me.n_out;
  return rc;
}

int cbh_vidx_find_video_coalesced(cbh_vidx* v, const int32_t* frames, const uint64_t* hashes, size_t n, uint32_t needle_id,
                                  int thresh, int skip_frames, int min_frames_matched, int min_frames_near,
                                  int filter_self, cbh_vmatch* out, size_t cap, size_t* n_out) {
  if (!v || !n_out || (cap && !out) || (n && (!frames || !hashes))) return CBH_E_INVAL;
  *n_out = 0;
  auto* co = combiner_of<Combiner<VideoReq>>(v);
  if (!co) return CBH_E_NOMEM;
  VideoReq me;
  me.frames = frames, me.hashes = hashes, me.n = n, me.needle_id = needle_id, me.thresh = thresh, me.skip = skip_frames;
  me.vfm = min_frames_matched, me.vfn = min_frames_near, me.filter_self = filter_self, me.out = out, me.cap = cap;
  int rc = co->submit(me, [&](std::vector<VideoReq*>& batch) {
    std::vector<int32_t> f;
    std::vector<uint64_t> h, offs(1, 0), oo(batch.size() + 1);
    std::vector<uint32_t> ids;
    for (VideoReq* r : batch) {
      f.insert(f.end(), r->frames, r->frames + r->n);
      h.insert(h.end(), r->hashes, r->hashes + r->n);
      offs.push_back(h.size());
      ids.push_back(r->needle_id);
    }
    const VideoReq& p = *batch[0];
    std::vector<cbh_vmatch> buf(std::max<size_t>(64, 16 * batch.size()));
    int rc2 = cbh_vidx_find_videos_batch(v, f.data(), h.data(), offs.data(), ids.data(), batch.size(), p.thresh, p.skip,
                                         p.vfm, p.vfn, p.filter_self, buf.data(), buf.size(), oo.data());
    if (rc2 == CBH_E_OVERFLOW) {
      buf.resize((size_t)oo[batch.size()]);
      rc2 = cbh_vidx_find_videos_batch(v, f.data(), h.data(), offs.data(), ids.data(), batch.size(), p.thresh, p.skip, p.vfm,
                                       p.vfn, p.filter_self, buf.data(), buf.size(), oo.data());
    }
    scatter(batch, rc2, buf, oo);
  });
  *n_out = me.n_out;
  return rc;
}

int cbh_idx256_find_coalesced(cbh_idx256* ix, const uint8_t* needle_rows, size_t n_desc, int thresh, int k, cbh_match* out,
                              size_t cap, size_t* n_out) {
  if (!ix || !n_out || (cap && !out) || (n_desc && !needle_rows) || k <= 0) return CBH_E_INVAL;
  *n_out = 0;
  auto* co = combiner_of<Combiner<OrbReq>>(ix);
  if (!co) return CBH_E_NOMEM;
  OrbReq me;
  me.rows = needle_rows, me.n_desc = n_desc, me.thresh = thresh, me.k = k, me.out = out, me.cap = cap;
  int rc = co->submit(me, [&](std::vector<OrbReq*>& batch) {
    std::vector<uint8_t> rows;
    std::vector<uint64_t> offs(1, 0), oo(batch.size() + 1);
    for (OrbReq* r : batch) {
      rows.insert(rows.end(), r->rows, r->rows + r->n_desc * 32);
      offs.push_back(rows.size() / 32);
    }
    std::vector<cbh_match> buf(rows.size() / 32 * (size_t)batch[0]->k + 1);
    int rc2 = cbh_idx256_find_batch(ix, rows.data(), offs.data(), batch.size(), batch[0]->thresh, batch[0]->k, buf.data(),
                                    buf.size(), oo.data());
    scatter(batch, rc2, buf, oo);
  });
  *n_out = me.n_out;
  return rc;
}

int cbh_color_find_coalesced(cbh_color* c, const void* needle_desc, cbh_match* out, size_t cap, size_t* n_out) {
  if (!c || !needle_desc || !n_out || (cap && !out)) return CBH_E_INVAL;
  *n_out = 0;
  auto* co = combiner_of<Combiner<ColorReq>>(c);
  if (!co) return CBH_E_NOMEM;
  ColorReq me;
  me.desc = (const uint8_t*)needle_desc, me.out = out, me.cap = cap;
  me.weight = std::max<size_t>(1, cbh_color_count(c));  // every entry of the index comes back per needle
  int rc = co->submit(me, [&](std::vector<ColorReq*>& batch) {
    std::vector<uint8_t> d;
    std::vector<uint64_t> oo(batch.size() + 1);
    for (ColorReq* r : batch) d.insert(d.end(), r->desc, r->desc + CBH_COLOR_DESC_BYTES);
    std::vector<cbh_match> buf(std::max<size_t>(1, batch.size() * cbh_color_count(c)));
    int rc2 = cbh_color_find_all_batch(c, d.data(), batch.size(), buf.data(), buf.size(), oo.data());
    scatter(batch, rc2, buf, oo);
  });
  *n_out = me.n_out;
  return rc;
}

/* rounds = combined searches, finds = calls: finds / rounds needles per device round trip */
int cbh_combine_stats(const void* handle, uint64_t* finds, uint64_t* rounds) {
  if (!handle || !finds || !rounds) return CBH_E_INVAL;
  *finds = *rounds = 0;
  Registry& R = registry();
  std::lock_guard<std::mutex> lk(R.mu);
  auto it = R.by_handle.find(handle);
  if (it == R.by_handle.end()) return CBH_OK;
  CombinerBase* c = it->second;
  std::lock_guard<std::mutex> l2(c->mu);
  *finds = c->st.finds, *rounds = c->st.rounds;
  return CBH_OK;
}

}  // extern "C"
