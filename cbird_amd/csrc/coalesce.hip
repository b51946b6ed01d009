// coalesce.hip -- drop-in throughput for an UNMODIFIED caller: concurrent find() calls share scans.
//
// cbird's all-pairs search is Database::similar: QtConcurrent::map over the haystack, one
// searchIndex() -> index->find(needle) per item, from as many pool threads as the host has cores, under a read lock
// (src/database.cpp:1400-1432, 1698-1700).  Each call is synchronous, so a GPU index that serves them one by one is
// latency-bound (one 27 us round trip per needle).  Two mechanisms inside cbh_idx64_find_coalesced change that
// without touching the caller (hard part 6: "internally batch/queue"):
//
//  1. COMBINING.  Callers that arrive while a scan is in flight queue up; one of them (the leader) takes everything
//     that is queued -- up to kMaxBatch needles -- and serves it with ONE scan (needles of different thresholds:
//     one scan per threshold), then hands each caller its own sorted match list.  T blocking callers = T needles per
//     round trip.
//  2. SELF-JOIN CACHE.  The needles of Database::similar are the index's own entries.  The leader keeps a bill of
//     the wall time it has spent serving rounds for a threshold; once that exceeds the estimated cost of scanning
//     the WHOLE index against itself (N x N, the 10-20 ms batch job of bench.py at N = 1M), it does exactly that,
//     once, keeps the sorted records on the host with a needle-hash -> row table, and from then on a find() whose
//     needle hash is an index entry is a table lookup (ski-rental rule: never more than about twice the better of
//     the two strategies; a handful of interactive -similar-to queries never triggers it).  load/add/remove bump
//     the index generation and drop the cache.
//
// Results are those of cbh_idx64_find bit for bit: all entries with hamm64 < thresh, id != 0, ascending (score, id).
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <thread>

#include "cbh_index.h"

namespace cbh {

namespace {
constexpr size_t kMaxBatch = 4096;       // needles per combined scan
constexpr size_t kSpecRecs = 2048;       // records fetched together with the count (one synchronisation per round)
constexpr size_t kJoinMaxRecords = (size_t)1 << 27;  // self-join results kept on the host: at most 1 GB

struct Req {
  uint64_t q;
  int thresh;
  cbh_match* out;
  size_t cap;
  size_t n_out = 0;
  int rc = CBH_OK;
  bool done = false;
};

struct SelfJoin {
  std::vector<uint32_t> off;     // n + 1 row offsets into rec (row = index slot)
  std::vector<cbh_record> rec;   // sorted ascending: (slot, score, id)
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// Everything a cache hit needs, immutable once published: readers take it without any lock.
struct Snapshot {
  std::vector<std::pair<int, std::shared_ptr<const SelfJoin>>> joins;  // by threshold (a handful)
  // needle hash -> first slot holding it (open addressing over a host mirror of the hashes)
  std::shared_ptr<const std::vector<uint64_t>> h_hashes;
  std::shared_ptr<const std::vector<uint32_t>> table;
  uint64_t table_mask = 0;
  long slot_of(uint64_t q) const {
    const std::vector<uint32_t>& t = *table;
    const std::vector<uint64_t>& h = *h_hashes;
    for (uint64_t i = (q * 0x9E3779B97F4A7C15ull) >> 20;; ++i) {
      const uint32_t s = t[i & table_mask];
      if (!s) return -1;
      if (h[s - 1] == q) return (long)s - 1;
    }
  }
};

struct Coalescer {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Req*> pending;
  bool leader = false;
  uint64_t epoch = ~0ull;
  // lock-free read side: snap is valid for index generation snap_epoch.  Replaced snapshots are retired, not freed,
  // until the next invalidation -- which follows an exclusive writer (load/add/remove), so no reader that saw a
  // matching snap_epoch can still be inside one (the reference's rw-lock contract, src/database.cpp:371,1698).
  std::atomic<uint64_t> snap_epoch{~0ull};
  std::atomic<const Snapshot*> snap{nullptr};
  std::vector<const Snapshot*> retired;
  // the bill that decides when a threshold's self-join is built
  std::map<int, double> spent;
  std::map<int, bool> too_big;
  // pinned staging of the leader
  uint64_t* h_q = nullptr;
  cbh_record* h_spec = nullptr;
  unsigned long long* h_total = nullptr;
  // statistics: sharded so that a million cache hits from 64 threads do not fight over one cache line
  struct alignas(64) Shard {
    std::atomic<uint64_t> finds{0}, hits{0};
  };
  Shard shards[64];
  uint64_t rounds = 0, scanned = 0, self_joins = 0;  // leader only, under mu
  int join_enabled = 1;

  ~Coalescer() {
    delete snap.load();
    for (const Snapshot* p : retired) delete p;
    if (h_q) (void)hipHostFree(h_q);
    if (h_spec) (void)hipHostFree(h_spec);
    if (h_total) (void)hipHostFree(h_total);
  }
  int staging() {
    if (h_total) return CBH_OK;  // (the last of the three: all or nothing)
    uint64_t* q = nullptr;
    cbh_record* spec = nullptr;
    unsigned long long* total = nullptr;
    hipError_t e = hipHostMalloc(&q, kMaxBatch * sizeof(uint64_t));
    if (e == hipSuccess) e = hipHostMalloc(&spec, kSpecRecs * sizeof(cbh_record));
    if (e == hipSuccess) e = hipHostMalloc(&total, sizeof(unsigned long long));
    if (e != hipSuccess) {  // a later call starts over; nothing half-made stays behind
      if (q) (void)hipHostFree(q);
      if (spec) (void)hipHostFree(spec);
      CBH_HIP(e);
    }
    h_q = q, h_spec = spec, h_total = total;
    return CBH_OK;
  }
  void invalidate(uint64_t gen) {  // under mu
    snap_epoch.store(~0ull, std::memory_order_release);
    delete snap.exchange(nullptr);
    for (const Snapshot* p : retired) delete p;
    retired.clear();
    spent.clear();
    too_big.clear();
    epoch = gen;
  }
  void retire_current(uint64_t gen) {  // under mu; readers may still hold the snapshot
    snap_epoch.store(~0ull, std::memory_order_release);
    const Snapshot* old = snap.exchange(nullptr, std::memory_order_acq_rel);
    if (old) retired.push_back(old);
    spent.clear();
    too_big.clear();
    epoch = gen;
  }
  void publish(const Snapshot* s, uint64_t gen) {  // under mu
    const Snapshot* old = snap.exchange(s, std::memory_order_acq_rel);
    if (old) retired.push_back(old);
    snap_epoch.store(gen, std::memory_order_release);
  }
  static Shard& my_shard(Shard* sh) {
    static thread_local const unsigned k = (unsigned)(std::hash<std::thread::id>()(std::this_thread::get_id()) % 64);
    return sh[k];
  }
};

void coalescer_free(Coalescer* c) { delete c; }

namespace {

void deliver(Req* r, const cbh_record* rec, size_t n) {
  r->n_out = n;
  const size_t m = std::min(n, r->cap);
  for (size_t i = 0; i < m; ++i) {
    r->out[i].id = CBH_REC_ID(rec[i]);
    r->out[i].score = CBH_REC_DIST(rec[i]);
  }
  r->rc = CBH_OK;
}

// one combined scan: every request of `batch` has the same threshold.  Records of all needles -> host, ordered.
int serve_by_scan(cbh_idx64* idx, Coalescer* co, Workspace* ws, std::vector<Req*>& batch) {
  const size_t nq = batch.size();
  const int thresh = batch[0]->thresh;
  int rc = co->staging();
  if (rc) return rc;
  if ((rc = ws->ensure_records(Workspace::kFindRecs))) return rc;
  if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, nq))) return rc;
  for (size_t i = 0; i < nq; ++i) co->h_q[i] = batch[i]->q;
  hipStream_t s = ws->stream;
  CBH_HIP(hipMemcpyAsync(ws->d_q, co->h_q, nq * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  unsigned long long total = 0;
  const size_t spec = std::min(kSpecRecs, ws->rec_cap);
  if (idx->shards) {  // every shard scans; scan_all merges their blocks into this workspace
    if ((rc = scan_all(idx, ws, ws->d_q, nq, thresh, s, &total))) return rc;
    if (total && total <= spec) {
      CBH_HIP(hipMemcpyAsync(co->h_spec, ws->d_rec, (size_t)total * sizeof(cbh_record), hipMemcpyDeviceToHost, s));
      CBH_HIP(hipStreamSynchronize(s));
    }
  } else {
    CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), s));
    rc = launch_hamm64_scan(idx->d_hashes, idx->d_ids, idx->n, ws->d_q, nq, thresh, ws->d_rec, ws->rec_cap, ws->d_total,
                            s, 0, nullptr);
    if (rc) return rc;
    CBH_HIP(hipMemcpyAsync(co->h_total, ws->d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    CBH_HIP(hipMemcpyAsync(co->h_spec, ws->d_rec, spec * sizeof(cbh_record), hipMemcpyDeviceToHost, s));
    CBH_HIP(hipStreamSynchronize(s));
    total = *co->h_total;
    {
      std::lock_guard<std::mutex> lk(idx->stats_mu);
      idx->stats.scan_launches += 1;
      idx->stats.scan_pairs += (uint64_t)idx->n * (uint64_t)nq;
    }
  }
  std::vector<cbh_record> big;
  cbh_record* recs = co->h_spec;
  if (total > spec) {
    if (!idx->shards && total > ws->rec_cap) {  // did not fit: the general path grows the buffer and rescans
      rc = scan_all(idx, ws, ws->d_q, nq, thresh, s, &total);
      if (rc) return rc;
    }
    big.resize((size_t)total);
    CBH_HIP(hipMemcpyAsync(big.data(), ws->d_rec, (size_t)total * sizeof(cbh_record), hipMemcpyDeviceToHost, s));
    CBH_HIP(hipStreamSynchronize(s));
    recs = big.data();
  }
  std::sort(recs, recs + total);  // (needle, score, id)
  size_t a = 0;
  for (size_t j = 0; j < nq; ++j) {
    size_t b = a;
    while (b < total && CBH_REC_QUERY(recs[b]) == j) ++b;
    deliver(batch[j], recs + a, b - a);
    a = b;
  }
  return CBH_OK;
}

// the whole index against itself at `thresh`; builds the hash -> slot table too when `base` has none yet.
// Returns a new snapshot = base + this join.
int build_self_join(cbh_idx64* idx, const Snapshot* base, Workspace* ws, int thresh, Snapshot** out, bool* too_big) {
  *too_big = false;
  *out = nullptr;
  const size_t n = idx->n;
  if (n > CBH_MAX_QUERIES_PER_CALL - 1) {
    *too_big = true;
    return CBH_OK;
  }
  hipStream_t s = ws->stream;
  unsigned long long total = 0;
  const bool need_table = !base || !base->table;
  auto hh = std::make_shared<std::vector<uint64_t>>();
  // the needles are the index entries themselves, in slot order.  A sharded index has no such array on one device:
  // its host mirror (needed for the table anyway) is uploaded as the needle list
  const uint64_t* d_needles = idx->d_hashes;
  int rc;
  if (idx->shards) {
    hh->resize(n);
    if ((rc = cbh_idx64_download(idx, hh->data(), nullptr, n))) return rc;
    if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, n))) return rc;
    CBH_HIP(hipMemcpyAsync(ws->d_q, hh->data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    d_needles = ws->d_q;
  }
  // bounded: an index full of near-duplicates (1e5 equal hashes = 1e10 pairs) must not make one find() ask for tens
  // of GB -- scan_all learns the count from its first pass and gives up before it grows anything past the limit
  rc = scan_all(idx, ws, d_needles, n, thresh, s, &total, 0, nullptr, kJoinMaxRecords);
  if (rc == CBH_E_OVERFLOW || (!rc && total > kJoinMaxRecords)) {
    *too_big = true;
    return CBH_OK;
  }
  if (rc) return rc;
  if ((rc = ws->ensure_sort())) return rc;
  if ((rc = launch_sort_records(ws->d_rec, ws->d_alt, (size_t)total, n, ws->d_tmp, ws->tmp_bytes, s))) return rc;
  auto sj = std::make_shared<SelfJoin>();
  sj->rec.resize((size_t)total);
  if (total)
    CBH_HIP(hipMemcpyAsync(sj->rec.data(), ws->d_rec, (size_t)total * sizeof(cbh_record), hipMemcpyDeviceToHost, s));
  if (need_table && !idx->shards) {
    hh->resize(n);
    CBH_HIP(hipMemcpyAsync(hh->data(), idx->d_hashes, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  }
  CBH_HIP(hipStreamSynchronize(s));
  // the workspace goes back to the index's pool: do not let it keep a self-join-sized block (and its sort scratch)
  ws->shrink_records(std::max<size_t>(idx->rec_cap_default, Workspace::kFindRecs));
  sj->off.assign(n + 1, 0);
  for (size_t i = 0; i < (size_t)total; ++i) sj->off[(size_t)CBH_REC_QUERY(sj->rec[i]) + 1]++;
  for (size_t j = 0; j < n; ++j) sj->off[j + 1] += sj->off[j];
  std::unique_ptr<Snapshot> sn(new (std::nothrow) Snapshot);
  if (!sn) return CBH_E_NOMEM;
  if (base) *sn = *base;
  if (need_table) {
    size_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    auto tab = std::make_shared<std::vector<uint32_t>>(cap, 0u);
    const uint64_t mask = cap - 1;
    const std::vector<uint64_t>& h = *hh;
    for (size_t sl = 0; sl < n; ++sl) {
      const uint64_t q = h[sl];
      if (!q) continue;  // removed slot / null hash: never a needle
      for (uint64_t i = (q * 0x9E3779B97F4A7C15ull) >> 20;; ++i) {
        uint32_t& e = (*tab)[i & mask];
        if (!e) {
          e = (uint32_t)sl + 1;
          break;
        }
        if (h[e - 1] == q) break;  // same hash already present: equal needles have equal rows
      }
    }
    sn->h_hashes = hh;
    sn->table = tab;
    sn->table_mask = mask;
  }
  sn->joins.emplace_back(thresh, sj);
  *out = sn.release();
  return CBH_OK;
}

}  // namespace
}  // namespace cbh

using namespace cbh;

static Coalescer* get_coalescer(cbh_idx64* idx) {
  Coalescer* co = idx->coalescer.load(std::memory_order_acquire);
  if (co) return co;
  std::lock_guard<std::mutex> lk(idx->ws_mu);
  co = idx->coalescer.load(std::memory_order_acquire);
  if (!co) {
    co = new (std::nothrow) Coalescer;
    idx->coalescer.store(co, std::memory_order_release);
  }
  return co;
}

int cbh_idx64_find_coalesced(cbh_idx64* idx, uint64_t q, int thresh, cbh_match* out, size_t cap, size_t* n_out) {
  if (!idx || !n_out || (cap && !out)) return CBH_E_INVAL;
  *n_out = 0;
  if (q == 0 || idx->n == 0 || thresh <= 0) return CBH_OK;  // as cbh_idx64_find
  Coalescer* co = get_coalescer(idx);
  if (!co) return CBH_E_NOMEM;
  Req me{q, thresh, out, cap};
  Coalescer::Shard& shard = Coalescer::my_shard(co->shards);
  shard.finds.fetch_add(1, std::memory_order_relaxed);
  // ---- cached self-join: a lookup, no lock, no device work ----
  const uint64_t gen0 = idx->generation.load(std::memory_order_acquire);
  if (co->snap_epoch.load(std::memory_order_acquire) == gen0) {
    const Snapshot* sn = co->snap.load(std::memory_order_acquire);
    if (sn)
      for (const auto& kv : sn->joins)
        if (kv.first == thresh) {
          const long sl = sn->slot_of(q);
          if (sl < 0) break;  // not an index entry: combined scan below
          const SelfJoin& sj = *kv.second;
          deliver(&me, sj.rec.data() + sj.off[(size_t)sl], sj.off[(size_t)sl + 1] - sj.off[(size_t)sl]);
          shard.hits.fetch_add(1, std::memory_order_relaxed);
          *n_out = me.n_out;
          return CBH_OK;
        }
  }
  std::unique_lock<std::mutex> lk(co->mu);
  if (co->epoch != gen0) co->invalidate(gen0);
  co->pending.push_back(&me);
  while (!me.done) {
    if (co->leader) {
      co->cv.wait(lk);
      continue;
    }
    // become the leader: serve everything that is queued (this request included), round after round, until this
    // request is done; then hand the role to whoever is still waiting
    co->leader = true;
    while (!me.done) {
      std::vector<Req*> batch;
      const size_t take = std::min(co->pending.size(), kMaxBatch);
      batch.assign(co->pending.begin(), co->pending.begin() + (long)take);
      co->pending.erase(co->pending.begin(), co->pending.begin() + (long)take);
      const uint64_t gen = co->epoch;
      const Snapshot* base = co->snap.load(std::memory_order_acquire);
      const bool joins_on = co->join_enabled != 0;
      lk.unlock();
      // ---- device work, no lock held ----
      int rc = CBH_OK;
      DeviceGuard g(idx->device);
      Workspace* ws = nullptr;
      if (!g.ok) rc = CBH_E_NODEVICE;
      if (!rc) ws = idx->acquire(&rc);
      std::map<int, std::vector<Req*>> by_thr;
      for (Req* r : batch) by_thr[r->thresh].push_back(r);
      std::map<int, double> cost;
      for (auto& kv : by_thr) {
        const double t1 = now_s();
        int r2 = rc;
        if (!r2) r2 = serve_by_scan(idx, co, ws, kv.second);
        if (r2)
          for (Req* r : kv.second) r->rc = r2, r->n_out = 0;
        cost[kv.first] = now_s() - t1;
      }
      // ski-rental: has serving this threshold round by round cost as much as one self-join would?
      Snapshot* fresh = nullptr;
      std::vector<int> big;
      if (!rc && joins_on) {
        const double n = (double)idx->n;
        const double est = n * n / 5.0e13 + n * 4.0e-8 + 2.0e-3;
        for (auto& kv : cost) {
          bool have = false;
          const Snapshot* cur = fresh ? fresh : base;
          if (cur)
            for (const auto& j : cur->joins) have |= (j.first == kv.first);
          double bill;
          bool skip;
          {
            std::lock_guard<std::mutex> l2(co->mu);
            bill = (co->spent[kv.first] += kv.second);
            skip = have || co->too_big[kv.first] || co->epoch != gen;
          }
          if (skip || bill < 0.5 * est) continue;
          Snapshot* sn = nullptr;
          bool tb = false;
          if (build_self_join(idx, cur, ws, kv.first, &sn, &tb) == CBH_OK) {
            if (tb) big.push_back(kv.first);
            if (sn) {
              delete fresh;
              fresh = sn;
            }
          }
        }
      }
      if (ws) idx->give_back(ws);
      // ---- publish ----
      lk.lock();
      co->rounds += 1;
      co->scanned += batch.size();
      if (co->epoch == gen && idx->generation.load() == gen) {
        if (fresh) {
          co->self_joins += fresh->joins.size() - (base ? base->joins.size() : 0);
          co->publish(fresh, gen);
          fresh = nullptr;
        }
        for (int t : big) co->too_big[t] = true;
      }
      delete fresh;
      for (Req* r : batch) r->done = true;
      co->cv.notify_all();
    }
    co->leader = false;
    co->cv.notify_all();
  }
  lk.unlock();
  *n_out = me.n_out;
  return me.rc;
}

int cbh_idx64_coalesce_stats(cbh_idx64* idx, cbh_coalesce_stats* out) {
  if (!idx || !out) return CBH_E_INVAL;
  *out = cbh_coalesce_stats{0, 0, 0, 0, 0};
  Coalescer* co = idx->coalescer.load(std::memory_order_acquire);
  if (!co) return CBH_OK;
  std::lock_guard<std::mutex> l2(co->mu);
  for (auto& sh : co->shards) out->finds += sh.finds.load(), out->cache_hits += sh.hits.load();
  out->rounds = co->rounds, out->scanned_needles = co->scanned, out->self_joins = co->self_joins;
  return CBH_OK;
}

int cbh_idx64_coalesce_set_self_join(cbh_idx64* idx, int enabled) {
  if (!idx) return CBH_E_INVAL;
  Coalescer* co = get_coalescer(idx);
  if (!co) return CBH_E_NOMEM;
  std::lock_guard<std::mutex> l2(co->mu);
  co->join_enabled = enabled ? 1 : 0;
  // switching the cache off while finds are running (this is a reader-side call, no writer lock protects it): a reader
  // may be walking the snapshot right now, so it is RETIRED -- unpublished, freed by the next load/add/remove or with
  // the index -- never deleted here
  if (!enabled) co->retire_current(idx->generation.load());
  return CBH_OK;
}
