// cbird_hip.hip -- host side of the C-ABI in include/cbird_hip.h.
//
// Holds the DctHashIndex state on the device (SoA hashes/ids exactly like
// src/dcthashindex.h:60-64), a pool of per-call workspaces so that find() is re-entrant for
// cbird's thread-pool callers (src/database.cpp:1400-1432), and the sequencing of
// scan -> sort -> select.  There is no CPU compute path in this file: if no gfx950 device is
// usable every compute entry point returns CBH_E_NODEVICE.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "cbh_internal.h"

namespace cbh {

static thread_local std::string t_last_error;
static thread_local int t_last_code = CBH_OK;  // the CBH_E_* that goes with t_last_error (cbh_last_error_code)

void set_last_error(const char* where, hipError_t e) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
  t_last_error = buf;
  t_last_code = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
}

void set_last_error_text(const char* text) {
  t_last_error = text ? text : "";
  t_last_code = CBH_E_UNSUPPORTED;
}
void clear_last_error() {
  t_last_error.clear();
  t_last_code = CBH_OK;
}
void set_last_error_code(int code) { t_last_code = code; }
void* fail_handle(int code, const char* why) {
  // a HIP / RCCL failure underneath has already said what happened (and NOMEM is worth keeping: callers retry on it)
  if (t_last_code == CBH_OK || why) t_last_code = code;
  if (why) t_last_error = why;
  else if (t_last_error.empty()) t_last_error = cbh_strerror(code);
  return nullptr;
}

// ---- stream-ordered scratch (see cbh_internal.h) --------------------------------------------------------------------
// The library's own arena.  Blocks are plain hipMalloc memory, cached per (device, stream); a block
// freed with free_async(p, s) is at once available again -- but only to allocations on the SAME stream s, which by
// stream order run after everything that used it.  A block changes streams only when its stream is known idle or gone
// (stream_destroy, eviction, trim), and goes back to the driver only then.  Nothing here depends on ROCm's
// hipMallocAsync, whose pools (tools/ubench/pool_cross_stream.hip, profiles/r03_pool_cross_stream.jsonl) hand out
// memory that is still in use (rounds 2-5 kept them selectable for the A/B soak: r05's "scratch_alloc" 0 / 1).
namespace {
constexpr size_t kMaxStreamCaches = 32;  // live (device, stream) caches; beyond that dead / idle streams give theirs up
struct Block {
  void* p;
  size_t bytes;
};
struct StreamCache {
  std::vector<Block> free;
  size_t free_bytes = 0;
  uint64_t last_use = 0;
};
struct LiveInfo {
  size_t bytes;
  int dev;
};
struct Arena {
  std::mutex mu;
  std::map<std::pair<int, hipStream_t>, StreamCache> caches;
  std::map<int, std::vector<Block>> orphan;  // blocks whose stream is idle or gone: any stream may take them
  std::map<int, size_t> orphan_bytes;
  std::map<void*, LiveInfo> live;  // blocks handed out
  // blocks a live stream's cache gave up because it exceeded the budget: work queued on that stream may still use them,
  // so each waits for an event recorded behind that work and goes back to the driver once it has completed
  // (one event per trim: the blocks a single free_async gave up wait behind the same point of the stream)
  struct Pending {
    std::vector<Block> blocks;
    size_t bytes;
    int dev;
    hipEvent_t ev;
  };
  std::vector<Pending> pending;
  uint64_t n_trimmed_live = 0, n_oom_retry_stream = 0, n_oom_retry_device = 0, n_oom_retry_persistent = 0;
  uint64_t clock = 0;
  uint64_t n_malloc = 0, n_reuse = 0, n_adopt = 0, n_evicted_dead = 0, n_evicted_idle = 0, n_released = 0;
};
Arena& arena() {
  static Arena* a = new Arena;  // never destroyed: calls may arrive during process teardown
  return *a;
}
uint64_t g_pool_keep_bytes = (uint64_t)16 << 30;  // cached scratch that outlives its stream, per device ("pool_keep_mb")
// What a LIVE stream's cache may hold ("pool_live_keep_mb").  A budget of its own: the working set of one call can be
// far above what is worth keeping for streams that are gone (ColorDescriptor::create takes ~50 GB of scratch per 10^5
// images, one block of it > 16 GiB; released and re-mapped every call it costs 2.3 s instead of 0.31 s per call, NOTES 7),
// and whatever is cached stays reclaimable -- an allocation the driver refuses gives the caches back before it fails.
// 0 = a quarter of the device's memory (72 GB of an MI355X's 288), never below the orphan budget's default.
uint64_t g_pool_live_keep_bytes = 0;
// (per ordinal, from hipDeviceTotalMem: a property of `dev`, whichever device is current -- and asked for in
// malloc_async before the arena's lock is taken, so free_async finds it cached)
std::atomic<uint64_t> g_live_auto[32];
uint64_t live_budget(int dev) {
  if (g_pool_live_keep_bytes) return g_pool_live_keep_bytes;
  if (dev < 0 || dev >= 32) return (uint64_t)16 << 30;
  uint64_t v = g_live_auto[dev].load(std::memory_order_relaxed);
  if (!v) {
    size_t tot = 0;
    if (hipDeviceTotalMem(&tot, dev) != hipSuccess) {
      (void)hipGetLastError();
      return (uint64_t)16 << 30;
    }
    v = std::max<uint64_t>((uint64_t)16 << 30, (uint64_t)tot / 4);
    g_live_auto[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// fault injection (cbh_internal.h): countdowns, -1 = disarmed
std::atomic<long> g_fault_alloc{-1}, g_fault_driver{-1}, g_fault_persist{-1};
std::atomic<unsigned long> g_fault_fired{0}, g_alloc_calls{0};
std::atomic<int> g_fault_sticky{0};  // "fault_alloc_sticky": once the countdown has run out every allocation fails
hipError_t countdown(std::atomic<long>& c) {
  long v = c.load(std::memory_order_relaxed);
  while (v >= 0) {
    if (v == 0 && g_fault_sticky.load() && &c == &g_fault_alloc) {  // stays armed at zero
      g_fault_fired++;
      return hipErrorOutOfMemory;
    }
    if (c.compare_exchange_weak(v, v - 1)) {
      if (v == 0) {
        g_fault_fired++;
        return hipErrorOutOfMemory;
      }
      return hipSuccess;
    }
  }
  return hipSuccess;
}

size_t round_size(size_t b) {
  if (b <= 256) return 256;
  if (b <= ((size_t)1 << 20)) {  // next power of two
    size_t r = 256;
    while (r < b) r <<= 1;
    return r;
  }
  return (b + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);  // 2 MiB granules
}

// smallest cached block that fits without wasting more than half of it (+ 2 MiB); -1 if none
long best_fit(const std::vector<Block>& v, size_t need) {
  long best = -1;
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i].bytes >= need && v[i].bytes <= 2 * need + ((size_t)2 << 20) && (best < 0 || v[i].bytes < v[(size_t)best].bytes))
      best = (long)i;
  return best;
}

// under A.mu, current device = dev.  The cache of an idle / dead stream moves to the orphan list (its blocks are
// unused: everything queued behind them has completed, or the stream no longer exists), what exceeds the budget is
// released.  Dead streams always, idle ones oldest first while there are too many caches (or all of them: `all`).
void evict(Arena& A, int dev, bool all, std::vector<void*>* to_free) {
  std::vector<std::pair<uint64_t, std::pair<int, hipStream_t>>> idle_live;
  auto orphanize = [&](StreamCache& c) {
    for (Block& b : c.free) {
      A.orphan[dev].push_back(b);
      A.orphan_bytes[dev] += b.bytes;
    }
    c.free.clear();
    c.free_bytes = 0;
  };
  for (auto it = A.caches.begin(); it != A.caches.end();) {
    if (it->first.first != dev) {
      ++it;
      continue;
    }
    const hipError_t q = hipStreamQuery(it->first.second);
    if (q == hipErrorNotReady) {
      ++it;
      continue;
    }
    if (q != hipSuccess) {  // the handle is not a stream any more
      (void)hipGetLastError();
      orphanize(it->second);
      A.n_evicted_dead++;
      it = A.caches.erase(it);
      continue;
    }
    idle_live.push_back({it->second.last_use, it->first});
    ++it;
  }
  std::sort(idle_live.begin(), idle_live.end());
  for (size_t i = 0; i < idle_live.size() && (all || A.caches.size() >= kMaxStreamCaches); ++i) {
    auto it = A.caches.find(idle_live[i].second);
    orphanize(it->second);
    A.n_evicted_idle++;
    A.caches.erase(it);
  }
  // orphans beyond the budget go back to the driver (largest first)
  std::vector<Block>& o = A.orphan[dev];
  std::sort(o.begin(), o.end(), [](const Block& x, const Block& y) { return x.bytes < y.bytes; });
  while (!o.empty() && A.orphan_bytes[dev] > g_pool_keep_bytes) {
    to_free->push_back(o.back().p);
    A.orphan_bytes[dev] -= o.back().bytes;
    o.pop_back();
    A.n_released++;
  }
}
}  // namespace

hipError_t fault_gate() {
  g_alloc_calls++;
  return countdown(g_fault_alloc);
}
hipError_t fault_gate_driver() { return countdown(g_fault_driver); }
void set_fault_alloc_after(int n) { g_fault_alloc = n < 0 ? -1 : n; }
void set_fault_driver_oom(int n) { g_fault_driver = n < 0 ? -1 : n; }
void set_fault_persist_oom(int n) { g_fault_persist = n < 0 ? -1 : n; }
void set_fault_alloc_sticky(int v) { g_fault_sticky = v ? 1 : 0; }
long get_fault_alloc_after() { return g_fault_alloc.load(); }
unsigned long get_fault_fired() { return g_fault_fired.load(); }
unsigned long get_alloc_calls() { return g_alloc_calls.load(); }

namespace {
// under A.mu: blocks of the pending list whose event has completed (to be hipFree'd outside the lock)
void reap_pending(Arena& A, std::vector<void*>* to_free, std::vector<hipEvent_t>* evs) {
  for (size_t i = 0; i < A.pending.size();) {
    if (hipEventQuery(A.pending[i].ev) == hipErrorNotReady) {
      ++i;
      continue;
    }
    (void)hipGetLastError();
    for (Block& b : A.pending[i].blocks) to_free->push_back(b.p);
    A.n_released += A.pending[i].blocks.size();
    evs->push_back(A.pending[i].ev);
    A.pending[i] = std::move(A.pending.back());
    A.pending.pop_back();
  }
}
// the driver's allocation of one arena block, through the "fault_driver_oom" gate
hipError_t driver_malloc(void** p, size_t bytes) {
  hipError_t e = fault_gate_driver();
  if (e != hipSuccess) {
    *p = nullptr;
    return e;
  }
  return (hipMalloc)(p, bytes);
}
}  // namespace

namespace {
// Every cached block of `dev`, whatever stream holds it, goes back to the driver: taken out of the caches first (nobody
// can be handed one any more), then the device is synchronised (whatever was queued behind them has run), then freed.
// The caller has made `dev` current.
void release_device(Arena& A, int dev) {
  std::vector<void*> all;
  std::vector<hipEvent_t> pev;
  {
    std::lock_guard<std::mutex> lk(A.mu);
    for (auto& kv : A.caches)
      if (kv.first.first == dev) {
        for (Block& b : kv.second.free) all.push_back(b.p);
        kv.second.free.clear();
        kv.second.free_bytes = 0;
      }
    for (Block& b : A.orphan[dev]) all.push_back(b.p);
    A.orphan[dev].clear();
    A.orphan_bytes[dev] = 0;
    for (size_t i = 0; i < A.pending.size();)
      if (A.pending[i].dev == dev) {
        for (Block& b : A.pending[i].blocks) all.push_back(b.p);
        pev.push_back(A.pending[i].ev);
        A.pending[i] = std::move(A.pending.back());
        A.pending.pop_back();
      } else {
        ++i;
      }
  }
  (void)hipDeviceSynchronize();
  for (void* q : all) (void)hipFree(q);
  for (hipEvent_t ev : pev) (void)hipEventDestroy(ev);
  (void)hipGetLastError();
}
}  // namespace

// Index storage, result workspaces, tables: memory that stays with a handle (hipMalloc(...) in the library's sources is
// this, through gated_malloc).  The arena may be sitting on most of the device -- a live stream keeps up to a quarter of
// it cached -- and a plain hipMalloc knows nothing of that: refused, it gives the arena's cached and pending blocks of
// the device back and tries once more, like the arena's own path does ("arena_oom_retry_persistent" counts).
hipError_t persistent_malloc(void** p, size_t bytes) {
  hipError_t e = countdown(g_fault_persist);
  if (e == hipSuccess) e = (hipMalloc)(p, bytes);
  if (e != hipErrorOutOfMemory) return e;
  (void)hipGetLastError();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipErrorOutOfMemory;
  Arena& A = arena();
  {
    std::lock_guard<std::mutex> lk(A.mu);
    A.n_oom_retry_persistent++;
  }
  release_device(A, dev);
  return (hipMalloc)(p, bytes);
}

void set_pool_live_keep_mb(int mb) { g_pool_live_keep_bytes = mb < 0 ? ~0ull : (uint64_t)mb << 20; }  // 0 = automatic
void set_pool_keep_mb(int mb) { g_pool_keep_bytes = mb < 0 ? ~0ull : (uint64_t)mb << 20; }

hipError_t malloc_async(void** p, size_t bytes, hipStream_t s) {
  hipError_t e = fault_gate();
  if (e != hipSuccess) {
    *p = nullptr;
    return e;
  }
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  (void)live_budget(dev);  // (cached per ordinal before the lock below: free_async reads it under the lock)
  Arena& A = arena();
  const size_t need = round_size(bytes);
  std::vector<void*> to_free;
  std::vector<hipEvent_t> evs;
  {
    std::lock_guard<std::mutex> lk(A.mu);
    A.n_malloc++;
    if (!A.pending.empty()) reap_pending(A, &to_free, &evs);
    auto it = A.caches.find({dev, s});
    if (it == A.caches.end()) {
      if (A.caches.size() >= kMaxStreamCaches) evict(A, dev, false, &to_free);
      it = A.caches.emplace(std::make_pair(dev, s), StreamCache{}).first;
    }
    StreamCache& c = it->second;
    c.last_use = ++A.clock;
    long i = best_fit(c.free, need);
    if (i >= 0) {
      Block b = c.free[(size_t)i];
      c.free.erase(c.free.begin() + i);
      c.free_bytes -= b.bytes;
      A.live[b.p] = LiveInfo{b.bytes, dev};
      A.n_reuse++;
      *p = b.p;
    } else if ((i = best_fit(A.orphan[dev], need)) >= 0) {
      std::vector<Block>& o = A.orphan[dev];
      Block b = o[(size_t)i];
      o.erase(o.begin() + i);
      A.orphan_bytes[dev] -= b.bytes;
      A.live[b.p] = LiveInfo{b.bytes, dev};
      A.n_adopt++;
      *p = b.p;
    } else {
      *p = nullptr;
    }
  }
  for (void* q : to_free) (void)hipFree(q);
  for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
  if (*p) return hipSuccess;
  e = driver_malloc(p, need);
  if (e == hipErrorOutOfMemory) {
    // 1. the calling stream's own cache: its blocks fit nothing (best_fit refused them), and after a synchronise of
    //    this stream nothing uses them
    (void)hipGetLastError();
    std::vector<Block> mine;
    {
      std::lock_guard<std::mutex> lk(A.mu);
      auto it = A.caches.find({dev, s});
      if (it != A.caches.end()) {
        mine.swap(it->second.free);
        it->second.free_bytes = 0;
      }
      A.n_oom_retry_stream++;
    }
    (void)hipStreamSynchronize(s);
    for (Block& b : mine) (void)hipFree(b.p);
    e = driver_malloc(p, need);
  }
  if (e == hipErrorOutOfMemory) {
    // 2. every cached block of the device, whatever stream holds it
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(A.mu);
      A.n_oom_retry_device++;
    }
    release_device(A, dev);
    e = driver_malloc(p, need);
  }
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(A.mu);
  A.live[*p] = LiveInfo{need, dev};
  return hipSuccess;
}

hipError_t free_async(void* p, hipStream_t s) {
  if (!p) return hipSuccess;
  Arena& A = arena();
  {
    std::lock_guard<std::mutex> lk(A.mu);
    auto it = A.live.find(p);
    if (it == A.live.end()) return hipErrorInvalidValue;  // not a block of this arena
    const LiveInfo info = it->second;
    A.live.erase(it);
    StreamCache& c = A.caches[{info.dev, s}];
    c.last_use = ++A.clock;
    c.free.push_back(Block{p, info.bytes});
    c.free_bytes += info.bytes;
    // A live stream keeps what it has used (the next call of the same caller finds its buffers mapped) up to the
    // budget of "pool_live_keep_mb"; beyond it the blocks freed longest ago leave the cache.  Work queued on s may
    // still use them, so they wait in the pending list behind ONE event recorded now.
    const uint64_t budget = live_budget(info.dev);
    if (c.free_bytes > budget) {
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(ev, s) == hipSuccess) {
          Arena::Pending pd{{}, 0, info.dev, ev};
          size_t k = 0;
          while (k < c.free.size() && c.free_bytes > budget) {
            pd.blocks.push_back(c.free[k]);
            pd.bytes += c.free[k].bytes;
            c.free_bytes -= c.free[k].bytes;
            A.n_trimmed_live++;
            ++k;
          }
          c.free.erase(c.free.begin(), c.free.begin() + (long)k);
          A.pending.push_back(std::move(pd));
        } else {
          (void)hipEventDestroy(ev);
        }
      }
      (void)hipGetLastError();
    }
  }
  return hipSuccess;
}

void stream_destroy(hipStream_t s) {
  if (!s) return;
  (void)hipStreamSynchronize(s);
  int dev = 0;
  std::vector<void*> to_free;
  if (hipGetDevice(&dev) == hipSuccess) {
    Arena& A = arena();
    std::lock_guard<std::mutex> lk(A.mu);
    auto it = A.caches.find({dev, s});
    if (it != A.caches.end()) {  // the stream is idle: its blocks may serve any stream from now on
      for (Block& b : it->second.free) {
        A.orphan[dev].push_back(b);
        A.orphan_bytes[dev] += b.bytes;
      }
      A.caches.erase(it);
      std::vector<Block>& o = A.orphan[dev];
      std::sort(o.begin(), o.end(), [](const Block& x, const Block& y) { return x.bytes < y.bytes; });
      while (!o.empty() && A.orphan_bytes[dev] > g_pool_keep_bytes) {
        to_free.push_back(o.back().p);
        A.orphan_bytes[dev] -= o.back().bytes;
        o.pop_back();
        A.n_released++;
      }
    }
  }
  (void)hipStreamDestroy(s);
  for (void* q : to_free) (void)hipFree(q);
}

// cbh_trim (the caller has synchronised the device): every cached block of `device` goes back to the driver, caches
// of streams that are gone are dropped, ROCm pools (modes 0 / 1) are trimmed
int trim_pools(int device, unsigned long long* released_bytes) {
  Arena& A = arena();
  std::vector<void*> to_free;
  std::vector<hipEvent_t> evs;
  unsigned long long rel = 0;
  {
    std::lock_guard<std::mutex> lk(A.mu);
    const uint64_t keep = g_pool_keep_bytes;
    g_pool_keep_bytes = 0;
    evict(A, device, true, &to_free);
    g_pool_keep_bytes = keep;
    reap_pending(A, &to_free, &evs);  // (the caller synchronised the device: every event has completed)
  }
  // (sizes of what evict() released are not tracked per pointer: measure through the driver)
  size_t f0 = 0, f1 = 0, tot = 0;
  (void)hipMemGetInfo(&f0, &tot);
  for (void* q : to_free) (void)hipFree(q);
  for (hipEvent_t ev : evs) (void)hipEventDestroy(ev);
  (void)hipMemGetInfo(&f1, &tot);
  rel = f1 > f0 ? f1 - f0 : 0;
  if (released_bytes) *released_bytes = rel;
  return CBH_OK;
}

// cbh_get_tuning("arena_<name>"): counters and sizes of the scratch arena, all devices together
int arena_counter(const char* name, long long* value) {
  Arena& A = arena();
  std::lock_guard<std::mutex> lk(A.mu);
  unsigned long long cached = 0, pend = 0, live = 0;
  for (auto& kv : A.caches) cached += kv.second.free_bytes;
  for (auto& kv : A.orphan_bytes) cached += kv.second;
  for (auto& b : A.pending) pend += b.bytes;
  for (auto& kv : A.live) live += kv.second.bytes;
  if (!strcmp(name, "cached_bytes")) return *value = (long long)cached, CBH_OK;
  if (!strcmp(name, "pending_bytes")) return *value = (long long)pend, CBH_OK;
  if (!strcmp(name, "live_bytes")) return *value = (long long)live, CBH_OK;
  if (!strcmp(name, "live_blocks")) return *value = (long long)A.live.size(), CBH_OK;
  if (!strcmp(name, "trimmed_live")) return *value = (long long)A.n_trimmed_live, CBH_OK;
  if (!strcmp(name, "oom_retry_stream")) return *value = (long long)A.n_oom_retry_stream, CBH_OK;
  if (!strcmp(name, "oom_retry_device")) return *value = (long long)A.n_oom_retry_device, CBH_OK;
  if (!strcmp(name, "oom_retry_persistent")) return *value = (long long)A.n_oom_retry_persistent, CBH_OK;
  if (!strcmp(name, "released")) return *value = (long long)A.n_released, CBH_OK;
  return CBH_E_INVAL;
}

}  // namespace cbh

#include "cbh_index.h"

namespace cbh {  // sharded.hip
int sharded_load(cbh_idx64* idx, const void* hashes, const void* ids, size_t n, bool on_device, hipStream_t stream);
int sharded_find_one(cbh_idx64* idx, uint64_t q, int thresh, std::vector<cbh_record>* recs, bool* fits);
int sharded_add(cbh_idx64* idx, const uint64_t* hashes, const uint32_t* ids, size_t n);
int sharded_remove(cbh_idx64* idx, const uint32_t* ids, size_t n, int zero_hash);
}  // namespace cbh

namespace cbh {
namespace {
// window = the first 4096 bytes of buf (num_records); lane l reads one dword at voffset 4 l (+ 2048 for the upper lanes)
// with scalar offsets so[i]; out[i * 64 + l] = what came back
__global__ void k_selftest_buffer_range(const unsigned* buf, const unsigned* so, unsigned n_so, unsigned* out) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, 4096, 0x27000);
  const unsigned lane = threadIdx.x & 63u;
  const unsigned voff = 4u * lane + (lane >= 32u ? 2048u : 0u);
  for (unsigned i = 0; i < n_so; ++i) {
    const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)so[i]);
    out[i * 64u + lane] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)s, 0);
  }
}
}  // namespace
}  // namespace cbh

extern "C" {

int cbh_version(void) { return CBH_VERSION; }

int cbh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int usable = 0;
  for (int d = 0; d < n; ++d)
    if (device_usable(d)) ++usable;
  return usable;
}

uint32_t cbh_usable_device_mask(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  uint32_t mask = 0;
  for (int d = 0; d < n && d < 32; ++d)
    if (device_usable(d)) mask |= 1u << d;
  return mask;
}

const char* cbh_strerror(int code) {
  switch (code) {
    case CBH_OK: return "ok";
    case CBH_E_INVAL: return "invalid argument";
    case CBH_E_UNSUPPORTED: return "unsupported by this build";
    case CBH_E_NODEVICE: return "no usable gfx950 device";
    case CBH_E_NOMEM: return "out of memory";
    case CBH_E_HIP: return "HIP runtime error";
    case CBH_E_OVERFLOW: return "result does not fit the record buffer";
    case CBH_E_NOTLOADED: return "index not loaded";
    default: return "unknown error";
  }
}

const char* cbh_last_error(void) { return t_last_error.c_str(); }

int cbh_last_error_code(void) { return t_last_code; }
void cbh_clear_error(void) { clear_last_error(); }

int cbh_trim(int device, unsigned long long* released_bytes) {
  if (released_bytes) *released_bytes = 0;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  (void)hipDeviceSynchronize();  // every queued hipFreeAsync has happened: what is free is really free
  return trim_pools(device, released_bytes);
}

/* ---- hashing ---------------------------------------------------------------------------- */

int cbh_dcthash_batch_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride, void* d_out, int device, void* stream) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (n && (!d_imgs || !d_out)) return CBH_E_INVAL;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_dcthash((const uint8_t*)d_imgs, n, w, h, row_stride, img_stride,
                          (uint64_t*)d_out, s);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_dcthash_tiles_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride, void* d_out, void* d_tiles, int device, void* stream) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (n && (!d_imgs || !d_out || !d_tiles)) return CBH_E_INVAL;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_dcthash((const uint8_t*)d_imgs, n, w, h, row_stride, img_stride,
                          (uint64_t*)d_out, s, (uint8_t*)d_tiles);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_dcthash_batch(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride,
                      size_t img_stride, uint64_t* out, int device) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!imgs || !out || w <= 0 || h <= 0 || row_stride < (size_t)w) return CBH_E_INVAL;
  if (img_stride < (size_t)(h - 1) * row_stride + (size_t)w && n > 1) return CBH_E_INVAL;
  if (w > 8192 || h > 8192) return CBH_E_UNSUPPORTED;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  // decoded tiles are staged in chunks of <= 256 MiB; strides are preserved on the device
  const size_t span1 = (size_t)(h - 1) * row_stride + (size_t)w;
  size_t per_chunk = std::max<size_t>(1, ((size_t)256 << 20) / std::max<size_t>(img_stride, span1));
  per_chunk = std::min(per_chunk, n);
  const size_t chunk_bytes = (per_chunk - 1) * img_stride + span1;
  uint8_t* d_imgs = nullptr;
  uint64_t* d_out = nullptr;
  hipStream_t s = nullptr;
  int rc = CBH_OK;
  hipError_t e;
  if ((e = hipMalloc(&d_imgs, chunk_bytes)) != hipSuccess ||
      (e = hipMalloc(&d_out, per_chunk * sizeof(uint64_t))) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) {
    set_last_error("dcthash_batch setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  for (size_t i0 = 0; rc == CBH_OK && i0 < n; i0 += per_chunk) {
    const size_t m = std::min(per_chunk, n - i0);
    const size_t bytes = (m - 1) * img_stride + span1;
    if ((e = hipMemcpyAsync(d_imgs, imgs + i0 * img_stride, bytes, hipMemcpyHostToDevice, s)) !=
        hipSuccess) {
      set_last_error("hipMemcpyAsync(H2D images)", e);
      rc = CBH_E_HIP;
      break;
    }
    rc = launch_dcthash(d_imgs, m, w, h, row_stride, img_stride, d_out, s);
    if (rc) break;
    if ((e = hipMemcpyAsync(out + i0, d_out, m * sizeof(uint64_t), hipMemcpyDeviceToHost, s)) !=
            hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      set_last_error("dcthash_batch D2H", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) cbh::stream_destroy(s);
  if (d_imgs) (void)hipFree(d_imgs);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

/* ---- Media::makeKeyPointHashes: src/media.cpp:874-923 ---------------------------------------- */

long long cbh_keypoint_rects(int cols, int rows, const float* kp, size_t nkp, int32_t* rects) {
  if (nkp && (!kp || !rects)) return CBH_E_INVAL;
  long long n = 0;
  for (size_t i = 0; i < nkp; ++i) {
    const float size = kp[3 * i + 2];
    if (!(size >= 31)) continue;  // "if resulting rectangle is too small dct hash is worthless" (:885)
    const float x0 = kp[3 * i], y0 = kp[3 * i + 1];
    const float x1 = x0 + size, y1 = y0 + size;
    if (x0 > 0 && y0 > 0 && x1 < cols - 2 && y1 < rows - 2) {
      rects[3 * n] = (int32_t)std::floor(x0);
      rects[3 * n + 1] = (int32_t)std::floor(y0);
      rects[3 * n + 2] = (int32_t)std::ceil(size);
      ++n;
    }
  }
  return n;
}

int cbh_keypoint_hashes_dev(void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                            const uint32_t* img_h, const uint32_t* img_row_stride, const float* kp,
                            const uint32_t* kp_first, void* d_out, uint32_t* out_first, int device, void* stream) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (!out_first) return CBH_E_INVAL;
  if (n == 0) {
    out_first[0] = 0;
    return CBH_OK;
  }
  if (!d_imgs || !img_off || !img_w || !img_h || !img_row_stride || !kp_first || (kp_first[n] && !kp) || !d_out)
    return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (kp_first[i + 1] < kp_first[i] || img_w[i] == 0 || img_h[i] == 0 || img_w[i] > 8192 || img_h[i] > 8192 ||
        img_row_stride[i] < img_w[i])
      return CBH_E_INVAL;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  return launch_keypoint_hashes((uint8_t*)d_imgs, n, img_off, img_w, img_h, img_row_stride, kp, kp_first,
                                (uint64_t*)d_out, out_first, (hipStream_t)stream);
}

namespace {

// stage the packed images, run k_rect_hashes, fetch hashes (and the modified images)
int rect_hashes_host(const uint8_t* imgs, size_t imgs_bytes, const std::vector<cbh::RectImageDesc>& images,
                     const std::vector<int>& rects, int write_back, uint64_t* out_hashes, uint8_t* imgs_after,
                     int device) {
  const size_t total = rects.size() / 4;
  if (total && !out_hashes) return CBH_E_INVAL;
  if (total == 0) {
    if (imgs_after) memcpy(imgs_after, imgs, imgs_bytes);
    return CBH_OK;
  }
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  uint8_t* d_imgs = nullptr;
  uint64_t* d_out = nullptr;
  hipStream_t s = nullptr;
  hipError_t e;
  int rc = CBH_OK;
  if ((e = hipMalloc(&d_imgs, imgs_bytes)) != hipSuccess ||
      (e = hipMalloc(&d_out, total * sizeof(uint64_t))) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMemcpyAsync(d_imgs, imgs, imgs_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) {
    set_last_error("rect hashes setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  if (rc == CBH_OK) rc = launch_rect_hashes(d_imgs, images, rects, write_back, d_out, s);
  if (rc == CBH_OK) {
    if ((e = hipMemcpyAsync(out_hashes, d_out, total * sizeof(uint64_t), hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (imgs_after && (e = hipMemcpyAsync(imgs_after, d_imgs, imgs_bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      set_last_error("rect hashes D2H", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) cbh::stream_destroy(s);
  if (d_imgs) (void)hipFree(d_imgs);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

int check_images(size_t n, size_t imgs_bytes, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
                 const uint32_t* img_row_stride) {
  for (size_t i = 0; i < n; ++i) {
    if (img_w[i] == 0 || img_h[i] == 0 || img_w[i] > 8192 || img_h[i] > 8192 || img_row_stride[i] < img_w[i])
      return CBH_E_INVAL;
    const unsigned long long end = img_off[i] + (unsigned long long)(img_h[i] - 1) * img_row_stride[i] + img_w[i];
    if (end > imgs_bytes) return CBH_E_INVAL;
  }
  return CBH_OK;
}

}  // namespace

int cbh_keypoint_hashes(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off,
                        const uint32_t* img_w, const uint32_t* img_h, const uint32_t* img_row_stride,
                        const float* kp, const uint32_t* kp_first, uint64_t* out_hashes, uint32_t* out_first,
                        uint8_t* imgs_after, int device) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (!out_first) return CBH_E_INVAL;
  if (n == 0) {
    out_first[0] = 0;
    return CBH_OK;
  }
  if (!imgs || !img_off || !img_w || !img_h || !img_row_stride || !kp_first || (kp_first[n] && !kp)) return CBH_E_INVAL;
  int rc = check_images(n, imgs_bytes, img_off, img_w, img_h, img_row_stride);
  if (rc) return rc;
  for (size_t i = 0; i < n; ++i)
    if (kp_first[i + 1] < kp_first[i]) return CBH_E_INVAL;
  const size_t nkp = kp_first[n];
  if (nkp && !out_hashes) return CBH_E_INVAL;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  uint8_t* d_imgs = nullptr;
  uint64_t* d_out = nullptr;
  hipStream_t s = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&d_imgs, imgs_bytes)) != hipSuccess ||
      (e = hipMalloc(&d_out, std::max<size_t>(nkp, 1) * sizeof(uint64_t))) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMemcpyAsync(d_imgs, imgs, imgs_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) {
    set_last_error("keypoint_hashes setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  if (rc == CBH_OK)
    rc = launch_keypoint_hashes(d_imgs, n, img_off, img_w, img_h, img_row_stride, kp, kp_first, d_out, out_first, s);
  if (rc == CBH_OK) {
    const size_t total = out_first[n];
    if ((total && (e = hipMemcpyAsync(out_hashes, d_out, total * sizeof(uint64_t), hipMemcpyDeviceToHost, s)) !=
                      hipSuccess) ||
        (imgs_after && (e = hipMemcpyAsync(imgs_after, d_imgs, imgs_bytes, hipMemcpyDeviceToHost, s)) != hipSuccess) ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      set_last_error("keypoint_hashes D2H", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) cbh::stream_destroy(s);
  if (d_imgs) (void)hipFree(d_imgs);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

int cbh_dcthash_rects(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off,
                      const uint32_t* img_w, const uint32_t* img_h, const uint32_t* img_row_stride,
                      const int32_t* rects, const uint32_t* rect_first, int in_place, uint64_t* out_hashes,
                      uint8_t* imgs_after, int device) {
  if (!device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!imgs || !img_off || !img_w || !img_h || !img_row_stride || !rect_first || (rect_first[n] && !rects))
    return CBH_E_INVAL;
  int rc = check_images(n, imgs_bytes, img_off, img_w, img_h, img_row_stride);
  if (rc) return rc;
  if (rect_first[0] != 0) return CBH_E_INVAL;
  std::vector<cbh::RectImageDesc> images(n);
  for (size_t i = 0; i < n; ++i) {
    if (rect_first[i + 1] < rect_first[i]) return CBH_E_INVAL;
    images[i] = cbh::RectImageDesc{img_off[i], (int)img_w[i], (int)img_h[i], img_row_stride[i], rect_first[i],
                                   rect_first[i + 1] - rect_first[i]};
  }
  std::vector<int> r(rects, rects + 4 * (size_t)rect_first[n]);
  return rect_hashes_host(imgs, imgs_bytes, images, r, in_place ? 1 : 0, out_hashes, imgs_after, device);
}

/* ---- DctHashIndex ----------------------------------------------------------------------- */

cbh_idx64* cbh_idx64_create(int device) {
  clear_last_error();
  if (!device_usable(device)) return (cbh_idx64*)fail_handle(CBH_E_NODEVICE, "cbh_idx64_create: no usable gfx950 device at that ordinal");
  cbh_idx64* idx = new (std::nothrow) cbh_idx64;
  if (!idx) return (cbh_idx64*)fail_handle(CBH_E_NOMEM, "cbh_idx64_create: host allocation failed");
  idx->device = device;
  return idx;
}

void cbh_idx64_destroy(cbh_idx64* idx) {
  if (!idx) return;
  cbh::combiner_drop(idx);  // combine.hip: the queue of cbh_*_find_coalesced callers
  DeviceGuard g(idx->device);
  for (Workspace* w : idx->ws_free) {
    w->release();
    delete w;
  }
  if (idx->d_hashes) (void)hipFree(idx->d_hashes);
  if (idx->d_ids) (void)hipFree(idx->d_ids);
  if (idx->coalescer.load()) coalescer_free(idx->coalescer.load());
  if (idx->shards) shardset_free(idx->shards);
  delete idx;
}

static int idx_append(cbh_idx64* idx, const void* hashes, const void* ids, size_t n,
                      hipMemcpyKind kind, hipStream_t s) {
  if (n == 0) return CBH_OK;
  if (!hashes || !ids) return CBH_E_INVAL;
  if (idx->n + n > 0xfffffff0ull) return CBH_E_INVAL;
  int rc = idx->reserve(idx->n + n);
  if (rc) return rc;
  CBH_HIP(hipMemcpyAsync(idx->d_hashes + idx->n, hashes, n * sizeof(uint64_t), kind, s));
  CBH_HIP(hipMemcpyAsync(idx->d_ids + idx->n, ids, n * sizeof(uint32_t), kind, s));
  CBH_HIP(hipStreamSynchronize(s));
  idx->n += n;
  idx->generation++;
  {
    std::lock_guard<std::mutex> lk(idx->tree_mu);
    idx->tree_valid = false;  // the HammingTree shape depends on the contents
  }
  return CBH_OK;
}

int cbh_idx64_load(cbh_idx64* idx, const uint64_t* hashes, const uint32_t* ids, size_t n) {
  if (!idx) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  idx->n = 0;  // load() on a loaded index is a no-op in the reference (:75); here it reloads
  idx->generation++;
  idx->loaded = true;
  if (idx->shards) {
    std::lock_guard<std::mutex> lk(idx->tree_mu);
    idx->tree_valid = false;
    return sharded_load(idx, hashes, ids, n, false, nullptr);
  }
  return idx_append(idx, hashes, ids, n, hipMemcpyHostToDevice, nullptr);
}

int cbh_idx64_load_dev(cbh_idx64* idx, const void* d_hashes, const void* d_ids, size_t n,
                       void* stream) {
  if (!idx) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  idx->n = 0;
  idx->generation++;
  idx->loaded = true;
  if (idx->shards) {
    std::lock_guard<std::mutex> lk(idx->tree_mu);
    idx->tree_valid = false;
    return sharded_load(idx, d_hashes, d_ids, n, true, (hipStream_t)stream);
  }
  return idx_append(idx, d_hashes, d_ids, n, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int cbh_idx64_is_loaded(const cbh_idx64* idx) { return idx && idx->loaded; }

int cbh_idx64_add(cbh_idx64* idx, const uint64_t* hashes, const uint32_t* ids, size_t n) {
  if (!idx) return CBH_E_INVAL;
  if (!idx->loaded) return CBH_E_NOTLOADED;  // "it is an error to call this if isLoaded() is false"
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  if (idx->shards) {
    int rc = sharded_add(idx, hashes, ids, n);
    if (!rc && n) {
      idx->generation++;
      std::lock_guard<std::mutex> lk(idx->tree_mu);
      idx->tree_valid = false;
    }
    return rc;
  }
  return idx_append(idx, hashes, ids, n, hipMemcpyHostToDevice, nullptr);
}

static int idx_remove(cbh_idx64* idx, const uint32_t* ids, size_t n, int zero_hash);

int cbh_idx64_remove(cbh_idx64* idx, const uint32_t* ids, size_t n) {
  if (idx) {
    std::lock_guard<std::mutex> lk(idx->tree_mu);
    idx->tree_valid = false;
  }
  return idx_remove(idx, ids, n, 1);
}

int cbh_idx64_remove_ids_only(cbh_idx64* idx, const uint32_t* ids, size_t n) {
  return idx_remove(idx, ids, n, 0);
}

static int idx_remove(cbh_idx64* idx, const uint32_t* ids, size_t n, int zero_hash) {
  if (!idx) return CBH_E_INVAL;
  if (!idx->loaded) return CBH_OK;  // `if (!isLoaded()) return;` (:176)
  if (n == 0 || idx->n == 0) return CBH_OK;
  if (!ids) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  idx->generation++;
  if (idx->shards) return sharded_remove(idx, ids, n, zero_hash);
  std::vector<uint32_t> rm(ids, ids + n);
  std::sort(rm.begin(), rm.end());
  rm.erase(std::unique(rm.begin(), rm.end()), rm.end());
  uint32_t* d_rm = nullptr;
  CBH_HIP(hipMalloc(&d_rm, rm.size() * sizeof(uint32_t)));
  int rc = CBH_OK;
  hipError_t e = hipMemcpy(d_rm, rm.data(), rm.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_last_error("hipMemcpy(remove ids)", e);
    rc = CBH_E_HIP;
  }
  if (!rc) rc = launch_remove_ids(idx->d_hashes, idx->d_ids, idx->n, d_rm, rm.size(), nullptr, zero_hash);
  if (!rc && (e = hipStreamSynchronize(nullptr)) != hipSuccess) {
    set_last_error("remove sync", e);
    rc = CBH_E_HIP;
  }
  (void)hipFree(d_rm);
  return rc;
}

size_t cbh_idx64_count(const cbh_idx64* idx) { return idx ? idx->n : 0; }

size_t cbh_idx64_memory_usage(const cbh_idx64* idx) {
  return idx ? (sizeof(uint64_t) + sizeof(uint32_t)) * idx->n : 0;
}

int cbh_idx64_download(const cbh_idx64* idx, uint64_t* hashes, uint32_t* ids, size_t cap) {
  if (!idx) return CBH_E_INVAL;
  size_t m = std::min(cap, idx->n);
  if (m == 0) return CBH_OK;
  if (idx->shards) return sharded_download(idx, hashes, ids, m);
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  if (hashes) CBH_HIP(hipMemcpy(hashes, idx->d_hashes, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (ids) CBH_HIP(hipMemcpy(ids, idx->d_ids, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return CBH_OK;
}

int cbh_idx64_media_ids(const cbh_idx64* idx, uint32_t* out, size_t cap, size_t* n_out) {
  if (!idx || !n_out) return CBH_E_INVAL;
  // bookkeeping, not a hot path: one download + host filter (reference: a loop over the SoA)
  std::vector<uint64_t> h(idx->n);
  std::vector<uint32_t> id(idx->n);
  int rc = cbh_idx64_download(idx, h.data(), id.data(), idx->n);
  if (rc) return rc;
  size_t m = 0;
  for (size_t i = 0; i < idx->n; ++i)
    if (h[i] != 0) {
      if (out && m < cap) out[m] = id[i];
      ++m;
    }
  *n_out = m;
  return CBH_OK;
}

cbh_idx64* cbh_idx64_slice(const cbh_idx64* idx, const uint32_t* ids, size_t n) {
  clear_last_error();
  if (!idx || !idx->loaded)  // Q_ASSERT(isLoaded()) (:223)
    return (cbh_idx64*)fail_handle(CBH_E_INVAL, "cbh_idx64_slice: the index is not loaded");
  std::vector<uint64_t> h(idx->n);
  std::vector<uint32_t> id(idx->n);
  if (int rc = cbh_idx64_download(idx, h.data(), id.data(), idx->n)) return (cbh_idx64*)fail_handle(rc, nullptr);
  std::vector<uint32_t> want(ids, ids + (ids ? n : 0));
  std::sort(want.begin(), want.end());
  std::vector<uint64_t> sh;
  std::vector<uint32_t> si;
  for (size_t i = 0; i < idx->n; ++i)
    if (std::binary_search(want.begin(), want.end(), id[i])) {
      sh.push_back(h[i]);
      si.push_back(id[i]);
    }
  // a slice of a sharded index is sharded the same way (Database::similarTo searches it like the whole index)
  cbh_idx64* out = idx->shards ? cbh_idx64_create_sharded(cbh_idx64_device_mask(idx), cbh_idx64_shards_per_device(idx))
                               : cbh_idx64_create(idx->device);
  if (!out) return nullptr;  // (the create call has set the code)
  if (int rc = cbh_idx64_load(out, sh.data(), si.data(), sh.size())) {
    cbh_idx64_destroy(out);
    return (cbh_idx64*)fail_handle(rc, nullptr);
  }
  return out;
}

int cbh_idx64_get_stats(const cbh_idx64* idx, cbh_stats* out) {
  if (!idx || !out) return CBH_E_INVAL;
  cbh_idx64* m = const_cast<cbh_idx64*>(idx);
  std::lock_guard<std::mutex> lk(m->stats_mu);
  *out = m->stats;
  return CBH_OK;
}

int cbh_idx64_reset_stats(cbh_idx64* idx) {
  if (!idx) return CBH_E_INVAL;
  std::lock_guard<std::mutex> lk(idx->stats_mu);
  idx->stats = cbh_stats{0, 0, 0.0};
  return CBH_OK;
}

int cbh_idx64_set_record_capacity(cbh_idx64* idx, size_t records) {
  if (!idx || records == 0) return CBH_E_INVAL;
  idx->rec_cap_default = records;
  for (int i = 0, r = cbh_idx64_shard_count(idx); idx->shards && i < r; ++i)  // a shard's block holds its share
    cbh_idx64_shard(idx, i)->rec_cap_default = std::max<size_t>(1024, records / (size_t)r);
  return CBH_OK;
}

int cbh_idx64_find(cbh_idx64* idx, uint64_t q, int thresh, cbh_match* out, size_t cap,
                   size_t* n_out) {
  if (!idx || !n_out || (cap && !out)) return CBH_E_INVAL;
  *n_out = 0;
  if (q == 0 || idx->n == 0 || thresh <= 0) return CBH_OK;  // null needle / empty tree (:196-205)
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  Workspace* ws = L.ws;
  rc = Workspace::grow(&ws->d_q, &ws->q_cap, 1);
  if (rc) return rc;
  // Fast path (the interactive -similar-to query, Engine::query): ONE kernel launch per shard, needle as a kernel argument,
  // matches straight into pinned host memory, completion by polling (cbh_internal.h: launch_find_one).  Larger results
  // (more than LoneBlock::kRecs matches) fall through to the general path below.
  {
    std::vector<cbh_record> recs;
    bool fits = false;
    if (idx->shards) {
      rc = sharded_find_one(idx, q, thresh, &recs, &fits);
      if (rc) return rc;
    } else {
      if ((rc = ws->ensure_lone())) return rc;
      const unsigned long long seq = ++ws->lone_seq;
      rc = launch_find_one(idx->d_hashes, idx->d_ids, idx->n, q, thresh, ws->d_lone, ws->h_lone, seq, ws->stream);
      if (rc) return rc;
      if ((rc = wait_find_one(ws->h_lone, seq, ws->stream))) return rc;
      const unsigned long long t = ws->h_lone->count;
      fits = t <= LoneBlock::kRecs;
      if (fits) recs.assign(ws->h_lone->recs, ws->h_lone->recs + t);
    }
    {
      std::lock_guard<std::mutex> lk(idx->stats_mu);
      idx->stats.scan_launches += 1;
      idx->stats.scan_pairs += (uint64_t)idx->n;
    }
    if (fits) {
      *n_out = recs.size();
      std::sort(recs.begin(), recs.end());
      const size_t m = std::min(recs.size(), cap);
      for (size_t i = 0; i < m; ++i) {
        out[i].id = CBH_REC_ID(recs[i]);
        out[i].score = CBH_REC_DIST(recs[i]);
      }
      return CBH_OK;
    }
  }
  rc = ws->ensure_records(Workspace::kFindRecs);  // 512 KB; only a result that overflows it grows the workspace (scan_all)
  if (rc) return rc;
  ws->h_small[Workspace::kSmallRecs] = (cbh_record)q;
  CBH_HIP(hipMemcpyAsync(ws->d_q, &ws->h_small[Workspace::kSmallRecs], sizeof q, hipMemcpyHostToDevice, ws->stream));
  unsigned long long total = 0;
  rc = scan_all(idx, ws, ws->d_q, 1, thresh, ws->stream, &total);
  if (rc) return rc;
  *n_out = (size_t)total;
  if (total == 0) return CBH_OK;
  std::vector<cbh_record> recs;
  if (total <= 65536) {  // small result: order on the host
    recs.resize((size_t)total);
    CBH_HIP(hipMemcpyAsync(recs.data(), ws->d_rec, total * sizeof(cbh_record),
                           hipMemcpyDeviceToHost, ws->stream));
    CBH_HIP(hipStreamSynchronize(ws->stream));
    std::sort(recs.begin(), recs.end());
  } else {
    if ((rc = ws->ensure_sort())) return rc;
    rc = launch_sort_records(ws->d_rec, ws->d_alt, (size_t)total, 1, ws->d_tmp, ws->tmp_bytes,
                             ws->stream);
    if (rc) return rc;
    recs.resize(std::min<size_t>((size_t)total, cap));
    if (!recs.empty())
      CBH_HIP(hipMemcpyAsync(recs.data(), ws->d_rec, recs.size() * sizeof(cbh_record),
                             hipMemcpyDeviceToHost, ws->stream));
    CBH_HIP(hipStreamSynchronize(ws->stream));
  }
  const size_t m = std::min(recs.size(), cap);
  for (size_t i = 0; i < m; ++i) {
    out[i].id = CBH_REC_ID(recs[i]);
    out[i].score = CBH_REC_DIST(recs[i]);
  }
  return CBH_OK;
}

static int find_batch_core(cbh_idx64* idx, Workspace* ws, const uint64_t* d_q, size_t nq,
                           int thresh, int k, cbh_match* d_out, uint32_t* d_counts,
                           hipStream_t s, unsigned long long* total,
                           const uint64_t* d_qmask = nullptr) {
  *total = 0;
  int rc = ws->ensure_records(std::max<size_t>(idx->rec_cap_default, 1024));  // (d_total must exist when n == 0)
  if (rc) return rc;
  if (idx->n && thresh > 0) {
    rc = scan_all(idx, ws, d_q, nq, thresh, s, total, 0, d_qmask);
    if (rc) return rc;
  } else {
    CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), s));
  }
  if (k <= kTopkMaxK && *total < ((unsigned long long)1 << 32)) {
    // K4 counting select (topk.hip): the workspace's { count, records } block is its input as it stands
    void* scratch = nullptr;
    unsigned* d_status = nullptr;
    const size_t ncap = std::min<size_t>(ws->rec_cap, (size_t)*total + 1);  // slots past the count are never read
    CBH_HIP(cbh::malloc_async(&scratch, topk_scratch_bytes(nq, ncap) + 16, s));
    d_status = (unsigned*)((char*)scratch + topk_scratch_bytes(nq, ncap));
    rc = topk_scratch_init(scratch, nq, s);
    if (!rc) rc = launch_records_topk(ws->d_total, 1, 0, ncap, nq, k, d_out, d_counts, d_status, scratch, s);
    (void)cbh::free_async(scratch, s);
    return rc;
  }
  if ((rc = ws->ensure_sort())) return rc;
  rc = launch_sort_records(ws->d_rec, ws->d_alt, (size_t)*total, nq, ws->d_tmp, ws->tmp_bytes, s);
  if (rc) return rc;
  return launch_select_records(ws->d_rec, (size_t)*total, nq, k, d_out, d_counts, s);
}

int cbh_idx64_find_batch(cbh_idx64* idx, const uint64_t* q, size_t nq, int thresh,
                         int max_per_query, cbh_match* out, uint32_t* counts) {
  return cbh_idx64_find_batch_masked(idx, q, nullptr, nq, thresh, max_per_query, out, counts);
}

int cbh_idx64_find_batch_masked(cbh_idx64* idx, const uint64_t* q, const uint64_t* qmask, size_t nq,
                                int thresh, int max_per_query, cbh_match* out, uint32_t* counts) {
  if (!idx || max_per_query < 0) return CBH_E_INVAL;
  if (nq == 0) return CBH_OK;
  if (!q || !counts || (max_per_query && !out) || nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  Workspace* ws = L.ws;
  const size_t k = (size_t)max_per_query;
  if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, nq))) return rc;
  if ((rc = Workspace::grow(&ws->d_out, &ws->out_cap, std::max<size_t>(1, nq * k)))) return rc;
  if ((rc = Workspace::grow(&ws->d_counts, &ws->counts_cap, nq))) return rc;
  CBH_HIP(hipMemcpyAsync(ws->d_q, q, nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  if (qmask) {
    if ((rc = Workspace::grow(&ws->d_qmask, &ws->qmask_cap, nq))) return rc;
    CBH_HIP(hipMemcpyAsync(ws->d_qmask, qmask, nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  }
  unsigned long long total = 0;
  rc = find_batch_core(idx, ws, ws->d_q, nq, thresh, max_per_query, ws->d_out, ws->d_counts,
                       ws->stream, &total, qmask ? ws->d_qmask : nullptr);
  if (rc) return rc;
  if (k)
    CBH_HIP(hipMemcpyAsync(out, ws->d_out, nq * k * sizeof(cbh_match), hipMemcpyDeviceToHost,
                           ws->stream));
  CBH_HIP(hipMemcpyAsync(counts, ws->d_counts, nq * sizeof(uint32_t), hipMemcpyDeviceToHost,
                         ws->stream));
  CBH_HIP(hipStreamSynchronize(ws->stream));
  return CBH_OK;
}

int cbh_idx64_find_batch_dev(cbh_idx64* idx, const void* d_q, size_t nq, int thresh,
                             int max_per_query, void* d_out, void* d_counts, uint64_t* total_out,
                             void* stream) {
  if (!idx || max_per_query < 0) return CBH_E_INVAL;
  if (total_out) *total_out = 0;
  if (nq == 0) return CBH_OK;
  if (!d_q || !d_counts || (max_per_query && !d_out) || nq > CBH_MAX_QUERIES_PER_CALL)
    return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : L.ws->stream;
  unsigned long long total = 0;
  rc = find_batch_core(idx, L.ws, (const uint64_t*)d_q, nq, thresh, max_per_query,
                       (cbh_match*)d_out, (uint32_t*)d_counts, s, &total);
  if (rc) return rc;
  CBH_HIP(hipStreamSynchronize(s));  // the workspace goes back to the pool: its buffers must be idle
  if (total_out) *total_out = total;
  return CBH_OK;
}

int cbh_idx64_scan_dev(cbh_idx64* idx, const void* d_q, size_t nq, int thresh, void* d_records,
                       size_t cap, void* d_total, void* stream) {
  if (!idx || !d_total || (cap && !d_records)) return CBH_E_INVAL;
  if (idx->shards) return CBH_E_UNSUPPORTED;  // the shard-local step: call it on cbh_idx64_shard(idx, i)
  if (nq == 0 || idx->n == 0) return CBH_OK;
  if (!d_q || nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_hamm64_scan(idx->d_hashes, idx->d_ids, idx->n, (const uint64_t*)d_q, nq, thresh,
                              (cbh_record*)d_records, cap, (unsigned long long*)d_total, s);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_records_topk_dev(const void* d_blocks, size_t n_blocks, size_t block_stride, size_t cap, size_t nq,
                         int max_per_query, void* d_out, void* d_counts, void* d_status, int device, void* stream) {
  if (nq == 0) return CBH_OK;
  if (!d_counts || !d_status || (max_per_query && !d_out) || max_per_query < 0 || (n_blocks && !d_blocks) ||
      nq > CBH_MAX_QUERIES_PER_CALL || n_blocks > 1024 || (n_blocks > 1 && block_stride < cap + 1) ||
      n_blocks * cap >= ((size_t)1 << 32))
    return CBH_E_INVAL;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  void* scratch = nullptr;  // stream-ordered, recycled by this stream's pool between calls
  hipError_t e = cbh::malloc_async(&scratch, topk_scratch_bytes(nq, n_blocks * cap), s);
  if (e != hipSuccess) {
    set_last_error("malloc_async(topk scratch)", e);
    return CBH_E_NOMEM;
  }
  int rc = topk_scratch_init(scratch, nq, s);
  if (!rc)
    rc = launch_records_topk((const unsigned long long*)d_blocks, (unsigned)n_blocks, block_stride, cap, nq,
                             max_per_query, (cbh_match*)d_out, (uint32_t*)d_counts, (unsigned*)d_status, scratch, s);
  (void)cbh::free_async(scratch, s);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_sort_records_dev(void* d_records, size_t n, size_t nq, int device, void* stream) {
  if (n < 2) return CBH_OK;
  if (!d_records) return CBH_E_INVAL;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  cbh_record* alt = nullptr;
  void* tmp = nullptr;
  const size_t tmp_bytes = sort_records_scratch_bytes(n);
  // stream-ordered scratch: no device-wide synchronisation, the pool recycles the blocks between calls
  CBH_HIP(cbh::malloc_async((void**)&alt, n * sizeof(cbh_record), s));
  hipError_t e = cbh::malloc_async(&tmp, tmp_bytes ? tmp_bytes : 16, s);
  if (e != hipSuccess) {
    (void)cbh::free_async(alt, s);
    set_last_error("cbh::malloc_async(sort scratch)", e);
    return CBH_E_NOMEM;
  }
  int rc = launch_sort_records((cbh_record*)d_records, alt, n, nq, tmp, tmp_bytes, s);
  (void)cbh::free_async(alt, s);
  (void)cbh::free_async(tmp, s);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_select_records_dev(const void* d_sorted_records, size_t n, size_t nq, int max_per_query,
                           void* d_out, void* d_counts, int device, void* stream) {
  if (nq == 0) return CBH_OK;
  if (!d_counts || (max_per_query && !d_out) || max_per_query < 0 || (n && !d_sorted_records))
    return CBH_E_INVAL;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_select_records((const cbh_record*)d_sorted_records, n, nq, max_per_query,
                                 (cbh_match*)d_out, (uint32_t*)d_counts, s);
  if (rc) return rc;
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

// every knob the library has (include/cbird_hip.h documents each): name -> setter
int cbh_set_tuning(const char* key, int value) {
  if (!key) return CBH_E_INVAL;
  static const struct {
    const char* name;
    void (*set)(int);
  } kKnobs[] = {
      {"scan_mfma", [](int v) { set_scan_mfma(v); }},
      {"scan_mfma_pre_max", [](int v) { set_scan_pre_max(v); }},
      {"scan_pre_rate_e9", [](int v) { set_scan_pre_rate(v); }},
      {"scan256_mfma", [](int v) { set_scan256_mfma(v); }},
      {"scan256_small", [](int v) { set_scan256_small(v); }},
      {"hash_mfma", [](int v) { (void)g_hash_mfma_set(v); }},
      {"hash_band_area", [](int v) { set_hash_band_area(v); }},
      {"hash_fuse", [](int v) { set_hash_fuse(v); }},
      {"hash_stream", [](int v) { set_hash_stream(v); }},
      {"kp_lds_side", [](int v) { set_kp_lds_side(v); }},
      {"kp_blur_side", [](int v) { set_kp_blur_side(v); }},
      {"color_fma", [](int v) { set_color_fma(v); }},
      {"color_create_chunk_mb", [](int v) { set_cd_chunk_mb(v); }},
      {"fdct_host_vote", [](int v) { g_fdct_host_vote = v; }},
      {"video_host_reduce", [](int v) { g_video_host_reduce = v; }},
      {"orb_retain_order", [](int v) { set_orb_retain_order(v); }},
      {"pool_keep_mb", [](int v) { set_pool_keep_mb(v); }},
      {"pool_live_keep_mb", [](int v) { set_pool_live_keep_mb(v); }},
      {"shard_force_rccl", [](int v) { set_shard_force_rccl(v); }},
      {"shard_exchange", [](int v) { set_shard_exchange(v); }},
      {"fault_alloc_after", [](int v) { set_fault_alloc_after(v); }},
      {"fault_alloc_sticky", [](int v) { set_fault_alloc_sticky(v); }},
      {"fault_driver_oom", [](int v) { set_fault_driver_oom(v); }},
      {"fault_persist_oom", [](int v) { set_fault_persist_oom(v); }},
      {"fault_rccl", [](int v) { set_fault_rccl(v); }},
  };
  if (!strcmp(key, "orb_retain_order") && value != 0 && value != 1) return CBH_E_INVAL;
  for (const auto& k : kKnobs)
    if (!strcmp(key, k.name)) {
      k.set(value);
      return CBH_OK;
    }
  return CBH_E_INVAL;
}

int cbh_get_tuning(const char* key, long long* value) {
  if (!key || !value) return CBH_E_INVAL;
  if (!strcmp(key, "fault_alloc_after")) return *value = get_fault_alloc_after(), CBH_OK;
  if (!strcmp(key, "fault_fired")) return *value = (long long)get_fault_fired(), CBH_OK;
  if (!strcmp(key, "alloc_calls")) return *value = (long long)get_alloc_calls(), CBH_OK;
  if (!strncmp(key, "arena_", 6)) return arena_counter(key + 6, value);
  if (!strcmp(key, "scan_pre_mask")) return *value = get_scan_pre_mask(), CBH_OK;
  if (!strcmp(key, "scan_probes")) return *value = get_scan_probes(), CBH_OK;
  if (!strcmp(key, "scan_joins")) return *value = get_scan_joins(), CBH_OK;
  if (!strcmp(key, "scan_probe_rate_e9")) return *value = get_scan_probe_rate_e9(), CBH_OK;
  if (!strcmp(key, "scan_probe_true_e9")) return *value = get_scan_probe_true_e9(), CBH_OK;
  return CBH_E_INVAL;
}

/* ---- measurement ------------------------------------------------------------------------ */

int cbh_idx64_time_scan_dev(cbh_idx64* idx, const void* d_q, size_t nq, int thresh,
                            void* d_records, size_t cap, void* d_total, int iters, float* ms_avg) {
  if (!idx || !ms_avg || iters <= 0 || !d_total) return CBH_E_INVAL;
  if (idx->shards) return CBH_E_UNSUPPORTED;  // time a shard: cbh_idx64_shard(idx, i)
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  hipStream_t s = L.ws->stream;
  hipEvent_t e0, e1;
  CBH_HIP(hipEventCreate(&e0));
  CBH_HIP(hipEventCreate(&e1));
  CBH_HIP(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s));
  CBH_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) {
    rc = launch_hamm64_scan(idx->d_hashes, idx->d_ids, idx->n, (const uint64_t*)d_q, nq, thresh,
                            (cbh_record*)d_records, cap, (unsigned long long*)d_total, s);
    if (rc) break;
  }
  CBH_HIP(hipEventRecord(e1, s));
  CBH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  CBH_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_avg = ms / (float)iters;
  return rc;
}

int cbh_selftest_buffer_range(int device, int* ok) {
  if (!ok) return CBH_E_INVAL;
  *ok = 0;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  const unsigned so[] = {0u, 1024u, 2044u, 4092u, 4096u, 5120u, 8192u, 1u << 20, 0x7ffffff0u};
  const unsigned n_so = sizeof so / sizeof so[0];
  std::vector<unsigned> host(2048), got(n_so * 64);
  for (unsigned i = 0; i < 2048; ++i) host[i] = i < 1024 ? 0x10000u + i : 0xDEADBEEFu;  // data | poison behind the window
  unsigned *d_buf = nullptr, *d_so = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(&d_buf, 8192);
  if (e == hipSuccess) e = hipMalloc(&d_so, sizeof so);
  if (e == hipSuccess) e = hipMalloc(&d_out, got.size() * 4);
  if (e == hipSuccess) e = hipMemcpy(d_buf, host.data(), 8192, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_so, so, sizeof so, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_buffer_range, dim3(1), dim3(64), 0, 0, d_buf, d_so, n_so, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
  for (void* p : {(void*)d_buf, (void*)d_so, (void*)d_out})
    if (p) (void)hipFree(p);
  CBH_HIP(e);
  bool good = true;
  for (unsigned i = 0; i < n_so; ++i)
    for (unsigned l = 0; l < 64; ++l) {
      const unsigned long long off = (unsigned long long)so[i] + 4u * l + (l >= 32 ? 2048u : 0u);
      const unsigned want = off + 4 <= 4096 ? 0x10000u + (unsigned)(off / 4) : 0u;
      good = good && got[i * 64 + l] == want;
    }
  *ok = good ? 1 : 0;
  return CBH_OK;
}

int cbh_time_dcthash_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                         size_t img_stride, void* d_out, int device, int iters, float* ms_avg) {
  if (!ms_avg || iters <= 0) return CBH_E_INVAL;
  if (!device_usable(device)) return CBH_E_NODEVICE;
  DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s;
  CBH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CBH_HIP(hipEventCreate(&e0));
  CBH_HIP(hipEventCreate(&e1));
  int rc = CBH_OK;
  CBH_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters && !rc; ++i)
    rc = launch_dcthash((const uint8_t*)d_imgs, n, w, h, row_stride, img_stride, (uint64_t*)d_out, s);
  CBH_HIP(hipEventRecord(e1, s));
  CBH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  CBH_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cbh::stream_destroy(s);
  *ms_avg = ms / (float)iters;
  return rc;
}

}  // extern "C"
