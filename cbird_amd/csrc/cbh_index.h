// cbh_index.h -- internal: device-resident index state shared by the host-side translation units
// (cbird_hip.hip, fdct.hip, video.hip).  Not part of the C-ABI.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "cbh_internal.h"
#include "cbh_shard.h"

namespace cbh {

// The kernels' per-call scratch (FP4 needle tiles, blur planes, sort buffers) comes from the stream-ordered
// allocator.  Its default pool gives freed memory back to the driver at the next synchronisation, so every call
// would map its scratch anew (milliseconds for GB-sized blur planes): keep it cached instead.
inline void keep_pool_memory(int dev) {
  static std::once_flag once[16];
  if (dev < 0 || dev >= 16) return;
  std::call_once(once[dev], [dev] {
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
      uint64_t keep = (uint64_t)1 << 30;  // (a finite mark: freed scratch above 1 GB returns to the driver; cbh_trim)
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
  });
}

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev == dev) {
      ok = true, prev = -1;  // already current (every logical shard of a one-device handle): nothing to set or restore
    } else {
      ok = hipSetDevice(dev) == hipSuccess;
    }
    if (ok) keep_pool_memory(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

inline bool device_usable(int dev) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    (void)hipGetLastError();  // do not leave "invalid device" behind: the next launch check of this thread would see it
    return false;
  }
  return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

struct Workspace {
  hipStream_t stream = nullptr;
  cbh_record* d_rec = nullptr;
  cbh_record* d_alt = nullptr;
  size_t rec_cap = 0;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  unsigned long long* d_total = nullptr;
  unsigned long long* h_total = nullptr;  // pinned
  cbh_record* h_small = nullptr;          // pinned: needle in, first kSmallRecs records out (single-needle find)
  static constexpr size_t kSmallRecs = 512;
  static constexpr size_t kFindRecs = 65536;  // record capacity a single-needle find starts with
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the lone-needle path (cbh_internal.h: launch_find_one): its pinned result block, two device words, a call counter
  LoneBlock* h_lone = nullptr;
  unsigned* d_lone = nullptr;
  unsigned long long lone_seq = 0;
  int ensure_lone() {
    if (h_lone && d_lone) return CBH_OK;
    if (!h_lone) {
      CBH_HIP(hipHostMalloc(&h_lone, sizeof(LoneBlock), hipHostMallocCoherent));
      h_lone->done = 0;
      h_lone->count = 0;
    }
    if (!d_lone) {
      CBH_HIP(hipMalloc(&d_lone, 2 * sizeof(unsigned)));
      // (on THIS stream: hipMemset on the null stream may return before it has run, and the workspace's stream is
      // non-blocking -- the first find's kernel would count on garbage and never publish)
      hipError_t e = hipMemsetAsync(d_lone, 0, 2 * sizeof(unsigned), stream);
      if (e != hipSuccess) {
        (void)hipFree(d_lone);
        d_lone = nullptr;
        CBH_HIP(e);
      }
    }
    return CBH_OK;
  }
  uint64_t* d_q = nullptr;
  size_t q_cap = 0;
  uint64_t* d_qmask = nullptr;  // per-needle equal-bits masks (tree / bucket compatible searches)
  size_t qmask_cap = 0;
  cbh_match* d_out = nullptr;
  size_t out_cap = 0;
  uint32_t* d_counts = nullptr;
  size_t counts_cap = 0;

  int init() {
    CBH_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CBH_HIP(hipHostMalloc(&h_total, sizeof(unsigned long long)));
    CBH_HIP(hipHostMalloc(&h_small, (kSmallRecs + 1) * sizeof(cbh_record)));
    CBH_HIP(hipEventCreate(&ev0));
    CBH_HIP(hipEventCreate(&ev1));
    return CBH_OK;
  }
  int ensure_records(size_t cap) {
    if (cap <= rec_cap) return CBH_OK;
    if (d_total) (void)hipFree(d_total);
    d_rec = nullptr;
    d_total = nullptr;
    rec_cap = 0;
    // one block { u64 count; records[cap] }: the layout the counting select (topk.hip) and the multi-GPU exchange take
    CBH_HIP(hipMalloc(&d_total, (cap + 1) * sizeof(cbh_record)));
    d_rec = reinterpret_cast<cbh_record*>(d_total) + 1;
    rec_cap = cap;
    return CBH_OK;
  }
  // ping-pong buffer + radix scratch of the global sort: only the paths that order ALL records need them (single-needle
  // find with a large result, fdct, video); the batched cut runs on the counting select and never allocates these
  size_t alt_cap = 0;
  int ensure_sort() {
    if (alt_cap >= rec_cap && d_alt) return CBH_OK;
    if (d_alt) (void)hipFree(d_alt);
    if (d_tmp) (void)hipFree(d_tmp);
    d_alt = nullptr;
    d_tmp = nullptr;
    alt_cap = 0;
    CBH_HIP(hipMalloc(&d_alt, std::max<size_t>(rec_cap, 16) * sizeof(cbh_record)));
    tmp_bytes = sort_records_scratch_bytes(std::max<size_t>(rec_cap, 16));
    CBH_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    alt_cap = rec_cap;
    return CBH_OK;
  }
  // give an oversized record block (and the sort buffers sized after it) back: a workspace returns to the pool of its
  // index and would otherwise hold its high-water mark for good (one whole-index self-join = up to 1 GB + sort scratch)
  void shrink_records(size_t keep) {
    if (rec_cap <= keep) return;
    if (d_total) (void)hipFree(d_total);
    if (d_alt) (void)hipFree(d_alt);
    if (d_tmp) (void)hipFree(d_tmp);
    d_total = nullptr, d_rec = nullptr, d_alt = nullptr, d_tmp = nullptr;
    rec_cap = 0, alt_cap = 0, tmp_bytes = 0;
  }
  // exchange buffers of a sharded index (cbh_shard.h): [0] a device's concatenated block, [1] the all-gathered blocks
  XBuf x[2];
  template <typename T>
  static int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap) return CBH_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t n = std::max<size_t>(need, 1024);
    CBH_HIP(hipMalloc(p, n * sizeof(T)));
    *cap = n;
    return CBH_OK;
  }
  void release() {
    if (d_alt) (void)hipFree(d_alt);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_total) (void)hipFree(d_total);  // (d_rec lives in the same allocation)
    if (h_total) (void)hipHostFree(h_total);
    if (h_small) (void)hipHostFree(h_small);
    if (h_lone) (void)hipHostFree(h_lone);
    if (d_lone) (void)hipFree(d_lone);
    if (d_q) (void)hipFree(d_q);
    if (d_qmask) (void)hipFree(d_qmask);
    if (d_out) (void)hipFree(d_out);
    if (d_counts) (void)hipFree(d_counts);
    for (XBuf& b : x) b.release();
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) cbh::stream_destroy(stream);
  }
};

struct Coalescer;                 // coalesce.hip: combining of concurrent find() callers + self-join cache
void coalescer_free(Coalescer*);
struct ShardSet;                  // sharded.hip: the children of an index that spans several shards / devices
void shardset_free(ShardSet*);

}  // namespace cbh

using namespace cbh;

struct cbh_idx64 {
  int device = 0;
  bool loaded = false;
  uint64_t* d_hashes = nullptr;
  uint32_t* d_ids = nullptr;
  size_t n = 0;
  size_t cap = 0;
  size_t rec_cap_default = (size_t)1 << 24;
  std::mutex stats_mu;
  cbh_stats stats = {0, 0, 0.0};
  std::mutex ws_mu;
  std::vector<Workspace*> ws_free;
  // HammingTree shape of the current contents (fdct tree-compatible search): node (depth d, prefix p) is
  // internal iff more than 8192 values share the low d bits p; key = d << 58 | p.  Rebuilt lazily.
  std::mutex tree_mu;
  bool tree_valid = false;
  std::unordered_set<uint64_t> tree_internal;
  std::atomic<uint64_t> generation{0};  // bumped by load/add/remove: caches derived from the contents check it
  std::atomic<Coalescer*> coalescer{nullptr};  // created on the first cbh_idx64_find_coalesced
  // cbh_idx64_create_sharded: this handle owns no slots itself (d_hashes == nullptr, n = total over the shards); its
  // children hold contiguous shares on their devices and every search goes through scan_all below (sharded.hip)
  ShardSet* shards = nullptr;

  Workspace* acquire(int* rc) {
    {
      std::lock_guard<std::mutex> lk(ws_mu);
      if (!ws_free.empty()) {
        Workspace* w = ws_free.back();
        ws_free.pop_back();
        *rc = CBH_OK;
        return w;
      }
    }
    Workspace* w = new (std::nothrow) Workspace;
    if (!w) {
      *rc = CBH_E_NOMEM;
      return nullptr;
    }
    *rc = w->init();
    if (*rc) {
      w->release();
      delete w;
      return nullptr;
    }
    return w;
  }
  void give_back(Workspace* w) {
    std::lock_guard<std::mutex> lk(ws_mu);
    ws_free.push_back(w);
  }
  int reserve(size_t need) {
    if (need <= cap) return CBH_OK;
    size_t ncap = std::max<size_t>(need, cap + cap / 2);
    ncap = (ncap + 1023) / 1024 * 1024;  // the reference grows in 1024-row chunks (:85-95)
    uint64_t* nh = nullptr;
    uint32_t* ni = nullptr;
    CBH_HIP(hipMalloc(&nh, ncap * sizeof(uint64_t)));
    hipError_t e = hipMalloc(&ni, ncap * sizeof(uint32_t));
    if (e != hipSuccess) {
      (void)hipFree(nh);
      set_last_error("hipMalloc(ids)", e);
      return CBH_E_NOMEM;
    }
    if (n) {
      e = hipMemcpy(nh, d_hashes, n * sizeof(uint64_t), hipMemcpyDeviceToDevice);
      if (e == hipSuccess) e = hipMemcpy(ni, d_ids, n * sizeof(uint32_t), hipMemcpyDeviceToDevice);
      if (e != hipSuccess) {  // keep the old arrays, drop the new ones
        (void)hipFree(nh);
        (void)hipFree(ni);
        set_last_error("hipMemcpy(reserve)", e);
        return CBH_E_HIP;
      }
    }
    if (d_hashes) (void)hipFree(d_hashes);
    if (d_ids) (void)hipFree(d_ids);
    d_hashes = nh;
    d_ids = ni;
    cap = ncap;
    return CBH_OK;
  }
};

struct WsLease {
  cbh_idx64* idx;
  Workspace* ws;
  WsLease(cbh_idx64* i, int* rc) : idx(i), ws(i->acquire(rc)) {}
  ~WsLease() {
    if (ws) idx->give_back(ws);
  }
};

namespace cbh {
// the same contract for an index made of shards: every shard scans its share on its own device and stream, the
// per-shard { count, records } blocks are exchanged (device-to-device copies inside one device, RCCL all-gather
// between devices) and end up as ONE block in the root workspace (sharded.hip)
int sharded_scan_all(cbh_idx64* idx, Workspace* ws, const uint64_t* d_q, size_t nq, int thresh, hipStream_t stream,
                     unsigned long long* total, unsigned flags, const uint64_t* d_qmask, size_t max_records);
int sharded_download(const cbh_idx64* idx, uint64_t* hashes, uint32_t* ids, size_t cap);
}  // namespace cbh

// scan into the workspace record buffer, growing it until every record fits.
// On return *total = number of matching pairs, all of them present in ws->d_rec.
// max_records: a result larger than this is not materialised -- CBH_E_OVERFLOW with *total set, and the workspace is
// never grown past it (the self-join cache asks before it commits a GB-sized buffer).
inline int scan_all(cbh_idx64* idx, Workspace* ws, const uint64_t* d_q, size_t nq, int thresh,
                    hipStream_t stream, unsigned long long* total, unsigned flags = 0,
                    const uint64_t* d_qmask = nullptr, size_t max_records = ~(size_t)0) {
  if (idx->shards) return sharded_scan_all(idx, ws, d_q, nq, thresh, stream, total, flags, d_qmask, max_records);
  int rc = ws->ensure_records(std::min(std::max<size_t>(idx->rec_cap_default, 1024), std::max<size_t>(max_records, 1024)));
  if (rc) return rc;
  for (int attempt = 0; attempt < 3; ++attempt) {
    CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), stream));
    CBH_HIP(hipEventRecord(ws->ev0, stream));
    rc = launch_hamm64_scan(idx->d_hashes, idx->d_ids, idx->n, d_q, nq, thresh, ws->d_rec,
                            ws->rec_cap, ws->d_total, stream, flags, d_qmask);
    if (rc) return rc;
    CBH_HIP(hipEventRecord(ws->ev1, stream));
    CBH_HIP(hipMemcpyAsync(ws->h_total, ws->d_total, sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, stream));
    CBH_HIP(hipStreamSynchronize(stream));
    *total = *ws->h_total;
    {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ws->ev0, ws->ev1) == hipSuccess) {
        std::lock_guard<std::mutex> lk(idx->stats_mu);
        idx->stats.scan_launches += 1;
        idx->stats.scan_pairs += (uint64_t)idx->n * (uint64_t)nq;
        idx->stats.scan_ms += (double)ms;
      }
    }
    if (*total <= ws->rec_cap) return CBH_OK;
    if (*total > max_records) return CBH_E_OVERFLOW;  // the caller does not want a result of this size
    // every match must be materialised to be ordered: grow and rescan
    CBH_HIP(hipStreamSynchronize(stream));
    rc = ws->ensure_records((size_t)*total + 1024);
    if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
  }
  return CBH_E_OVERFLOW;
}
