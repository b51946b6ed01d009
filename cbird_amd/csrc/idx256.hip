// idx256.hip -- CvFeaturesIndex (src/cvfeaturesindex.{h,cpp}): 256-bit ORB/BRIEF descriptors, k nearest
// neighbours in Hamming space, brute force on gfx950.
//
// Reference: all descriptors live in one N x 32-byte matrix (cvfeaturesindex.h:73); find() asks a FLANN LSH
// index for the 10 nearest rows of every needle descriptor (cvfeaturesindex.cpp:497), keeps distance <
// cvThresh (:508), maps row -> mediaId through the first-row map (:511-516, removed media have id 0 and are
// skipped AFTER the knn cut, :518), and scores each media by median distance * 1000 / votes (:564-596).
// LSH is approximate; this is the exact search it approximates (results are a superset).
//
// k_hamm256_scan has the shape of k_hamm64_scan: each lane keeps H rows (first 128 bits, 4 VGPRs per row) in
// registers, needles are wave-uniform SGPR operands.  Since the k nearest are only ever used below a threshold,
// it is a threshold scan: the first 128 bits give a sound lower bound (4 xor + 4 bcnt per pair, min3 over
// pairs), and only slots whose bound drops under the threshold load their second half and evaluate all 256 bits.
// Records q<<41 | dist<<32 | row are then ordered and cut at k per needle descriptor.

#include <map>

#include "cbh_index.h"

namespace {

constexpr int kThreads = 256;
constexpr int kH = 8;
constexpr int kQB = 4;

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }

__device__ __forceinline__ uint32_t popc128(uint4 a, uint4 b) {
  return __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
}

template <int H, int QB>
__global__ __launch_bounds__(kThreads) void k_hamm256_scan(
    const uint4* __restrict__ rows /* 2 x uint4 per row */, uint32_t n, const uint4* __restrict__ q, uint32_t nq,
    uint32_t q_chunk, uint32_t thresh, unsigned long long* __restrict__ rec, unsigned long long cap,
    unsigned long long* __restrict__ total) {
  const uint32_t base_idx = blockIdx.x * (uint32_t)(kThreads * H) + threadIdx.x;
  uint4 h[H];
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const uint32_t idx = base_idx + (uint32_t)j * kThreads;
    h[j] = idx < n ? rows[(size_t)idx * 2] : make_uint4(0u, 0u, 0u, 0u);
  }
  const uint32_t q0 = blockIdx.y * q_chunk;
  const uint32_t q1 = min(nq, q0 + q_chunk);
  for (uint32_t qb = q0; qb < q1; qb += QB) {
    uint4 cur[QB];
#pragma unroll
    for (int i = 0; i < QB; ++i) cur[i] = q[(size_t)min(qb + i, q1 - 1) * 2];  // wave-uniform -> SMEM
    uint32_t acc[H];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = 0xffffu;
#pragma unroll
    for (int i = 0; i < QB; i += 2) {
#pragma unroll
      for (int j = 0; j < H; ++j) acc[j] = min3u(acc[j], popc128(h[j], cur[i]), popc128(h[j], cur[i + 1]));
    }
    uint32_t m = acc[0];
#pragma unroll
    for (int j = 1; j < H; ++j) m = min(m, acc[j]);
    if (m < thresh) {
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (acc[j] < thresh) {
          const uint32_t idx = base_idx + (uint32_t)j * kThreads;
          if (idx < n) {
            const uint4 h2 = rows[(size_t)idx * 2 + 1];
#pragma unroll 1
            for (uint32_t qi = qb; qi < min(qb + QB, q1); ++qi) {
              const uint32_t d = popc128(h[j], q[(size_t)qi * 2]) + popc128(h2, q[(size_t)qi * 2 + 1]);
              if (d < thresh) {
                const unsigned long long slot = atomicAdd(total, 1ull);
                if (slot < cap)
                  rec[slot] = ((unsigned long long)qi << 41) | ((unsigned long long)d << 32) | idx;
              }
            }
          }
        }
      }
    }
  }
}

// first k records of every needle descriptor from the sorted list: (row, dist)
__global__ __launch_bounds__(256) void k_select256(const unsigned long long* __restrict__ rec, size_t n,
                                                   uint32_t nq, int k, uint32_t* __restrict__ out_row,
                                                   uint16_t* __restrict__ out_dist,
                                                   uint32_t* __restrict__ counts) {
  const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= nq) return;
  auto lower = [&](unsigned long long key) {
    size_t lo = 0, hi = n;
    while (lo < hi) {
      size_t mid = (lo + hi) >> 1;
      if (rec[mid] < key)
        lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  };
  const size_t a = lower((unsigned long long)qi << 41);
  const size_t b = lower(((unsigned long long)qi + 1) << 41);
  const size_t cnt = b - a;
  counts[qi] = cnt > 0xffffffffull ? 0xffffffffu : (uint32_t)cnt;
  for (int j = 0; j < k; ++j) {  // places past the count are zeroed (the buffers are reused between calls)
    const unsigned long long r = (size_t)j < cnt ? rec[a + (size_t)j] : 0ull;
    out_row[(size_t)qi * k + j] = (uint32_t)r;
    out_dist[(size_t)qi * k + j] = (uint16_t)((r >> 32) & 0x1ff);
  }
}

int sig_bits256(size_t nq) {
  int b = 0;
  while (b < 23 && ((size_t)1 << b) < nq) ++b;
  return 41 + b;
}

}  // namespace

struct Shards256;  // below: the children of a CvFeaturesIndex that spans several shards / devices

struct cbh_idx256 {
  int device = 0;
  bool loaded = false;
  Shards256* shards = nullptr;  // cbh_idx256_create_sharded: this handle keeps the maps, its children the rows
  cbh::XBuf x[2];               // exchange buffers of a child (cbh_shard.h)
  uint8_t* d_rows = nullptr;  // N x 32 B
  size_t n = 0, cap = 0;
  // _indexMap (cvfeaturesindex.h:77): first row -> mediaId (0 = removed), ascending; sentinel (n, 0)
  std::vector<uint32_t> first_row;
  std::vector<uint32_t> media_id;
  std::map<uint32_t, uint32_t> id_to_first;  // _idMap
  // scratch (one search at a time per index; guarded)
  std::mutex mu;
  hipStream_t stream = nullptr;
  unsigned long long *d_rec = nullptr, *d_alt = nullptr, *d_total = nullptr, *h_total = nullptr;
  void* d_tmp = nullptr;
  size_t rec_cap = 0, tmp_bytes = 0;
  uint8_t* d_q = nullptr;
  size_t q_cap = 0;
  uint32_t *d_out_row = nullptr, *d_counts = nullptr;
  uint16_t* d_out_dist = nullptr;
  size_t out_cap = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double scan_ms = 0;
  uint64_t scan_pairs = 0, scan_launches = 0;

  uint32_t media_of_row(uint32_t row) const {  // upper_bound(index) - 1 (:514-516)
    auto it = std::upper_bound(first_row.begin(), first_row.end(), row);
    if (it == first_row.begin()) return 0;
    return media_id[(size_t)(it - first_row.begin()) - 1];
  }
};

namespace {

int ensure_scratch(cbh_idx256* ix, size_t nq, size_t rec_cap, int k, bool scan_only = false) {
  if (!ix->stream) {
    CBH_HIP(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    CBH_HIP(hipMalloc(&ix->d_total, 8));
    CBH_HIP(hipHostMalloc(&ix->h_total, 8));
    CBH_HIP(hipEventCreate(&ix->ev0));
    CBH_HIP(hipEventCreate(&ix->ev1));
  }
  if (rec_cap > ix->rec_cap) {
    if (ix->d_rec) (void)hipFree(ix->d_rec);
    if (ix->d_alt) (void)hipFree(ix->d_alt);
    if (ix->d_tmp) (void)hipFree(ix->d_tmp);
    ix->d_rec = ix->d_alt = nullptr;
    ix->d_tmp = nullptr;
    ix->rec_cap = 0;
    CBH_HIP(hipMalloc(&ix->d_rec, rec_cap * 8));
    if (!scan_only) {  // (a shard only scans: the sort and the cut run on the parent)
      CBH_HIP(hipMalloc(&ix->d_alt, rec_cap * 8));
      ix->tmp_bytes = cbh::sort_records_scratch_bytes(rec_cap);
      CBH_HIP(hipMalloc(&ix->d_tmp, ix->tmp_bytes ? ix->tmp_bytes : 16));
    }
    ix->rec_cap = rec_cap;
  }
  if (nq > ix->q_cap) {
    if (ix->d_q) (void)hipFree(ix->d_q);
    ix->d_q = nullptr;
    ix->q_cap = 0;
    CBH_HIP(hipMalloc(&ix->d_q, nq * 32));
    ix->q_cap = nq;
  }
  if (!scan_only && nq * (size_t)k > ix->out_cap) {
    if (ix->d_out_row) (void)hipFree(ix->d_out_row);
    if (ix->d_out_dist) (void)hipFree(ix->d_out_dist);
    if (ix->d_counts) (void)hipFree(ix->d_counts);
    ix->d_out_row = ix->d_counts = nullptr;
    ix->d_out_dist = nullptr;
    ix->out_cap = 0;
    CBH_HIP(hipMalloc(&ix->d_out_row, nq * (size_t)k * 4));
    CBH_HIP(hipMalloc(&ix->d_out_dist, nq * (size_t)k * 2));
    CBH_HIP(hipMalloc(&ix->d_counts, nq * 4));
    ix->out_cap = nq * (size_t)k;
  }
  return CBH_OK;
}

int launch_scan256(cbh_idx256* ix, const uint8_t* d_q, size_t nq, int thresh) {
  if (cbh::scan256_mfma_wanted(ix->n, nq, thresh))
    return cbh::launch_scan256_mfma(ix->d_rows, ix->n, d_q, nq, thresh, ix->d_rec, ix->rec_cap, ix->d_total,
                                    ix->stream);
  const uint32_t tile = kThreads * kH;
  const uint32_t tiles = (uint32_t)((ix->n + tile - 1) / tile);
  uint32_t q_chunk = 4096;
  while (q_chunk > 256 && (uint64_t)tiles * ((nq + q_chunk - 1) / q_chunk) < 8192) q_chunk >>= 1;
  uint32_t chunks = (uint32_t)((nq + q_chunk - 1) / q_chunk);
  if (chunks > 65535) {
    q_chunk = (uint32_t)((nq + 65534) / 65535);
    q_chunk = (q_chunk + kQB - 1) / kQB * kQB;
    chunks = (uint32_t)((nq + q_chunk - 1) / q_chunk);
  }
  hipLaunchKernelGGL((k_hamm256_scan<kH, kQB>), dim3(tiles, chunks), dim3(kThreads), 0, ix->stream,
                     reinterpret_cast<const uint4*>(ix->d_rows), (uint32_t)ix->n,
                     reinterpret_cast<const uint4*>(d_q), (uint32_t)nq, q_chunk, (uint32_t)thresh, ix->d_rec,
                     (unsigned long long)ix->rec_cap, ix->d_total);
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

// every record of nq needle rows (host memory) against the rows of ONE device-resident index, in ix->d_rec on
// ix->stream; the buffer grows until all of them fit
int scan_records(cbh_idx256* ix, const uint8_t* needles, size_t nq, int k, int thresh, unsigned long long* total_out,
                 bool scan_only = false) {
  int rc;
  CBH_HIP(hipMemcpyAsync(ix->d_q, needles, nq * 32, hipMemcpyHostToDevice, ix->stream));
  unsigned long long total = 0;
  for (int attempt = 0;; ++attempt) {
    CBH_HIP(hipMemsetAsync(ix->d_total, 0, 8, ix->stream));
    CBH_HIP(hipEventRecord(ix->ev0, ix->stream));
    rc = launch_scan256(ix, ix->d_q, nq, thresh);
    if (rc) return rc;
    CBH_HIP(hipEventRecord(ix->ev1, ix->stream));
    CBH_HIP(hipMemcpyAsync(ix->h_total, ix->d_total, 8, hipMemcpyDeviceToHost, ix->stream));
    CBH_HIP(hipStreamSynchronize(ix->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, ix->ev0, ix->ev1) == hipSuccess) {
      ix->scan_ms += ms;
      ix->scan_pairs += (uint64_t)ix->n * nq;
      ix->scan_launches++;
    }
    total = *ix->h_total;
    if (total <= ix->rec_cap) break;
    if (attempt >= 2) return CBH_E_OVERFLOW;
    rc = ensure_scratch(ix, nq, (size_t)total + 1024, k, scan_only);
    if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
  }
  *total_out = total;
  return CBH_OK;
}

}  // namespace

// ---- one CvFeaturesIndex over several shards / GPUs (cbh_idx256_create_sharded) ------------------------------------
// Sharded BY IMAGE: a media's descriptor rows stay together on one shard; the parent keeps the
// first-row -> mediaId maps in GLOBAL row numbers exactly as the one-device index does, the children hold rows only.
// Rows arrive media by media (add), so a shard takes media until it has received kShardRun rows, then the emptiest
// shard takes over: the global row order is the add order, a shard's rows are runs of it (a segment table per shard).
// A search scans every shard on its own device and stream, rewrites the LOCAL row of every record to the global one
// (k_rows_to_global: the tie-break of the knn is (distance, global row), and the maps are global), brings the records
// to the parent with ShardComm::exchange (cbh_shard.h: copies inside a device, ncclAllGather between devices) and
// sorts / cuts / scores there as the one-device index does.
struct Shards256 {
  cbh::ShardComm comm;
  std::vector<cbh_idx256*> child;
  struct Seg {
    uint32_t shard;
    size_t local, global, len;
  };
  std::vector<Seg> segs;  // in global order
  size_t cur = 0, cur_run = 0;
  static constexpr size_t kShardRun = 16384;
  // per shard: the segment table on its device (local start ascending, global - local), rebuilt when rows were added
  std::vector<uint32_t*> d_seg_local;
  std::vector<long long*> d_seg_delta;
  std::vector<uint32_t> n_seg;
  bool dirty = true;
};

namespace {

__global__ __launch_bounds__(256) void k_rows_to_global(unsigned long long* __restrict__ rec, unsigned long long n,
                                                        const uint32_t* __restrict__ seg_local,
                                                        const long long* __restrict__ seg_delta, uint32_t nseg) {
  const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
  if (i >= n) return;
  const unsigned long long r = rec[i];
  const uint32_t row = (uint32_t)r;
  uint32_t lo = 0, hi = nseg;  // last segment whose local start is <= row
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (seg_local[mid] <= row)
      lo = mid;
    else
      hi = mid;
  }
  rec[i] = (r & 0xffffffff00000000ull) | (uint32_t)((long long)row + seg_delta[lo]);
}

int upload_segments(cbh_idx256* ix) {
  Shards256* S = ix->shards;
  if (!S->dirty) return CBH_OK;
  const size_t R = S->child.size();
  for (size_t s = 0; s < R; ++s) {
    std::vector<uint32_t> loc;
    std::vector<long long> delta;
    for (const Shards256::Seg& g : S->segs)
      if (g.shard == s) {  // (global order is also local order inside a shard)
        loc.push_back((uint32_t)g.local);
        delta.push_back((long long)g.global - (long long)g.local);
      }
    cbh::DeviceGuard dg(S->child[s]->device);
    if (!dg.ok) return CBH_E_NODEVICE;
    if (S->d_seg_local[s]) (void)hipFree(S->d_seg_local[s]);
    if (S->d_seg_delta[s]) (void)hipFree(S->d_seg_delta[s]);
    S->d_seg_local[s] = nullptr, S->d_seg_delta[s] = nullptr;
    S->n_seg[s] = (uint32_t)loc.size();
    if (loc.empty()) continue;
    CBH_HIP(hipMalloc(&S->d_seg_local[s], loc.size() * 4));
    CBH_HIP(hipMalloc(&S->d_seg_delta[s], loc.size() * 8));
    CBH_HIP(hipMemcpy(S->d_seg_local[s], loc.data(), loc.size() * 4, hipMemcpyHostToDevice));
    CBH_HIP(hipMemcpy(S->d_seg_delta[s], delta.data(), delta.size() * 8, hipMemcpyHostToDevice));
  }
  S->dirty = false;
  return CBH_OK;
}

// as scan_records, over the shards: all records, rows global, in the parent's d_rec on the parent's stream
int scan_records_sharded(cbh_idx256* ix, const uint8_t* needles, size_t nq, int k, int thresh,
                         unsigned long long* total_out) {
  Shards256* S = ix->shards;
  const size_t R = S->child.size();
  int rc = upload_segments(ix);
  if (rc) return rc;
  std::vector<unsigned long long> count(R, 0);
  std::vector<char> todo(R, 0);
  // needles to every shard's device, first scans
  for (size_t s = 0; s < R; ++s) {
    cbh_idx256* c = S->child[s];
    if (!c->n) continue;
    cbh::DeviceGuard dg(c->device);
    if (!dg.ok) return CBH_E_NODEVICE;
    if ((rc = ensure_scratch(c, nq, std::max<size_t>(c->rec_cap, std::max<size_t>(65536, ((size_t)1 << 22) / R)), k, true)))
      return rc;
    CBH_HIP(hipMemcpyAsync(c->d_q, needles, nq * 32, hipMemcpyHostToDevice, c->stream));
    todo[s] = 1;
  }
  float scan_ms = 0;
  for (int attempt = 0; attempt < 3; ++attempt) {
    bool any = false;
    for (size_t s = 0; s < R; ++s) {
      if (!todo[s]) continue;
      any = true;
      cbh_idx256* c = S->child[s];
      cbh::DeviceGuard dg(c->device);
      CBH_HIP(hipMemsetAsync(c->d_total, 0, 8, c->stream));
      CBH_HIP(hipEventRecord(c->ev0, c->stream));
      if ((rc = launch_scan256(c, c->d_q, nq, thresh))) return rc;
      CBH_HIP(hipEventRecord(c->ev1, c->stream));
      CBH_HIP(hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, c->stream));
      S->comm.n_scans++;
      if (attempt) S->comm.n_rescans++;
    }
    if (!any) break;
    float worst = 0;
    for (size_t s = 0; s < R; ++s) {
      if (!todo[s]) continue;
      cbh_idx256* c = S->child[s];
      cbh::DeviceGuard dg(c->device);
      CBH_HIP(hipStreamSynchronize(c->stream));
      count[s] = *c->h_total;
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) worst = std::max(worst, ms);
      todo[s] = 0;
      if (count[s] > c->rec_cap) {  // this shard alone grows its buffer and scans again
        rc = ensure_scratch(c, nq, (size_t)count[s] + 1024, k, true);
        if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
        todo[s] = 1;
      }
    }
    scan_ms += worst;
  }
  for (size_t s = 0; s < R; ++s)
    if (todo[s]) return CBH_E_OVERFLOW;
  ix->scan_ms += scan_ms;
  ix->scan_pairs += (uint64_t)ix->n * nq;
  ix->scan_launches++;
  unsigned long long sum = 0;
  for (size_t s = 0; s < R; ++s) sum += count[s];
  if (sum > ix->rec_cap) {
    cbh::DeviceGuard dg(ix->device);
    rc = ensure_scratch(ix, nq, (size_t)sum + 1024, k);
    if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
  }
  // local rows -> global rows, then the exchange
  std::vector<cbh::ShardPart> parts(R);
  for (size_t s = 0; s < R; ++s) {
    cbh_idx256* c = S->child[s];
    cbh::DeviceGuard dg(c->device);
    if (count[s]) {
      hipLaunchKernelGGL(k_rows_to_global, dim3((unsigned)((count[s] + 255) / 256)), dim3(256), 0, c->stream, c->d_rec,
                         count[s], S->d_seg_local[s], S->d_seg_delta[s], S->n_seg[s]);
      CBH_HIP(hipGetLastError());
    }
    if (!c->stream) {  // an empty shard that never scanned still takes part in a collective
      if ((rc = ensure_scratch(c, 1, 1024, k, true))) return rc;
    }
    parts[s].dev_pos = S->comm.dev_pos_of_shard(s);
    parts[s].stream = c->stream;
    parts[s].d_rec = c->d_rec;
    parts[s].count = count[s];
    parts[s].ev = c->ev1;
    parts[s].x = c->x;
    parts[s].h_word = c->h_total;
  }
  cbh::DeviceGuard dg(ix->device);
  if ((rc = S->comm.exchange(parts, ix->stream, ix->d_rec))) return rc;
  CBH_HIP(hipStreamSynchronize(ix->stream));
  for (size_t s = 0; s < R; ++s) {  // the shards' side of a collective has finished too before their buffers are reused
    cbh_idx256* c = S->child[s];
    cbh::DeviceGuard dg2(c->device);
    if (c->stream) CBH_HIP(hipStreamSynchronize(c->stream));
  }
  *total_out = sum;
  return CBH_OK;
}

// knn (k per needle descriptor, below thresh) for nq needle rows on the host side of the index;
// out_row/out_dist [nq*k], counts[nq] (full number under thresh)
int knn_core(cbh_idx256* ix, const uint8_t* needles, size_t nq, int k, int thresh, std::vector<uint32_t>* row,
             std::vector<uint16_t>* dist, std::vector<uint32_t>* counts,
             std::vector<unsigned long long>* all_records = nullptr) {
  if (!all_records) {
    row->assign(nq * (size_t)k, 0);
    dist->assign(nq * (size_t)k, 0);
    counts->assign(nq, 0);
  } else {
    all_records->clear();
  }
  if (nq == 0 || ix->n == 0 || thresh <= 0 || k <= 0) return CBH_OK;
  if (nq >= (1u << 23)) return CBH_E_INVAL;
  cbh::DeviceGuard g(ix->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(ix->mu);
  int rc = ensure_scratch(ix, nq, std::max<size_t>(ix->rec_cap, (size_t)1 << 22), k);
  if (rc) return rc;
  unsigned long long total = 0;
  if ((rc = ix->shards ? scan_records_sharded(ix, needles, nq, k, thresh, &total) : scan_records(ix, needles, nq, k, thresh, &total)))
    return rc;
  if (total > 1) {
    unsigned long long* sorted = nullptr;
    if ((rc = cbh::sort_keys64_db(ix->d_rec, ix->d_alt, (size_t)total, (unsigned)sig_bits256(nq), ix->d_tmp, ix->tmp_bytes,
                                  ix->stream, &sorted)))
      return rc;
    if (sorted != ix->d_rec)
      CBH_HIP(hipMemcpyAsync(ix->d_rec, sorted, total * 8, hipMemcpyDeviceToDevice, ix->stream));
  }
  if (all_records) {  // radius search: every record, already in (needle, distance, row) order
    all_records->resize((size_t)total);
    if (total)
      CBH_HIP(hipMemcpyAsync(all_records->data(), ix->d_rec, total * 8, hipMemcpyDeviceToHost, ix->stream));
    CBH_HIP(hipStreamSynchronize(ix->stream));
    return CBH_OK;
  }
  hipLaunchKernelGGL(k_select256, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ix->stream, ix->d_rec,
                     (size_t)total, (uint32_t)nq, k, ix->d_out_row, ix->d_out_dist, ix->d_counts);
  CBH_HIP(hipGetLastError());
  CBH_HIP(hipMemcpyAsync(row->data(), ix->d_out_row, nq * (size_t)k * 4, hipMemcpyDeviceToHost, ix->stream));
  CBH_HIP(hipMemcpyAsync(dist->data(), ix->d_out_dist, nq * (size_t)k * 2, hipMemcpyDeviceToHost, ix->stream));
  CBH_HIP(hipMemcpyAsync(counts->data(), ix->d_counts, nq * 4, hipMemcpyDeviceToHost, ix->stream));
  CBH_HIP(hipStreamSynchronize(ix->stream));
  return CBH_OK;
}

// votes and scores of one needle image (cvfeaturesindex.cpp:499-596) from a knn table that already carries the
// mediaId of every candidate (0 = deleted/removed item, :518)
void score_media(const uint32_t* media, const uint16_t* dist, const uint32_t* counts, size_t d0, size_t d1, int k,
                 std::vector<cbh_match>* out) {
  std::map<uint32_t, std::vector<int>> matches;  // QMap<uint32_t, Match_>: ascending mediaId
  for (size_t j = d0; j < d1; ++j) {
    const uint32_t len = std::min<uint32_t>((uint32_t)k, counts[j]);
    for (uint32_t t = 0; t < len; ++t) {
      const uint32_t mediaId = media[j * (size_t)k + t];
      if (!mediaId) continue;
      matches[mediaId].push_back((int)dist[j * (size_t)k + t]);
    }
  }
  for (auto& kv : matches) {
    std::vector<int>& scores = kv.second;
    std::sort(scores.begin(), scores.end());
    int score;
    const size_t middle = scores.size() / 2;
    if (scores.size() < 2)
      score = scores[0];
    else if (scores.size() % 2 == 0)
      score = (scores[middle - 1] + scores[middle]) / 2;
    else
      score = scores[middle];
    score = score * 1000 / (int)scores.size();
    out->push_back(cbh_match{kv.first, score});
  }
}

void score256(const cbh_idx256* ix, const uint32_t* row, const uint16_t* dist, const uint32_t* counts, size_t d0,
              size_t d1, int k, std::vector<cbh_match>* out) {
  std::vector<uint32_t> media((d1 - d0) * (size_t)k, 0);
  for (size_t j = d0; j < d1; ++j) {
    const uint32_t len = std::min<uint32_t>((uint32_t)k, counts[j]);
    for (uint32_t t = 0; t < len; ++t) media[(j - d0) * (size_t)k + t] = ix->media_of_row(row[j * (size_t)k + t]);
  }
  score_media(media.data(), dist + d0 * (size_t)k, counts + d0, 0, d1 - d0, k, out);
}

}  // namespace

extern "C" {

cbh_idx256* cbh_idx256_create(int device) {
  cbh::clear_last_error();
  if (!cbh::device_usable(device)) return (cbh_idx256*)cbh::fail_handle(CBH_E_NODEVICE, "cbh_idx256_create: no usable gfx950 device at that ordinal");
  cbh_idx256* ix = new (std::nothrow) cbh_idx256;
  if (!ix) return (cbh_idx256*)cbh::fail_handle(CBH_E_NOMEM, "cbh_idx256_create: host allocation failed");
  ix->device = device;
  ix->first_row.push_back(0);
  ix->media_id.push_back(0);
  return ix;
}

cbh_idx256* cbh_idx256_create_sharded(uint32_t device_mask, int shards_per_device) {
  cbh::clear_last_error();
  Shards256* S = new (std::nothrow) Shards256;
  if (!S) return (cbh_idx256*)cbh::fail_handle(CBH_E_NOMEM, "cbh_idx256_create_sharded: host allocation failed");
  if (!S->comm.init(device_mask, shards_per_device)) {
    delete S;
    return (cbh_idx256*)cbh::fail_handle(CBH_E_INVAL, "cbh_idx256_create_sharded: empty mask, a device of the mask is not usable, or shards_per_device out of range");
  }
  cbh_idx256* ix = cbh_idx256_create(S->comm.devices[0]);
  if (!ix) {
    delete S;
    return nullptr;  // (cbh_idx256_create has set the code)
  }
  ix->shards = S;
  const size_t R = S->comm.shard_count();
  for (size_t s = 0; s < R; ++s) {
    cbh_idx256* c = cbh_idx256_create(S->comm.device_of_shard(s));
    if (!c) {
      cbh_idx256_destroy(ix);
      return nullptr;
    }
    S->child.push_back(c);
  }
  S->d_seg_local.assign(R, nullptr);
  S->d_seg_delta.assign(R, nullptr);
  S->n_seg.assign(R, 0);
  return ix;
}

int cbh_idx256_shard_count(const cbh_idx256* ix) { return !ix ? 0 : ix->shards ? (int)ix->shards->child.size() : 1; }

size_t cbh_idx256_shard_rows(const cbh_idx256* ix, int i) {
  if (!ix) return 0;
  if (!ix->shards) return i == 0 ? ix->n : 0;
  return i >= 0 && (size_t)i < ix->shards->child.size() ? ix->shards->child[(size_t)i]->n : 0;
}

int cbh_idx256_shard_stats(const cbh_idx256* ix, cbh_shard_stats* out) {
  if (!ix || !out) return CBH_E_INVAL;
  memset(out, 0, sizeof *out);
  out->shards = 1, out->devices = 1;
  if (!ix->shards) return CBH_OK;
  const Shards256* S = ix->shards;
  out->shards = (uint32_t)S->child.size();
  out->devices = (uint32_t)S->comm.devices.size();
  out->device_mask = S->comm.mask;
  out->segments = S->segs.size();
  out->scans = S->comm.n_scans.load();
  out->rescans = S->comm.n_rescans.load();
  out->collectives = S->comm.n_collectives.load();
  out->peer_copies = S->comm.n_peer_copies.load();
  out->local_copies = S->comm.n_local_copies.load();
  return CBH_OK;
}

void cbh_idx256_destroy(cbh_idx256* ix) {
  if (!ix) return;
  cbh::combiner_drop(ix);  // combine.hip: the queue of cbh_*_find_coalesced callers
  if (ix->shards) {
    Shards256* S = ix->shards;
    S->comm.destroy_comms();
    for (size_t s = 0; s < S->child.size(); ++s) {
      cbh::DeviceGuard g(S->child[s]->device);
      if (s < S->d_seg_local.size() && S->d_seg_local[s]) (void)hipFree(S->d_seg_local[s]);
      if (s < S->d_seg_delta.size() && S->d_seg_delta[s]) (void)hipFree(S->d_seg_delta[s]);
      cbh_idx256_destroy(S->child[s]);
    }
    delete S;
    ix->shards = nullptr;
  }
  cbh::DeviceGuard g(ix->device);
  for (cbh::XBuf& b : ix->x) b.release();
  for (void* p : {(void*)ix->d_rows, (void*)ix->d_rec, (void*)ix->d_alt, (void*)ix->d_tmp, (void*)ix->d_total,
                  (void*)ix->d_q, (void*)ix->d_out_row, (void*)ix->d_out_dist, (void*)ix->d_counts})
    if (p) (void)hipFree(p);
  if (ix->h_total) (void)hipHostFree(ix->h_total);
  if (ix->ev0) (void)hipEventDestroy(ix->ev0);
  if (ix->ev1) (void)hipEventDestroy(ix->ev1);
  if (ix->stream) cbh::stream_destroy(ix->stream);
  delete ix;
}

/* add(): append one media's descriptor rows (cvfeaturesindex.cpp:122-150); n_rows == 0 is skipped with no
 * map entry ("no descriptors for ..."), as in the reference.  load() is add() per `matrix` row. */
int cbh_idx256_add(cbh_idx256* ix, uint32_t media_id, const uint8_t* rows, size_t n_rows) {
  if (!ix) return CBH_E_INVAL;
  ix->loaded = true;
  if (n_rows == 0) return CBH_OK;
  if (!rows) return CBH_E_INVAL;
  if (ix->n + n_rows > 0xfffffff0ull) return CBH_E_INVAL;
  cbh::DeviceGuard g(ix->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->shards) {  // the rows go to a shard, the maps stay here in global row numbers
    Shards256* S = ix->shards;
    if (S->cur_run >= Shards256::kShardRun) {  // the current shard has had its run: the emptiest one takes over
      size_t best = 0;
      for (size_t s = 1; s < S->child.size(); ++s)
        if (S->child[s]->n < S->child[best]->n) best = s;
      S->cur = best;
      S->cur_run = 0;
    }
    cbh_idx256* c = S->child[S->cur];
    const size_t local = c->n;
    int rc = cbh_idx256_add(c, media_id, rows, n_rows);
    if (rc) return rc;
    if (!S->segs.empty() && S->segs.back().shard == S->cur && S->segs.back().local + S->segs.back().len == local &&
        S->segs.back().global + S->segs.back().len == ix->n)
      S->segs.back().len += n_rows;
    else
      S->segs.push_back(Shards256::Seg{(uint32_t)S->cur, local, ix->n, n_rows});
    S->cur_run += n_rows;
    S->dirty = true;
  } else if (ix->n + n_rows > ix->cap) {
    size_t ncap = std::max<size_t>(ix->n + n_rows, ix->cap + ix->cap / 2 + 65536);
    uint8_t* nr = nullptr;
    CBH_HIP(hipMalloc(&nr, ncap * 32));
    if (ix->n) CBH_HIP(hipMemcpy(nr, ix->d_rows, ix->n * 32, hipMemcpyDeviceToDevice));
    if (ix->d_rows) (void)hipFree(ix->d_rows);
    ix->d_rows = nr;
    ix->cap = ncap;
  }
  if (!ix->shards) CBH_HIP(hipMemcpy(ix->d_rows + ix->n * 32, rows, n_rows * 32, hipMemcpyHostToDevice));
  // _idMap[mid] = numDesc; _indexMap[numDesc] = mid; sentinel (numDesc + rows) -> 0
  ix->first_row.back() = (uint32_t)ix->n;
  ix->media_id.back() = media_id;
  ix->id_to_first[media_id] = (uint32_t)ix->n;
  ix->n += n_rows;
  ix->first_row.push_back((uint32_t)ix->n);
  ix->media_id.push_back(0);
  return CBH_OK;
}

/* remove(): the media's map entry becomes id 0; its descriptors stay in the matrix (:152-165) */
int cbh_idx256_remove(cbh_idx256* ix, const uint32_t* ids, size_t n) {
  if (!ix || (n && !ids)) return CBH_E_INVAL;
  std::lock_guard<std::mutex> lk(ix->mu);
  for (size_t i = 0; i < n; ++i) {
    auto it = ix->id_to_first.find(ids[i]);
    if (it == ix->id_to_first.end()) continue;
    auto p = std::lower_bound(ix->first_row.begin(), ix->first_row.end() - 1, it->second);
    if (p != ix->first_row.end() - 1 && *p == it->second) ix->media_id[(size_t)(p - ix->first_row.begin())] = 0;
  }
  return CBH_OK;
}

int cbh_idx256_is_loaded(const cbh_idx256* ix) { return ix && ix->loaded; }
size_t cbh_idx256_count(const cbh_idx256* ix) { return ix ? ix->n : 0; }  // _descriptors.rows (:103)
size_t cbh_idx256_memory_usage(const cbh_idx256* ix) { return ix ? ix->n * 32 * 2 : 0; }  // (:105-120)

/* descriptorsForMediaId (:421-436): row range of one media (0,0 when unknown) */
int cbh_idx256_rows_of(const cbh_idx256* ix, uint32_t media_id, size_t* first, size_t* count) {
  if (!ix || !first || !count) return CBH_E_INVAL;
  *first = *count = 0;
  auto it = ix->id_to_first.find(media_id);
  if (it == ix->id_to_first.end()) return CBH_OK;
  auto p = std::lower_bound(ix->first_row.begin(), ix->first_row.end(), it->second);
  *first = it->second;
  *count = (size_t)(*(p + 1) - *p);
  return CBH_OK;
}

int cbh_idx256_download_rows(const cbh_idx256* ix, size_t first, size_t count, uint8_t* out) {
  if (!ix || (count && !out) || first + count > ix->n) return CBH_E_INVAL;
  if (!count) return CBH_OK;
  if (ix->shards) {  // the range may span runs on several shards
    for (const Shards256::Seg& sg : ix->shards->segs) {
      const size_t a = std::max(first, sg.global), b = std::min(first + count, sg.global + sg.len);
      if (a >= b) continue;
      const cbh_idx256* c = ix->shards->child[sg.shard];
      cbh::DeviceGuard g(c->device);
      if (!g.ok) return CBH_E_NODEVICE;
      CBH_HIP(hipMemcpy(out + (a - first) * 32, c->d_rows + (sg.local + (a - sg.global)) * 32, (b - a) * 32,
                        hipMemcpyDeviceToHost));
    }
    return CBH_OK;
  }
  cbh::DeviceGuard g(ix->device);
  if (!g.ok) return CBH_E_NODEVICE;
  CBH_HIP(hipMemcpy(out, ix->d_rows + first * 32, count * 32, hipMemcpyDeviceToHost));
  return CBH_OK;
}

/* exact knnSearch(needles, k) restricted to distance < thresh: out_row/out_dist [nq*k] in (distance, row)
 * order, counts[nq] = number of rows under thresh (may exceed k) */
int cbh_idx256_knn(cbh_idx256* ix, const uint8_t* needles, size_t nq, int k, int thresh, uint32_t* out_row,
                   uint16_t* out_dist, uint32_t* counts) {
  if (!ix || (nq && (!needles || !out_row || !out_dist || !counts))) return CBH_E_INVAL;
  std::vector<uint32_t> row, cnt;
  std::vector<uint16_t> dist;
  int rc = knn_core(ix, needles, nq, k, thresh, &row, &dist, &cnt);
  if (rc) return rc;
  memcpy(out_row, row.data(), row.size() * 4);
  memcpy(out_dist, dist.data(), dist.size() * 2);
  memcpy(counts, cnt.data(), cnt.size() * 4);
  return CBH_OK;
}

/* knn + the mediaId of every candidate row (0 = removed): the shard-local step of the multi-GPU path, where the
 * first-row -> mediaId map is local to the shard */
int cbh_idx256_knn_media(cbh_idx256* ix, const uint8_t* needles, size_t nq, int k, int thresh, uint32_t* out_row,
                         uint16_t* out_dist, uint32_t* out_media, uint32_t* counts) {
  if (!ix || (nq && (!needles || !out_row || !out_dist || !out_media || !counts))) return CBH_E_INVAL;
  int rc = cbh_idx256_knn(ix, needles, nq, k, thresh, out_row, out_dist, counts);
  if (rc) return rc;
  for (size_t j = 0; j < nq; ++j) {
    const uint32_t len = std::min<uint32_t>((uint32_t)std::max(k, 0), counts[j]);
    for (uint32_t t = 0; t < (uint32_t)std::max(k, 0); ++t)
      out_media[j * (size_t)k + t] = t < len ? ix->media_of_row(out_row[j * (size_t)k + t]) : 0u;
  }
  return CBH_OK;
}

/* the scoring half of find() (:499-596) on a knn table with mediaIds (host code): what every rank runs on the
 * merged candidate lists in the multi-GPU path */
int cbh_cvfeatures_score(const uint32_t* media, const uint16_t* dist, const uint32_t* counts, const uint64_t* offsets,
                         size_t n_needles, int k, cbh_match* out, size_t cap, uint64_t* out_offsets) {
  if (!offsets || !out_offsets || (cap && !out) || k <= 0) return CBH_E_INVAL;
  if (n_needles && offsets[n_needles] && (!media || !dist || !counts)) return CBH_E_INVAL;
  uint64_t pos = 0;
  for (size_t i = 0; i < n_needles; ++i) {
    out_offsets[i] = pos;
    std::vector<cbh_match> res;
    score_media(media, dist, counts, (size_t)offsets[i], (size_t)offsets[i + 1], k, &res);
    for (auto& m : res) {
      if (pos < cap) out[pos] = m;
      ++pos;
    }
  }
  out_offsets[n_needles] = pos;
  return pos > cap ? CBH_E_OVERFLOW : CBH_OK;
}

/* CvFeaturesIndex::find (:438-604) for one needle with n_desc descriptor rows */
int cbh_idx256_find(cbh_idx256* ix, const uint8_t* needle_rows, size_t n_desc, int thresh, int k, cbh_match* out,
                    size_t cap, size_t* n_out) {
  if (!ix || !n_out || (cap && !out) || (n_desc && !needle_rows)) return CBH_E_INVAL;
  *n_out = 0;
  std::vector<uint32_t> row, cnt;
  std::vector<uint16_t> dist;
  int rc = knn_core(ix, needle_rows, n_desc, k, thresh, &row, &dist, &cnt);
  if (rc) return rc;
  std::vector<cbh_match> res;
  score256(ix, row.data(), dist.data(), cnt.data(), 0, n_desc, k, &res);
  *n_out = res.size();
  for (size_t i = 0; i < res.size() && i < cap; ++i) out[i] = res[i];
  return CBH_OK;
}

int cbh_idx256_find_batch(cbh_idx256* ix, const uint8_t* needle_rows, const uint64_t* offsets, size_t n_needles,
                          int thresh, int k, cbh_match* out, size_t cap, uint64_t* out_offsets) {
  if (!ix || !offsets || !out_offsets || (cap && !out)) return CBH_E_INVAL;
  const size_t nq = n_needles ? (size_t)offsets[n_needles] : 0;
  std::vector<uint32_t> row, cnt;
  std::vector<uint16_t> dist;
  int rc = knn_core(ix, needle_rows, nq, k, thresh, &row, &dist, &cnt);
  if (rc) return rc;
  uint64_t pos = 0;
  for (size_t i = 0; i < n_needles; ++i) {
    out_offsets[i] = pos;
    std::vector<cbh_match> res;
    score256(ix, row.data(), dist.data(), cnt.data(), (size_t)offsets[i], (size_t)offsets[i + 1], k, &res);
    for (auto& m : res) {
      if (pos < cap) out[pos] = m;
      ++pos;
    }
  }
  out_offsets[n_needles] = pos;
  return pos > cap ? CBH_E_OVERFLOW : CBH_OK;
}

/* cv::BFMatcher(NORM_HAMMING).radiusMatch(queryDescriptors, matches, maxDistance) against the rows of the index as
 * the train set (src/templatematcher.cpp:134,217): every (query, train) pair with distance <= max_dist, grouped by
 * query in ascending (distance, train row) order -- OpenCV sorts each query's list by distance and leaves equal
 * distances in unspecified order.  out_first[q] .. out_first[q+1] delimit query q's matches (nq + 1 entries). */
int cbh_idx256_radius_match(cbh_idx256* ix, const uint8_t* queries, size_t nq, int max_dist, cbh_dmatch* out,
                            size_t cap, uint64_t* out_first) {
  if (!ix || !out_first || (nq && !queries) || (cap && !out)) return CBH_E_INVAL;
  if (max_dist < 0) max_dist = -1;
  if (max_dist > 256) max_dist = 256;
  std::vector<unsigned long long> rec;
  int rc = knn_core(ix, queries, nq, 1, max_dist + 1, nullptr, nullptr, nullptr, &rec);
  if (rc) return rc;
  size_t p = 0;
  for (size_t q = 0; q < nq; ++q) {
    out_first[q] = p;
    while (p < rec.size() && (size_t)(rec[p] >> 41) == q) {
      if (p < cap) out[p] = cbh_dmatch{(int32_t)q, (int32_t)(uint32_t)rec[p], (int32_t)((rec[p] >> 32) & 0x1ff)};
      ++p;
    }
  }
  out_first[nq] = p;
  return p > cap ? CBH_E_OVERFLOW : CBH_OK;
}

int cbh_idx256_get_stats(const cbh_idx256* ix, cbh_stats* out) {
  if (!ix || !out) return CBH_E_INVAL;
  out->scan_launches = ix->scan_launches;
  out->scan_pairs = ix->scan_pairs;
  out->scan_ms = ix->scan_ms;
  return CBH_OK;
}

}  // extern "C"
