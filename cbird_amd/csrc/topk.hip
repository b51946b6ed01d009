// topk.hip -- K4: per-needle counting select over unordered scan records.
//
// Replaces the global radix sort + binary-search cut of records.hip on the batched 64-bit path.  What cbird does per
// needle in Database::searchIndex is std::sort(matches) by score and a cut at maxMatches (src/database.cpp:1729-1737,
// operator< src/index.h:284); over a batch that is a SEGMENTED selection, and every key is small (distance 0..64,
// then mediaId as the tie-break that fixes the reference's unspecified tie order), so no comparison sort of the whole
// record list is needed:
//
//   count    cnt[needle] += 1 for every record                                   (one atomic per record)
//   scan     off = exclusive prefix sum of cnt                                   (needle-major segment offsets)
//   scatter  seg[off[needle] + (--cnt[needle])] = distance<<32 | mediaId         (cnt returns to all-zero: the scratch
//                                                                                 cleans itself for the next call)
//   select   per needle: its first k entries in ascending (distance, mediaId) order by repeated minimum -- one lane
//            per needle for the short segments that dominate (a 1M x 1M dht-8 sweep averages 1.2 records per
//            needle), one workgroup per needle for the long ones (duplicates, videos)
//
// Input is a list of BLOCKS: block b = { u64 count; u64 records[cap]; } at d_blocks + b*stride.  The scan kernels write
// exactly that (count and records adjacent), and one all_gather_into_tensor of R such blocks is the whole multi-GPU
// exchange (cbird_amd/dist.py): no count ever has to visit the host, so nothing on this path synchronises.  A block
// whose count exceeds cap (records were dropped by the scan) sets bit 0 of *d_status; the caller rescans with a
// larger capacity when it eventually looks.
#include "cbh_internal.h"

namespace cbh {
namespace {

constexpr int kShortMax = 64;   // segments up to this length are selected by one lane
constexpr int kScanTile = 2048;  // elements per workgroup in the prefix-sum kernels (256 lanes x 8)

__device__ __forceinline__ bool slot_record(const unsigned long long* __restrict__ blocks, unsigned nb,
                                            size_t stride, size_t cap, size_t t, unsigned long long* rec) {
  const size_t b = t / cap, i = t - b * cap;
  if (b >= nb) return false;
  const unsigned long long* blk = blocks + b * stride;
  const unsigned long long c = blk[0];
  if (i >= (c < cap ? c : cap)) return false;
  *rec = blk[1 + i];
  return true;
}

__global__ __launch_bounds__(256) void k_topk_count(const unsigned long long* __restrict__ blocks, unsigned nb,
                                                    size_t stride, size_t cap, unsigned nq,
                                                    unsigned* __restrict__ cnt, unsigned* __restrict__ status) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nb && blocks[t * stride] > cap) atomicOr(status, 1u);  // this block lost records: results incomplete
  unsigned long long r;
  if (!slot_record(blocks, nb, stride, cap, t, &r)) return;
  const unsigned long long q = r >> 39;
  if (q < nq) atomicAdd(&cnt[q], 1u);
}

// exclusive prefix sum of cnt[0..n) into off[0..n], three small kernels (n <= 2^25 -> <= 16384 tiles)
__global__ __launch_bounds__(256) void k_scan_tiles(const unsigned* __restrict__ cnt, unsigned n,
                                                    unsigned* __restrict__ tile_sum) {
  __shared__ unsigned s[256];
  const unsigned base = blockIdx.x * kScanTile + threadIdx.x * 8;
  unsigned v = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (base + i < n) v += cnt[base + i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void k_scan_tile_sums(unsigned* __restrict__ tile_sum, unsigned nt,
                                                         unsigned* __restrict__ off_total) {
  // one workgroup: exclusive scan of up to 16384 tile sums in place (16 per lane)
  __shared__ unsigned s[1024];
  unsigned loc[16];
  unsigned v = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned j = threadIdx.x * 16 + i;
    loc[i] = j < nt ? tile_sum[j] : 0u;
    v += loc[i];
  }
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const unsigned add = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned run = s[threadIdx.x] - v;  // exclusive prefix of this lane's 16
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned j = threadIdx.x * 16 + i;
    if (j < nt) tile_sum[j] = run;
    run += loc[i];
  }
  if (threadIdx.x == 1023) *off_total = s[1023];
}

__global__ __launch_bounds__(256) void k_scan_apply(const unsigned* __restrict__ cnt, unsigned n,
                                                    const unsigned* __restrict__ tile_off,
                                                    unsigned* __restrict__ off /* n + 1 */,
                                                    const unsigned* __restrict__ off_total) {
  __shared__ unsigned s[256];
  const unsigned base = blockIdx.x * kScanTile + threadIdx.x * 8;
  unsigned loc[8], v = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    loc[i] = base + i < n ? cnt[base + i] : 0u;
    v += loc[i];
  }
  s[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const unsigned add = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0u;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned run = tile_off[blockIdx.x] + s[threadIdx.x] - v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (base + i < n) off[base + i] = run;
    run += loc[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = *off_total;
}

__global__ __launch_bounds__(256) void k_topk_scatter(const unsigned long long* __restrict__ blocks, unsigned nb,
                                                      size_t stride, size_t cap, unsigned nq,
                                                      unsigned* __restrict__ cnt, const unsigned* __restrict__ off,
                                                      unsigned long long* __restrict__ seg) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long r;
  if (!slot_record(blocks, nb, stride, cap, t, &r)) return;
  const unsigned long long q = r >> 39;
  if (q >= nq) return;
  const unsigned p = atomicSub(&cnt[q], 1u) - 1u;
  seg[off[q] + p] = r & ((1ull << 39) - 1);
}

// Emit the first k of seg[0..len) in ascending order (duplicates kept): repeatedly take the smallest value above the
// previous one together with its multiplicity.
__device__ __forceinline__ void emit_match(cbh_match* __restrict__ out, unsigned long long v) {
  cbh_match m;
  m.id = (uint32_t)(v & 0xffffffffull);
  m.score = (int32_t)(v >> 32);
  *out = m;
}

__global__ __launch_bounds__(256) void k_topk_select(const unsigned long long* __restrict__ seg,
                                                     const unsigned* __restrict__ off, unsigned nq, int k,
                                                     cbh_match* __restrict__ out, uint32_t* __restrict__ counts,
                                                     unsigned* __restrict__ long_list, unsigned* __restrict__ n_long) {
  const unsigned j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nq) return;
  const unsigned a = off[j], len = off[j + 1] - a;
  counts[j] = len;
  cbh_match* o = out + (size_t)j * (size_t)k;
  if (len > (unsigned)kShortMax && k > 0) {
    long_list[atomicAdd(n_long, 1u)] = j;
    return;
  }
  int emitted = 0;
  if (len == 1) {  // by far the most common non-empty case
    if (k > 0) emit_match(o, seg[a]), emitted = 1;
  } else if (len > 1) {
    unsigned long long last = 0;
    bool have_last = false;
    while (emitted < k && (unsigned)emitted < len) {
      unsigned long long best = ~0ull;
      unsigned mult = 0;
      for (unsigned i = 0; i < len; ++i) {
        const unsigned long long v = seg[a + i];
        if (have_last && v <= last) continue;
        if (v < best) best = v, mult = 1;
        else if (v == best) ++mult;
      }
      for (unsigned c = 0; c < mult && emitted < k; ++c) emit_match(o + emitted++, best);
      last = best, have_last = true;
    }
  }
  for (; emitted < k; ++emitted) {
    cbh_match z;
    z.id = 0, z.score = 0;
    o[emitted] = z;
  }
}

// long segments: one workgroup per needle, k rounds of a workgroup-wide (minimum, multiplicity) reduction
__global__ __launch_bounds__(256) void k_topk_long(const unsigned long long* __restrict__ seg,
                                                   const unsigned* __restrict__ off, int k,
                                                   cbh_match* __restrict__ out,
                                                   const unsigned* __restrict__ long_list,
                                                   const unsigned* __restrict__ n_long) {
  __shared__ unsigned long long s_best[256];
  __shared__ unsigned s_mult[256];
  const unsigned nl = *n_long;
  for (unsigned li = blockIdx.x; li < nl; li += gridDim.x) {
    const unsigned j = long_list[li];
    const unsigned a = off[j], len = off[j + 1] - a;
    cbh_match* o = out + (size_t)j * (size_t)k;
    int emitted = 0;
    unsigned long long last = 0;
    bool have_last = false;
    while (emitted < k && (unsigned)emitted < len) {
      unsigned long long best = ~0ull;
      unsigned mult = 0;
      for (unsigned i = threadIdx.x; i < len; i += 256) {
        const unsigned long long v = seg[a + i];
        if (have_last && v <= last) continue;
        if (v < best) best = v, mult = 1;
        else if (v == best) ++mult;
      }
      s_best[threadIdx.x] = best, s_mult[threadIdx.x] = mult;
      __syncthreads();
      for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
          const unsigned long long ob = s_best[threadIdx.x + d];
          const unsigned om = s_mult[threadIdx.x + d];
          if (ob < s_best[threadIdx.x]) s_best[threadIdx.x] = ob, s_mult[threadIdx.x] = om;
          else if (ob == s_best[threadIdx.x]) s_mult[threadIdx.x] += om;
        }
        __syncthreads();
      }
      best = s_best[0], mult = s_mult[0];
      __syncthreads();
      if (threadIdx.x == 0)
        for (unsigned c = 0; c < mult && emitted + (int)c < k; ++c) emit_match(o + emitted + c, best);
      emitted += (int)(mult < (unsigned)(k - emitted) ? mult : (unsigned)(k - emitted));
      last = best, have_last = true;
    }
    // (len > kShortMax >= ... so emitted == k unless k > len: pad)
    if (threadIdx.x == 0)
      for (int e = emitted; e < k; ++e) {
        cbh_match z;
        z.id = 0, z.score = 0;
        o[e] = z;
      }
  }
}

}  // namespace

// scratch layout (u32 words unless noted), all inside one allocation the caller keeps between calls:
//   cnt[nq] (kept all-zero between calls) | off[nq+1] | tile[16384] | misc[4] = {off_total, n_long, -, -} |
//   long_list[nq] | seg[total_cap] u64
size_t topk_scratch_bytes(size_t nq, size_t total_cap) {
  const size_t words = nq + (nq + 1) + 16384 + 4 + nq;
  return ((words * 4 + 15) & ~(size_t)15) + total_cap * 8 + 16;
}

// The first call on a fresh scratch must find cnt zeroed: topk_scratch_init does that.
int topk_scratch_init(void* d_scratch, size_t nq, hipStream_t stream) {
  CBH_HIP(hipMemsetAsync(d_scratch, 0, nq * sizeof(unsigned), stream));
  return CBH_OK;
}

// count -> scan -> scatter: groups the records of the blocks by needle.  On return (stream-ordered) off[0..nq] holds the
// segment offsets and seg[off[q] .. off[q+1]) the records of needle q as distance<<32 | payload, unordered.
int launch_records_group(const unsigned long long* d_blocks, unsigned nb, size_t stride, size_t cap, size_t nq,
                         unsigned* d_status, void* d_scratch, const unsigned** d_off, const unsigned long long** d_seg,
                         hipStream_t stream) {
  const size_t total_cap = (size_t)nb * cap;
  if (nq == 0 || nq > ((size_t)1 << 25) || total_cap >= ((size_t)1 << 32)) return CBH_E_INVAL;
  unsigned* cnt = (unsigned*)d_scratch;
  unsigned* off = cnt + nq;
  unsigned* tile = off + nq + 1;
  unsigned* misc = tile + 16384;
  const size_t words = nq + (nq + 1) + 16384 + 4 + nq;
  unsigned long long* seg = (unsigned long long*)((char*)d_scratch + ((words * 4 + 15) & ~(size_t)15));
  const unsigned nt = (unsigned)((nq + kScanTile - 1) / kScanTile);
  const unsigned gslots = (unsigned)((std::max<size_t>(total_cap, nb) + 255) / 256);
  CBH_HIP(hipMemsetAsync(misc, 0, 4 * sizeof(unsigned), stream));
  if (total_cap)
    hipLaunchKernelGGL(k_topk_count, dim3(gslots), dim3(256), 0, stream, d_blocks, nb, stride, cap, (unsigned)nq, cnt,
                       d_status);
  hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, stream, cnt, (unsigned)nq, tile);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(1024), 0, stream, tile, nt, misc);
  hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(256), 0, stream, cnt, (unsigned)nq, tile, off, misc);
  if (total_cap)
    hipLaunchKernelGGL(k_topk_scatter, dim3(gslots), dim3(256), 0, stream, d_blocks, nb, stride, cap, (unsigned)nq,
                       cnt, off, seg);
  CBH_HIP(hipGetLastError());
  if (d_off) *d_off = off;
  if (d_seg) *d_seg = seg;
  return CBH_OK;
}

int launch_records_topk(const unsigned long long* d_blocks, unsigned nb, size_t stride, size_t cap, size_t nq, int k,
                        cbh_match* d_out, uint32_t* d_counts, unsigned* d_status, void* d_scratch,
                        hipStream_t stream) {
  if (nq == 0) return CBH_OK;
  if (k < 0) return CBH_E_INVAL;
  const unsigned* off = nullptr;
  const unsigned long long* seg = nullptr;
  int rc = launch_records_group(d_blocks, nb, stride, cap, nq, d_status, d_scratch, &off, &seg, stream);
  if (rc) return rc;
  unsigned* misc = (unsigned*)d_scratch + nq + (nq + 1) + 16384;
  unsigned* long_list = misc + 4;
  hipLaunchKernelGGL(k_topk_select, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, seg, off, (unsigned)nq,
                     k, d_out, d_counts, long_list, misc + 1);
  if (k > 0)
    hipLaunchKernelGGL(k_topk_long, dim3(1024), dim3(256), 0, stream, seg, off, k, d_out, long_list, misc + 1);
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace cbh
