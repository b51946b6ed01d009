// records.hip -- K4: ordering and truncation of scan records; in-place slot removal.
//
// The scan emits unordered cbh_record = query<<39 | distance<<32 | mediaId.  An ascending
// u64 sort therefore yields exactly the order cbird produces after Database::searchIndex's
// std::sort on score (src/database.cpp:1729, operator< src/index.h:284), with the unspecified
// tie order of the reference fixed to ascending mediaId.  Distances span 7 bits, so this is a
// counting problem; the batched per-needle cut no longer sorts at all (topk.hip).  What still orders ALL records --
// a single needle's complete match list, cuts with k > 64 -- uses rocPRIM's radix sort, called directly and
// restricted to the significant bits.  Selection (first max_per_query of each needle, database.cpp:1735)
// is one binary search per needle over the sorted list.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cbh_internal.h"

namespace cbh {
namespace {

int sig_bits(size_t nq) {
  int b = 0;
  while (b < 25 && ((size_t)1 << b) < nq) ++b;
  return 39 + b;
}

__device__ __forceinline__ size_t lower_bound_rec(const cbh_record* __restrict__ r, size_t n,
                                                  cbh_record key) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    size_t mid = (lo + hi) >> 1;
    if (r[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_select_records(const cbh_record* __restrict__ rec,
                                                        size_t n, uint32_t nq, int k,
                                                        cbh_match* __restrict__ out,
                                                        uint32_t* __restrict__ counts) {
  const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= nq) return;
  const size_t a = lower_bound_rec(rec, n, (cbh_record)qi << 39);
  const size_t b = lower_bound_rec(rec, n, ((cbh_record)qi + 1) << 39);
  const size_t cnt = b - a;
  counts[qi] = cnt > 0xffffffffull ? 0xffffffffu : (uint32_t)cnt;
  for (int j = 0; j < k; ++j) {
    cbh_match m;
    if ((size_t)j < cnt) {
      const cbh_record r = rec[a + (size_t)j];
      m.id = CBH_REC_ID(r);
      m.score = CBH_REC_DIST(r);
    } else {
      m.id = 0;
      m.score = 0;
    }
    out[(size_t)qi * (size_t)k + (size_t)j] = m;
  }
}

// DctHashIndex::remove (src/dcthashindex.cpp:175-191): slots whose id is in the removed set
// get id = 0 and hash = 0 in place.  rm[] is sorted ascending; one binary search per slot.
__global__ __launch_bounds__(256) void k_remove_ids(uint64_t* __restrict__ hashes,
                                                    uint32_t* __restrict__ ids, size_t n,
                                                    const uint32_t* __restrict__ rm, size_t n_rm,
                                                    int zero_hash) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = ids[i];
  size_t lo = 0, hi = n_rm;
  while (lo < hi) {
    size_t mid = (lo + hi) >> 1;
    if (rm[mid] < id)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < n_rm && rm[lo] == id) {
    ids[i] = 0;
    if (zero_hash) hashes[i] = 0;  // HammingTree::remove keeps the hash (hammingtree.h:349-361)
  }
}

}  // namespace

// ---- the library's radix sorts: rocPRIM, instantiated HERE and nowhere else (every caller sorts 64-bit keys; one copy of
// the sort kernels in the code object instead of one per translation unit) -------------------------------------------
size_t sort_records_scratch_bytes(size_t n) {
  size_t bytes = 0;
  rocprim::double_buffer<unsigned long long> db(nullptr, nullptr);  // (cbh_record is the same 64 bits under another name)
  (void)rocprim::radix_sort_keys(nullptr, bytes, db, n, 0, 64, (hipStream_t)0);
  return bytes;
}

// keys [0, n) ascending on bits [0, end_bit); d_alt = a second buffer of n keys, d_tmp / tmp_bytes from
// sort_records_scratch_bytes(>= n).  *sorted = whichever of the two buffers holds the result.
int sort_keys64_db(unsigned long long* d_keys, unsigned long long* d_alt, size_t n, unsigned end_bit, void* d_tmp,
                   size_t tmp_bytes, hipStream_t stream, unsigned long long** sorted) {
  *sorted = d_keys;
  if (n < 2) return CBH_OK;
  rocprim::double_buffer<unsigned long long> db(d_keys, d_alt);
  CBH_HIP(rocprim::radix_sort_keys(d_tmp, tmp_bytes, db, n, 0, end_bit, stream));
  *sorted = db.current();
  return CBH_OK;
}

int launch_sort_records(cbh_record* d_rec, cbh_record* d_alt, size_t n, size_t nq, void* d_tmp,
                        size_t tmp_bytes, hipStream_t stream) {
  static_assert(sizeof(cbh_record) == sizeof(unsigned long long), "record width");
  unsigned long long* cur = nullptr;
  int rc = sort_keys64_db(reinterpret_cast<unsigned long long*>(d_rec), reinterpret_cast<unsigned long long*>(d_alt), n,
                          (unsigned)sig_bits(nq), d_tmp, tmp_bytes, stream, &cur);
  if (rc) return rc;
  if (cur != reinterpret_cast<unsigned long long*>(d_rec))
    CBH_HIP(hipMemcpyAsync(d_rec, cur, n * sizeof(cbh_record), hipMemcpyDeviceToDevice, stream));
  return CBH_OK;
}

// in place, scratch from the stream-ordered allocator
int sort_keys_u64(unsigned long long* d_keys, size_t n, int end_bit, hipStream_t s) {
  if (n < 2) return CBH_OK;
  unsigned long long* alt = nullptr;
  void* tmp = nullptr;
  const size_t bytes = sort_records_scratch_bytes(n);
  cbh::Scratch scratch(s);
  CBH_HIP(scratch.get(&alt, n * 8));
  CBH_HIP(scratch.get(&tmp, bytes ? bytes : 16));
  unsigned long long* cur = nullptr;
  int rc = sort_keys64_db(d_keys, alt, n, (unsigned)end_bit, tmp, bytes, s, &cur);
  if (rc) return rc;
  if (cur != d_keys) CBH_HIP(hipMemcpyAsync(d_keys, cur, n * 8, hipMemcpyDeviceToDevice, s));
  return CBH_OK;
}

int sort_pairs_u64_u32(unsigned long long* d_keys, uint32_t* d_vals, size_t n, int end_bit, hipStream_t s) {
  if (n < 2) return CBH_OK;
  unsigned long long* kalt = nullptr;
  uint32_t* valt = nullptr;
  void* tmp = nullptr;
  size_t bytes = 0;
  CBH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_keys, kalt, d_vals, valt, n, 0, (unsigned)end_bit, s));
  cbh::Scratch scratch(s);
  CBH_HIP(scratch.get(&kalt, n * 8));
  CBH_HIP(scratch.get(&valt, n * 4));
  CBH_HIP(scratch.get(&tmp, bytes ? bytes : 16));
  CBH_HIP(rocprim::radix_sort_pairs(tmp, bytes, d_keys, kalt, d_vals, valt, n, 0, (unsigned)end_bit, s));
  CBH_HIP(hipMemcpyAsync(d_keys, kalt, n * 8, hipMemcpyDeviceToDevice, s));
  CBH_HIP(hipMemcpyAsync(d_vals, valt, n * 4, hipMemcpyDeviceToDevice, s));
  return CBH_OK;
}

int launch_select_records(const cbh_record* d_sorted, size_t n, size_t nq, int k, cbh_match* d_out,
                          uint32_t* d_counts, hipStream_t stream) {
  if (nq == 0) return CBH_OK;
  dim3 grid((unsigned)((nq + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_select_records, grid, block, 0, stream, d_sorted, n, (uint32_t)nq, k, d_out,
                     d_counts);
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

int launch_remove_ids(uint64_t* d_hashes, uint32_t* d_ids, size_t n, const uint32_t* d_sorted_rm,
                      size_t n_rm, hipStream_t stream, int zero_hash) {
  if (n == 0 || n_rm == 0) return CBH_OK;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_remove_ids, grid, block, 0, stream, d_hashes, d_ids, n, d_sorted_rm, n_rm,
                     zero_hash);
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace cbh
