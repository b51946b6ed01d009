// hamm64_scan.hip -- K3: all-pairs 64-bit Hamming threshold scan for gfx950 (CDNA4).
//
// Replaces the per-needle tree walk behind DctHashIndex::find (src/dcthashindex.cpp:193-220;
// VpTree::thresholdSearch src/tree/vptree.h:228-255) with the exact brute-force predicate the
// reference states at dcthashindex.cpp:210-217:  hamm64(q, hash[i]) < thresh  &&  id[i] != 0
// (hamm64 = popcountll(a ^ b), src/hamm.h:24-26), evaluated for a whole batch of needles.
//
// Mapping to the machine (see NOTES.md "k_hamm64_scan")
//  * haystack slots live in VGPRs: each lane owns H=8 slots (16 VGPRs), a 256-thread workgroup
//    owns a tile of 2048 slots, loaded once with coalesced 8-B loads;
//  * needles are wave-uniform: they stream through the scalar cache (s_load_dwordx16 = 8 needles)
//    and feed the VALU as SGPR operands, so the inner loop has no vector memory traffic at all;
//  * per (needle, slot): v_xor_b32 + v_bcnt_u32_b32 on the low word (plus xor+bcnt-accumulate on
//    the high word in the FULL variant); two needles fold into one v_min3_u32 against a running
//    per-slot minimum.  After QB=8 needles one compare decides whether anything in the
//    8x8x64 block can be under threshold; only then the exact 64-bit distances are recomputed
//    and records are appended (wave-aggregated atomic).  PRE (low-word prefilter) is exact
//    because popc(lo) <= popc(lo)+popc(hi): a pair whose low-word distance is already >= thresh
//    cannot match.  It is used for small thresholds where the low word alone rejects almost
//    every block; FULL is used otherwise.
//  * grid = (slot tiles) x (needle chunks); consecutive blockIdx.x share a needle chunk, so the
//    workgroups resident at one time stream the same few needle chunks out of L2.
#include "cbh_internal.h"

namespace cbh {
namespace {

constexpr int kThreads = 256;
constexpr int kH = 8;   // haystack slots per lane
constexpr int kQB = 8;  // needles per check block (one s_load_dwordx16)

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
  return min(min(a, b), c);  // -> v_min3_u32
}

__device__ __forceinline__ void emit(cbh_record* __restrict__ rec, unsigned long long cap,
                                     unsigned long long* __restrict__ total, uint32_t qidx,
                                     uint32_t dist, uint32_t id) {
  unsigned long long slot = atomicAdd(total, 1ull);  // compiler aggregates per wave
  if (slot < cap) rec[slot] = ((cbh_record)qidx << 39) | ((cbh_record)dist << 32) | id;
}

// exact evaluation of needles [qa, qb) (read back from memory) against this lane's H slots
template <int H>
__device__ __forceinline__ void exact_block(const uint2 (&h)[H], uint32_t base_idx, uint32_t n,
                                            const uint32_t* __restrict__ ids,
                                            const uint64_t* __restrict__ q, uint32_t qa,
                                            uint32_t qb, uint32_t thresh,
                                            cbh_record* __restrict__ rec, unsigned long long cap,
                                            unsigned long long* __restrict__ total,
                                            uint32_t keep0, const uint2* __restrict__ qmask) {
#pragma unroll 1
  for (uint32_t qi = qa; qi < qb; ++qi) {
    const uint64_t qq = q[qi];
    if (qq == 0) continue;  // null needle: DctHashIndex::find returns nothing (:196-200)
    const uint32_t ql = (uint32_t)qq, qh = (uint32_t)(qq >> 32);
    const uint2 mk = qmask ? qmask[qi] : make_uint2(0u, 0u);  // bits that must be equal (tree/bucket modes)
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const uint32_t d = __popc(h[j].x ^ ql) + __popc(h[j].y ^ qh);
      if (d < thresh && (((h[j].x ^ ql) & mk.x) | ((h[j].y ^ qh) & mk.y)) == 0) {
        const uint32_t idx = base_idx + (uint32_t)j * kThreads;
        if (idx < n) {
          const uint32_t id = ids[idx];
          if (id != 0 || keep0) emit(rec, cap, total, qi, d, id);
        }
      }
    }
  }
}

// second level of the filter: only slots whose running minimum fell under the threshold are
// re-evaluated exactly against the QB needles of the block.  The needles are still in SGPRs
// (cur[]), so this costs ~5 VALU ops per needle and no memory access; the media id is fetched
// only for a pair that really matches (about one in 10^6 on distinct images).
template <int H, int QB>
__device__ __forceinline__ void refine_block(const uint2 (&h)[H], const uint32_t (&acc)[H],
                                             const uint2 (&cur)[QB], uint32_t base_idx, uint32_t n,
                                             const uint32_t* __restrict__ ids, uint32_t qb,
                                             uint32_t thresh, cbh_record* __restrict__ rec,
                                             unsigned long long cap,
                                             unsigned long long* __restrict__ total,
                                             uint32_t keep0, const uint2* __restrict__ qmask) {
#pragma unroll
  for (int j = 0; j < H; ++j) {
    if (acc[j] < thresh) {
      const uint32_t idx = base_idx + (uint32_t)j * kThreads;
      // exact distances of the QB needles to slot j; one branch for the (rare) real match
      uint32_t d[QB];
#pragma unroll
      for (int i = 0; i < QB; ++i) d[i] = __popc(h[j].x ^ cur[i].x) + __popc(h[j].y ^ cur[i].y);
      uint32_t m = d[0];
#pragma unroll
      for (int i = 1; i + 1 < QB; i += 2) m = min3u(m, d[i], d[i + 1]);
      if (QB % 2 == 0) m = min(m, d[QB - 1]);
      if (m < thresh && idx < n) {
        const uint32_t id = ids[idx];
        if (id != 0 || keep0) {
#pragma unroll
          for (int i = 0; i < QB; ++i)
            if (d[i] < thresh && (cur[i].x | cur[i].y) != 0) {
              const uint2 mk = qmask ? qmask[qb + (uint32_t)i] : make_uint2(0u, 0u);
              if ((((h[j].x ^ cur[i].x) & mk.x) | ((h[j].y ^ cur[i].y) & mk.y)) == 0)
                emit(rec, cap, total, qb + (uint32_t)i, d[i], id);
            }
        }
      }
    }
  }
}

enum { MODE_PRE = 0, MODE_FULL = 1, MODE_EQ = 2 };

template <int H, int QB, int MODE, bool GROUP>
__global__ __launch_bounds__(kThreads) void k_hamm64_scan(
    const uint2* __restrict__ hay, const uint32_t* __restrict__ ids, uint32_t n,
    const uint64_t* __restrict__ q, uint32_t nq, uint32_t q_chunk, uint32_t thresh,
    cbh_record* __restrict__ rec, unsigned long long cap, unsigned long long* __restrict__ total,
    uint32_t keep0, const uint2* __restrict__ qmask) {
  const uint32_t base_idx = blockIdx.x * (uint32_t)(kThreads * H) + threadIdx.x;
  uint2 h[H];
#pragma unroll
  for (int j = 0; j < H; ++j) {
    const uint32_t idx = base_idx + (uint32_t)j * kThreads;
    h[j] = idx < n ? hay[idx] : make_uint2(0u, 0u);
  }
  const uint32_t q0 = blockIdx.y * q_chunk;
  const uint32_t q1 = min(nq, q0 + q_chunk);
  uint32_t qb = q0;
  // needles as (lo,hi) dword pairs; 64-bit pointer bump keeps the 8 loads of a block at constant
  // offsets from one SGPR base (wave-uniform -> SMEM)
  const uint2* __restrict__ qp = reinterpret_cast<const uint2*>(q) + (size_t)q0;

  uint2 cur[QB];
  if (qb + QB <= q1) {
#pragma unroll
    for (int i = 0; i < QB; ++i) cur[i] = qp[i];
  }
  for (; qb + QB <= q1; qb += QB) {
    // prefetch the next block of needles while this one is being compared
    uint2 nxt[QB];
    qp += QB;
    if (qb + 2 * QB <= q1) {
#pragma unroll
      for (int i = 0; i < QB; ++i) nxt[i] = qp[i];
    } else {
#pragma unroll
      for (int i = 0; i < QB; ++i) nxt[i] = make_uint2(0u, 0u);
    }
    if (MODE == MODE_EQ) {
      // dht == 1: distance < 1 is equality -- one v_cmp_eq_u64 per pair, OR-ed on the scalar unit
      unsigned long long any = 0;
#pragma unroll
      for (int i = 0; i < QB; ++i) {
        const unsigned long long qq = ((unsigned long long)cur[i].y << 32) | cur[i].x;
#pragma unroll
        for (int j = 0; j < H; ++j) {
          const unsigned long long hh = ((unsigned long long)h[j].y << 32) | h[j].x;
          any |= __ballot(hh == qq);
        }
      }
      if (any) exact_block<H>(h, base_idx, n, ids, q, qb, qb + QB, thresh, rec, cap, total, keep0, qmask);
    } else {
      uint32_t acc[H];
#pragma unroll
      for (int j = 0; j < H; ++j) acc[j] = 0xffu;
      if (GROUP) {
        // Issue-rate shaping (tools/ubench/valu_rate.hip): VGPR-only v_xor_b32 runs at 32 lanes/clk
        // only inside long runs of such ops, while v_bcnt/v_min3 (and any op with an SGPR source)
        // run at 16 lanes/clk and cost a ~25-cycle mode switch when interleaved.  So: broadcast
        // the needles into VGPRs, do all QB*H xors back to back, then all the popcounts/minima.
        uint32_t ql[QB], qh[QB];
#pragma unroll
        for (int i = 0; i < QB; ++i) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(ql[i]) : "s"(cur[i].x));
          if (MODE == MODE_FULL) asm volatile("v_mov_b32 %0, %1" : "=v"(qh[i]) : "s"(cur[i].y));
        }
        uint32_t x[QB][H];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < QB; ++i)
#pragma unroll
          for (int j = 0; j < H; ++j) x[i][j] = h[j].x ^ ql[i];
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == MODE_FULL) {
#pragma unroll
          for (int i = 0; i < QB; ++i)
#pragma unroll
            for (int j = 0; j < H; ++j) x[i][j] = (uint32_t)__popc(x[i][j]);
          uint32_t y[QB][H];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < QB; ++i)
#pragma unroll
            for (int j = 0; j < H; ++j) y[i][j] = h[j].y ^ qh[i];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < QB; i += 2)
#pragma unroll
            for (int j = 0; j < H; ++j)
              acc[j] = min3u(acc[j], x[i][j] + (uint32_t)__popc(y[i][j]),
                             x[i + 1][j] + (uint32_t)__popc(y[i + 1][j]));
        } else {
#pragma unroll
          for (int i = 0; i < QB; i += 2)
#pragma unroll
            for (int j = 0; j < H; ++j)
              acc[j] = min3u(acc[j], (uint32_t)__popc(x[i][j]), (uint32_t)__popc(x[i + 1][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
      for (int i = 0; i < QB; i += 2) {
        const uint2 qa = cur[i];
        const uint2 qc = cur[i + 1];
#pragma unroll
        for (int j = 0; j < H; ++j) {
          uint32_t c0 = __popc(h[j].x ^ qa.x);
          uint32_t c1 = __popc(h[j].x ^ qc.x);
          if (MODE == MODE_FULL) {
            c0 += __popc(h[j].y ^ qa.y);
            c1 += __popc(h[j].y ^ qc.y);
          }
          acc[j] = min3u(acc[j], c0, c1);
        }
      }
      }
      uint32_t m = acc[0];
#pragma unroll
      for (int j = 1; j + 1 < H; j += 2) m = min3u(m, acc[j], acc[j + 1]);
      if (H % 2 == 0) m = min(m, acc[H - 1]);
      if (m < thresh) refine_block<H, QB>(h, acc, cur, base_idx, n, ids, qb, thresh, rec, cap, total, keep0, qmask);
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) cur[i] = nxt[i];
  }
  if (qb < q1) exact_block<H>(h, base_idx, n, ids, q, qb, q1, thresh, rec, cap, total, keep0, qmask);
}

// the lone needle: see cbh_internal.h.  A thread takes 8 slots 256 apart (coalesced 8-byte loads).
constexpr unsigned kLoneSlots = 8;
__global__ __launch_bounds__(256) void k_find_one(const uint2* __restrict__ hay, const uint32_t* __restrict__ ids,
                                                  uint32_t n, uint32_t qlo, uint32_t qhi, uint32_t thresh,
                                                  unsigned* __restrict__ d_state, LoneBlock* __restrict__ host,
                                                  unsigned long long seq) {
  const uint32_t base = blockIdx.x * (256u * kLoneSlots) + threadIdx.x;
  bool wrote = false;
#pragma unroll
  for (unsigned k = 0; k < kLoneSlots; ++k) {
    const uint32_t i = base + k * 256u;
    if (i >= n) break;
    const uint2 hv = hay[i];
    const uint32_t d = (uint32_t)__popc(hv.x ^ qlo) + (uint32_t)__popc(hv.y ^ qhi);
    if (d < thresh) {
      const uint32_t id = ids[i];
      if (id != 0) {
        const unsigned slot = atomicAdd(&d_state[0], 1u);
        if (slot < LoneBlock::kRecs) {
          host->recs[slot] = ((cbh_record)d << 32) | id;
          wrote = true;
        }
      }
    }
  }
  if (wrote) __threadfence_system();  // this lane's records are in host memory before its workgroup reports
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&d_state[1], 1u) == gridDim.x - 1u) {  // the last workgroup: everyone's matches are counted and written
      __threadfence();
      host->count = atomicExch(&d_state[0], 0u);
      d_state[1] = 0u;
      __threadfence_system();
      __hip_atomic_store(const_cast<unsigned long long*>(&host->done), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

constexpr int kPreMax = 7;  // largest threshold served by the low-word prefilter variant (r01: wins up to 7 on this kernel)

}  // namespace

int launch_find_one(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n, uint64_t q, int thresh, unsigned* d_state,
                    LoneBlock* h_block, unsigned long long seq, hipStream_t stream) {
  if (n == 0 || n > 0xfffffff0ull || !d_state || !h_block || thresh <= 0) return CBH_E_INVAL;
  const unsigned per_wg = 256u * kLoneSlots;
  hipLaunchKernelGGL(k_find_one, dim3((unsigned)((n + per_wg - 1) / per_wg)), dim3(256), 0, stream,
                     reinterpret_cast<const uint2*>(d_hashes), d_ids, (uint32_t)n, (uint32_t)q, (uint32_t)(q >> 32),
                     (uint32_t)thresh, d_state, h_block, seq);
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

int wait_find_one(const LoneBlock* h_block, unsigned long long seq, hipStream_t stream) {
  const unsigned long long* flag = const_cast<const unsigned long long*>(&h_block->done);
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return CBH_OK;
    if ((spins & 0xfffu) == 0xfffu) {  // every 4096 polls: is the stream still busy at all?
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) {  // the kernel has retired: its last store is visible by now, or something went wrong
        return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? CBH_OK : CBH_E_HIP;
      }
      if (q != hipErrorNotReady) {
        set_last_error("k_find_one", q);
        return CBH_E_HIP;
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

int launch_hamm64_scan(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n,
                       const uint64_t* d_q, size_t nq, int thresh, cbh_record* d_rec, size_t cap,
                       unsigned long long* d_total, hipStream_t stream, unsigned flags,
                       const uint64_t* d_qmask, const void* qx_given) {
  if (n == 0 || nq == 0 || thresh <= 0) return CBH_OK;
  if (n > 0xfffffff0ull || nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  // thresholds <= 8, "scan_mfma" 3: the bucketed join (hamm64_join.hip) when its candidate count says it is cheaper than
  // looking at every pair (only asked where a scan would take >= 1 ms); 4: whenever it can represent the call (the parity
  // suite).  As shipped (1) every pair is compared: the join avoids comparisons, it does not make them faster.
  {
    const int mode = get_scan_mfma();
    const double scan_ms = (double)n * (double)nq * (thresh <= 6 ? 8.8e-12 : 16.3e-12);  // (prefilter / three-field kernel)
    if ((mode == 4 || (mode == 3 && scan_ms >= 1.0)) && scan_join_possible(n, nq, thresh, flags, d_qmask)) {
      const int rc = launch_hamm64_join(d_hashes, d_ids, n, d_q, nq, thresh, d_rec, cap, d_total, stream, flags, mode == 4,
                                        scan_ms);
      if (rc == CBH_OK || mode == 4 || (rc != CBH_E_UNSUPPORTED && rc != CBH_E_NOMEM)) return rc;
      // (its scratch is all taken before the first record is written: a join that could not get it, like one whose count
      // said no, leaves the call to the scan)
      if (rc == CBH_E_NOMEM) cbh_clear_error();
    }
  }
  if (scan_mfma_wanted(n, nq, thresh))
    return launch_hamm64_scan_mfma(d_hashes, d_ids, n, d_q, nq, thresh, d_rec, cap, d_total, stream,
                                   flags, d_qmask, qx_given);
  const uint32_t tile = kThreads * kH;
  const uint32_t tiles = (uint32_t)((n + tile - 1) / tile);
  // needle chunk: enough workgroups to fill 256 CUs x 8 waves/SIMD several times over, but each
  // workgroup amortises its 16 KB tile load over >= 1024 needles when there are that many.
  uint32_t q_chunk = 16384;
  while (q_chunk > 1024 && (uint64_t)tiles * ((nq + q_chunk - 1) / q_chunk) < 8192) q_chunk >>= 1;
  uint32_t chunks = (uint32_t)((nq + q_chunk - 1) / q_chunk);
  if (chunks > 65535) {
    q_chunk = (uint32_t)((nq + 65534) / 65535);
    q_chunk = (q_chunk + kQB - 1) / kQB * kQB;
    chunks = (uint32_t)((nq + q_chunk - 1) / q_chunk);
  }
  dim3 grid(tiles, chunks), block(kThreads);
  const uint2* hay = reinterpret_cast<const uint2*>(d_hashes);
#define CBH_SCAN(MODE, GROUP)                                                                 \
  hipLaunchKernelGGL((k_hamm64_scan<kH, kQB, MODE, GROUP>), grid, block, 0, stream, hay,      \
                     d_ids, (uint32_t)n, d_q, (uint32_t)nq, q_chunk, (uint32_t)thresh, d_rec, \
                     (unsigned long long)cap, d_total, (uint32_t)(flags & 1u),                \
                     reinterpret_cast<const uint2*>(d_qmask))
  // (issue-rate-shaped variants only: GROUP; thresholds of 1 compare for equality)
  if (thresh == 1)
    CBH_SCAN(MODE_EQ, false);
  else if (thresh <= kPreMax)
    CBH_SCAN(MODE_PRE, true);
  else
    CBH_SCAN(MODE_FULL, true);
#undef CBH_SCAN
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace cbh
