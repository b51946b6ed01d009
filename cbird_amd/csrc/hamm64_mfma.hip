// hamm64_mfma.hip -- K3m: the all-pairs 64-bit Hamming threshold scan on the gfx950 matrix cores.
//
// Same contract as k_hamm64_scan (hamm64_scan.hip): every pair with
//   hamm64(q, hash[i]) < thresh  &&  id[i] != 0  &&  q != 0
// (src/dcthashindex.cpp:196-217, hamm64 = popcountll(a ^ b), src/hamm.h:24-26) is appended as
// a cbh_record.  Only the arithmetic differs.
//
// Why the matrix cores.  PMC shows the VALU scan is bound by integer-VALU issue (one
// v_bcnt_u32_b32 per 32 bits per pair), not by memory: 0.8 GB of HBM traffic per 10^12 pairs.
// The distance is also a dot product of sign vectors (fp4_sign.h),
//   dot(s(a), s(b)) = 64 - 2 * hamm64(a, b),
// +-1.0 are exact in FP4, so ONE v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 = one hash) yields
// the exact distances of 32 haystack rows x 32 needles: 1024 pairs in ~32 matrix-core cycles,
// against ~8.3 (prefilter) / 14.3 (full) VALU cycles per 64 pairs.  All sums are small integers,
// so the f32 accumulation is exact and results stay bit-identical.
//
// Keeping the VALU out of the way.  16 f32 results per lane per MFMA would cost 8 v_max3_f32
// (32 cycles) to reduce -- as much as the MFMA itself.  Two tricks halve that:
//   * two needle tiles share one accumulator: the second is multiplied by the MX block scale
//     2^15 on top of C0 = 2^23 + 0x4040 + 64*2^15.  In [2^23, 2^24) one f32 ulp is 1, so the
//     mantissa holds  (0x4040 + dotA) + 2^15 * (64 + dotB)  exactly, i.e. the f32 bit pattern is
//     hi16 = 0x4B00 + (64 + dotB)/2,  lo16 = 0x4040 + dotA  (dotB even => bit 15 is 0): two
//     distances per register, each half monotone in its distance;
//   * both halves are positive normal f16 bit patterns, so v_pk_maximum3_f16 (new on gfx950) takes
//     the per-half maximum of three registers at once: 4 ops per MFMA instead of 8.
//
// Three variants (all exact), chosen by the launcher (pick_pre below: per launch, from the candidate rate of the data):
//   FULL3 (k_hamm64_mfma3, thresholds up to 64 that the prefilter does not take) three needle tiles per accumulator,
//         detection by OR of flag bits -- described at the kernel below.
//   FULL2 (thresh 65 only) K = the 64 bits of one hash; tile B is a second MFMA accumulated
//         onto tile A's.  hi16 = 0x4B40 - distB, lo16 = 0x4080 - 2*distA.  Hits are real matches.
//   PRE   (small thresholds while candidates are rare) 32-bit prefilter at twice the pair rate.  The 32-bit word is the FOLD
//         f(x) = lo(x) ^ hi(x): bit i of f(a) ^ f(b) is the XOR of bits i and i + 32 of a ^ b, so
//         popc(f(a) ^ f(b)) <= popc(a ^ b) -- a lower bound on the distance that looks at all 64 bits.  (Round 1-4
//         used the low word alone: also a lower bound, but the low-frequency coefficients of images agree far more
//         often than chance; on image-derived hashes the fold passes 2-3x fewer false candidates -- exactly the rate
//         of uniform random words -- NOTES 11.)  The block scale is per lane and K block, so lanes 0-31 (K 0..31)
//         carry the folds of one needle tile and lanes 32-63 (K 32..63) the folds of the next, against the
//         haystack's folds in both K blocks: ONE MFMA = 2048 fold distances.  Two such MFMAs are chained into one
//         accumulator with block scales 2^-1 | 2^5 and 2^11 | 2^17: four 6-bit flag-bit fields per register (the
//         top one flags by carrying into the exponent), reduced with v_or3_b32 -- one result VGPR per 256
//         comparisons.  Candidates are re-evaluated on the full 64 bits.
//
// Hits.  After the MFMAs of a group of G haystack tiles one compare of the reduced flags decides
// whether anything is under the threshold.  Then the lanes that hold flagged results list them in wave-private
// LDS and the whole wave works the list off, one candidate per lane.  RECORDS are not written one by one either:
// every append to the result block moves its counter, ONE address for the whole device, and the L2 takes ~10 ns
// per atomic on one address whatever its operand -- 10^6 self matches of a self-join are 10 ms of serial atomics
// beside a 9-16 ms scan, 5 x 10^6 duplicate matches tripled the three-field kernel's time (profiles/
// r06_adaptive_ab_before.jsonl).  A wave parks its records in LDS (kOutCap of them) and appends them with one atomic.
//
// Layout.  A workgroup is 4 waves; each wave keeps HT haystack tiles (32 rows each) expanded
// to FP4 in VGPRs (4 VGPRs per tile: lane (r, half) holds word `half` of row r; PRE: the fold
// lo ^ hi in both halves) and streams needle tiles -- pre-expanded once per call by
// k_expand_needles into a 32-byte-per-needle scratch -- through 16-byte loads that the 4 waves
// share in L1/L2.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "cbh_internal.h"
#include "fp4_sign.h"

namespace cbh {
namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kHT = 8;  // haystack tiles a wave keeps in registers (256 rows)
constexpr int kG = 2;   // tiles per accumulator group
// 2^23 + 0x4040 + 64 * 2^15
constexpr float kC0 = 8388608.0f + 16448.0f + 2097152.0f;
constexpr int kScale15 = 0x8e8e8e8e;  // E8M0 142 = 2^15
constexpr uint32_t kQueue = 2048;     // words of wave-private LDS behind the prefilter / two-field kernels
// PRE keeps FOUR prefilter-word distances per accumulator register as 6-bit fields at bits 0, 6, 12, 18 (two chained
// MFMAs; see the kernel), biased so that "under the threshold" is bit 5 of the field; the top field's flag is the carry
// into the f32 exponent (bit 23 of the pattern).  OR-ing accumulators preserves "some flag is set".
constexpr uint32_t kFlagMaskPre = (1u << 5) | (1u << 11) | (1u << 17) | (1u << 23);
constexpr int kScaleHalf = 0x7e7e7e7e;       // E8M0 126 = 2^-1
constexpr int kScale5 = (int)0x84848484;     // 2^5
constexpr int kScale11 = (int)0x8a8a8a8a;    // 2^11
constexpr int kScale17 = (int)0x90909090;    // 2^17

// needles -> FP4 scratch: needle j -> 2 x uint4 (low word, high word) at qx[2j], and behind those (qx[2 * nq_pad + j])
// the prefilter word lo ^ hi of needle j; j >= nq padded with hash 0
__global__ __launch_bounds__(256) void k_expand_needles(const uint64_t* __restrict__ q, uint32_t nq,
                                                        uint32_t nq_pad, uint4* __restrict__ qx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // one thread per output uint4
  if (i >= 3u * nq_pad) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
  if (i < 2u * nq_pad) {
    qx[i] = fp4_expand32((i >> 1) < nq ? w[i] : 0u);
  } else {
    const uint32_t j = i - 2u * nq_pad;
    qx[i] = fp4_expand32(j < nq ? w[2u * j] ^ w[2u * j + 1u] : 0u);
  }
}

__device__ __forceinline__ h2 as_h2(float f) { return __builtin_bit_cast(h2, f); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ h2 pkmax3(h2 a, h2 b, h2 c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);  // v_pk_maximum3_f16
}
// The queues below are wave-private and the LDS executes one wave's instructions in order, so a
// ds_read issued after a ds_write of another lane of the same wave sees it: only the COMPILER must
// be kept from reordering or caching LDS accesses across the hand-over points (a compiler-level memory
// clobber; `volatile` would make it wait for every outstanding needle prefetch at each access).
__device__ __forceinline__ void wave_order() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

struct HitParams {
  uint32_t lo_key, hi_key, lo_zero, hi_zero, thresh, n, nq, keep0;
  const uint64_t* q;
  const uint32_t* ids;
  cbh_record* rec;
  unsigned long long cap;
  unsigned long long* total;
  const uint2* hay;      // raw slot hashes, read only for the optional equal-bits filter
  const uint2* qmask;    // optional: bits of (needle ^ slot) that must be zero
};

// ---- records: parked per wave, appended with one atomic ---------------------------------------------------------------
// s_out = 2 * kOutCap words of the wave's LDS, nout = records parked (wave-uniform; callers keep it in an SGPR).
// All three functions must be reached by the WHOLE wave (uniform control flow: they ballot).
constexpr uint32_t kOutCap = 128;
// (not inlined, the flush costs the prefilter kernel 144 bytes of scratch around the call and 1.5 ms per launch)
__device__ __forceinline__ void out_flush(uint32_t* s_out, uint32_t& nout, const HitParams& hp) {
  if (nout == 0) return;
  wave_order();
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(hp.total, (unsigned long long)nout);
  const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
  base = ((unsigned long long)bhi << 32) | blo;
  for (uint32_t k = lane; k < nout; k += 64u) {
    const uint2 r = *reinterpret_cast<const uint2*>(&s_out[2u * k]);
    if (base + k < hp.cap) hp.rec[base + k] = ((cbh_record)r.y << 32) | r.x;  // (past cap: counted, not stored)
  }
  wave_order();
  nout = 0;
}
// every lane with `has` contributes the record  needle qidx << 39 | dist << 32 | id
__device__ __forceinline__ void out_push(uint32_t* s_out, uint32_t& nout, bool has, uint32_t qidx, uint32_t dist,
                                         uint32_t id, const HitParams& hp) {
  const uint64_t m = __builtin_amdgcn_ballot_w64(has);
  if (m == 0) return;
  const uint32_t c = (uint32_t)__popcll(m);
  if (nout + c > kOutCap) out_flush(s_out, nout, hp);
  if (has) {
    const uint32_t pos = nout + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    *reinterpret_cast<uint2*>(&s_out[2u * pos]) = make_uint2(id, (qidx << 7) | dist);
  }
  nout += c;
}

// the reference's approximate structures compare a needle only with entries sharing its low bits
__device__ __forceinline__ bool mask_ok(const HitParams& hp, uint32_t row, uint32_t qi, uint64_t nv) {
  if (!hp.qmask) return true;
  const uint2 hv = hp.hay[row], mk = hp.qmask[qi];
  return (((hv.x ^ (uint32_t)nv) & mk.x) | ((hv.y ^ (uint32_t)(nv >> 32)) & mk.y)) == 0;
}

// FULL2, cold path (inlined once per haystack tile of the step's call site).
// One haystack tile's 16 accumulators: the lanes holding flagged results append
//   dist<<11 | field<<10 | g<<6 | lane
// to the wave's LDS queue (ballot + mbcnt compaction, count in an SGPR), then the whole wave drains
// the queue, one candidate per lane.  C/D layout of the 32x32 MFMA: column = lane & 31 -> needle,
// row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5) -> haystack row in the tile.
__device__ __forceinline__ void handle_tile2(const v16f& c, uint32_t row0, uint32_t p, const HitParams& hp,
                                             uint32_t* s_queue, uint32_t* s_out, uint32_t& nout) {
  const uint32_t lane = threadIdx.x & 63u;
  {  // quick reject of the tile that did not cause the group's hit
    h2 m0 = {0, 0}, m1 = {0, 0};
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
      m0 = pkmax3(m0, as_h2(c[g]), as_h2(c[g + 1]));
      m1 = pkmax3(m1, as_h2(c[g + 2]), as_h2(c[g + 3]));
    }
    const uint32_t tb = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(m0, m1));
    if (__builtin_amdgcn_ballot_w64((tb << 16) >= hp.lo_key || tb >= hp.hi_key) == 0) return;
  }
  uint32_t cnt = 0;  // wave-uniform
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const uint32_t bits = as_u32(c[g]);
    const bool fh = bits >= hp.hi_key;
    const bool fl = (bits << 16) >= hp.lo_key;
    if (__builtin_amdgcn_ballot_w64(fh || fl) == 0) continue;  // scalar branch, rarely not taken
    const uint64_t mh = __builtin_amdgcn_ballot_w64(fh);
    if (fh)
      s_queue[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u))] =
          ((hp.hi_zero - (bits >> 16)) << 11) | (1u << 10) | ((uint32_t)g << 6) | lane;
    cnt += (uint32_t)__popcll(mh);
    const uint64_t ml = __builtin_amdgcn_ballot_w64(fl);
    if (fl)
      s_queue[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u))] =
          (((hp.lo_zero - (bits & 0xffffu)) >> 1) << 11) | ((uint32_t)g << 6) | lane;
    cnt += (uint32_t)__popcll(ml);
  }
  wave_order();
  // drain: one candidate per lane
  for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
    const uint32_t k = k0 + lane;
    const uint32_t e = s_queue[min(k, cnt - 1u)];
    const uint32_t src = e & 63u, g = (e >> 6) & 15u, field = (e >> 10) & 1u;
    const uint32_t rit = (g & 3u) + 8u * (g >> 2) + 4u * (src >> 5);  // row in tile
    const uint32_t row = row0 + rit;
    const uint32_t qi = p * 64u + field * 32u + (src & 31u);
    const uint32_t d = e >> 11;
    bool has = false;
    uint32_t id = 0;
    if (k < cnt && row < hp.n && qi < hp.nq) {
      const uint64_t nv = hp.q[qi];
      if (nv != 0 && d < hp.thresh && mask_ok(hp, row, qi, nv)) {
        id = hp.ids[row];
        has = id != 0 || hp.keep0;
      }
    }
    out_push(s_out, nout, has, qi, d, id, hp);
  }
  wave_order();
}

// PRE, the candidate path (hit lanes in rounds of kParkLanes; one round and one lane is the common case while candidates
// are rare: 8192 pairs per group x 1e-5 .. 6e-5 per pair).  A candidate costs the matrix pipe nothing but a handful of VALU
// slots when it is FOUND and is re-checked LATER, 64 at a time:
//   * each lane that holds a flag parks its accumulators in LDS (ds_write_b128 straight from the MFMA result registers,
//     under the lanes' own exec mask: LDS issue, no VALU); hit lane by hit lane, lane r reads register r back, ONE
//     ballot names the flagged registers, and those lanes append a descriptor {register pattern, register | tile |
//     lane | step} to the wave's pending list -- ~10 VALU instructions per hit lane, no global memory access, so the
//     wave is back at its MFMAs after two LDS round trips;
//   * when 64 descriptors are pending (and at the end of the wave's needle chunk) the wave drains the list, one
//     descriptor per lane: haystack hash from LDS, the needles of the flagged fields from global memory (64 lanes'
//     loads in flight together), popcount; real matches go to the wave's record buffer.
// (An immediate scalar re-check -- s_load + s_bcnt1 per candidate -- was built first: no VALU at all, but every event
//  stalled the wave for a scalar-cache miss, 15.9 ms at threshold 6 where this form takes ~11; NOTES 11.  The per-tile
//  queue path of rounds 1-4 -- ~750 cycles per candidate group -- is in the history: r05's "scan_pre_lean" 0.)
// The wave's 2048 words: pending list (2 words per descriptor) | record buffer | parking area.
constexpr uint32_t kParkLanes = 16;  // hit lanes parked at a time (a denser group takes several rounds)
constexpr uint32_t kPark = kQueue - kParkLanes * 32u;  // word offset of the parking area
constexpr uint32_t kOutOff = kPark - 2u * kOutCap;     // ... of the record buffer; pending: < 64 + 16 * 32 <= kOutOff / 2 descriptors
static_assert(kOutOff / 2u >= 64u + kParkLanes * 32u, "pending list");

// OR of accumulator registers [A, B) of a group (register r = tile r / 16, element r % 16), three and then two per
// v_or3_b32
template <int A, int B, int G>
__device__ __forceinline__ uint32_t or_regs(const v16f (&c)[G]) {
  static_assert(B - A >= 3, "range");
  uint32_t o = as_u32(c[A >> 4][A & 15]) | as_u32(c[(A + 1) >> 4][(A + 1) & 15]) | as_u32(c[(A + 2) >> 4][(A + 2) & 15]);
#pragma unroll
  for (int r = A + 3; r < B; r += 2)
    o |= as_u32(c[r >> 4][r & 15]) | (r + 1 < B ? as_u32(c[(r + 1) >> 4][(r + 1) & 15]) : 0u);
  return o;
}

// PRE = true: the prefilter kernel (4 workgroups per CU: 128 VGPRs; bound by VALU issue, and a fourth wave per SIMD hides
// more of it -- same box, compiled for 1 / 2 / 3 / 4: 10.8 / 10.8 / 10.1 / 9.8 ms, r05).  PRE = false: FULL2.
template <bool PRE>
__global__ __launch_bounds__(kThreads, PRE ? 4 : 1) void k_hamm64_mfma(
    const uint2* __restrict__ hay, const uint32_t* __restrict__ ids, uint32_t n,
    const uint64_t* __restrict__ q, const uint4* __restrict__ qx, uint32_t nq, uint32_t n_pairs,
    uint32_t pairs_per_chunk, uint32_t thresh, cbh_record* __restrict__ rec,
    unsigned long long cap, unsigned long long* __restrict__ total, uint32_t keep0,
    const uint2* __restrict__ qmask, const uint4* __restrict__ qf) {
  constexpr int HT = kHT, G = kG;
  __shared__ uint32_t s_queue_[kWaves][kQueue];
  __shared__ uint2 s_hay_[PRE ? kWaves : 1][PRE ? HT * 32 : 1];  // PRE: raw hashes for the re-check
  __shared__ uint32_t s_out2_[PRE ? 1 : kWaves][PRE ? 1 : 2 * kOutCap];  // FULL2: the record buffer (PRE: inside s_queue)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // uniform, and known to be
  const uint32_t r = lane & 31u, half = lane >> 5;
  const uint32_t tile0 = (blockIdx.x * kWaves + wave) * HT;
  if (tile0 * 32u >= n) return;  // whole wave past the end (no workgroup barriers in this kernel)
  uint32_t* s_queue = s_queue_[wave];
  uint2* s_hay = s_hay_[PRE ? wave : 0];
  uint32_t* s_out = PRE ? s_queue + kOutOff : s_out2_[PRE ? 0 : wave];
  uint32_t nout = 0;  // records parked in s_out (wave-uniform)

  v8i a[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const uint32_t row = (tile0 + t) * 32u + r;
    const uint2 hv = row < n ? hay[row] : make_uint2(0u, 0u);
    // PRE: the prefilter word lo ^ hi in both K blocks
    a[t] = fp4_operand(fp4_expand32(PRE ? hv.x ^ hv.y : (half ? hv.y : hv.x)));
    if (PRE && half == 0) s_hay[t * 32 + r] = hv;
  }
  wave_order();
  // FULL2: C0 = kC0 (two 16-bit fields compared against per-threshold keys).
  // PRE:   a step takes TWO needle pairs = four needle tiles: MFMA 1 carries tiles 0 | 1 in its K blocks with block
  //        scales 2^-1 | 2^5, MFMA 2 (accumulating onto it) tiles 2 | 3 with 2^11 | 2^17, and
  //        C0 = 2^23 + (16 + b)(1 + 2^6 + 2^12 + 2^18), b = thresh - 1, so that field i (6 bits at bit 6i) holds
  //        16 + b + dot_lo_i / 2 = 32 + b - dlo_i in [b, 32 + b]:  dlo_i <= b  <=>  field >= 32  <=>  bit 5 of the field;
  //        for the top field that is a carry out of the mantissa, i.e. bit 23 of the f32 pattern (the exponent goes
  //        from 150 to 151).  The +-0.5 products of the first K block are exact at an accumulator of 2^23 because the
  //        hardware adds the 32 products of a block (an integer) before it meets the accumulator -- checked on
  //        4.3e9 results incl. 1e7 hits by tools/ubench/mfma_half_exact.hip.  One result VGPR now answers 256
  //        prefilter comparisons instead of 128: half the v_or3_b32 per comparison.
  v16f c0;
#pragma unroll
  for (int g = 0; g < 16; ++g)
    c0[g] = PRE ? 8388608.0f + (float)((16u + (thresh - 1u)) * 266305u) : kC0;  // 266305 = 1 + 2^6 + 2^12 + 2^18
  asm volatile("" : "+v"(c0));  // keep C0 resident: otherwise it is rebuilt (16 v_mov) every trip
  // PRE block scales (per lane half = per K block): first MFMA 2^-1 | 2^5, second 2^11 | 2^17
  int scale_b = PRE ? (half ? kScale5 : kScaleHalf) : kScaleOne;
  int scale_b2 = PRE ? (half ? kScale17 : kScale11) : kScale15;
  asm volatile("" : "+v"(scale_b));
  asm volatile("" : "+v"(scale_b2));

  const uint32_t p0 = blockIdx.y * pairs_per_chunk;
  const uint32_t p1 = min(n_pairs, p0 + pairs_per_chunk);
  // pair p = needles [64p, 64p+64): tile A = first 32, tile B = last 32; 2 uint4 per needle.
  // FULL2: lane (c, half) reads word `half` of needle c of each tile; PRE: the prefilter word of needle
  // 64p + lane (tile A in K block 0, tile B in K block 1), one uint4 per needle in qf
  const uint4* __restrict__ qp = qx + ((size_t)p0 * 64u + r) * 2u + half;  // (FULL2; PRE addresses its tiles below)
  const uint32_t lo_zero = 0x4080u;                      // lo16 at distance 0
  const uint32_t hi_zero = 0x4B40u;                      // hi16 at distance 0
  const uint32_t lo_thr = lo_zero - 2u * (thresh - 1u);  // lo16 >= lo_thr  <=>  distA < thresh
  const uint32_t hi_thr = hi_zero - (thresh - 1u);       // hi16 >= hi_thr  <=>  distB < thresh
  const uint32_t lo_key = lo_thr << 16, hi_key = hi_thr << 16;
  const HitParams hp = {lo_key, hi_key, lo_zero, hi_zero, thresh, n, nq, keep0, q, ids, rec, cap, total, hay, qmask};

  uint32_t npend = 0;  // PRE: descriptors waiting in s_queue (wave-uniform)
  // one descriptor per lane: the flagged fields of a parked register against all 64 bits
  // (whole passes of 64 only -- the newest descriptors; the < 64 oldest wait for company, and for the chunk's end: a pass
  // costs its global-memory round trip whether one lane works in it or all of them)
  auto drain = [&](bool all) {
    wave_order();
    const uint32_t keep = all ? 0u : (npend & 63u);
    for (uint32_t k0 = keep; k0 < npend; k0 += 64u) {
      const uint32_t k = min(k0 + lane, npend - 1u);
      const uint32_t bits = s_queue[2u * k], w1 = s_queue[2u * k + 1u];
      const uint32_t g = w1 & 15u, tile = (w1 >> 4) & 7u, L = (w1 >> 7) & 63u, pp = p0 + 2u * (w1 >> 13);
      const uint32_t rit = (g & 3u) + 8u * (g >> 2) + 4u * (L >> 5);  // C/D layout, see handle_tile2
      const uint32_t row = (tile0 + tile) * 32u + rit;
      // a carry into the exponent (top field flagged) leaves the lower fields unreadable: all four are candidates
      uint32_t fields = ((bits >> 23) & 1u) ? 0xfu
                                            : (((bits >> 5) & 1u) | (((bits >> 11) & 1u) << 1) | (((bits >> 17) & 1u) << 2));
      if (row >= hp.n || k0 + lane >= npend) fields = 0;
      const uint2 hv = s_hay[tile * 32u + rit];
      // every lane takes its lowest flagged field per round (nearly always one round: one field per descriptor); the
      // loop is uniform -- the record buffer ballots
      while (__builtin_amdgcn_ballot_w64(fields != 0) != 0) {
        const uint32_t qi = pp * 64u + (uint32_t)__builtin_ctz(fields | 16u) * 32u + (L & 31u);
        bool has = false;
        uint32_t d = 0, id = 0;
        if (fields != 0 && qi < hp.nq) {
          const uint64_t nv = hp.q[qi];
          d = __popc(hv.x ^ (uint32_t)nv) + __popc(hv.y ^ (uint32_t)(nv >> 32));
          if (nv != 0 && d < hp.thresh && mask_ok(hp, row, qi, nv)) {
            id = hp.ids[row];
            has = id != 0 || hp.keep0;
          }
        }
        out_push(s_out, nout, has, qi, d, id, hp);
        fields &= fields - 1u;
      }
    }
    wave_order();
    npend = keep;
  };

  // one needle-tile pair against the HT resident haystack tiles
  auto step = [&](const uint32_t p, const uint4& nA, const uint4& nB) __attribute__((always_inline)) {
    const v8i bA = fp4_operand(nA);
    const v8i bB = fp4_operand(nB);
    // G tiles at a time: 2*G MFMAs in flight, G*16 accumulator registers live
#pragma unroll
    for (int t0 = 0; t0 < HT; t0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], bA, c0, 4, 4, 0, kScaleOne,
                                                               0, scale_b);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], bB, c[t], 4, 4, 0,
                                                               kScaleOne, 0, scale_b2);
      if constexpr (PRE) {
        // flag bits survive OR: v_or3_b32 takes two more registers per op (plain VGPR-only ops, cheaper to issue than
        // the packed max); two chains of 8 ops
        constexpr int R = G * 16;
        // (two chains of 17 and 15 registers: 8 + 7 v_or3_b32; 16 + 16 would take 8 + 8)
        constexpr int RA = R / 2 + 1;
        const uint32_t half0 = or_regs<0, RA, G>(c), half1 = or_regs<RA, R, G>(c);
        const uint32_t flags = (half0 | half1) & kFlagMaskPre;
        const uint64_t hm = __builtin_amdgcn_ballot_w64(flags != 0);
        if (hm != 0) {
          // wave-uniform from here: some lane holds a candidate (one group in ~10 at threshold 5, every other at 6)
          static_assert(R == 32, "the parking slots below are 32 registers");
          auto park4 = [&](uint32_t at, int t, int k) {  // registers 16 t + 4 k .. + 3 of the lane: one ds_write_b128
            *reinterpret_cast<float4*>(&s_queue[at + (uint32_t)(t * 16 + 4 * k)]) =
                make_float4(c[t][4 * k], c[t][4 * k + 1], c[t][4 * k + 2], c[t][4 * k + 3]);
          };
          auto park = [&](uint32_t at) {
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
              for (int k = 0; k < 4; ++k) park4(at, t, k);
          };
          const uint32_t w1c = ((uint32_t)t0 << 4) | (((p - p0) >> 1) << 13);  // tile of register 0, step
          // lane r < 32 reads register r of a hit lane back (the lanes above read on into the next slot, or the first words
          // behind the wave's queue: inside the workgroup's LDS, and discarded by the mask / lane < 32); `ok` = the registers
          // whose parking slot was written.  Appends {pattern, register | tile | hit lane | step} for every flagged one.
          auto list = [&](uint32_t at, uint32_t L, uint32_t ok) {
            // (the lane id through an opaque copy: the compiler otherwise hoists this path's lane-derived values -- the
            // parking area's address, 1 << lane -- out of the chunk loop, runs out of its 128 registers and SPILLS them:
            // two scratch loads + a wait in front of every list(), i.e. in every other group at threshold 6)
            uint32_t ln = lane;
            asm volatile("" : "+v"(ln));
            const uint32_t v = s_queue[at + ln];
            const bool pred = (v & kFlagMaskPre) != 0 && ((ok >> (ln & 31u)) & 1u) != 0u;
            const uint32_t bm = (uint32_t)(__builtin_amdgcn_ballot_w64(pred) & 0xffffffffull);
            if (pred && ln < (uint32_t)R) {
              // lane = 16 t + g: bits 0-3 the register, bits 4-6 (t0 + t) the wave's tile; bits 7-12 the hit lane
              *reinterpret_cast<uint2*>(&s_queue[2u * (npend + __builtin_amdgcn_mbcnt_lo(bm, 0u))]) =
                  make_uint2(v, ln | (w1c | (L << 7)));
            }
            npend += (uint32_t)__popc(bm);
          };
          if ((hm & (hm - 1)) == 0) {
            // ONE hit lane (nine events in ten), a fixed address -- and nearly always one flagged register: only the
            // registers of the reduction chain(s) that hold a flag are parked -- 0..16 (five ds_write_b128) or 17..31
            // (four) instead of all eight; a ds_write_b128 moves 1 KB whatever its exec mask.  (Masks, not bools: a
            // uniform bool comes back as v_cndmask + v_cmp.)
            const uint64_t b0 = __builtin_amdgcn_ballot_w64((half0 & kFlagMaskPre) != 0);
            const uint64_t b1 = __builtin_amdgcn_ballot_w64((half1 & kFlagMaskPre) != 0);
            if (flags != 0) {
              if (b0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) park4(kPark, 0, k);
              }
              park4(kPark, 1, 0);
              if (b1) {
#pragma unroll
                for (int k = 1; k < 4; ++k) park4(kPark, 1, k);
              }
            }
            const uint32_t ok = (b0 ? 0x000fffffu : 0u) | (b1 ? 0xffff0000u : 0u);
            wave_order();
            list(kPark, (uint32_t)__builtin_ctzll(hm), ok);
            wave_order();
            // (drained at the end of the step: a one-lane event adds at most 32 descriptors, four groups cannot overflow the
            //  list -- and the drain's code sits once per step instead of once per group: 4.3k instead of 5.5k instructions,
            //  threshold 6 1.5 % faster, profiles/r06_pre_drain_sites_ab.json)
          } else {
            // several hit lanes, in chunks of kParkLanes (one chunk unless the group is dense: duplicates, video frames);
            // the k-th hit lane of a chunk parks all its registers at kPark + 32 k
            uint64_t rest = hm;
            do {
              uint64_t cm = rest;
              if ((uint32_t)__popcll(rest) > kParkLanes) {
                uint64_t rem = rest;
                for (uint32_t i = 0; i < kParkLanes; ++i) rem &= rem - 1;
                cm = rest ^ rem;  // the lowest kParkLanes hit lanes
              }
              rest ^= cm;
              if ((cm >> lane) & 1ull)
                park(kPark + 32u * __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u)));
              wave_order();
              uint32_t at = kPark;
              for (uint64_t m = cm; m; m &= m - 1, at += 32u) list(at, (uint32_t)__builtin_ctzll(m), 0xffffffffu);
              wave_order();
              if (npend >= 64u) drain(false);
            } while (rest);
          }
        }
      } else {
        // packed per-half maximum of the group's G*16 results: 8 v_pk_maximum3_f16 per tile
        h2 m0 = {0, 0}, m1 = {0, 0};
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int g = 0; g < 16; g += 4) {
            m0 = pkmax3(m0, as_h2(c[t][g]), as_h2(c[t][g + 1]));
            m1 = pkmax3(m1, as_h2(c[t][g + 2]), as_h2(c[t][g + 3]));
          }
        const uint32_t mb = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(m0, m1));
        const bool hit = (mb << 16) >= lo_key || mb >= hi_key;
        if (__builtin_amdgcn_ballot_w64(hit) != 0) {
          // wave-uniform from here: something in this group is under the threshold (rare)
#pragma unroll
          for (int t = 0; t < G; ++t) handle_tile2(c[t], (tile0 + t0 + t) * 32u, p, hp, s_queue, s_out, nout);
        }
      }
    }
    if constexpr (PRE)
      if (npend >= 64u) drain(false);
  };

  if constexpr (PRE) {
    // A step takes two needle pairs (four tiles); the chunk length is even, only the call's last pair can be single -- its
    // partner slot is fed the same tiles again (chosen by ADDRESS, so that both loads are issued back to back and stay in
    // flight during the MFMAs) and its candidates fall out at qi >= nq.
    // Two steps per trip with explicit double buffers (round 5): the single call site of rounds 1-4 cost eight VALU
    // register moves per step (next -> current, and the default of a prefetch that may not happen) + two 64-bit vector
    // address updates -- a tenth of the loop's VALU instructions, in a kernel bound by VALU issue.
    // (raw buffer loads: descriptor base = pair p0's tiles, scalar byte offset of the pair, constant per-lane offset)
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(qf + (size_t)p0 * 64u), 0, (int)0xffffffffu, 0x27000);
    const uint32_t voff = lane * 16u;
    auto ldt = [&](uint32_t rel, uint32_t partner) -> uint4 {  // tiles of pair p0 + rel (+ 1: the partner, if it exists)
      const uint32_t pr = min(rel + partner, n_pairs - 1u - p0);  // (s_min_u32: a select here became vector code)
      const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(pr * 1024u), 0);
      return make_uint4(v.x, v.y, v.z, v.w);
    };
    const uint32_t np = p1 - p0;
    uint4 a0 = ldt(0, 0), a1 = ldt(0, 1);
#pragma unroll 1
    for (uint32_t rel = 0; rel < np; rel += 4) {  // (two call sites of step(): its cold paths are large)
      const uint4 b0 = ldt(min(rel + 2, np - 1), 0), b1 = ldt(min(rel + 2, np - 1), 1);
      step(p0 + rel, a0, a1);
      if (rel + 2 < np) {
        a0 = ldt(min(rel + 4, np - 1), 0);
        a1 = ldt(min(rel + 4, np - 1), 1);
        step(p0 + rel + 2, b0, b1);
      }
    }
    if (npend) drain(true);
  } else {
    // two pairs per trip with explicit double buffers: the loads of the next pair are in flight
    // while the 2*HT MFMAs of the current one run
    uint4 x0 = qp[0], x1 = qp[64];
    uint32_t p = p0;
    for (; p + 1 < p1; p += 2) {
      const uint4 y0 = qp[128], y1 = qp[192];
      step(p, x0, x1);
      qp += 256;
      if (p + 2 < p1) {
        x0 = qp[0];
        x1 = qp[64];
      }
      step(p + 1, y0, y1);
    }
    if (p < p1) step(p, x0, x1);
  }
  out_flush(s_out, nout, hp);
}


// ---- FULL3: three needle tiles per accumulator, detection by OR instead of maximum ------------------------
// The VALU reduction shares the issue port with the MFMAs (measured: an FP4 32x32x64 MFMA blocks it ~24 of
// its ~40 cycles, every VALU op adds 4), so fewer reduction ops per MFMA is the lever.  With w = 64 - dist
// and b = thresh - 1 the accumulator is built as
//   2^23 + 2 * sum_{i<3} 2^(7i) * (w_i + b)        (B scales 2^0, 2^7, 2^14; C0 carries 2^23 + the b terms)
// i.e. three 7-bit fields (w + b <= 127 for thresh <= 64), and  dist_i < thresh  <=>  w_i + b >= 64  <=>
// bit 6 of field i = bit 7 + 7i of the f32 pattern.  OR-ing registers keeps "some flag bit is set", so
// v_or3_b32 reduces two registers per op for THREE MFMAs' worth of results: 2.7 VALU ops per MFMA instead
// of 4, which makes the kernel matrix-core bound.
constexpr uint32_t kFlagMask3 = (1u << 7) | (1u << 14) | (1u << 21);
constexpr int kScale7 = (int)0x86868686;   // E8M0 134 = 2^7
constexpr int kScale14 = (int)0x8d8d8d8d;  // E8M0 141 = 2^14

__device__ __forceinline__ void handle_tile3(const v16f& c, uint32_t row0, uint32_t p3, const HitParams& hp,
                                             uint32_t* s_queue, uint32_t* s_out, uint32_t& nout) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t any = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) any |= as_u32(c[g]);
  if (__builtin_amdgcn_ballot_w64((any & kFlagMask3) != 0) == 0) return;  // the other tile of the group
  const uint32_t b = hp.thresh - 1u;
  // one pass per field: the queue holds 16 registers x 64 lanes (4 KB per wave, so that LDS never limits the waves per
  // SIMD); the records of a tile come out field by field -- their order in the block is free
#pragma unroll 1
  for (uint32_t f = 0; f < 3; ++f) {
    uint32_t cnt = 0;  // wave-uniform
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const uint32_t bits = as_u32(c[g]);
      const bool fl = (bits >> (7u + 7u * f)) & 1u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(fl);
      if (m == 0) continue;
      if (fl)  // entry: dist<<12 | field<<10 | g<<6 | lane
        s_queue[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
            ((64u + b - ((bits >> (1u + 7u * f)) & 0x7fu)) << 12) | (f << 10) | ((uint32_t)g << 6) | lane;
      cnt += (uint32_t)__popcll(m);
    }
    wave_order();
    for (uint32_t k0 = 0; k0 < cnt; k0 += 64u) {
      const uint32_t k = k0 + lane;
      const uint32_t e = s_queue[min(k, cnt - 1u)];
      const uint32_t src = e & 63u, g = (e >> 6) & 15u, field = (e >> 10) & 3u, d = e >> 12;
      const uint32_t row = row0 + (g & 3u) + 8u * (g >> 2) + 4u * (src >> 5);
      const uint32_t qi = p3 * 96u + field * 32u + (src & 31u);
      const uint64_t nv = (k < cnt && row < hp.n && qi < hp.nq) ? hp.q[qi] : 0;
      bool has = false;
      uint32_t id = 0;
      if (nv != 0 && mask_ok(hp, row, qi, nv)) {
        id = hp.ids[row];
        has = id != 0 || hp.keep0;
      }
      out_push(s_out, nout, has, qi, d, id, hp);
    }
    wave_order();
  }
}

// (2 workgroups per CU as the minimum: with at most 256 registers per lane the compiler keeps the accumulators in
//  VGPRs -- given 512 it puts them in AGPRs and adds a v_accvgpr_read_b32 for every register the OR reduction touches,
//  71 instead of 39 VALU instructions per six MFMAs; compiled for 4 workgroups per CU it spills: 35 ms.)
__global__ __launch_bounds__(kThreads, 2) void k_hamm64_mfma3(
    const uint2* __restrict__ hay, const uint32_t* __restrict__ ids, uint32_t n,
    const uint64_t* __restrict__ q, const uint4* __restrict__ qx, uint32_t nq, uint32_t n_triples,
    uint32_t triples_per_chunk, uint32_t thresh, cbh_record* __restrict__ rec,
    unsigned long long cap, unsigned long long* __restrict__ total, uint32_t keep0,
    const uint2* __restrict__ qmask) {
  constexpr int HT = kHT, G = kG;
  __shared__ uint32_t s_queue_[kWaves][16 * 64];
  __shared__ uint32_t s_out_[kWaves][2 * kOutCap];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t r = lane & 31u, half = lane >> 5;
  const uint32_t tile0 = (blockIdx.x * kWaves + wave) * HT;
  if (tile0 * 32u >= n) return;
  uint32_t* s_queue = s_queue_[wave];
  uint32_t* s_out = s_out_[wave];
  uint32_t nout = 0;

  v8i a[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const uint32_t row = (tile0 + t) * 32u + r;
    const uint2 hv = row < n ? hay[row] : make_uint2(0u, 0u);
    a[t] = fp4_operand(fp4_expand32(half ? hv.y : hv.x));
  }
  // C0 = 2^23 + (64 + 2b) * (1 + 2^7 + 2^14): every field starts at 2 * (32 + b) and gains dot_i = 2 * (w_i - 32)
  const uint32_t b = thresh - 1u;
  v16f c0;
#pragma unroll
  for (int g = 0; g < 16; ++g) c0[g] = 8388608.0f + (float)((64u + 2u * b) * 16513u);
  asm volatile("" : "+v"(c0));
  const HitParams hp = {0, 0, 0, 0, thresh, n, nq, keep0, q, ids, rec, cap, total, hay, qmask};

  const uint32_t p0 = blockIdx.y * triples_per_chunk;
  const uint32_t p1 = min(n_triples, p0 + triples_per_chunk);
  // triple p = needles [96p, 96p+96): three tiles of 32; lane (c, half) reads word `half` of needle c (below)

  auto step = [&](const uint32_t p, const uint4& n0, const uint4& n1, const uint4& n2) __attribute__((always_inline)) {
    const v8i b0 = fp4_operand(n0), b1 = fp4_operand(n1), b2 = fp4_operand(n2);
#pragma unroll
    for (int t0 = 0; t0 < HT; t0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], b0, c0, 4, 4, 0, kScaleOne, 0,
                                                               kScaleOne);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], b1, c[t], 4, 4, 0, kScaleOne, 0,
                                                               kScale7);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], b2, c[t], 4, 4, 0, kScaleOne, 0,
                                                               kScale14);
      // one v_or3_b32 per two result registers, the last register and the mask in one v_bitop3_b32
      if (__builtin_amdgcn_ballot_w64((or_regs<0, G * 16, G>(c) & kFlagMask3) != 0) != 0) {
#pragma unroll
        for (int t = 0; t < G; ++t) handle_tile3(c[t], (tile0 + t0 + t) * 32u, p, hp, s_queue, s_out, nout);
      }
    }
  };

  // Two triples (12 * HT MFMAs) per trip with explicit double buffers, the next triple's three tile loads in flight
  // meanwhile (round 5; as in the prefilter kernel: the single call site cost 12 register moves and three vector address
  // updates per triple, and this kernel too is bound by what the VALU issues beside the MFMAs -- T = 24 M + 4 V cycles).
  // Raw buffer loads: descriptor base = triple p0, scalar byte offset of the tile, constant per-lane offset.
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(qx + (size_t)p0 * 192u), 0, (int)0xffffffffu, 0x27000);
  const uint32_t voff = (2u * r + half) * 16u;
  auto ldt = [&](uint32_t rel, uint32_t tile) -> uint4 {  // tile 0..2 of triple p0 + rel
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)(rel * 3072u + tile * 1024u), 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  const uint32_t np = p1 - p0;
  uint4 x0 = ldt(0, 0), x1 = ldt(0, 1), x2 = ldt(0, 2);
#pragma unroll 1
  for (uint32_t rel = 0; rel < np; rel += 2) {  // (two call sites of step())
    const uint32_t ry = min(rel + 1, np - 1);
    const uint4 y0 = ldt(ry, 0), y1 = ldt(ry, 1), y2 = ldt(ry, 2);
    step(p0 + rel, x0, x1, x2);
    if (rel + 1 < np) {
      const uint32_t rx = min(rel + 2, np - 1);
      x0 = ldt(rx, 0);
      x1 = ldt(rx, 1);
      x2 = ldt(rx, 2);
      step(p0 + rel + 1, y0, y1, y2);
    }
  }
  out_flush(s_out, nout, hp);
}

// ---- which kernel: the candidate rate of THIS launch's data ---------------------------------------------------------
// The prefilter kernel is twice as fast as the three-field kernel while its candidates are rare and loses to it when
// they are not: every candidate costs a park / list / re-check (~110 SIMD cycles per event).  How many there are is a
// property of the data -- r_cand(t) = P[popc(fold(a) ^ fold(b)) < t] over the launch's needle x slot pairs: 5.7e-5 at
// t = 6 and 2.7e-4 at t = 7 for unrelated hashes, but anything for a library of scans of one form, blank frames or a video
// against itself.  The three-field kernel in turn pays for every TRUE match (a flagged group goes through three passes
// of sixteen ballots): ~4x what a candidate costs the prefilter, so where the candidates are mostly true matches -- a
// dense cluster of near-identical hashes -- the prefilter wins again, at any rate.  Measured per 10^12 pairs
// (tools/ab/adaptive_ab.py, profiles/r06_adaptive_ab*.jsonl):  T_pre = 8.8 ms + 6e4 ms x r_cand,  T_full = 16.3 ms +
// 2.5e5 ms x r_true.  k_fold_probe counts both rates on kProbeS x kProbeS evenly spaced (slot, needle) samples -- a few
// microseconds and one host round trip, against launches of milliseconds -- and pick_pre takes the prefilter while
//   r_cand - kTrueWeight x r_true <= kPreRateMax          (the rate at which the two kernels tie: 1.25e-4).
// Launches too small to pay for the round trip, and a probe that cannot allocate, take the fixed rule of round 5
// (thresholds <= 6).
constexpr uint32_t kProbeS = 2048;     // samples per side
constexpr int kProbeT = 8;             // thresholds 1..8 are counted (the prefilter never pays beyond: 1e-3 per pair at 8)
constexpr int kPreStatic = 6;          // the fixed rule
int g_pre_max_thresh = -1;             // "scan_mfma_pre_max": -1 = by candidate rate (default), 0 = never the prefilter,
                                       // t > 0 = thresholds <= t take it whatever the data (tests, A/B)
int g_pre_rate_max_e9 = 125000;        // "scan_pre_rate_e9": kPreRateMax x 1e9
constexpr double kTrueWeight = 4.0;
constexpr uint64_t kProbeMinPairs = 1ull << 31;  // ~20 us of scan: below this the probe's round trip is not worth it

// grid (sq / 256, sh / 64): thread = one needle sample against 64 slot samples; counts[t - 1] += pairs with fold
// distance < t, counts[kProbeT + t - 1] += pairs with 64-bit distance < t.  The samples are pseudo-random rows / needles
// (a 32-bit mix of the sample number): evenly spaced ones meet the diagonal of a self-join far more often than its share
// -- a shard of 125 000 slots against its index's 10^6 needles counted 256 self matches among 4.2 x 10^6 sampled pairs,
// sixty times their true rate, which is how dht 7 first came to take the prefilter on a sharded handle.
__device__ __forceinline__ uint32_t probe_mix(uint32_t x) {
  x = ((x >> 16) ^ x) * 0x45d9f3bu;
  x = ((x >> 16) ^ x) * 0x45d9f3bu;
  return (x >> 16) ^ x;
}
__global__ __launch_bounds__(256) void k_fold_probe(const uint2* __restrict__ hay, uint32_t n, const uint2* __restrict__ q,
                                                    uint32_t nq, uint32_t sh, uint32_t sq, uint32_t* __restrict__ counts) {
  __shared__ uint2 s_h[64];
  __shared__ uint32_t s_cnt[2 * kProbeT];
  const uint32_t t = threadIdx.x;
  if (t < 64) {
    const uint32_t i = blockIdx.y * 64u + t;
    s_h[t] = i < sh ? hay[sh == n ? i : probe_mix(i) % n] : make_uint2(0u, 0u);
  }
  if (t < 2 * kProbeT) s_cnt[t] = 0;
  __syncthreads();
  const uint32_t j = blockIdx.x * 256u + t;
  const uint32_t nslots = min(64u, sh - blockIdx.y * 64u);
  uint32_t cnt[kProbeT] = {}, cnt64[kProbeT] = {};
  if (j < sq) {
    const uint2 nv = q[sq == nq ? j : probe_mix(j ^ 0x9e3779b9u) % nq];
    const uint32_t f = nv.x ^ nv.y;
    for (uint32_t k = 0; k < nslots; ++k) {
      const uint2 hv = s_h[k];
      const uint32_t d = (uint32_t)__popc(f ^ hv.x ^ hv.y);
#pragma unroll
      for (int th = 0; th < kProbeT; ++th) cnt[th] += d < (uint32_t)(th + 1) ? 1u : 0u;
      if (d < (uint32_t)kProbeT) {  // (rare: a true match is a candidate first)
        const uint32_t d64 = (uint32_t)__popc(nv.x ^ hv.x) + (uint32_t)__popc(nv.y ^ hv.y);
#pragma unroll
        for (int th = 0; th < kProbeT; ++th) cnt64[th] += d64 < (uint32_t)(th + 1) ? 1u : 0u;
      }
    }
  }
#pragma unroll
  for (int th = 0; th < 2 * kProbeT; ++th) {
    uint32_t v = th < kProbeT ? cnt[th % kProbeT] : cnt64[th % kProbeT];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((t & 63u) == 0 && v) atomicAdd(&s_cnt[th], v);
  }
  __syncthreads();
  if (t < 2 * kProbeT && s_cnt[t]) atomicAdd(&counts[t], s_cnt[t]);
}

// pinned words for the probe's answer: a free list (a slot is in use only inside one synchronous probe)
std::mutex g_probe_mu;
std::vector<uint32_t*> g_probe_free;
uint32_t* probe_slot_get() {
  {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    if (!g_probe_free.empty()) {
      uint32_t* p = g_probe_free.back();
      g_probe_free.pop_back();
      return p;
    }
  }
  uint32_t* p = nullptr;
  if (hipHostMalloc(&p, 2 * kProbeT * sizeof(uint32_t)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void probe_slot_put(uint32_t* p) {
  std::lock_guard<std::mutex> lk(g_probe_mu);
  g_probe_free.push_back(p);
}

std::atomic<uint64_t> g_pre_mask{0};          // bit t: the most recent matrix-core launch at threshold t took the prefilter
std::atomic<uint64_t> g_n_probe{0};           // probes run
std::atomic<long long> g_last_rate_e9{-1};    // candidate rate x 1e9 the last probe found for its threshold
std::atomic<long long> g_last_true_e9{-1};    // ... and the rate of true (64-bit) matches

// true = this launch takes the prefilter kernel.  n_total = the slots the call scans in all (a sharded handle probes one
// shard's slots on behalf of all of them)
bool pick_pre(const uint64_t* d_hashes, size_t n, size_t n_total, const uint64_t* d_q, size_t nq, int thresh,
              hipStream_t stream) {
  if (thresh > 32) return false;
  if (g_pre_max_thresh >= 0) return thresh <= g_pre_max_thresh;
  if (thresh > kProbeT) return false;
  if ((uint64_t)n_total * (uint64_t)nq < kProbeMinPairs) return thresh <= kPreStatic;
  uint32_t* d_cnt = nullptr;
  if (cbh::malloc_async((void**)&d_cnt, 2 * kProbeT * sizeof(uint32_t), stream) != hipSuccess) {
    (void)hipGetLastError();
    return thresh <= kPreStatic;
  }
  uint32_t* h_cnt = probe_slot_get();
  bool ok = h_cnt != nullptr;
  const uint32_t sh = (uint32_t)std::min<size_t>(n, kProbeS), sq = (uint32_t)std::min<size_t>(nq, kProbeS);
  if (ok) {
    ok = hipMemsetAsync(d_cnt, 0, 2 * kProbeT * sizeof(uint32_t), stream) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(k_fold_probe, dim3((sq + 255u) / 256u, (sh + 63u) / 64u), dim3(256), 0, stream,
                         reinterpret_cast<const uint2*>(d_hashes), (uint32_t)n, reinterpret_cast<const uint2*>(d_q),
                         (uint32_t)nq, sh, sq, d_cnt);
      ok = hipGetLastError() == hipSuccess &&
           hipMemcpyAsync(h_cnt, d_cnt, 2 * kProbeT * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
           hipStreamSynchronize(stream) == hipSuccess;
    }
  }
  (void)cbh::free_async(d_cnt, stream);
  bool pre = thresh <= kPreStatic;
  if (ok) {
    const double pairs = (double)sh * (double)sq;
    const double r_cand = (double)h_cnt[thresh - 1] / pairs, r_true = (double)h_cnt[kProbeT + thresh - 1] / pairs;
    g_last_rate_e9 = (long long)(r_cand * 1e9);
    g_last_true_e9 = (long long)(r_true * 1e9);
    g_n_probe++;
    pre = (r_cand - kTrueWeight * r_true) * 1e9 <= (double)g_pre_rate_max_e9;
  } else {
    (void)hipGetLastError();
  }
  if (h_cnt) probe_slot_put(h_cnt);
  return pre;
}

int g_scan_mfma = 1;           // use the matrix-core scan when the batch is large enough
uint32_t g_mfma_min_nq = 256;  // below this the needle expansion + tile padding is not worth it

}  // namespace

void set_scan_mfma(int on) {
  if (on >= 0) g_scan_mfma = on;
}
int get_scan_mfma() { return g_scan_mfma; }
void set_scan_pre_max(int t) {
  if (t >= -1 && t <= 32) g_pre_max_thresh = t;
}
void set_scan_pre_rate(int e9) {
  if (e9 >= 0) g_pre_rate_max_e9 = e9;
}
// read-backs (cbh_get_tuning): "scan_pre_mask", "scan_probes", "scan_probe_rate_e9"
long long get_scan_pre_mask() { return (long long)g_pre_mask.load(); }
long long get_scan_probes() { return (long long)g_n_probe.load(); }
long long get_scan_probe_rate_e9() { return g_last_rate_e9.load(); }
long long get_scan_probe_true_e9() { return g_last_true_e9.load(); }

// the choice for a call that scans n_total slots in several launches (sharded.hip), made once on one shard's slots:
// SCAN_PRE_GIVEN | SCAN_PRE_VALUE bits for launch_hamm64_scan's flags
unsigned scan_pre_flags(const uint64_t* d_hashes, size_t n, size_t n_total, const uint64_t* d_q, size_t nq, int thresh,
                        hipStream_t stream) {
  return SCAN_PRE_GIVEN | (pick_pre(d_hashes, n, n_total, d_q, nq, thresh, stream) ? SCAN_PRE_VALUE : 0u);
}

bool scan_mfma_wanted(size_t n, size_t nq, int thresh) {
  if (g_scan_mfma == 2 || g_scan_mfma == 4) return thresh >= 1 && thresh <= 65;  // forced (tests); 3 sizes like 1
  return g_scan_mfma && nq >= g_mfma_min_nq && n >= 4096 && thresh >= 1 && thresh <= 65;
}

static uint32_t padded_needles(size_t nq) { return (uint32_t)((nq + 191) / 192) * 192u; }  // whole pairs (64) and triples (96)

int expand_needles_for_scan(const uint64_t* d_q, size_t nq, hipStream_t stream, void** qx_out) {
  *qx_out = nullptr;
  if (nq == 0 || nq > CBH_MAX_QUERIES_PER_CALL) return CBH_OK;
  const uint32_t nq_pad = padded_needles(nq);
  uint4* qx = nullptr;
  CBH_HIP(malloc_async((void**)&qx, (size_t)nq_pad * 48u, stream));  // 2 words + the prefilter word, 16 B each
  hipLaunchKernelGGL(k_expand_needles, dim3((3u * nq_pad + 255u) / 256u), dim3(256), 0, stream, d_q,
                     (uint32_t)nq, nq_pad, qx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    (void)free_async(qx, stream);
    CBH_HIP(e);
  }
  *qx_out = qx;
  return CBH_OK;
}

int launch_hamm64_scan_mfma(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n,
                            const uint64_t* d_q, size_t nq, int thresh, cbh_record* d_rec,
                            size_t cap, unsigned long long* d_total, hipStream_t stream,
                            unsigned flags, const uint64_t* d_qmask, const void* qx_given) {
  if (n == 0 || nq == 0 || thresh <= 0) return CBH_OK;
  if (n > 0xfffffff0ull || nq > CBH_MAX_QUERIES_PER_CALL || thresh > 65) return CBH_E_INVAL;
  const bool pre = (flags & SCAN_PRE_GIVEN) ? (flags & SCAN_PRE_VALUE) != 0 : pick_pre(d_hashes, n, n, d_q, nq, thresh, stream);
  if (thresh < 64) {
    if (pre) g_pre_mask |= 1ull << thresh; else g_pre_mask &= ~(1ull << thresh);
  }
  const uint32_t n_pairs = (uint32_t)((nq + 63) / 64);
  const uint32_t n_triples = (uint32_t)((nq + 95) / 96);
  const uint32_t nq_pad = padded_needles(nq);
  const uint4* qx = reinterpret_cast<const uint4*>(qx_given);
  void* qx_own = nullptr;
  if (!qx) {
    int rc = expand_needles_for_scan(d_q, nq, stream, &qx_own);
    if (rc) return rc;
    qx = reinterpret_cast<const uint4*>(qx_own);
  }
  const uint4* qf = qx + 2u * (size_t)nq_pad;
  const uint32_t rows_per_wg = 32u * kHT * kWaves;
  // launches that run side by side on this device (the shards of a sharded handle): the workgroups that fill the machine are
  // theirs together -- a shard of 125 000 slots alone cut its needles into chunks of 128 pairs to reach 8192 workgroups and
  // paid the shorter chunks' per-chunk costs (3 % of the sweep) for parallelism its seven siblings already supplied
  const uint32_t sib = std::max(1u, (flags >> SCAN_SIBLINGS_SHIFT) & 0xffu);
  const uint32_t wgs = (uint32_t)((n + rows_per_wg - 1) / rows_per_wg);
  if (!pre && thresh <= 64) {
    // needle chunk: >= 8192 workgroups in flight when there is that much work, but each wave amortises its tile expansion
    // over >= 11 needle-tile triples (172 triples = 16512 needles per chunk: 16.05 ms against 16.25-16.3 with 2-4x that,
    // tools/ab/scan_chunk_ab.py)
    uint32_t tpc = 172u;
    while (tpc > 11 && (uint64_t)wgs * sib * ((n_triples + tpc - 1) / tpc) < 8192) tpc = (tpc + 1) / 2;
    uint32_t ch3 = (n_triples + tpc - 1) / tpc;
    if (ch3 > 65535) {
      tpc = (n_triples + 65534) / 65535;
      ch3 = (n_triples + tpc - 1) / tpc;
    }
    hipLaunchKernelGGL(k_hamm64_mfma3, dim3(wgs, ch3), dim3(kThreads), 0, stream,
                       reinterpret_cast<const uint2*>(d_hashes), d_ids, (uint32_t)n, d_q, qx, (uint32_t)nq, n_triples, tpc,
                       (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total, (uint32_t)(flags & 1u),
                       reinterpret_cast<const uint2*>(d_qmask));
  } else {
    // (prefilter: 512 pairs = 32768 needles per chunk -- a wave drains its pending candidates at the end of its chunk,
    // mostly a short list: at threshold 6 chunks of 512 / 1024 pairs run 12.63 ms against 12.98 with 256 and 13.35 with 64)
    uint32_t ppc = pre ? 512u : 256u;
    while (ppc > 16 && (uint64_t)wgs * sib * ((n_pairs + ppc - 1) / ppc) < 8192) ppc >>= 1;
    uint32_t chunks = (n_pairs + ppc - 1) / ppc;
    if (chunks > 65535) {
      ppc = ((n_pairs + 65534) / 65535 + 1u) & ~1u;  // even: the prefilter variant steps two pairs at a time
      chunks = (n_pairs + ppc - 1) / ppc;
    }
#define CBH_MFMA(PRE)                                                                                                 \
  hipLaunchKernelGGL((k_hamm64_mfma<PRE>), dim3(wgs, chunks), dim3(kThreads), 0, stream,                              \
                     reinterpret_cast<const uint2*>(d_hashes), d_ids, (uint32_t)n, d_q, qx, (uint32_t)nq, n_pairs, ppc, \
                     (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total, (uint32_t)(flags & 1u),               \
                     reinterpret_cast<const uint2*>(d_qmask), qf)
    if (pre) CBH_MFMA(true); else CBH_MFMA(false);
#undef CBH_MFMA
  }
  hipError_t e = hipGetLastError();
  if (qx_own) (void)cbh::free_async(qx_own, stream);
  CBH_HIP(e);
  return CBH_OK;
}

}  // namespace cbh
