// hamm64_mfma.hip -- K3m: the all-pairs 64-bit Hamming threshold scan on the gfx950 matrix cores.
//
// Same contract as k_hamm64_scan (hamm64_scan.hip): every pair with
//   hamm64(q, hash[i]) < thresh  &&  id[i] != 0  &&  q != 0
// (src/dcthashindex.cpp:196-217, hamm64 = popcountll(a ^ b), src/hamm.h:24-26) is appended as
// a cbh_record.  Only the arithmetic differs.
//
// Why the matrix cores.  PMC shows the VALU scan is bound by integer-VALU issue (one
// v_bcnt_u32_b32 per 32 bits per pair), not by memory: 0.8 GB of HBM traffic per 10^12 pairs.
// The distance is also a dot product of sign vectors,
//   dot(s(a), s(b)) = 64 - 2 * hamm64(a, b),   s(x)_k = +1 if bit k of x is set, else -1,
// and +-1.0 are exact in FP4 (E2M1: 0x2 / 0xA), so ONE v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 =
// one hash) yields the exact distances of 32 haystack rows x 32 needles: 1024 pairs in ~32
// matrix-core cycles, against ~8.3 (prefilter) / 14.3 (full) VALU cycles per 64 pairs.  All sums
// are small integers, so the f32 accumulation is exact and results stay bit-identical.
//
// Keeping the VALU out of the way.  16 f32 results per lane per MFMA would cost 8 v_max3_f32
// (32 cycles) to reduce -- as much as the MFMA itself.  Two tricks halve that:
//   * the second needle tile of a pair is multiplied by the MX block scale 2^15 and accumulated
//     onto the first, on top of C0 = 2^23 + 0x4040 + 64*2^15.  In [2^23, 2^24) one f32 ulp is 1, so
//     the mantissa holds   (0x4040 + dotA) + 2^15 * (64 + dotB)   exactly, i.e. the f32 bit
//     pattern is   hi16 = 0x4B00 + (64 - distB),  lo16 = 0x4080 - 2*distA   (dotB even => bit 15
//     is 0): two distances per register, each half monotone in its distance;
//   * both halves are positive normal f16 bit patterns, so v_pk_maximum3_f16 (new on gfx950) takes
//     the per-half maximum of three registers at once: 4 ops per MFMA instead of 8.
// After the 2*HT MFMAs of a needle-tile pair one compare decides whether any of the
// 64 x (32*HT) x ... distances is under the threshold; only then the accumulators (still in
// registers) are decoded and records are emitted.
//
// Layout.  A workgroup is 4 waves; each wave keeps HT haystack tiles (32 rows each) expanded
// to FP4 in VGPRs (4 VGPRs per tile: lane (r, half) holds word `half` of row r) and streams needle
// tiles -- pre-expanded once per call by k_expand_needles into a 32-byte-per-needle scratch --
// through 16-byte loads that the 4 waves share in L1/L2.
#include "cbh_internal.h"
#include "fp4_sign.h"

namespace cbh {
namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kG = 2;   // tiles per accumulator group
constexpr uint32_t kLoZero = 0x4080u;  // lo16 at distance 0
constexpr uint32_t kHiZero = 0x4B40u;  // hi16 at distance 0
// 2^23 + 0x4040 + 64 * 2^15
constexpr float kC0 = 8388608.0f + 16448.0f + 2097152.0f;
constexpr int kScale15 = 0x8e8e8e8e;   // E8M0 142 = 2^15

// needles -> FP4 scratch: needle j -> 2 x uint4 (low word, high word); j >= nq padded with hash 0
__global__ __launch_bounds__(256) void k_expand_needles(const uint64_t* __restrict__ q, uint32_t nq,
                                                        uint32_t nq_pad, uint4* __restrict__ qx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // one thread per 32-bit word
  if (i >= 2u * nq_pad) return;
  const uint32_t j = i >> 1;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(q);
  qx[i] = fp4_expand32(j < nq ? w[i] : 0u);
}

__device__ __forceinline__ void emit(cbh_record* __restrict__ rec, unsigned long long cap,
                                     unsigned long long* __restrict__ total, uint32_t qidx,
                                     uint32_t dist, uint32_t id) {
  unsigned long long slot = atomicAdd(total, 1ull);
  if (slot < cap) rec[slot] = ((cbh_record)qidx << 39) | ((cbh_record)dist << 32) | id;
}

__device__ __forceinline__ h2 as_h2(float f) { return __builtin_bit_cast(h2, f); }
__device__ __forceinline__ h2 pkmax3(h2 a, h2 b, h2 c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);  // v_pk_maximum3_f16
}

template <int HT, int G>
__global__ __launch_bounds__(kThreads) void k_hamm64_mfma(
    const uint2* __restrict__ hay, const uint32_t* __restrict__ ids, uint32_t n,
    const uint64_t* __restrict__ q, const uint4* __restrict__ qx, uint32_t nq, uint32_t n_pairs,
    uint32_t pairs_per_chunk, uint32_t thresh, cbh_record* __restrict__ rec,
    unsigned long long cap, unsigned long long* __restrict__ total, uint32_t keep0) {
  __shared__ float s_c[kWaves][G * 16][64];  // refine scratch: one accumulator group per wave
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t r = lane & 31u, half = lane >> 5;
  const uint32_t tile0 = (blockIdx.x * kWaves + wave) * HT;
  if (tile0 * 32u >= n) return;  // whole wave past the end (no barriers in this kernel)

  v8i a[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const uint32_t row = (tile0 + t) * 32u + r;
    const uint2 hv = row < n ? hay[row] : make_uint2(0u, 0u);
    const uint4 e = fp4_expand32(half ? hv.y : hv.x);
    a[t] = v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
  }
  v16f c0;
#pragma unroll
  for (int g = 0; g < 16; ++g) c0[g] = kC0;
  asm volatile("" : "+v"(c0));  // keep C0 resident: otherwise it is rebuilt (16 v_mov) every trip

  const uint32_t p0 = blockIdx.y * pairs_per_chunk;
  const uint32_t p1 = min(n_pairs, p0 + pairs_per_chunk);
  // pair p = needles [64p, 64p+64): tile A = first 32, tile B = last 32; 2 uint4 per needle
  const uint4* __restrict__ qp = qx + ((size_t)p0 * 64u + r) * 2u + half;
  const uint32_t lo_thr = kLoZero - 2u * (thresh - 1u);  // lo16 >= lo_thr  <=>  distA < thresh
  const uint32_t hi_thr = kHiZero - (thresh - 1u);       // hi16 >= hi_thr  <=>  distB < thresh

  // one needle-tile pair against the HT resident haystack tiles
  auto step = [&](const uint32_t p, const uint4& nA, const uint4& nB) {
    const v8i bA = v8i{(int)nA.x, (int)nA.y, (int)nA.z, (int)nA.w, 0, 0, 0, 0};
    const v8i bB = v8i{(int)nB.x, (int)nB.y, (int)nB.z, (int)nB.w, 0, 0, 0, 0};
    // G tiles at a time: 2*G MFMAs in flight, G*16 accumulator registers live
#pragma unroll
    for (int t0 = 0; t0 < HT; t0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], bA, c0, 4, 4, 0, kScaleOne,
                                                               0, kScaleOne);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t], bB, c[t], 4, 4, 0,
                                                               kScaleOne, 0, kScale15);
      h2 m0 = {0, 0}, m1 = {0, 0};
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
          m0 = pkmax3(m0, as_h2(c[t][g]), as_h2(c[t][g + 1]));
          m1 = pkmax3(m1, as_h2(c[t][g + 2]), as_h2(c[t][g + 3]));
        }
      const uint32_t mb = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(m0, m1));
      if ((mb & 0xffffu) >= lo_thr || (mb >> 16) >= hi_thr) {
        // rare: park the accumulators in LDS (each lane reads back only its own values, so no
        // barrier) and walk them in a rolled loop -- keeps this cold path out of the hot loop's
        // instruction stream.  C/D layout of the 32x32 MFMA: column = lane & 31 -> needle,
        // row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5) -> haystack row in the tile.
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) s_c[wave][t * 16 + g][lane] = c[t][g];
#pragma unroll 1
        for (uint32_t e = 0; e < (uint32_t)G * 16u; ++e) {
          const uint32_t bits = __builtin_bit_cast(uint32_t, (float)s_c[wave][e][lane]);
          const uint32_t lo = bits & 0xffffu, hi = bits >> 16;
          if (lo >= lo_thr || hi >= hi_thr) {
            const uint32_t g = e & 15u;
            const uint32_t row = (tile0 + t0 + (e >> 4)) * 32u + (g & 3u) + 8u * (g >> 2) + 4u * half;
            if (row < n) {
              const uint32_t id = ids[row];
              if (id != 0 || keep0) {
                const uint32_t qa = p * 64u + r, qb = qa + 32u;
                if (lo >= lo_thr && qa < nq && q[qa] != 0)
                  emit(rec, cap, total, qa, (kLoZero - lo) >> 1, id);
                if (hi >= hi_thr && qb < nq && q[qb] != 0) emit(rec, cap, total, qb, kHiZero - hi, id);
              }
            }
          }
        }
      }
    }
  };

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

int g_scan_mfma = 1;        // use the matrix-core scan when the batch is large enough
int g_mfma_ht = 8;          // haystack tiles per wave (4 or 8)
uint32_t g_mfma_min_nq = 256;  // below this the needle expansion + tile padding is not worth it

}  // namespace

void set_scan_mfma(int on) {
  if (on >= 0) g_scan_mfma = on;
}
void set_scan_mfma_ht(int ht) {
  if (ht == 2 || ht == 4 || ht == 8) g_mfma_ht = ht;
}

bool scan_mfma_wanted(size_t n, size_t nq, int thresh) {
  if (g_scan_mfma == 2) return thresh >= 1 && thresh <= 65;  // forced (tests)
  return g_scan_mfma && nq >= g_mfma_min_nq && n >= 4096 && thresh >= 1 && thresh <= 65;
}

int launch_hamm64_scan_mfma(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n,
                            const uint64_t* d_q, size_t nq, int thresh, cbh_record* d_rec,
                            size_t cap, unsigned long long* d_total, hipStream_t stream,
                            unsigned flags) {
  if (n == 0 || nq == 0 || thresh <= 0) return CBH_OK;
  if (n > 0xfffffff0ull || nq > CBH_MAX_QUERIES_PER_CALL || thresh > 65) return CBH_E_INVAL;
  const uint32_t n_pairs = (uint32_t)((nq + 63) / 64);
  const uint32_t nq_pad = n_pairs * 64u;
  uint4* qx = nullptr;
  CBH_HIP(hipMallocAsync((void**)&qx, (size_t)nq_pad * 32u, stream));
  hipLaunchKernelGGL(k_expand_needles, dim3((2u * nq_pad + 255u) / 256u), dim3(256), 0, stream, d_q,
                     (uint32_t)nq, nq_pad, qx);
  const uint32_t ht = (uint32_t)g_mfma_ht;
  const uint32_t rows_per_wg = 32u * ht * kWaves;
  const uint32_t wgs = (uint32_t)((n + rows_per_wg - 1) / rows_per_wg);
  // needle chunk: >= 8192 workgroups in flight when there is that much work, but each wave
  // amortises its tile expansion over >= 16 needle-tile pairs
  uint32_t ppc = 256;  // 16384 needles
  while (ppc > 16 && (uint64_t)wgs * ((n_pairs + ppc - 1) / ppc) < 8192) ppc >>= 1;
  uint32_t chunks = (n_pairs + ppc - 1) / ppc;
  if (chunks > 65535) {
    ppc = (n_pairs + 65534) / 65535;
    chunks = (n_pairs + ppc - 1) / ppc;
  }
#define CBH_MFMA(HT)                                                                            \
  hipLaunchKernelGGL((k_hamm64_mfma<HT, kG>), dim3(wgs, chunks), dim3(kThreads), 0, stream,     \
                     reinterpret_cast<const uint2*>(d_hashes), d_ids, (uint32_t)n, d_q, qx,     \
                     (uint32_t)nq, n_pairs, ppc, (uint32_t)thresh, d_rec,                       \
                     (unsigned long long)cap, d_total, (uint32_t)(flags & 1u))
  if (ht == 8) CBH_MFMA(8);
  else if (ht == 2) CBH_MFMA(2);
  else CBH_MFMA(4);
#undef CBH_MFMA
  hipError_t e = hipGetLastError();
  (void)hipFreeAsync(qx, stream);
  CBH_HIP(e);
  return CBH_OK;
}

}  // namespace cbh
