// hamm256_mfma.hip -- K6m: 256-bit Hamming threshold scan (CvFeaturesIndex brute force,
// src/cvfeaturesindex.cpp:497-518: the exact search FLANN-LSH approximates) on the matrix cores.
//
// Same records as k_hamm256_scan (idx256.hip): one  q<<41 | dist<<32 | row  per (needle descriptor,
// index row) with popcount(xor of the 32 bytes) < thresh.
//
// A 256-bit distance is four chained FP4 sign-dot-product MFMAs (fp4_sign.h) on one accumulator:
// dot = 256 - 2 * dist, exact in f32.  A wave keeps HT row tiles (32 rows x 256 bits = 16 VGPRs per
// tile) expanded in registers and streams needle tiles (32 descriptors, pre-expanded once per call
// into a tile-major FP4 scratch so that every load is a contiguous 512 B per half-wave).  Per
// (row tile, needle tile): 4 MFMAs, then 8 v_max3_f32 over the 16 results and one compare; the rare
// hit parks the accumulators in LDS and decodes them in a rolled loop.  VALU work is 2 ops per
// MFMA, so the kernel runs at the matrix-core rate: 4 x ~43 cycles per 1024 pairs.
#include <algorithm>

#include "cbh_internal.h"
#include "fp4_sign.h"

namespace cbh {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;

// needle descriptors -> FP4 scratch, tile-major: uint4 index ((tile*4 + chunk)*2 + half)*32 + c
// holds the expansion of 32-bit word (2*chunk + half) of descriptor tile*32 + c
__global__ __launch_bounds__(256) void k_expand_needles256(const uint32_t* __restrict__ q, uint32_t nq,
                                                           uint32_t nq_pad, uint4* __restrict__ qx) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= 8u * nq_pad) return;
  const uint32_t c = i & 31u, word = (i >> 5) & 7u, tile = i >> 8;
  const uint32_t j = tile * 32u + c;
  qx[i] = fp4_expand32(j < nq ? q[(size_t)j * 8u + word] : 0u);
}

// KCH = 4: all 256 bits on the matrix cores.  KCH = 2: the first 128 bits only -- a sound lower bound of the
// distance (like k_hamm256_scan's first-half filter) at half the MFMA work; candidates under the threshold on
// 128 bits get their second halves evaluated from the raw rows.  For thresh <= kPre128MaxThresh the bound
// rejects all but ~10^-5 of the pairs of unrelated descriptors (128 fair bits: mean 64, sigma 5.7).
template <int HT, int G, int KCH>
__global__ __launch_bounds__(kThreads) void k_hamm256_mfma(
    const uint32_t* __restrict__ rows /* 8 words per row */, uint32_t n, const uint4* __restrict__ qx,
    const uint32_t* __restrict__ qraw /* 8 words per needle */, uint32_t nq, uint32_t n_tiles,
    uint32_t tiles_per_chunk, uint32_t thresh, unsigned long long* __restrict__ rec, unsigned long long cap,
    unsigned long long* __restrict__ total) {
  __shared__ float s_c[kWaves][G * 16][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t r = lane & 31u, half = lane >> 5;
  const uint32_t tile0 = (blockIdx.x * kWaves + wave) * HT;
  if (tile0 * 32u >= n) return;

  v8i a[HT][KCH];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const uint32_t row = (tile0 + t) * 32u + r;
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
      const uint32_t w = row < n ? rows[(size_t)row * 8u + 2u * k + half] : 0u;
      a[t][k] = fp4_operand(fp4_expand32(w));
    }
  }
  const uint32_t q0 = blockIdx.y * tiles_per_chunk;
  const uint32_t q1 = min(n_tiles, q0 + tiles_per_chunk);
  const uint4* __restrict__ qp = qx + (size_t)q0 * 256u + half * 32u + r;  // + chunk*64 per K chunk
  // dot over 64*KCH signs >= dot_thr  <=>  distance on those bits < thresh
  const float dot_thr = (float)(64 * KCH) - 2.0f * (float)(thresh - 1u);

  auto step = [&](const uint32_t qt, const uint4 (&nb)[KCH]) {
    v8i b[KCH];
#pragma unroll
    for (int k = 0; k < KCH; ++k) b[k] = fp4_operand(nb[k]);
#pragma unroll
    for (int t0 = 0; t0 < HT; t0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) c[t][g] = 0.0f;
#pragma unroll
      for (int k = 0; k < KCH; ++k)
#pragma unroll
        for (int t = 0; t < G; ++t)
          c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][k], b[k], c[t], 4, 4, 0,
                                                                 kScaleOne, 0, kScaleOne);
      float m0 = -512.0f, m1 = -512.0f;
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
          m0 = __builtin_fmaxf(__builtin_fmaxf(m0, c[t][g]), c[t][g + 1]);      // v_max3_f32
          m1 = __builtin_fmaxf(__builtin_fmaxf(m1, c[t][g + 2]), c[t][g + 3]);
        }
      if (__builtin_fmaxf(m0, m1) >= dot_thr) {
        // rare: decode through LDS in a rolled loop (C/D layout: column = lane & 31 -> needle,
        // row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5) -> index row in the tile)
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) s_c[wave][t * 16 + g][lane] = c[t][g];
#pragma unroll 1
        for (uint32_t e = 0; e < (uint32_t)G * 16u; ++e) {
          const float dot = s_c[wave][e][lane];
          if (dot >= dot_thr) {
            const uint32_t g = e & 15u;
            const uint32_t row = (tile0 + t0 + (e >> 4)) * 32u + (g & 3u) + 8u * (g >> 2) + 4u * half;
            const uint32_t qi = qt * 32u + r;
            if (row < n && qi < nq) {
              uint32_t d = (uint32_t)(64 * KCH - (int)dot) >> 1;
              if (KCH < 4) {  // the remaining 64-bit chunks from the raw data
#pragma unroll
                for (int wd = 2 * KCH; wd < 8; ++wd) d += __popc(rows[(size_t)row * 8u + wd] ^ qraw[(size_t)qi * 8u + wd]);
              }
              if (d < thresh) {
                const unsigned long long slot = atomicAdd(total, 1ull);
                if (slot < cap)
                  rec[slot] = ((unsigned long long)qi << 41) | ((unsigned long long)d << 32) | row;
              }
            }
          }
        }
      }
    }
  };

  // two needle tiles per trip, explicit double buffers
  uint4 x[KCH], y[KCH];
#pragma unroll
  for (int k = 0; k < KCH; ++k) x[k] = qp[k * 64];
  uint32_t qt = q0;
  for (; qt + 1 < q1; qt += 2) {
#pragma unroll
    for (int k = 0; k < KCH; ++k) y[k] = qp[256 + k * 64];
    step(qt, x);
    qp += 512;
    if (qt + 2 < q1) {
#pragma unroll
      for (int k = 0; k < KCH; ++k) x[k] = qp[k * 64];
    }
    step(qt + 1, y);
  }
  if (qt < q1) step(qt, x);
}

// ---- THREE needle tiles per accumulator (first-128-bit prefilter, thresh <= kPre128MaxThresh) --------------------------
// k_hamm256_mfma<.., 2> spends 8 result registers per MFMA on one needle tile: 4 v_max3_f32 per MFMA, and on this
// machine every VALU op costs the issue port 4 of the ~32 cycles an FP4 MFMA has (hamm64_mfma.hip, FULL3): the matrix
// pipe idles a quarter of the time.  The fields of hamm64_mfma's FULL3 / PRE carry over: with b = thresh - 1 and the B
// block scales 2^-1, 2^7, 2^15 for the three tiles of a triple (both 64-bit chunks of a tile under the same scale, C0 =
// 2^23 + (64 + b)(1 + 2^8 + 2^16)) the accumulator ends as
//     2^23 + sum_i 2^(8i) * (128 + b - d_i),      d_i = distance on the first 128 bits to tile i's descriptor,
// three 8-bit fields in [b, 128 + b] (no borrow, no carry for b <= 127), and  d_i <= b  <=>  bit 7 of field i = bits 7,
// 15 of the f32 pattern and, for the top field, the carry out of the mantissa = bit 23 (exponent 150 -> 151; the ulp is
// then 2: the lowest bit of the sum is rounded away, which can only turn an odd field 0 into a neighbouring even one --
// a field of exactly 128 is even and stays, so no pair under the threshold is lost).  The +-0.5 products are exact
// because the hardware adds the 32 products of a K block (an integer) before it meets the accumulator
// (tools/ubench/mfma_half_exact.hip).  OR keeps "some flag is set": 16 result registers per SIX MFMAs, 1.33 v_or3_b32
// per MFMA.  A flagged register's candidates (all three fields when the exponent moved, else the flagged ones) are
// evaluated on all 256 bits from the raw rows -- the records are those of every other 256-bit kernel.
constexpr int kS256Half = 0x7e7e7e7e;        // E8M0 126 = 2^-1
constexpr int kS256_7 = (int)0x86868686;     // 2^7
constexpr int kS256_15 = (int)0x8e8e8e8e;    // 2^15
constexpr uint32_t kFlag256 = (1u << 7) | (1u << 15) | (1u << 23);

template <int HT, int G>
__global__ __launch_bounds__(kThreads, 2) void k_hamm256_mfma3(  // (2: accumulators in VGPRs, see k_hamm64_mfma3)
    const uint32_t* __restrict__ rows /* 8 words per row */, uint32_t n, const uint4* __restrict__ qx,
    const uint32_t* __restrict__ qraw /* 8 words per needle */, uint32_t nq, uint32_t n_triples,
    uint32_t triples_per_chunk, uint32_t thresh, unsigned long long* __restrict__ rec, unsigned long long cap,
    unsigned long long* __restrict__ total) {
  __shared__ uint32_t s_c[kWaves][G * 16][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t r = lane & 31u, half = lane >> 5;
  const uint32_t tile0 = (blockIdx.x * kWaves + wave) * HT;
  if (tile0 * 32u >= n) return;

  v8i a[HT][2];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const uint32_t row = (tile0 + t) * 32u + r;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t w = row < n ? rows[(size_t)row * 8u + 2u * k + half] : 0u;
      a[t][k] = fp4_operand(fp4_expand32(w));
    }
  }
  v16f c0;
#pragma unroll
  for (int g = 0; g < 16; ++g) c0[g] = 8388608.0f + (float)((64u + (thresh - 1u)) * 65793u);  // 65793 = 1 + 2^8 + 2^16
  asm volatile("" : "+v"(c0));
  const uint32_t p0 = blockIdx.y * triples_per_chunk;
  const uint32_t p1 = min(n_triples, p0 + triples_per_chunk);
  // triple p = needle tiles 3p, 3p + 1, 3p + 2; a tile is 256 uint4 of scratch, chunk k of it at + 64 k
  const uint4* __restrict__ qp = qx + (size_t)p0 * 768u + half * 32u + r;

  auto step = [&](const uint32_t p, const uint4 (&nb)[6]) {
    v8i b[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = fp4_operand(nb[k]);
#pragma unroll
    for (int t0 = 0; t0 < HT; t0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][0], b[0], c0, 4, 4, 0, kScaleOne, 0, kS256Half);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][1], b[1], c[t], 4, 4, 0, kScaleOne, 0, kS256Half);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][0], b[2], c[t], 4, 4, 0, kScaleOne, 0, kS256_7);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][1], b[3], c[t], 4, 4, 0, kScaleOne, 0, kS256_7);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][0], b[4], c[t], 4, 4, 0, kScaleOne, 0, kS256_15);
#pragma unroll
      for (int t = 0; t < G; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t0 + t][1], b[5], c[t], 4, 4, 0, kScaleOne, 0, kS256_15);
      uint32_t o0 = 0, o1 = 0;
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
          o0 |= __float_as_uint(c[t][g]) | __float_as_uint(c[t][g + 1]);  // v_or3_b32
          o1 |= __float_as_uint(c[t][g + 2]) | __float_as_uint(c[t][g + 3]);
        }
      if (__builtin_amdgcn_ballot_w64(((o0 | o1) & kFlag256) != 0) != 0) {
        // rare: every lane walks its own results (parked in LDS so that the loop stays rolled)
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) s_c[wave][t * 16 + g][lane] = __float_as_uint(c[t][g]);
#pragma unroll 1
        for (uint32_t e = 0; e < (uint32_t)G * 16u; ++e) {
          const uint32_t bits = s_c[wave][e][lane];
          if ((bits & kFlag256) == 0u) continue;
          const uint32_t g = e & 15u;
          const uint32_t row = (tile0 + t0 + (e >> 4)) * 32u + (g & 3u) + 8u * (g >> 2) + 4u * half;
          if (row >= n) continue;
          const bool moved = (bits >> 23) & 1u;  // exponent 151: the lower fields are shifted and rounded -- check all
#pragma unroll 1
          for (uint32_t f = 0; f < 3u; ++f) {
            if (!moved && ((bits >> (7u + 8u * f)) & 1u) == 0u) continue;
            const uint32_t qi = (p * 3u + f) * 32u + r;
            if (qi >= nq) continue;
            uint32_t d = 0;
#pragma unroll
            for (int wd = 0; wd < 8; ++wd) d += __popc(rows[(size_t)row * 8u + wd] ^ qraw[(size_t)qi * 8u + wd]);
            if (d < thresh) {
              const unsigned long long slot = atomicAdd(total, 1ull);
              if (slot < cap) rec[slot] = ((unsigned long long)qi << 41) | ((unsigned long long)d << 32) | row;
            }
          }
        }
      }
    }
  };

  // one triple (6 * HT MFMAs) per trip, the next triple's six tile-chunk loads in flight meanwhile
  uint4 x[6], y[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = qp[(k >> 1) * 256 + (k & 1) * 64];
#pragma unroll 1
  for (uint32_t p = p0; p < p1; ++p) {
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = x[k];
    if (p + 1 < p1) {
      qp += 768;
#pragma unroll
      for (int k = 0; k < 6; ++k) y[k] = qp[(k >> 1) * 256 + (k & 1) * 64];
    }
    step(p, x);
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = y[k];
  }
}

// ---- few needles (<= 512 descriptors: one ORB needle image, the reference's query shape, cvfeaturesindex.cpp:497) ----
// With 16 needle tiles the kernel above gives each wave 6 row tiles and 192 MFMAs of work behind a prologue (row loads,
// FP4 expansion) that nothing overlaps: 1.5 ms for 5*10^7 rows where the matrix cores need 0.9.  Here the roles are
// swapped: ALL needle tiles are the stationary operand (first 128 bits: NT x 2 operands of 4 VGPRs), a persistent wave
// streams row tiles -- the raw words of the next tile are in flight while the current one runs its NT x 2 MFMAs -- and
// the grid is sized for the machine, not for the needle chunks.  First-128-bit prefilter only (thresh <= 40) with the
// three-tiles-per-accumulator fields of k_hamm256_mfma3.  Records identical to k_hamm256_mfma / k_hamm256_scan.
// The table (round 4): every streamed row tile has to be expanded to FP4 -- 2 words x 4 dwords x 9 shift / mask instructions
// per lane, 3.1 of the kernel's 4.6 VALU instructions per MFMA (profiles/r04_pmc_mfma_utilisation.md).  A 256-entry table
// in LDS (byte -> its 8 sign nibbles) turns that into 8 ds_read_b32 + their addresses.
template <int NT>
__global__ __launch_bounds__(kThreads, 2) void k_hamm256_small(  // (2: accumulators in VGPRs, see k_hamm64_mfma3)
    const uint32_t* __restrict__ rows /* 8 words per row */, uint32_t n, const uint4* __restrict__ qx,
    const uint32_t* __restrict__ qraw, uint32_t nq, uint32_t thresh, unsigned long long* __restrict__ rec,
    unsigned long long cap, unsigned long long* __restrict__ total) {
  // needle tiles 3j, 3j+1, 3j+2 share accumulator j (the 8-bit flag fields of k_hamm256_mfma3; the last accumulator holds
  // the one or two tiles that remain): 16 result registers per up to six MFMAs
  constexpr int NA = (NT + 2) / 3;
  constexpr int G = NA % 3 == 0 ? 3 : 2;  // accumulators in flight (independent MFMA chains)
  __shared__ uint32_t s_c[kWaves][G * 16][64];
  __shared__ uint32_t s_lut[256];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // uniform, and known to be
  const uint32_t r = lane & 31u, half = lane >> 5;
  for (uint32_t i = threadIdx.x; i < 256u; i += kThreads) s_lut[i] = fp4_expand32(i).x;
  __syncthreads();
  auto expand = [&](uint32_t w) -> uint4 {
    return make_uint4(s_lut[w & 0xffu], s_lut[(w >> 8) & 0xffu], s_lut[(w >> 16) & 0xffu], s_lut[w >> 24]);
  };
  v8i b[NT][2];
#pragma unroll
  for (int q = 0; q < NT; ++q)
#pragma unroll
    for (int k = 0; k < 2; ++k) b[q][k] = fp4_operand(qx[(size_t)q * 256u + (uint32_t)k * 64u + half * 32u + r]);
  v16f c0;
#pragma unroll
  for (int g = 0; g < 16; ++g) c0[g] = 8388608.0f + (float)((64u + (thresh - 1u)) * 65793u);
  asm volatile("" : "+v"(c0));
  const uint32_t n_row_tiles = (n + 31u) / 32u;
  const uint32_t stride = gridDim.x * kWaves;
  uint32_t tile0 = blockIdx.x * kWaves + wave;
  // Raw words of a row tile through raw buffer loads (round 5): descriptor over the n rows, per-lane offset (row in the
  // tile, word) constant, the tile's offset scalar -- no 64-bit vector address arithmetic, and rows past the end read as
  // zero by the descriptor's range check instead of a select per word (the launcher keeps n * 32 below 2^32).
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(rows), 0, (int)(n * 32u), 0x27000);
  const uint32_t voff = r * 32u + half * 4u;
  auto load_raw = [&](uint32_t t, uint32_t& w0, uint32_t& w1) {
    const uint32_t so = min(t, n_row_tiles) * 1024u;  // (a tile past the end: out of range, zeros)
    w0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)so, 0);
    w1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(voff + 8u), (int)so, 0);
  };
  constexpr int kAhead = 4;  // row tiles in flight (a tile's MFMAs take ~0.6 us, an HBM load under load 1-2 us)
  uint32_t p0[kAhead], p1[kAhead];
#pragma unroll
  for (int u = 0; u < kAhead; ++u) load_raw(tile0 + (uint32_t)u * stride, p0[u], p1[u]);
  // kAhead tiles per trip, each from its own pair of registers (round 5: the shifting ring cost six moves per tile)
#pragma unroll 1
  for (; tile0 < n_row_tiles; tile0 += (uint32_t)kAhead * stride) {
#pragma unroll
  for (int u = 0; u < kAhead; ++u) {
    const uint32_t tile = tile0 + (uint32_t)u * stride;
    if (tile >= n_row_tiles) break;  // (uniform)
    const v8i a0 = fp4_operand(expand(p0[u])), a1 = fp4_operand(expand(p1[u]));
    load_raw(tile + (uint32_t)kAhead * stride, p0[u], p1[u]);  // behind the MFMAs of kAhead tiles
#pragma unroll
    for (int j0 = 0; j0 < NA; j0 += G) {
      v16f c[G];
#pragma unroll
      for (int t = 0; t < G; ++t) c[t] = c0;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const int sc = f == 0 ? kS256Half : (f == 1 ? kS256_7 : kS256_15);
#pragma unroll
        for (int t = 0; t < G; ++t) {
          const int q = 3 * (j0 + t) + f;
          if (j0 + t < NA && q < NT)
            c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b[q][0], c[t], 4, 4, 0, kScaleOne, 0, sc);
        }
#pragma unroll
        for (int t = 0; t < G; ++t) {
          const int q = 3 * (j0 + t) + f;
          if (j0 + t < NA && q < NT)
            c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b[q][1], c[t], 4, 4, 0, kScaleOne, 0, sc);
        }
      }
      // (a field without a tile stays at its C0 value 64 + b < 128: never flagged for b < 64)
      uint32_t o0 = 0, o1 = 0;
#pragma unroll
      for (int t = 0; t < G; ++t)
        if (j0 + t < NA) {
#pragma unroll
          for (int g = 0; g < 16; g += 4) {
            o0 |= __float_as_uint(c[t][g]) | __float_as_uint(c[t][g + 1]);  // v_or3_b32
            o1 |= __float_as_uint(c[t][g + 2]) | __float_as_uint(c[t][g + 3]);
          }
        }
      if (__builtin_amdgcn_ballot_w64(((o0 | o1) & kFlag256) != 0) != 0) {  // rare: as in k_hamm256_mfma3
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) s_c[wave][t * 16 + g][lane] = (j0 + t < NA) ? __float_as_uint(c[t][g]) : 0u;
#pragma unroll 1
        for (uint32_t e = 0; e < (uint32_t)G * 16u; ++e) {
          const uint32_t bits = s_c[wave][e][lane];
          if ((bits & kFlag256) == 0u) continue;
          const uint32_t g = e & 15u;
          const uint32_t row = tile * 32u + (g & 3u) + 8u * (g >> 2) + 4u * half;
          if (row >= n) continue;
          const bool moved = (bits >> 23) & 1u;
#pragma unroll 1
          for (uint32_t f = 0; f < 3u; ++f) {
            if (!moved && ((bits >> (7u + 8u * f)) & 1u) == 0u) continue;
            const uint32_t qt = 3u * ((uint32_t)j0 + (e >> 4)) + f;
            const uint32_t qi = qt * 32u + r;
            if (qt >= (uint32_t)NT || qi >= nq) continue;
            uint32_t d = 0;
#pragma unroll
            for (int wd = 0; wd < 8; ++wd) d += __popc(rows[(size_t)row * 8u + wd] ^ qraw[(size_t)qi * 8u + wd]);
            if (d < thresh) {
              const unsigned long long slot = atomicAdd(total, 1ull);
              if (slot < cap) rec[slot] = ((unsigned long long)qi << 41) | ((unsigned long long)d << 32) | row;
            }
          }
        }
      }
    }
  }
  }
}

int g_scan256_small = 1;   // "scan256_small": the stationary-needle kernel for <= 512 needle descriptors (default on)
int g_scan256_mfma = 1;
constexpr int kPre128MaxThresh = 40;  // thresholds up to this take a first-128-bit prefilter variant

}  // namespace

void set_scan256_mfma(int on) {
  if (on >= 0) g_scan256_mfma = on;
}
void set_scan256_small(int v) {
  if (v == 0 || v == 1) g_scan256_small = v;
}

bool scan256_mfma_wanted(size_t n, size_t nq, int thresh) {
  if (thresh < 1 || thresh > 257) return false;
  if (g_scan256_mfma == 2) return true;  // forced (tests)
  return g_scan256_mfma && nq >= 64 && n >= 4096;
}

int launch_scan256_mfma(const uint8_t* d_rows, size_t n, const uint8_t* d_q, size_t nq, int thresh,
                        unsigned long long* d_rec, size_t cap, unsigned long long* d_total,
                        hipStream_t stream) {
  if (n == 0 || nq == 0 || thresh <= 0) return CBH_OK;
  if (n > 0xfffffff0ull || nq >= (1u << 23) || thresh > 257) return CBH_E_INVAL;
  const uint32_t n_tiles = (uint32_t)((nq + 31) / 32);
  const uint32_t nq_pad = n_tiles * 32u;
  uint4* qx = nullptr;
  CBH_HIP(cbh::malloc_async((void**)&qx, (size_t)nq_pad * 128u, stream));
  hipLaunchKernelGGL(k_expand_needles256, dim3((8u * nq_pad + 255u) / 256u), dim3(256), 0, stream,
                     reinterpret_cast<const uint32_t*>(d_q), (uint32_t)nq, nq_pad, qx);
  if (g_scan256_small && thresh <= kPre128MaxThresh && n_tiles <= 16 && (n >= 4096 || g_scan256_mfma == 2) &&
      n <= ((size_t)1 << 27) - 64) {  // (its buffer descriptor spans n * 32 bytes)
    // needle tiles padded to the template's count read zero descriptors from the scratch (rows of zero bits never pass:
    // qi >= nq is dropped in the hit path)
    const uint32_t nt = n_tiles <= 4 ? 4 : n_tiles <= 8 ? 8 : 16;
    if (nt != n_tiles) {  // the scratch must hold nt tiles
      (void)cbh::free_async(qx, stream);
      qx = nullptr;
      CBH_HIP(cbh::malloc_async((void**)&qx, (size_t)nt * 32u * 128u, stream));
      hipLaunchKernelGGL(k_expand_needles256, dim3((8u * nt * 32u + 255u) / 256u), dim3(256), 0, stream,
                         reinterpret_cast<const uint32_t*>(d_q), (uint32_t)nq, nt * 32u, qx);
    }
    const uint32_t row_tiles = (uint32_t)((n + 31) / 32);
    uint32_t wgs_s = 2048u;  // workgroups of the persistent grid; measured: 512 / 1024 / 2048 / 8192 = 1.22 / 1.10 / 1.07 / 1.07 ms
    wgs_s = std::min(wgs_s, (row_tiles + kWaves - 1) / kWaves);
#define CBH_SMALL(NTT)                                                                                               \
  hipLaunchKernelGGL((k_hamm256_small<NTT>), dim3(wgs_s), dim3(kThreads), 0, stream,                                   \
                     reinterpret_cast<const uint32_t*>(d_rows), (uint32_t)n, qx, reinterpret_cast<const uint32_t*>(d_q), \
                     (uint32_t)nq, (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total)
    if (nt == 4) CBH_SMALL(4); else if (nt == 8) CBH_SMALL(8); else CBH_SMALL(16);
#undef CBH_SMALL
    hipError_t es = hipGetLastError();
    (void)cbh::free_async(qx, stream);
    CBH_HIP(es);
    return CBH_OK;
  }
  if (thresh <= kPre128MaxThresh && n_tiles >= 3) {
    const uint32_t n_triples = (n_tiles + 2u) / 3u;
    if (n_triples * 3u != n_tiles) {  // the scratch must hold whole triples (zero descriptors: dropped at qi >= nq)
      (void)cbh::free_async(qx, stream);
      qx = nullptr;
      CBH_HIP(cbh::malloc_async((void**)&qx, (size_t)n_triples * 96u * 128u, stream));
      hipLaunchKernelGGL(k_expand_needles256, dim3((8u * n_triples * 96u + 255u) / 256u), dim3(256), 0, stream,
                         reinterpret_cast<const uint32_t*>(d_q), (uint32_t)nq, n_triples * 96u, qx);
    }
    // shape: 12 row tiles per wave, groups of 2 (measured, 1e7 rows x 32 000 needles: 6/2, 6/3, 8/2, 12/2, 12/3 =
    // 12.2 / 11.6 / 11.3 / 10.9 / 10.9 ms; one tile per accumulator 13.4)
    constexpr int ht3 = 12;
    const uint32_t rows_per_wg3 = 32u * (uint32_t)ht3 * kWaves;
    const uint32_t wgs3 = (uint32_t)((n + rows_per_wg3 - 1) / rows_per_wg3);
    uint32_t tpc3 = 43;  // needle triples per chunk (4128 descriptors)
    while (tpc3 > 2 && (uint64_t)wgs3 * ((n_triples + tpc3 - 1) / tpc3) < 8192) tpc3 = (tpc3 + 1) >> 1;
    uint32_t chunks3 = (n_triples + tpc3 - 1) / tpc3;
    if (chunks3 > 65535) {
      tpc3 = (n_triples + 65534) / 65535;
      chunks3 = (n_triples + tpc3 - 1) / tpc3;
    }
#define CBH_256F3(HT, GG)                                                                               \
  hipLaunchKernelGGL((k_hamm256_mfma3<HT, GG>), dim3(wgs3, chunks3), dim3(kThreads), 0, stream,         \
                     reinterpret_cast<const uint32_t*>(d_rows), (uint32_t)n, qx,                        \
                     reinterpret_cast<const uint32_t*>(d_q), (uint32_t)nq, n_triples, tpc3,             \
                     (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total)
    CBH_256F3(12, 2);
#undef CBH_256F3
    hipError_t e3 = hipGetLastError();
    (void)cbh::free_async(qx, stream);
    CBH_HIP(e3);
    return CBH_OK;
  }
  // fewer than three needle tiles, or thresholds beyond the prefilter's range: one tile per accumulator
  constexpr int ht = 6;
  const uint32_t rows_per_wg = 32u * (uint32_t)ht * kWaves;
  const uint32_t wgs = (uint32_t)((n + rows_per_wg - 1) / rows_per_wg);
  uint32_t tpc = 128;  // needle tiles per chunk (4096 descriptors)
  while (tpc > 4 && (uint64_t)wgs * ((n_tiles + tpc - 1) / tpc) < 8192) tpc >>= 1;
  uint32_t chunks = (n_tiles + tpc - 1) / tpc;
  if (chunks > 65535) {
    tpc = (n_tiles + 65534) / 65535;
    chunks = (n_tiles + tpc - 1) / tpc;
  }
  const bool pre128 = thresh <= kPre128MaxThresh;
#define CBH_256(HT, GG, KC)                                                                       \
  hipLaunchKernelGGL((k_hamm256_mfma<HT, GG, KC>), dim3(wgs, chunks), dim3(kThreads), 0, stream,  \
                     reinterpret_cast<const uint32_t*>(d_rows), (uint32_t)n, qx,                  \
                     reinterpret_cast<const uint32_t*>(d_q), (uint32_t)nq, n_tiles, tpc,          \
                     (uint32_t)thresh, d_rec, (unsigned long long)cap, d_total)
  if (pre128) CBH_256(6, 3, 2); else CBH_256(6, 3, 4);
#undef CBH_256
  hipError_t e = hipGetLastError();
  (void)cbh::free_async(qx, stream);
  CBH_HIP(e);
  return CBH_OK;
}

}  // namespace cbh
