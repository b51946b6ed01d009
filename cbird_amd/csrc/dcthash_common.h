// dcthash_common.h -- what the dctHash64 kernels of dcthash.hip (whole images) and kphash.hip (rectangles and keypoint
// squares inside an image) share: the stage 3-6 tails, the area tables' entry types, the exact blur arithmetic of the
// lane-per-8-columns kernels.  Everything device-side sits in an anonymous namespace: each translation unit compiles its
// own copy.
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <mutex>
#include <tuple>
#include <vector>

#include "cbh_internal.h"
#include "cv_dct32_dev.h"

namespace cbh {

struct DctTables {
  unsigned char zz[64];  // zig-zag positions 6..69 -> index into the 9x9 block (row*9+col)
  CvDct32Tabs cv;        // stages 3 / 5: cv::dct / cv::sum as OpenCV 2.4 evaluates them (cv_dct32_dev.h)
};

struct AreaTab {
  int si, di;
  float alpha;
};
// the y table of make_area_tab seen from a SOURCE row (k_blur_area_regs<.., FUSE>): a row contributes to one or two
// consecutive output rows (scale >= 1).  info: bits 0..7 di of the first entry, bit 8 = that entry opens its cell,
// bit 9 = it closes it, bit 10 = a second entry exists (cell di + 1), bit 11 / 12 = opens / closes for that one
// k0 = 0.f when the first entry opens its cell, 1.f when it continues one: sum = fma(sum, k0, a0 * v) is `t0` or `sum + t0`
// (one rounding either way) without a select (k_band_area)
struct YRow {
  float a0, a1;
  int info;
  float k0;
};

namespace {

constexpr int kThreads = 256;


__device__ __forceinline__ int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    p = p < 0 ? -p : 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// stages 3-6 from a 32x32 u8 tile in LDS, called by all 256 threads of a workgroup (wave 0 works); tile must be 16-byte
// aligned.  cv::dct as OpenCV 2.4 evaluates it (cv_dct32_dev.h): 32 row transforms on the lanes of wave 0, then the nine
// column transforms; cv::sum's grouping for the threshold.
__device__ __forceinline__ void hash_from_tile(const unsigned char* __restrict__ tile /*LDS*/,
                                               const unsigned char* __restrict__ sZ /*LDS 64*/,
                                               float* __restrict__ sT /*LDS 288*/,
                                               float* __restrict__ sY /*LDS 81*/,
                                               uint64_t* __restrict__ out,
                                               const DctTables* __restrict__ tabs) {
  const int tid = threadIdx.x;
  {
    if (tid < 32) {
      const uint4* trow = reinterpret_cast<const uint4*>(tile + tid * 32);
      const uint4 p0 = trow[0], p1 = trow[1];
      const unsigned px[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
      float x[32], y[9];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        x[4 * q] = (float)(px[q] & 0xffu);
        x[4 * q + 1] = (float)((px[q] >> 8) & 0xffu);
        x[4 * q + 2] = (float)((px[q] >> 16) & 0xffu);
        x[4 * q + 3] = (float)(px[q] >> 24);
      }
      cvdct::dct32_first9(x, &tabs->cv, y);
#pragma unroll
      for (int k = 0; k < 9; ++k) sT[tid * 9 + k] = y[k];
    }
    __syncthreads();
    if (tid < 9) {
      float x[32], y[9];
#pragma unroll
      for (int r = 0; r < 32; ++r) x[r] = sT[r * 9 + tid];
      cvdct::dct32_first9(x, &tabs->cv, y);
#pragma unroll
      for (int u = 0; u < 9; ++u) sY[u * 9 + tid] = y[u];
    }
    __syncthreads();
    if (tid < 64) {
      const float c = sY[sZ[tid]];
      const float thr = (float)cvdct::sum64_lanes(__builtin_bit_cast(int, c)) / 64;
      const unsigned long long b = __ballot(tid >= 1 && c > thr);
      if (tid == 0) *out = b ? b : 1ull;
    }
  }
}




typedef float f32_lds __attribute__((may_alias));
// stages 3-6 for one image per HALF-WAVE (both halves of a wave work on their own image): tile = the image's 32 x 32 bytes
// in LDS, sT / sY = 288 / 84 floats of LDS of its own.  Lane l32 = lane & 31: row transform of tile row l32, then the nine
// column transforms, then selected coefficients l32 and 32 + l32; the threshold's double sum in cv::sum's grouping runs
// on lane broadcasts for both images of the wave at once.  Every lane of the workgroup must call it (two barriers
// inside); returns the hash (valid in every lane of the half-wave).
__device__ __forceinline__ unsigned long long hash_halfwave(const unsigned char* tile, f32_lds* sT, f32_lds* sY,
                                                            const DctTables* __restrict__ tabs, int lane) {
  const int l32 = lane & 31, hw = (lane >> 5) & 1;
  {
    float x[32], y[9];
    const uint4* trow = reinterpret_cast<const uint4*>(tile + l32 * 32);
    const uint4 a = trow[0], b = trow[1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[4 * i + 0] = (float)(w[i] & 0xffu);
      x[4 * i + 1] = (float)((w[i] >> 8) & 0xffu);
      x[4 * i + 2] = (float)((w[i] >> 16) & 0xffu);
      x[4 * i + 3] = (float)(w[i] >> 24);
    }
    cvdct::dct32_first9(x, &tabs->cv, y);  // cv::dct's own evaluation (cv_dct32_dev.h)
#pragma unroll
    for (int k = 0; k < 9; ++k) sT[l32 * 9 + k] = y[k];
  }
  __syncthreads();
  if (l32 < 9) {  // nine column transforms per image
    float x[32], y[9];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = sT[r * 9 + l32];
    cvdct::dct32_first9(x, &tabs->cv, y);
#pragma unroll
    for (int u = 0; u < 9; ++u) sY[u * 9 + l32] = y[u];
  }
  __syncthreads();
  const float c0 = sY[tabs->zz[l32]];
  const float c1 = sY[tabs->zz[l32 + 32]];
  const int cb0 = __builtin_bit_cast(int, c0), cb1 = __builtin_bit_cast(int, c1);
  const double sumA = cvdct::sum64_halfwave(cb0, cb1, 0), sumB = cvdct::sum64_halfwave(cb0, cb1, 32);
  const float thr = (float)(hw ? sumB : sumA) / 64;
  const unsigned long long b0 = __ballot(c0 > thr);
  const unsigned long long b1 = __ballot(c1 > thr);
  const int sh = hw * 32;
  unsigned long long hv = ((b0 >> sh) & 0xffffffffull) | (((b1 >> sh) & 0xffffffffull) << 32);
  hv &= ~1ull;  // bit 0 is never encoded (cvutil.cpp:537)
  return hv == 0 ? 1ull : hv;
}


__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) {
  return __builtin_amdgcn_udot4(a, b, c, false);
}


template <int K>
struct BlurK;  // nearest(S / K^2) = ((S + add) * m) >> 24, exact for S <= K^2 * 255; (S + add) * m < 2^32
template <>
struct BlurK<3> {
  static constexpr unsigned m = 1864136u, add = 4u;  // 9 * m - 2^24 = 8: error < 2299 * 8 / (9 * 2^24) << 1/9
};
template <>
struct BlurK<5> {
  static constexpr unsigned m = 671089u, add = 12u;  // 25 * m - 2^24 = 9
};
template <>
struct BlurK<7> {
  static constexpr unsigned m = 342393u, add = 24u;  // 49 * m - 2^24 = 41
};

// (u16 half of a dword) * m in one VALU op (SDWA word select); operands < 2^24
__device__ __forceinline__ unsigned mul24_word0(unsigned p, unsigned m) {
  unsigned r;
  asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"
      : "=v"(r)
      : "v"(p), "v"(m));
  return r;
}
__device__ __forceinline__ unsigned mul24_word1(unsigned p, unsigned m) {
  unsigned r;
  asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
      : "=v"(r)
      : "v"(p), "v"(m));
  return r;
}

// the eight blurred pixels of a lane from its four packed column-sum pairs: nearest(S / K^2) = byte 3 of (S + add) * m
// -- one SDWA multiply per pixel (word select), then the quotient bytes are gathered with v_perm_b32 (selector
// 0x0c = zero byte): 14 ops per 8 pixels (and/shift + multiply + shift + shift/or packing took ~30)
template <int K>
__device__ __forceinline__ uint2 blur_quotients(const unsigned (&S)[4]) {
  unsigned pr[8];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    pr[2 * c] = mul24_word0(S[c], BlurK<K>::m);
    pr[2 * c + 1] = mul24_word1(S[c], BlurK<K>::m);
  }
  uint2 qo;
  qo.x = __builtin_amdgcn_perm(pr[1], pr[0], 0x0c0c0703u) | __builtin_amdgcn_perm(pr[3], pr[2], 0x07030c0cu);
  qo.y = __builtin_amdgcn_perm(pr[5], pr[4], 0x0c0c0703u) | __builtin_amdgcn_perm(pr[7], pr[6], 0x07030c0cu);
  return qo;
}

constexpr int kBlurRB = 16;  // output rows per workgroup

// byte mask of window dword d (bytes 4d..4d+3 of the 16-byte window whose byte 4 is the lane's pixel 0) for the
// K-tap sum centred on the lane's pixel i
constexpr unsigned tap_mask(int R, int i, int d) {
  unsigned m = 0;
  for (int b = 0; b < 4; ++b) {
    const int byte = 4 * d + b;
    if (byte >= 4 + i - R && byte <= 4 + i + R) m |= 1u << (8 * b);
  }
  return m;
}
template <int R, int I>
__device__ __forceinline__ unsigned hsum_tap(const unsigned (&W)[4]) {
  unsigned acc = 0;
  if constexpr (tap_mask(R, I, 0) != 0) acc = udot4(W[0], tap_mask(R, I, 0), acc);
  if constexpr (tap_mask(R, I, 1) != 0) acc = udot4(W[1], tap_mask(R, I, 1), acc);
  if constexpr (tap_mask(R, I, 2) != 0) acc = udot4(W[2], tap_mask(R, I, 2), acc);
  if constexpr (tap_mask(R, I, 3) != 0) acc = udot4(W[3], tap_mask(R, I, 3), acc);
  return acc;
}

// the eight K-tap sums of a lane's 16-byte window, packed in pairs.  K = 7 shares the two full-dword sums between the
// outputs like k_dcthash_256 does: 14 v_dot4_u32_u8 instead of 20.
template <int R>
__device__ __forceinline__ void hsum_pairs(const unsigned (&W)[4], unsigned (&P)[4]) {
  if constexpr (R == 3) {  // output i = window bytes i+1 .. i+7
    const unsigned T1 = udot4(W[1], 0x01010101u, 0u), T2 = udot4(W[2], 0x01010101u, 0u);
    const unsigned H0 = udot4(W[0], 0x01010100u, T1);
    const unsigned H1 = udot4(W[0], 0x01010000u, udot4(W[2], 0x00000001u, T1));
    const unsigned H2 = udot4(W[0], 0x01000000u, udot4(W[2], 0x00000101u, T1));
    const unsigned H3 = udot4(W[2], 0x00010101u, T1);
    const unsigned H4 = udot4(W[1], 0x01010100u, T2);
    const unsigned H5 = udot4(W[1], 0x01010000u, udot4(W[3], 0x00000001u, T2));
    const unsigned H6 = udot4(W[1], 0x01000000u, udot4(W[3], 0x00000101u, T2));
    const unsigned H7 = udot4(W[3], 0x00010101u, T2);
    P[0] = H0 | (H1 << 16), P[1] = H2 | (H3 << 16), P[2] = H4 | (H5 << 16), P[3] = H6 | (H7 << 16);
  } else {
    P[0] = hsum_tap<R, 0>(W) | (hsum_tap<R, 1>(W) << 16), P[1] = hsum_tap<R, 2>(W) | (hsum_tap<R, 3>(W) << 16);
    P[2] = hsum_tap<R, 4>(W) | (hsum_tap<R, 5>(W) << 16), P[3] = hsum_tap<R, 6>(W) | (hsum_tap<R, 7>(W) << 16);
  }
}

typedef unsigned u32_any_align __attribute__((aligned(1)));

}  // namespace

// host side (dcthash.hip)
bool area_fast(int w, int h);  // cv::resize takes its integer-ratio INTER_AREA path for this geometry
std::vector<AreaTab> make_area_tab(int ssize, int dsize, std::vector<int>* first);  // computeResizeAreaTab
int get_tables(const DctTables** out);  // the current device's copy of the stage 3-6 tables

}  // namespace cbh
