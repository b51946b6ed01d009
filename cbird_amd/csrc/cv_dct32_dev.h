// cv_dct32_dev.h -- ONE LABELLED UNIT for the third-party arithmetic of dctHash64's stages 3 and 5 on the device:
//
//     cv::dct(freq, freq)         src/cvutil.cpp:476-477   (32x32 CV_32F, forward)
//     float(cv::sum(freq)[0])     src/cvutil.cpp:528       (1x64 CV_32F)
//
// as OpenCV 2.4.13.7 (cbird.pri:148-152) evaluates them: modules/core/src/dxt.cpp DCT_32f -> RealDFT -> DFT
// (bit reversal + two radix-4 passes for the 16-point complex transform, twiddle tables from DFTInit / DCTInit) and
// modules/core/src/stat.cpp sum_<float,double> (groups of four summed in float, accumulated in double).  OpenCV is
// not available in this image, so this is the published algorithm AS RECALLED ("parity unpinned" until
// tools/gen_golden_opencv.cpp has run against the real library); the CPU twin with the same operation order is
// oracle/cv_dct32.c, and the device code below must stay operation-for-operation identical to it (products rounded
// before sums: the library is built without FMA, this file is compiled with -ffp-contract=off).
//
// Selected by the tuning knob "hash_dct" (1 = this unit, the default; 0 = the canonical 9x32 matrix form documented in
// NOTES.md section 3).  tools/hash_at_risk.py measures how often the two and a float64 evaluation disagree on a bit.
//
// Everything is written for full unrolling: a lane owns one 32-point transform in registers, every index is a
// compile-time constant, the twiddles are wave-uniform (scalar loads), and only the first nine outputs are kept --
// the dead half of the post-processing disappears at compile time without changing the operations of the live half.
#pragma once

namespace cbh {

struct CvDct32Tabs {
  float dft_re[32], dft_im[32];  // exp(-2*pi*i*k/32): DFTInit's double recurrence, stored as float
  float dct_re[17], dct_im[17];  // 0.25 * exp(-i*pi*k/64): DCTInit's double recurrence, stored as float
};

// host: the tables exactly as DFTInit(32) / DCTInit(32) build them (see oracle/cv_dct32.c for the prose)
inline void cv_dct32_make_tabs(CvDct32Tabs* t) {
  const double two_pi = 6.283185307179586476925286766559;
  {
    const double c = __builtin_cos(two_pi / 32), s = __builtin_sin(two_pi / 32);
    double w_re = c, w_im = -s;
    const double w1_re = c, w1_im = -s;
    t->dft_re[0] = 1.f, t->dft_im[0] = 0.f;
    t->dft_re[16] = -1.f, t->dft_im[16] = 0.f;
    for (int i = 1; i < 16; ++i) {
      t->dft_re[i] = (float)w_re, t->dft_im[i] = (float)w_im;
      t->dft_re[32 - i] = (float)w_re, t->dft_im[32 - i] = (float)-w_im;
      const double tt = w_re * w1_re - w_im * w1_im;
      w_im = w_re * w1_im + w_im * w1_re;
      w_re = tt;
    }
  }
  {
    const double c = __builtin_cos(two_pi / 128), s = __builtin_sin(two_pi / 128);
    const double w1_re = c, w1_im = -s;
    double w_re = (float)(2 * 0.125), w_im = 0.f;
    for (int i = 0; i <= 16; ++i) {
      t->dct_re[i] = (float)w_re, t->dct_im[i] = (float)w_im;
      const double tt = w_re * w1_re - w_im * w1_im;
      w_im = w_re * w1_im + w_im * w1_re;
      w_re = tt;
    }
  }
}

#if defined(__HIPCC__)
namespace cvdct {

constexpr int bitrev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

// one radix-4 pass of DFT<float> over 16 complex values (NX = 1: n 1 -> 4, DW0 = 8; NX = 4: n 4 -> 16, DW0 = 2)
template <int NX, int DW0>
__device__ __forceinline__ void radix4_pass(float (&re)[16], float (&im)[16], const CvDct32Tabs* __restrict__ t) {
  constexpr int n = NX * 4;
#pragma unroll
  for (int i = 0; i < 16; i += n) {
    {
      const int a = i, b = i + NX, c = i + 2 * NX, d = i + 3 * NX;  // v0[0], v0[nx], v1[0], v1[nx]
      float r0 = re[c], i0 = im[c];
      float r4 = re[d], i4 = im[d];
      const float r1 = r0 + r4, i1 = i0 + i4;
      const float r3 = i0 - i4, i3 = r4 - r0;
      float r2 = re[a], i2 = im[a];
      r4 = re[b], i4 = im[b];
      r0 = r2 + r4, i0 = i2 + i4;
      r2 -= r4, i2 -= i4;
      re[a] = r0 + r1, im[a] = i0 + i1;
      re[c] = r0 - r1, im[c] = i0 - i1;
      re[b] = r2 + r3, im[b] = i2 + i3;
      re[d] = r2 - r3, im[d] = i2 - i3;
    }
#pragma unroll
    for (int j = 1; j < NX; ++j) {
      const int dw = DW0 * j;
      const int a = i + j, b = a + NX, c = a + 2 * NX, d = a + 3 * NX;
      float r2 = re[b] * t->dft_re[dw * 2] - im[b] * t->dft_im[dw * 2];
      float i2 = re[b] * t->dft_im[dw * 2] + im[b] * t->dft_re[dw * 2];
      float r0 = re[c] * t->dft_im[dw] + im[c] * t->dft_re[dw];
      float i0 = re[c] * t->dft_re[dw] - im[c] * t->dft_im[dw];
      float r3 = re[d] * t->dft_im[dw * 3] + im[d] * t->dft_re[dw * 3];
      float i3 = re[d] * t->dft_re[dw * 3] - im[d] * t->dft_im[dw * 3];
      const float r1 = i0 + i3, i1 = r0 + r3;
      r3 = r0 - r3, i3 = i3 - i0;
      const float r4 = re[a], i4 = im[a];
      r0 = r4 + r2, i0 = i4 + i2;
      r2 = r4 - r2, i2 = i4 - i2;
      re[a] = r0 + r1, im[a] = i0 + i1;
      re[c] = r0 - r1, im[c] = i0 - i1;
      re[b] = r2 + r3, im[b] = i2 + i3;
      re[d] = r2 - r3, im[d] = i2 - i3;
    }
  }
}

// DCT_32f for n = 32, outputs 0..8 only.  x[32] is the input row/column in natural order.
__device__ __forceinline__ void dct32_first9(const float (&x)[32], const CvDct32Tabs* __restrict__ t,
                                             float (&out)[9]) {
  // 1. dft_src[j] = x[2j], dft_src[31-j] = x[2j+1]
  float buf[32];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    buf[j] = x[2 * j];
    buf[31 - j] = x[2 * j + 1];
  }
  // 2a. complex DFT(16) on pairs, input in bit-reversed order
  float re[16], im[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    re[i] = buf[2 * bitrev4(i)];
    im[i] = buf[2 * bitrev4(i) + 1];
  }
  radix4_pass<1, 8>(re, im, t);
  radix4_pass<4, 2>(re, im, t);
  float dst[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dst[2 * i] = re[i];
    dst[2 * i + 1] = im[i];
  }
  // 2b. RealDFT post-processing (scale = 1, scale2 = 0.5)
  {
    float tt = dst[0] - dst[1];
    dst[0] = (dst[0] + dst[1]) * 1.f;
    dst[1] = tt * 1.f;
    const float t0 = dst[16];
    tt = dst[31];
    dst[31] = dst[1];
#pragma unroll
    for (int j = 2; j < 16; j += 2) {
      const float wre = t->dft_re[j / 2], wim = t->dft_im[j / 2];
      float h2_re = 0.5f * (dst[j + 1] + tt);
      float h2_im = 0.5f * (dst[32 - j] - dst[j]);
      const float h1_re = 0.5f * (dst[j] + dst[32 - j]);
      const float h1_im = 0.5f * (dst[j + 1] - tt);
      tt = h2_re * wre - h2_im * wim;
      h2_im = h2_re * wim + h2_im * wre;
      h2_re = tt;
      tt = dst[32 - j - 1];
      dst[j - 1] = h1_re + h2_re;
      dst[32 - j - 1] = h1_re - h2_re;
      dst[j] = h1_im + h2_im;
      dst[32 - j] = h2_im - h1_im;
    }
    dst[15] = t0 * 1.f;
    dst[16] = -tt * 1.f;
  }
  // 3. rotation by the DCT twiddles; y[j], j = 0..8
  out[0] = (dst[0] * t->dct_re[0]) * 0.70710678118654752440084436210485f;
#pragma unroll
  for (int j = 1; j < 9; ++j) out[j] = t->dct_re[j] * dst[2 * j - 1] - t->dct_im[j] * dst[2 * j];
}

// cv::sum over 64 floats held one per lane of a wave (lane i = element i): ((a+b)+c)+d per group of four in float,
// groups accumulated in double, index order.  c_bits = the lane's value as int bits.  Wave-uniform result.
__device__ __forceinline__ double sum64_lanes(int c_bits) {
  double sum = 0.0;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(c_bits, 4 * g));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(c_bits, 4 * g + 1));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(c_bits, 4 * g + 2));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(c_bits, 4 * g + 3));
    sum += (double)(((a + b) + c) + d);
  }
  return sum;
}

// the same for an image held by a HALF-wave: lane base + i holds element i in c0_bits and element 32 + i in c1_bits
// (base = 0 or 32).  Wave-uniform result; no LDS round trips.
__device__ __forceinline__ double sum64_halfwave(int c0_bits, int c1_bits, int base) {
  double sum = 0.0;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int src = g < 8 ? c0_bits : c1_bits;
    const int l = base + 4 * (g & 7);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(src, l));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(src, l + 1));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(src, l + 2));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(src, l + 3));
    sum += (double)(((a + b) + c) + d);
  }
  return sum;
}

// the same from an array in memory (serial, one lane)
__device__ __forceinline__ double sum64_mem(const float* __restrict__ y, const unsigned char* __restrict__ zz) {
  double sum = 0.0;
  for (int g = 0; g < 16; ++g)
    sum += (double)(((y[zz[4 * g]] + y[zz[4 * g + 1]]) + y[zz[4 * g + 2]]) + y[zz[4 * g + 3]]);
  return sum;
}

}  // namespace cvdct
#endif

}  // namespace cbh
