/* oracle/cv_dct32.c -- TEST INFRASTRUCTURE ONLY (see cbird_oracle.c's header for the rules).
 *
 * ONE LABELLED TRANSLATION UNIT for the third-party arithmetic of stage 3 and stage 5 of dctHash64:
 *
 *     cv::dct(freq, freq)              src/cvutil.cpp:476-477   (32x32, CV_32F, forward, in place)
 *     float(cv::sum(freq)[0])          src/cvutil.cpp:528       (1x64, CV_32F)
 *
 * Both live in OpenCV 2.4.13.7 (pin: cbird.pri:148-152; modules/core/src/dxt.cpp and stat.cpp), which is neither
 * vendored under /root/reference nor installed in this image.  What follows restates the PUBLISHED algorithm of
 * that version AS RECALLED -- "parity unpinned" until tools/gen_golden_opencv.cpp has been run somewhere with the
 * real library (tests/test_opencv_golden.py consumes its output).  The structure is:
 *
 *   cv::dct, 2-D, even length n: stage 0 transforms every ROW, stage 1 every COLUMN of the result, each with the
 *   1-D routine DCT_32f:
 *     1. permute          dft_src[j] = x[2j], dft_src[n-1-j] = x[2j+1]                         (j < n/2)
 *     2. RealDFT(n)       = complex DFT of length n/2 on the pairs (dft_src[2k], dft_src[2k+1]) + one pass of
 *                           "split" butterflies with the DFT twiddles -> packed spectrum
 *                           Re X0, Re X1, Im X1, ..., Re X(n/2)
 *        complex DFT(16)  = bit-reversal permutation, then two radix-4 passes (n = 1->4->16), twiddles from a
 *                           table of n complex floats built by a double-precision recurrence (DFTInit)
 *     3. rotate           y[0] = X0 * w0 * sin45;  y[j] = w[j].re*ReXj - w[j].im*ImXj;
 *                         y[n-j] = -w[j].im*ReXj - w[j].re*ImXj;  y[n/2] = X(n/2) * w[n/2].re
 *                         with w[j] = sqrt(2/n) * exp(-i*pi*j/(2n)), same kind of recurrence (DCTInit)
 *   All arithmetic in f32, products rounded before they are added (the library is built for SSE2/SSE3 without FMA;
 *   its SSE3 radix-4 routine forms the same sums and products as the scalar one, so the scalar order is used here).
 *   ENABLE_FAST_MATH=1 (docker/build-opencv.sh:31) permits gcc to re-associate; nothing below has more than one
 *   legal association except the 4-term float sums of cv::sum, kept left to right.
 *
 *   cv::sum on CV_32F (stat.cpp, sum_<float,double>, CV_ENABLE_UNROLLED): the 64 values are taken four at a time,
 *   ((a+b)+c)+d evaluated IN FLOAT, and each group is added to a double accumulator.
 *
 * The canonical alternative (a separable 9x32 matrix product with a fixed fmaf order) is
 * orc_hash_from_tile32 variant 0 in cbird_oracle.c; tools/hash_at_risk.py measures how often the two -- and a
 * float64 evaluation -- disagree on a bit.
 *
 * Build: part of libcbird_oracle.so (oracle/Makefile), -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define N 32
#define N2 16

typedef struct {
  float re, im;
} cpx;

typedef struct {
  cpx dft_wave[N];     /* exp(-2*pi*i*k/32), DFTInit */
  cpx dct_wave[N2 + 1]; /* 0.25 * exp(-i*pi*k/64), DCTInit */
  int itab[N];          /* bit reversal of 5 bits, DFTInit */
  int init;
} cv_dct32_tabs;

static cv_dct32_tabs g_tabs;

/* DFTTab[m] = { cos(2*pi/2^m), sin(2*pi/2^m) } as doubles (dxt.cpp holds them as 17-digit literals) */
static void dft_tab(int m, double* c, double* s) {
  if (m == 0) {
    *c = 1.0;
    *s = 0.0;
  } else if (m == 1) {
    *c = -1.0;
    *s = 0.0;
  } else if (m == 2) {
    *c = 0.0;
    *s = 1.0;
  } else {
    const double a = 6.283185307179586476925286766559 / (double)(1 << m);
    *c = cos(a);
    *s = sin(a);
  }
}

/* DFTInit (n0 = 32, one factor 32) + DCTInit (n = 32): tables by complex recurrence in double, stored as float */
static void cv_dct32_init(cv_dct32_tabs* t) {
  /* itab: n <= 256 branch, shift = 10 - m on the byte-reversal table; for n = 32 this is the 5-bit reversal */
  for (int i = 0; i < N; ++i) {
    int r = 0;
    for (int b = 0; b < 5; ++b)
      if (i & (1 << b)) r |= 1 << (4 - b);
    t->itab[i] = r;
  }
  {
    double c, s;
    dft_tab(5, &c, &s);
    double w_re = c, w_im = -s;
    const double w1_re = c, w1_im = -s;
    const int n = (N + 1) / 2;
    t->dft_wave[0].re = 1.f;
    t->dft_wave[0].im = 0.f;
    t->dft_wave[n].re = -1.f;
    t->dft_wave[n].im = 0.f;
    for (int i = 1; i < n; ++i) {
      t->dft_wave[i].re = (float)w_re;
      t->dft_wave[i].im = (float)w_im;
      t->dft_wave[N - i].re = (float)w_re;
      t->dft_wave[N - i].im = (float)-w_im;
      const double tt = w_re * w1_re - w_im * w1_im;
      w_im = w_re * w1_im + w_im * w1_re;
      w_re = tt;
    }
  }
  {
    /* m = 5: scale = 2 * DctScale[5] = 2 * 0.125; w1 = conj(DFTTab[7]) = exp(-i*pi/64) */
    double c, s;
    dft_tab(7, &c, &s);
    const double scale = 2 * 0.125;
    const double w1_re = c, w1_im = -s;
    double w_re = (float)scale, w_im = 0.f; /* "w.re = (float)scale" */
    for (int i = 0; i <= N2; ++i) {
      t->dct_wave[i].re = (float)w_re;
      t->dct_wave[i].im = (float)w_im;
      const double tt = w_re * w1_re - w_im * w1_im;
      w_im = w_re * w1_im + w_im * w1_re;
      w_re = tt;
    }
  }
  t->init = 1;
}

static const cv_dct32_tabs* tabs(void) {
  if (!g_tabs.init) cv_dct32_init(&g_tabs);
  return &g_tabs;
}

/* exported for the GPU side's table upload test and for tools: 32 + 17 complex floats */
void orc_cv_dct32_tables(float* dft_wave /*64*/, float* dct_wave /*34*/) {
  const cv_dct32_tabs* t = tabs();
  memcpy(dft_wave, t->dft_wave, sizeof(t->dft_wave));
  memcpy(dct_wave, t->dct_wave, sizeof(t->dct_wave));
}

/* DFT<float>, n = 16, tab_size = 32, forward, in place; factors = {16}: permutation + two radix-4 passes */
static void cv_dft16(cpx* dst, const cv_dct32_tabs* t) {
  const cpx* wave = t->dft_wave;
  const int n0 = N2;
  {
    /* in-place shuffle (nf == 1, (n & 3) == 0): the net effect is dst[i] <- dst[bitrev4(i)]; itab is read with
       tab_step = 2 (table of 32 for a transform of 16) */
    cpx tmp[N2];
    memcpy(tmp, dst, sizeof(tmp));
    for (int i = 0; i < n0; ++i) dst[i] = tmp[t->itab[2 * i]];
  }
  int n = 1, dw0 = N;
  for (; n * 4 <= N2;) {
    const int nx = n;
    n *= 4;
    dw0 /= 4;
    for (int i = 0; i < n0; i += n) {
      cpx *v0, *v1;
      float r0, i0, r1, i1, r2, i2, r3, i3, r4, i4;
      v0 = dst + i;
      v1 = v0 + nx * 2;

      r0 = v1[0].re; i0 = v1[0].im;
      r4 = v1[nx].re; i4 = v1[nx].im;

      r1 = r0 + r4; i1 = i0 + i4;
      r3 = i0 - i4; i3 = r4 - r0;

      r2 = v0[0].re; i2 = v0[0].im;
      r4 = v0[nx].re; i4 = v0[nx].im;

      r0 = r2 + r4; i0 = i2 + i4;
      r2 -= r4; i2 -= i4;

      v0[0].re = r0 + r1; v0[0].im = i0 + i1;
      v1[0].re = r0 - r1; v1[0].im = i0 - i1;
      v0[nx].re = r2 + r3; v0[nx].im = i2 + i3;
      v1[nx].re = r2 - r3; v1[nx].im = i2 - i3;

      for (int j = 1, dw = dw0; j < nx; j++, dw += dw0) {
        v0 = dst + i + j;
        v1 = v0 + nx * 2;

        r2 = v0[nx].re * wave[dw * 2].re - v0[nx].im * wave[dw * 2].im;
        i2 = v0[nx].re * wave[dw * 2].im + v0[nx].im * wave[dw * 2].re;
        r0 = v1[0].re * wave[dw].im + v1[0].im * wave[dw].re;
        i0 = v1[0].re * wave[dw].re - v1[0].im * wave[dw].im;
        r3 = v1[nx].re * wave[dw * 3].im + v1[nx].im * wave[dw * 3].re;
        i3 = v1[nx].re * wave[dw * 3].re - v1[nx].im * wave[dw * 3].im;

        r1 = i0 + i3; i1 = r0 + r3;
        r3 = r0 - r3; i3 = i3 - i0;
        r4 = v0[0].re; i4 = v0[0].im;

        r0 = r4 + r2; i0 = i4 + i2;
        r2 = r4 - r2; i2 = i4 - i2;

        v0[0].re = r0 + r1; v0[0].im = i0 + i1;
        v1[0].re = r0 - r1; v1[0].im = i0 - i1;
        v0[nx].re = r2 + r3; v0[nx].im = i2 + i3;
        v1[nx].re = r2 - r3; v1[nx].im = i2 - i3;
      }
    }
  }
}

/* RealDFT<float>, n = 32, forward, real output packed, scale = 1, src == dst (in-place transform) */
static void cv_realdft32(float* dst, const cv_dct32_tabs* tb) {
  const int n = N, n2 = N2;
  const float scale = 1.f;
  const float scale2 = scale * 0.5f;
  const cpx* wave = tb->dft_wave;
  float t0, t, h1_re, h1_im, h2_re, h2_im;
  int j;

  cv_dft16((cpx*)dst, tb);

  t = dst[0] - dst[1];
  dst[0] = (dst[0] + dst[1]) * scale;
  dst[1] = t * scale;

  t0 = dst[n2];
  t = dst[n - 1];
  dst[n - 1] = dst[1];

  for (j = 2, wave++; j < n2; j += 2, wave++) {
    /* calc odd */
    h2_re = scale2 * (dst[j + 1] + t);
    h2_im = scale2 * (dst[n - j] - dst[j]);

    /* calc even */
    h1_re = scale2 * (dst[j] + dst[n - j]);
    h1_im = scale2 * (dst[j + 1] - t);

    /* rotate */
    t = h2_re * wave->re - h2_im * wave->im;
    h2_im = h2_re * wave->im + h2_im * wave->re;
    h2_re = t;
    t = dst[n - j - 1];

    dst[j - 1] = h1_re + h2_re;
    dst[n - j - 1] = h1_re - h2_re;
    dst[j] = h1_im + h2_im;
    dst[n - j] = h2_im - h1_im;
  }

  if (j <= n2) {
    dst[n2 - 1] = t0 * scale;
    dst[n2] = -t * scale;
  }
}

/* DCT_32f, n = 32: src/dst with element strides */
static void cv_dct32_1d(const float* src, int src_step, float* dst, int dst_step, const cv_dct32_tabs* tb) {
  static const float sin_45 = (float)0.70710678118654752440084436210485;
  const int n = N, n2 = N2;
  float buf[N]; /* dft_src == dft_dst: inplace_transform */
  const cpx* dct_wave = tb->dct_wave;
  float* dst1 = dst + (n - 1) * dst_step;
  int j;

  for (j = 0; j < n2; j++, src += src_step * 2) {
    buf[j] = src[0];
    buf[n - j - 1] = src[src_step];
  }

  cv_realdft32(buf, tb);
  const float* s = buf;

  dst[0] = (float)(s[0] * dct_wave->re * sin_45);
  dst += dst_step;
  for (j = 1, dct_wave++; j < n2; j++, dct_wave++, dst += dst_step, dst1 -= dst_step) {
    float t0 = dct_wave->re * s[j * 2 - 1] - dct_wave->im * s[j * 2];
    float t1 = -dct_wave->im * s[j * 2 - 1] - dct_wave->re * s[j * 2];
    dst[0] = t0;
    dst1[0] = t1;
  }

  dst[0] = s[n - 1] * dct_wave->re;
}

/* cv::dct on a continuous 32x32 CV_32F matrix, in place: rows (stage 0), then columns (stage 1) */
void orc_cv_dct32x32(float* m /* [32*32] */) {
  const cv_dct32_tabs* tb = tabs();
  for (int i = 0; i < N; ++i) cv_dct32_1d(m + i * N, 1, m + i * N, 1, tb);
  for (int i = 0; i < N; ++i) {
    /* the column is gathered into dft_src before anything is written, so in place is safe */
    float col[N];
    cv_dct32_1d(m + i, N, col, 1, tb);
    for (int r = 0; r < N; ++r) m[r * N + i] = col[r];
  }
}

/* one 1-D transform (tests: against a float64 DCT-II) */
void orc_cv_dct32_1d(const float* in, float* out) { cv_dct32_1d(in, 1, out, 1, tabs()); }

/* cv::sum of a continuous 1 x len CV_32F row: groups of four summed in float, accumulated in double */
double orc_cv_sum_f32(const float* src, int len) {
  double s0 = 0;
  int i = 0;
  for (; i <= len - 4; i += 4, src += 4) s0 += src[0] + src[1] + src[2] + src[3];
  for (; i < len; i++, src += 1) s0 += src[0];
  return s0;
}
