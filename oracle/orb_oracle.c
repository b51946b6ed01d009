/* oracle/orb_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker; never linked into the product).
 *
 * CPU restatement of SURVEY.md section 8 row a11: what cbird's
 *     Media::makeKeyPoints            /root/reference/src/media.cpp:859-866
 *         cv::OrbFeatureDetector(numKeyPoints, 1.2f, 12, 31, 0, 2, HARRIS_SCORE, 31).detect(gray)
 *     Media::makeKeyPointDescriptors  /root/reference/src/media.cpp:868-872
 *         cv::OrbDescriptorExtractor().compute(gray, keyPoints, descriptors)      (256-bit rBRIEF, WTA_K 2)
 * compute.  The algorithm lives in OpenCV 2.4.13.7 (pinned by /root/reference/cbird.pri:148-152:
 * modules/features2d/src/orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp; imgproc's resize / GaussianBlur;
 * core's fastAtan2), which is NOT vendored in the reference and not installed here, so what follows restates
 * the published algorithm of those files AS RECALLED.
 *
 *                      ***  PARITY UNPINNED versus the cbird binary  ***
 *
 * Two things can never be pinned from here and are stated where they occur:
 *   (1) rBRIEF's 256 test pairs are a LEARNED table (`bit_pattern_31_`, 1024 integers inside orb.cpp).  It cannot
 *       be derived; it is an INPUT here (orc_orb_set_pattern), exactly as in the product (cbh_orb_set_pattern).
 *       tools/orb_pattern_from_opencv.py extracts it from an OpenCV source tree a maintainer has.
 *   (2) KeyPointsFilter::retainBest uses std::nth_element + std::partition: which of several keypoints with EQUAL
 *       response survive, and the ORDER of the survivors, depend on the C++ library that built cbird.  Two orders are
 *       offered here and in the product (orc_orb_set_retain_order / cbh_set_tuning("orb_retain_order")):
 *         1 (default)  libstdc++'s, the library of cbird's Linux builds: oracle/retain_stl.cpp runs retainBest's few
 *                      lines on this image's real std::nth_element / std::partition, so the ORDER is pinned on the
 *                      library (retainBest's own lines remain recalled);
 *         0            canonical: every keypoint whose response is >= the n-th best response is kept, in raster order
 *                      (row, then column) within a pyramid level -- a superset of what any library leaves.
 *
 * Everything else is integer or strictly-ordered float arithmetic (no FMA: build with -ffp-contract=off), so the HIP
 * path is compared BIT FOR BIT with this file. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORB_NLEVELS 12        /* media.cpp:861 */
#define ORB_EDGE 31           /* edgeThreshold */
#define ORB_PATCH 31          /* patchSize */
#define ORB_HALF_PATCH 15
#define ORB_FAST_T 20         /* orb.cpp: FastFeatureDetector fd(20, true) */
#define ORB_HARRIS_BLOCK 7    /* orb.cpp: HarrisResponses(..., 7, HARRIS_K) */
#define ORB_HARRIS_K 0.04f

typedef struct {
  float x, y, size, angle, response;
  int octave;
} orc_keypoint; /* cv::KeyPoint without class_id */

static int cv_round_d(double v) { return (int)lrint(v); } /* cvRound: round half to even (SSE2 cvtsd2si) */
static int cv_floor_f(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}

/* ---- scales and level sizes: orb.cpp getScale(), ORB::operator() pyramid loop ------------------------------------ */
static float orb_get_scale(int level) { return (float)pow((double)1.2f, (double)level); } /* scaleFactor is a double
                                                                                             member holding 1.2f */
void orc_orb_level_size(int w, int h, int level, int* lw, int* lh) {
  float scale = 1 / orb_get_scale(level);
  *lw = cv_round_d((double)(w * scale));
  *lh = cv_round_d((double)(h * scale));
}
float orc_orb_scale(int level) { return orb_get_scale(level); }

/* features per level: orb.cpp computeKeyPoints() */
void orc_orb_features_per_level(int nfeatures, int* out /* ORB_NLEVELS */) {
  float factor = (float)(1.0 / (double)1.2f);
  float ndesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)ORB_NLEVELS));
  int sum = 0;
  for (int level = 0; level < ORB_NLEVELS - 1; ++level) {
    out[level] = cv_round_d((double)ndesired);
    sum += out[level];
    ndesired *= factor;
  }
  out[ORB_NLEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* ---- cv::resize, INTER_LINEAR, CV_8UC1 (imgproc/src/imgwarp.cpp: resize() coefficient loop, HResizeLinear,
 *      VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>) ----------------------------------------------------
 *   fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, 0); sx >= sw-1 -> (sw-1, 0)
 *   coefficients saturate_cast<short>(c * 2048)   (cvRound)
 *   rows: sy and sy+1 clipped to the image;  dst = (((b0*(D0>>4))>>16) + ((b1*(D1>>4))>>16) + 2) >> 2 */
static short sat_short(float v) {
  long r = lrintf(v);
  return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}
void orc_resize_linear_coeffs(int ssize, int dsize, int* ofs, short* c0, short* c1) {
  double inv_scale = (double)dsize / ssize;
  double scale = 1. / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_f(f);
    f -= s;
    if (s < 0) f = 0.f, s = 0;
    if (s >= ssize - 1) f = 0.f, s = ssize - 1;
    ofs[d] = s;
    c0[d] = sat_short((1.f - f) * 2048.f);
    c1[d] = sat_short(f * 2048.f);
  }
}
/* The vertical pass keeps the library's form exactly: no clamping of the coefficient for y (only x has the
 * xmin / xmax handling); the invoker clips BOTH source rows to [0, h-1], and
 * ((b0*(D>>4))>>16) + ((b1*(D>>4))>>16) is not ((2048*(D>>4))>>16) when both shifts truncate. */
static void resize_linear_ycoeffs(int ssize, int dsize, int* ofs, short* c0, short* c1) {
  double inv_scale = (double)dsize / ssize;
  double scale = 1. / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_f(f);
    f -= s;
    ofs[d] = s;
    c0[d] = sat_short((1.f - f) * 2048.f);
    c1[d] = sat_short(f * 2048.f);
  }
}
static int clipi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
int orc_resize_linear_u8_cv(const uint8_t* src, int w, int h, size_t stride, int dw, int dh, uint8_t* dst) {
  if (w < 1 || h < 1 || dw < 1 || dh < 1) return -1;
  int* xo = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
  short* xc = (short*)malloc(sizeof(short) * 2 * (size_t)(dw + dh));
  int* yo = xo + dw;
  short *xc0 = xc, *xc1 = xc + dw, *yc0 = xc + 2 * dw, *yc1 = xc + 2 * dw + dh;
  orc_resize_linear_coeffs(w, dw, xo, xc0, xc1);
  resize_linear_ycoeffs(h, dh, yo, yc0, yc1);
  for (int dy = 0; dy < dh; ++dy) {
    int sy0 = clipi(yo[dy], 0, h - 1), sy1 = clipi(yo[dy] + 1, 0, h - 1);
    const uint8_t* S0 = src + (size_t)sy0 * stride;
    const uint8_t* S1 = src + (size_t)sy1 * stride;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xo[dx], sx1 = sx + 1 < w ? sx + 1 : w - 1;
      int D0 = S0[sx] * xc0[dx] + S0[sx1] * xc1[dx];
      int D1 = S1[sx] * xc0[dx] + S1[sx1] * xc1[dx];
      int v = (((yc0[dy] * (D0 >> 4)) >> 16) + ((yc1[dy] * (D1 >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)dy * dw + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
  free(xo);
  free(xc);
  return 0;
}

/* ---- FAST-9/16 (features2d/src/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>) ------------------------------ */
static const int kFastOfs[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_corner_score(const uint8_t* ptr, const int* pixel, int threshold) {
  enum { K = 8, N = K * 3 + 1 };
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    a = a < d[k + 3] ? a : d[k + 3];
    if (a <= a0) continue;
    for (int j = 4; j <= 8; ++j) a = a < d[k + j] ? a : d[k + j];
    int t = a < d[k] ? a : d[k];
    a0 = a0 > t ? a0 : t;
    t = a < d[k + 9] ? a : d[k + 9];
    a0 = a0 > t ? a0 : t;
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    b = b > d[k + 3] ? b : d[k + 3];
    for (int j = 4; j <= 5; ++j) b = b > d[k + j] ? b : d[k + j];
    if (b >= b0) continue;
    for (int j = 6; j <= 8; ++j) b = b > d[k + j] ? b : d[k + j];
    int t = b > d[k] ? b : d[k];
    b0 = b0 < t ? b0 : t;
    t = b > d[k + 9] ? b : d[k + 9];
    b0 = b0 < t ? b0 : t;
  }
  return -b0 - 1;
}

/* scores[y*w + x] = corner score after the 3x3 non-maximum suppression (0 = no keypoint); the keypoints OpenCV emits
 * are the non-zero entries in raster order, response = the score */
void orc_fast_nms_scores(const uint8_t* img, int w, int h, size_t stride, uint8_t* scores) {
  memset(scores, 0, (size_t)w * h);
  if (w < 7 || h < 7) return;
  int pixel[25];
  for (int k = 0; k < 16; ++k) pixel[k] = kFastOfs[k][0] + kFastOfs[k][1] * (int)stride;
  for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
  uint8_t* raw = (uint8_t*)calloc((size_t)w * h, 1);
  const int threshold = ORB_FAST_T, K = 8, N = 25;
  for (int i = 3; i < h - 3; ++i) {
    for (int j = 3; j < w - 3; ++j) {
      const uint8_t* ptr = img + (size_t)i * stride + j;
      int v = ptr[0], found = 0;
      {
        int vt = v - threshold, count = 0;
        for (int k = 0; k < N; ++k) {
          int x = ptr[pixel[k]];
          if (x < vt) {
            if (++count > K) {
              found = 1;
              break;
            }
          } else
            count = 0;
        }
      }
      if (!found) {
        int vt = v + threshold, count = 0;
        for (int k = 0; k < N; ++k) {
          int x = ptr[pixel[k]];
          if (x > vt) {
            if (++count > K) {
              found = 1;
              break;
            }
          } else
            count = 0;
        }
      }
      if (found) raw[(size_t)i * w + j] = (uint8_t)fast_corner_score(ptr, pixel, threshold);
    }
  }
  for (int i = 3; i < h - 3; ++i)
    for (int j = 3; j < w - 3; ++j) {
      int s = raw[(size_t)i * w + j];
      if (!s) continue; /* not a corner (a corner's score is >= threshold = 20) */
      const uint8_t *p = raw + (size_t)(i - 1) * w + j, *c = raw + (size_t)i * w + j, *n = raw + (size_t)(i + 1) * w + j;
      if (s > c[1] && s > c[-1] && s > p[-1] && s > p[0] && s > p[1] && s > n[-1] && s > n[0] && s > n[1])
        scores[(size_t)i * w + j] = (uint8_t)s;
    }
  free(raw);
}

/* ---- HarrisResponses (orb.cpp) ------------------------------------------------------------------------------------ */
float orc_harris_response(const uint8_t* img, size_t stride, int x, int y) {
  const int blockSize = ORB_HARRIS_BLOCK, r = blockSize / 2;
  float scale = (1 << 2) * blockSize * 255.0f;
  scale = 1.0f / scale;
  float scale_sq_sq = scale * scale * scale * scale;
  const int step = (int)stride;
  const uint8_t* ptr0 = img + (ptrdiff_t)(y - r) * step + (x - r);
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < blockSize; ++i)
    for (int j = 0; j < blockSize; ++j) {
      const uint8_t* ptr = ptr0 + i * step + j;
      int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
      int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  return ((float)a * b - (float)c * c - ORB_HARRIS_K * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ---- orientation: IC_Angle + the u_max table (orb.cpp), cv::fastAtan2 (core/src/mathfuncs.cpp, 2.4) --------------- */
void orc_orb_umax(int* umax /* ORB_HALF_PATCH + 2 */) {
  const int half = ORB_HALF_PATCH;
  int v, v0, vmax = (int)floor(half * sqrt(2.f) / 2 + 1);
  int vmin = (int)ceil(half * sqrt(2.f) / 2);
  for (v = 0; v <= half + 1; ++v) umax[v] = 0;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)half * half - v * v));
  for (v = half, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

float orc_fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

float orc_ic_angle(const uint8_t* img, size_t stride, int x, int y) {
  int umax[ORB_HALF_PATCH + 2];
  orc_orb_umax(umax);
  const int half_k = ORB_HALF_PATCH, step = (int)stride;
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = img + (ptrdiff_t)y * step + x;
  for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
  for (int v = 1; v <= half_k; ++v) {
    int v_sum = 0, d = umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---- GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8U: getGaussianKernel(7, 2, CV_32F) -> fixed point
 *      (createSeparableLinearFilter: both kernels smooth+symmetrical, 8-bit in and out => bits = 8 per pass,
 *      row pass int sums, column pass (sum + (1 << 15)) >> 16, saturated) ------------------------------------------- */
void orc_gauss7_kernel(int* k /* 7 */) {
  const int n = 7;
  const double sigma = 2.0;
  float cf[7];
  double scale2X = -0.5 / (sigma * sigma), sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    double t = exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = (int)lrintf(cf[i] * 256.f);
  }
}
static int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
void orc_gauss7_blur_u8(const uint8_t* src, int w, int h, size_t stride, uint8_t* dst /* w*h */) {
  int k[7];
  orc_gauss7_kernel(k);
  int* rows = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int t = -3; t <= 3; ++t) s += k[t + 3] * src[(size_t)y * stride + reflect101(x + t, w)];
      rows[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int t = -3; t <= 3; ++t) s += k[t + 3] * rows[(size_t)reflect101(y + t, h) * w + x];
      int v = (s + (1 << 15)) >> 16;
      dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(rows);
}

/* ---- rBRIEF (orb.cpp computeOrbDescriptor, WTA_K == 2) ------------------------------------------------------------ */
static int8_t g_pattern[1024];
static int g_have_pattern = 0;
int orc_orb_set_pattern(const int8_t* xy /* 256 x (x0, y0, x1, y1) */) {
  for (int i = 0; i < 1024; ++i)
    if (xy[i] < -15 || xy[i] > 15) return -1;
  memcpy(g_pattern, xy, 1024);
  g_have_pattern = 1;
  return 0;
}
void orc_orb_descriptor(const uint8_t* blurred, size_t stride, int cx, int cy, float angle_deg, uint8_t* desc /* 32 */) {
  float angle = angle_deg;
  angle *= (float)(3.14159265358979323846 / 180.f);
  float a = (float)cos(angle), b = (float)sin(angle);
  const uint8_t* center = blurred + (ptrdiff_t)cy * (int)stride + cx;
  const int step = (int)stride;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int k = 0; k < 8; ++k) {
      const int8_t* p = g_pattern + (size_t)(8 * i + k) * 4;
      int t[2];
      for (int e = 0; e < 2; ++e) {
        int px = p[2 * e], py = p[2 * e + 1];
        int iy = cv_round_d((double)(px * b + py * a));
        int ix = cv_round_d((double)(px * a - py * b));
        t[e] = center[iy * step + ix];
      }
      val |= (t[0] < t[1]) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ---- the detector: ORB::operator()(image, noArray(), keypoints, noArray(), false) = makeKeyPoints ------------------ */
typedef struct {
  int w, h;
  uint8_t* px;
} orb_level;

static int build_pyramid(const uint8_t* img, int w, int h, size_t stride, int nlevels, orb_level* lv) {
  int built = 0;
  for (int l = 0; l < nlevels; ++l) {
    int lw, lh;
    orc_orb_level_size(w, h, l, &lw, &lh);
    if (lw < 1 || lh < 1) break;
    lv[l].w = lw, lv[l].h = lh;
    lv[l].px = (uint8_t*)malloc((size_t)lw * lh);
    if (l == 0)
      for (int y = 0; y < h; ++y) memcpy(lv[0].px + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
    else
      orc_resize_linear_u8_cv(lv[l - 1].px, lv[l - 1].w, lv[l - 1].h, (size_t)lv[l - 1].w, lw, lh, lv[l].px);
    built = l + 1;
  }
  return built;
}
static void free_pyramid(orb_level* lv, int n) {
  for (int l = 0; l < n; ++l) free(lv[l].px);
}

static int cmp_float_desc(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return x > y ? -1 : x < y ? 1 : 0;
}
/* KeyPointsFilter::retainBest, canonical form (header, (2)): keep[i] = response[i] >= n-th best */
static long retain_best(const float* resp, long cnt, int n, uint8_t* keep) {
  if (n < 0 || cnt <= n) {
    memset(keep, 1, (size_t)cnt);
    return cnt;
  }
  if (n == 0) {
    memset(keep, 0, (size_t)cnt);
    return 0;
  }
  float* s = (float*)malloc(sizeof(float) * (size_t)cnt);
  memcpy(s, resp, sizeof(float) * (size_t)cnt);
  qsort(s, (size_t)cnt, sizeof(float), cmp_float_desc);
  float thr = s[n - 1];
  free(s);
  long kept = 0;
  for (long i = 0; i < cnt; ++i) kept += (keep[i] = resp[i] >= thr);
  return kept;
}

/* retainBest's order.  1 (default): what OpenCV 2.4's retainBest leaves when the C++ library is libstdc++ --
 * oracle/retain_stl.cpp, the real std::nth_element and std::partition.  0: canonical -- every tie kept, raster order
 * (header, (2)). */
long orc_retain_best_stl(const float* resp, long cnt, int n_points, int depth_limit, int32_t* order);
static int g_retain_order = 1;
void orc_orb_set_retain_order(int mode) { g_retain_order = mode == 1; }

/* returns the number of keypoints (all of them are counted; at most cap are written) or < 0 */
long orc_orb_detect(const uint8_t* img, int w, int h, size_t stride, int nfeatures, orc_keypoint* out, long cap) {
  if (w < 1 || h < 1 || nfeatures < 0) return -1;
  int nper[ORB_NLEVELS];
  orc_orb_features_per_level(nfeatures, nper);
  orb_level lv[ORB_NLEVELS];
  int nl = build_pyramid(img, w, h, stride, ORB_NLEVELS, lv);
  long total = 0;
  for (int l = 0; l < nl; ++l) {
    const int lw = lv[l].w, lh = lv[l].h;
    if (lw <= 2 * ORB_EDGE || lh <= 2 * ORB_EDGE) continue; /* runByImageBorder clears the list */
    uint8_t* sc = (uint8_t*)malloc((size_t)lw * lh);
    orc_fast_nms_scores(lv[l].px, lw, lh, (size_t)lw, sc);
    /* runByImageBorder(edgeThreshold): Rect(31, 31, w-62, h-62).contains(pt) */
    long cnt = 0;
    for (int y = ORB_EDGE; y < lh - ORB_EDGE; ++y)
      for (int x = ORB_EDGE; x < lw - ORB_EDGE; ++x) cnt += sc[(size_t)y * lw + x] != 0;
    int* xs = (int*)malloc(sizeof(int) * 2 * (size_t)(cnt + 1));
    int* ys = xs + cnt + 1;
    float* resp = (float*)malloc(sizeof(float) * (size_t)(cnt + 1));
    uint8_t* keep = (uint8_t*)malloc((size_t)cnt + 1);
    long c = 0;
    for (int y = ORB_EDGE; y < lh - ORB_EDGE; ++y)
      for (int x = ORB_EDGE; x < lw - ORB_EDGE; ++x)
        if (sc[(size_t)y * lw + x]) xs[c] = x, ys[c] = y, resp[c] = (float)sc[(size_t)y * lw + x], ++c;
    if (g_retain_order == 1) { /* the same three steps, survivors in the order the library leaves them */
      int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(cnt + 1));
      int32_t* ord2 = ord + cnt + 1;
      const long k1 = orc_retain_best_stl(resp, cnt, 2 * nper[l], -1, ord);
      float* r2 = (float*)malloc(sizeof(float) * (size_t)(k1 + 1));
      for (long i = 0; i < k1; ++i) r2[i] = orc_harris_response(lv[l].px, (size_t)lw, xs[ord[i]], ys[ord[i]]);
      const long k2 = orc_retain_best_stl(r2, k1, nper[l], -1, ord2);
      const float sf1 = orb_get_scale(l);
      for (long i = 0; i < k2; ++i, ++total) {
        if (total >= cap) continue;
        const int x = xs[ord[ord2[i]]], y = ys[ord[ord2[i]]];
        orc_keypoint* k = out + total;
        k->octave = l;
        k->size = ORB_PATCH * sf1;
        k->response = r2[ord2[i]];
        k->angle = orc_ic_angle(lv[l].px, (size_t)lw, x, y);
        k->x = (float)x, k->y = (float)y;
        if (l != 0) k->x *= sf1, k->y *= sf1;
      }
      free(ord), free(r2), free(sc), free(xs), free(resp), free(keep);
      continue;
    }
    /* retainBest(2 * featuresNum) on the FAST score, HarrisResponses, retainBest(featuresNum) */
    retain_best(resp, cnt, 2 * nper[l], keep);
    long c2 = 0;
    for (long i = 0; i < cnt; ++i)
      if (keep[i]) {
        xs[c2] = xs[i], ys[c2] = ys[i];
        resp[c2] = orc_harris_response(lv[l].px, (size_t)lw, xs[i], ys[i]);
        ++c2;
      }
    retain_best(resp, c2, nper[l], keep);
    const float sf = orb_get_scale(l);
    for (long i = 0; i < c2; ++i)
      if (keep[i]) {
        if (total < cap) {
          orc_keypoint* k = out + total;
          k->octave = l;
          k->size = ORB_PATCH * sf;
          k->response = resp[i];
          k->angle = orc_ic_angle(lv[l].px, (size_t)lw, xs[i], ys[i]);
          k->x = (float)xs[i], k->y = (float)ys[i];
          if (l != 0) k->x *= sf, k->y *= sf; /* ORB::operator(): keypoint->pt *= scale */
        }
        ++total;
      }
    free(sc), free(xs), free(resp), free(keep);
  }
  free_pyramid(lv, nl);
  return total;
}

/* the extractor: ORB::operator()(image, Mat(), keypoints, descriptors, true) = makeKeyPointDescriptors.
 * kps is in/out like the reference's non-const KeyPointList&: keypoints too close to the border are REMOVED and the
 * survivors come back grouped by octave with pt = (pt * (1/scale)) * scale.  Returns the number kept (= descriptor
 * rows), or < 0 (no pattern set: -2). */
long orc_orb_compute(const uint8_t* img, int w, int h, size_t stride, orc_keypoint* kps, long nkp, uint8_t* desc) {
  if (!g_have_pattern) return -2;
  if (w < 1 || h < 1) return -1;
  /* DescriptorExtractor::compute: runByImageBorder(keypoints, size, 0) -- and ORB's own with edgeThreshold */
  long n = 0;
  int levels = 0;
  for (long i = 0; i < nkp; ++i) {
    /* Rect::contains(Point2f -> Point via cvRound) */
    int px = cv_round_d((double)kps[i].x), py = cv_round_d((double)kps[i].y);
    if (w <= 2 * ORB_EDGE || h <= 2 * ORB_EDGE) break;
    if (px >= ORB_EDGE && px < w - ORB_EDGE && py >= ORB_EDGE && py < h - ORB_EDGE) kps[n++] = kps[i];
  }
  for (long i = 0; i < n; ++i) {
    int o = kps[i].octave > 0 ? kps[i].octave : 0;
    if (o + 1 > levels) levels = o + 1;
  }
  if (n == 0) return 0;
  if (levels > ORB_NLEVELS) return -1;
  orb_level lv[ORB_NLEVELS];
  int nl = build_pyramid(img, w, h, stride, levels, lv);
  if (nl < levels) {
    free_pyramid(lv, nl);
    return -1;
  }
  orc_keypoint* tmp = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)n);
  long o = 0;
  for (int l = 0; l < levels; ++l) {
    long first = o;
    for (long i = 0; i < n; ++i)
      if (kps[i].octave == l) tmp[o++] = kps[i];
    if (o == first) continue;
    if (l != 0) {
      float scale = 1 / orb_get_scale(l);
      for (long i = first; i < o; ++i) tmp[i].x *= scale, tmp[i].y *= scale;
    }
    uint8_t* blurred = (uint8_t*)malloc((size_t)lv[l].w * lv[l].h);
    orc_gauss7_blur_u8(lv[l].px, lv[l].w, lv[l].h, (size_t)lv[l].w, blurred);
    for (long i = first; i < o; ++i)
      orc_orb_descriptor(blurred, (size_t)lv[l].w, cv_round_d((double)tmp[i].x), cv_round_d((double)tmp[i].y),
                         tmp[i].angle, desc + 32 * (size_t)i);
    free(blurred);
    if (l != 0) {
      float scale = orb_get_scale(l);
      for (long i = first; i < o; ++i) tmp[i].x *= scale, tmp[i].y *= scale;
    }
  }
  /* keypoints with a negative octave would be dropped by the clustering; ORB never produces them */
  memcpy(kps, tmp, sizeof(orc_keypoint) * (size_t)o);
  free(tmp);
  free_pyramid(lv, nl);
  return o;
}

/* stage exports for the tests */
int orc_orb_pyramid_level(const uint8_t* img, int w, int h, size_t stride, int level, uint8_t* dst /* lw*lh */) {
  orb_level lv[ORB_NLEVELS];
  if (level < 0 || level >= ORB_NLEVELS) return -1;
  int nl = build_pyramid(img, w, h, stride, level + 1, lv);
  int rc = -1;
  if (nl == level + 1) {
    memcpy(dst, lv[level].px, (size_t)lv[level].w * lv[level].h);
    rc = 0;
  }
  free_pyramid(lv, nl);
  return rc;
}
