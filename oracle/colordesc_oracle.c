/* oracle/colordesc_oracle.c -- TEST INFRASTRUCTURE ONLY (parity checker; never linked into the product).
 *
 * CPU restatement of SURVEY.md section 8 row a14: ColorDescriptor::create, /root/reference/src/cvutil.cpp:790-1099
 * (fully visible reference code) over the OpenCV 2.4.13.7 calls it makes, restated AS RECALLED:
 *     sizeLongestSide(rgb, 256, INTER_NEAREST)      cvutil.cpp:1932-1949, imgproc resizeNN
 *     cv::ellipse(mask, RotatedRect, 255, CV_FILLED) core/src/drawing.cpp: ellipse2Poly (SinTable), FillConvexPoly, Line2
 *     convertTo(CV_32F); luv *= 1/255; cvtColor(CV_BGR2Luv)   imgproc/src/color.cpp RGB2Luv_f (sRGB gamma + cube-root
 *                                                   tables as cubic splines, cvCbrt)
 *     cv::kmeans(samples, 32, TermCriteria(ITER|EPS, 100, 10), 1, KMEANS_PP_CENTERS)   core/src/matrix.cpp, cv::RNG
 *
 *                      ***  PARITY UNPINNED versus the cbird binary  ***
 *
 * and NOT PINNABLE in three places, by the reference's own construction:
 *   (1) kmeans draws from cv::theRNG(), a per-thread generator that is never reseeded: the descriptor of an image depends
 *       on how many images the worker thread has clustered before ("FIXME: there seems to be some randomness in the
 *       descriptor with identical input", cvutil.cpp:791).  Here every image starts from the state a fresh thread has
 *       (RNG() : state = 0xffffffff), i.e. the result the reference gives for the FIRST image a worker sees.
 *   (2) colours of equal frequency are ordered by std::sort over QHash::keys() (unspecified, hash-seed dependent):
 *       here by descending frequency, then ascending key.
 *   (3) what cv::ellipse paints at the rim is OpenCV's fixed-point polygon fill, restated from memory.
 * Float and double arithmetic is evaluated strictly left to right without FMA (build with -ffp-contract=off); every
 * accumulation runs in the reference's order (sample order), which is what the HIP path reproduces bit for bit. */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CD_K 32 /* ColorDescriptor::NUM_DESC_COLORS */

static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_d(double v) {
  int i = (int)v;
  return i - (v < i);
}

/* ---- sizeLongestSide(rgb, 256, INTER_NEAREST) ----------------------------------------------------------------------- */
void orc_cd_resized_dims(int w, int h, int* ow, int* oh) {
  *ow = w, *oh = h;
  if (h > 256 || w > 256) {
    float aspect = (float)w / h;
    if (w > h) {
      *ow = 256;
      *oh = (int)(256 / aspect);
    } else {
      *oh = 256;
      *ow = (int)(aspect * 256);
    }
  }
}
static void resize_nn_bgr(const uint8_t* src, int w, int h, size_t stride, int ch, int dw, int dh, uint8_t* dst /* 3 ch */) {
  double ifx = 1. / ((double)dw / w), ify = 1. / ((double)dh / h);
  for (int y = 0; y < dh; ++y) {
    int sy = cv_floor_d(y * ify);
    if (sy > h - 1) sy = h - 1;
    for (int x = 0; x < dw; ++x) {
      int sx = cv_floor_d(x * ifx);
      if (sx > w - 1) sx = w - 1;
      const uint8_t* p = src + (size_t)sy * stride + (size_t)sx * ch;
      uint8_t* q = dst + ((size_t)y * dw + x) * 3;
      q[0] = p[0], q[1] = p[1], q[2] = p[2]; /* BGRA -> BGR drops the fourth byte */
    }
  }
}

/* ---- cv::ellipse(mask, RotatedRect((c/2, r/2), (0.9c, 0.9r), 0), 255, CV_FILLED) ------------------------------------ */
#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)
typedef struct {
  int x, y;
} pt_t;

static float sin_table(int deg) { /* drawing.cpp SinTable[]: sin(deg) printed with seven decimals */
  char buf[32];
  snprintf(buf, sizeof buf, "%.7f", sin(deg * 3.14159265358979323846 / 180.0));
  return strtof(buf, NULL);
}
static void put_point(uint8_t* img, int w, int h, int x, int y) {
  if (0 <= x && x < w && 0 <= y && y < h) img[(size_t)y * w + x] = 255;
}
/* Line2: fixed-point 8-connected line between points given with XY_SHIFT fractional bits (no clipping needed: the
 * ellipse lies inside the image) */
static void line2(uint8_t* img, int w, int h, pt_t pt1, pt_t pt2) {
  int dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
  int j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
  int i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
  int x_step, y_step, ecount;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    pt1.x ^= pt2.x & j, pt2.x ^= pt1.x & j, pt1.x ^= pt2.x & j;
    pt1.y ^= pt2.y & j, pt2.y ^= pt1.y & j, pt1.y ^= pt2.y & j;
    x_step = XY_ONE;
    y_step = (int)(((int64_t)dy << XY_SHIFT) / (ax | 1));
    ecount = (pt2.x - pt1.x) >> XY_SHIFT;
  } else {
    dx = (dx ^ i) - i;
    pt1.x ^= pt2.x & i, pt2.x ^= pt1.x & i, pt1.x ^= pt2.x & i;
    pt1.y ^= pt2.y & i, pt2.y ^= pt1.y & i, pt1.y ^= pt2.y & i;
    x_step = (int)(((int64_t)dx << XY_SHIFT) / (ay | 1));
    y_step = XY_ONE;
    ecount = (pt2.y - pt1.y) >> XY_SHIFT;
  }
  pt1.x += (XY_ONE >> 1);
  pt1.y += (XY_ONE >> 1);
  put_point(img, w, h, (pt2.x + (XY_ONE >> 1)) >> XY_SHIFT, (pt2.y + (XY_ONE >> 1)) >> XY_SHIFT);
  if (ax > ay) {
    pt1.x >>= XY_SHIFT;
    while (ecount >= 0) {
      put_point(img, w, h, pt1.x, pt1.y >> XY_SHIFT);
      pt1.x++;
      pt1.y += y_step;
      ecount--;
    }
  } else {
    pt1.y >>= XY_SHIFT;
    while (ecount >= 0) {
      put_point(img, w, h, pt1.x >> XY_SHIFT, pt1.y);
      pt1.x += x_step;
      pt1.y++;
      ecount--;
    }
  }
  (void)x_step;
}
static void fill_convex_poly(uint8_t* img, int w, int h, const pt_t* v, int npts) {
  struct {
    int idx, di, x, dx, ye;
  } edge[2];
  const int shift = XY_SHIFT, delta = 1 << (shift - 1);
  int i, y, imin = 0, left = 0, right = 1, x1, x2;
  int edges = npts;
  int xmin, xmax, ymin, ymax;
  const int delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
  pt_t p0 = v[npts - 1];
  xmin = xmax = v[0].x;
  ymin = ymax = v[0].y;
  for (i = 0; i < npts; i++) {
    pt_t p = v[i];
    if (p.y < ymin) {
      ymin = p.y;
      imin = i;
    }
    if (p.y > ymax) ymax = p.y;
    if (p.x > xmax) xmax = p.x;
    if (p.x < xmin) xmin = p.x;
    line2(img, w, h, p0, p);
    p0 = p;
  }
  xmin = (xmin + delta) >> shift;
  xmax = (xmax + delta) >> shift;
  ymin = (ymin + delta) >> shift;
  ymax = (ymax + delta) >> shift;
  if (npts < 3 || xmax < 0 || ymax < 0 || xmin >= w || ymin >= h) return;
  if (ymax > h - 1) ymax = h - 1;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y = ymin;
  edge[0].di = 1;
  edge[1].di = npts - 1;
  edge[0].x = edge[1].x = edge[0].dx = edge[1].dx = 0;
  do {
    for (i = 0; i < 2; i++) {
      if (y >= edge[i].ye) {
        int idx = edge[i].idx, di = edge[i].di;
        int xs = 0, xe, ye, ty = 0;
        for (;;) {
          ty = (v[idx].y + delta) >> shift;
          if (ty > y || edges == 0) break;
          xs = v[idx].x;
          idx += di;
          idx -= ((idx < npts) - 1) & npts; /* idx -= idx >= npts ? npts : 0 */
          edges--;
        }
        ye = ty;
        xe = v[idx].x;
        if (y >= ye) return; /* no more edges */
        edge[i].ye = ye;
        edge[i].dx = ((xe - xs) * 2 + (ye - y)) / (2 * (ye - y));
        edge[i].x = xs;
        edge[i].idx = idx;
      }
    }
    if (edge[left].x > edge[right].x) {
      left ^= 1;
      right ^= 1;
    }
    x1 = edge[left].x;
    x2 = edge[right].x;
    if (y >= 0) {
      int xx1 = (x1 + delta1) >> XY_SHIFT;
      int xx2 = (x2 + delta2) >> XY_SHIFT;
      if (xx2 >= 0 && xx1 < w) {
        if (xx1 < 0) xx1 = 0;
        if (xx2 >= w) xx2 = w - 1;
        for (int x = xx1; x <= xx2; ++x) img[(size_t)y * w + x] = 255;
      }
    }
    x1 += edge[left].dx;
    x2 += edge[right].dx;
    edge[left].x = x1;
    edge[right].x = x2;
  } while (++y <= ymax);
}
void orc_cd_ellipse_mask(int cols, int rows, uint8_t* mask /* cols*rows */) {
  memset(mask, 0, (size_t)cols * rows);
  /* RotatedRect({cols*0.5f, rows*0.5f}, {cols*0.9f, rows*0.9f}, 0); ellipse(): fixed point with XY_SHIFT */
  const float cxf = cols * 0.5f, cyf = rows * 0.5f, swf = cols * 0.9f, shf = rows * 0.9f;
  pt_t center = {cv_round_d((double)(cxf * (1 << XY_SHIFT))), cv_round_d((double)(cyf * (1 << XY_SHIFT)))};
  int aw = abs(cv_round_d((double)(swf * (1 << (XY_SHIFT - 1))))), ah = abs(cv_round_d((double)(shf * (1 << (XY_SHIFT - 1)))));
  int delta = ((aw > ah ? aw : ah) + (XY_ONE >> 1)) >> XY_SHIFT;
  delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
  /* ellipse2Poly(center, axes, 0, 0, 360, delta) */
  pt_t pts[80];
  int n = 0;
  const float alpha = sin_table(450 - 0), beta = sin_table(0);
  const double size_a = aw, size_b = ah, cx = center.x, cy = center.y;
  pt_t prev = {INT_MIN, INT_MIN};
  for (int i = 0; i < 360 + delta; i += delta) {
    int angle = i;
    if (angle > 360) angle = 360;
    double x = size_a * sin_table(450 - angle), y = size_b * sin_table(angle);
    pt_t pt = {cv_round_d(cx + x * alpha - y * beta), cv_round_d(cy + x * beta + y * alpha)};
    if (pt.x != prev.x || pt.y != prev.y) {
      pts[n++] = pt;
      prev = pt;
    }
  }
  if (n == 1) pts[n++] = pts[0];
  fill_convex_poly(mask, cols, rows, pts, n);
}

/* ---- cvtColor(CV_BGR2Luv) on CV_32FC3: RGB2Luv_f with the sRGB gamma and cube-root spline tables ------------------- */
#define TAB_SIZE 1024
static float g_gamma_tab[TAB_SIZE * 4], g_cbrt_tab[TAB_SIZE * 4];
static int g_tabs_ready = 0;

float orc_cv_cbrt(float value) { /* cvCbrt, core/src/mathfuncs.cpp */
  float fr;
  union {
    int i;
    float f;
  } v, m;
  int ix, s, ex, shx;
  v.f = value;
  ix = v.i & 0x7fffffff;
  s = v.i & 0x80000000;
  ex = (ix >> 23) - 127;
  shx = ex % 3;
  shx -= shx >= 0 ? 3 : 0;
  ex = (ex - shx) / 3; /* exponent of cube root */
  v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
  fr = v.f;
  /* 0.125 <= fr < 1.0; quartic rational polynomial with error < 2^(-24) */
  fr = (float)(((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                  119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                0.1636161226585754240958355063) /
               ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                  168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0));
  m.f = value;
  v.f = fr;
  v.i = (int)(((unsigned)v.i + ((unsigned)ex << 23) + (unsigned)s) & (((unsigned)m.i << 1) != 0u ? ~0u : 0u));
  return v.f;
}
static void spline_build(const float* f, int n, float* tab) {
  float cn = 0;
  tab[0] = tab[1] = 0.f;
  for (int i = 1; i < n - 1; i++) {
    float t = 3 * (f[i + 1] - 2 * f[i] + f[i - 1]);
    float l = 1 / (4 - tab[(i - 1) * 4]);
    tab[i * 4] = l;
    tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
  }
  for (int i = n - 1; i >= 0; i--) {
    float c = tab[i * 4 + 1] - tab[i * 4] * cn;
    float b = f[i + 1] - f[i] - (cn + c * 2) * (float)0.3333333333333333;
    float d = (cn - c) * (float)0.3333333333333333;
    tab[i * 4] = f[i];
    tab[i * 4 + 1] = b;
    tab[i * 4 + 2] = c;
    tab[i * 4 + 3] = d;
    cn = c;
  }
}
static float spline_interpolate(float x, const float* tab, int n) {
  int ix = cv_floor_d((double)x);
  ix = ix < 0 ? 0 : ix > n - 1 ? n - 1 : ix;
  x -= ix;
  tab += ix * 4;
  return ((tab[3] * x + tab[2]) * x + tab[1]) * x + tab[0];
}
static void init_lab_tabs(void) {
  if (g_tabs_ready) return;
  static float f[TAB_SIZE + 1], g[TAB_SIZE + 1];
  const float LabCbrtTabScale = TAB_SIZE / 1.5f, GammaTabScale = (float)TAB_SIZE;
  float scale = 1.f / LabCbrtTabScale;
  for (int i = 0; i <= TAB_SIZE; i++) {
    float x = i * scale;
    f[i] = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : orc_cv_cbrt(x);
  }
  spline_build(f, TAB_SIZE, g_cbrt_tab);
  scale = 1.f / GammaTabScale;
  for (int i = 0; i <= TAB_SIZE; i++) {
    float x = i * scale;
    g[i] = x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow((double)(x + 0.055) * (1. / 1.055), 2.4);
  }
  spline_build(g, TAB_SIZE, g_gamma_tab);
  g_tabs_ready = 1;
}
void orc_cd_tables(float* gamma_tab, float* cbrt_tab) { /* 4096 floats each (tests) */
  init_lab_tabs();
  memcpy(gamma_tab, g_gamma_tab, sizeof g_gamma_tab);
  memcpy(cbrt_tab, g_cbrt_tab, sizeof g_cbrt_tab);
}
void orc_cd_bgr2luv(float b, float g, float r, float* luv) {
  init_lab_tabs();
  /* sRGB2XYZ_D65 with the R and B columns swapped for blueIdx 0; D65 white point */
  static const float M[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                             0.072169f, 0.019334f, 0.119193f, 0.950227f};
  static const float D65[3] = {0.950456f, 1.f, 1.088754f};
  const float C0 = M[2], C1 = M[1], C2 = M[0], C3 = M[5], C4 = M[4], C5 = M[3], C6 = M[8], C7 = M[7], C8 = M[6];
  float d0 = 1.f / (D65[0] + D65[1] * 15 + D65[2] * 3);
  const float un = 4 * D65[0] * d0, vn = 9 * D65[1] * d0;
  const float _un = 13 * un, _vn = 13 * vn;
  const float gscale = (float)TAB_SIZE, LabCbrtTabScale = TAB_SIZE / 1.5f;
  float R = b, G = g, B = r; /* src[0], src[1], src[2] with the coefficient columns already swapped */
  R = spline_interpolate(R * gscale, g_gamma_tab, TAB_SIZE);
  G = spline_interpolate(G * gscale, g_gamma_tab, TAB_SIZE);
  B = spline_interpolate(B * gscale, g_gamma_tab, TAB_SIZE);
  float X = R * C0 + G * C1 + B * C2;
  float Y = R * C3 + G * C4 + B * C5;
  float Z = R * C6 + G * C7 + B * C8;
  float L = spline_interpolate(Y * LabCbrtTabScale, g_cbrt_tab, TAB_SIZE);
  L = 116.f * L - 16.f;
  float t = X + 15 * Y + 3 * Z;
  float d = (4 * 13) / (t > FLT_EPSILON ? t : FLT_EPSILON);
  luv[0] = L;
  luv[1] = L * (X * d - _un);
  luv[2] = L * ((9 * 0.25f) * Y * d - _vn);
}

/* ---- cv::kmeans(samples, 32, labels, (ITER|EPS, 100, 10), 1, KMEANS_PP_CENTERS, centers) ---------------------------- */
typedef struct {
  uint64_t state;
} cv_rng;
static unsigned rng_next(cv_rng* r) {
  r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
  return (unsigned)r->state;
}
static double rng_double(cv_rng* r) {
  unsigned t = rng_next(r);
  return (double)(((uint64_t)t << 32) | rng_next(r)) * 5.4210108624275221700372640043497e-20;
}
static float norm_l2sqr3(const float* a, const float* b) { /* normL2Sqr_ with n = 3: the scalar tail loop */
  float d = 0.f;
  for (int j = 0; j < 3; j++) {
    float t = a[j] - b[j];
    d += t * t;
  }
  return d;
}
static void generate_centers_pp(const float* data, int N, float* out_centers, int K, cv_rng* rng, int trials) {
  int centers[CD_K];
  float* buf = (float*)malloc(sizeof(float) * 3 * (size_t)N);
  float *dist = buf, *tdist = buf + N, *tdist2 = tdist + N;
  double sum0 = 0;
  centers[0] = (int)(rng_next(rng) % (unsigned)N);
  for (int i = 0; i < N; i++) {
    dist[i] = norm_l2sqr3(data + 3 * (size_t)i, data + 3 * (size_t)centers[0]);
    sum0 += dist[i];
  }
  for (int k = 1; k < K; k++) {
    double bestSum = DBL_MAX;
    int bestCenter = -1;
    for (int j = 0; j < trials; j++) {
      double p = rng_double(rng) * sum0, s = 0;
      int i;
      for (i = 0; i < N - 1; i++)
        if ((p -= dist[i]) <= 0) break;
      int ci = i;
      for (i = 0; i < N; i++) {
        float d = norm_l2sqr3(data + 3 * (size_t)i, data + 3 * (size_t)ci);
        tdist2[i] = d < dist[i] ? d : dist[i]; /* std::min(d, dist[i]) */
        s += tdist2[i];
      }
      if (s < bestSum) {
        bestSum = s;
        bestCenter = ci;
        float* t = tdist;
        tdist = tdist2;
        tdist2 = t;
      }
    }
    centers[k] = bestCenter;
    sum0 = bestSum;
    float* t = dist;
    dist = tdist;
    tdist = t;
  }
  for (int k = 0; k < K; k++)
    for (int j = 0; j < 3; j++) out_centers[3 * k + j] = data[3 * (size_t)centers[k] + j];
  free(buf);
}
/* returns the number of iterations run (>= 1) */
int orc_cd_kmeans(const float* data, int N, int* labels, float* centers_out /* 32*3 */) {
  const int K = CD_K;
  cv_rng rng = {0xffffffffu}; /* RNG(): a fresh thread's state -- header, (1) */
  float cbuf[2][CD_K * 3];
  float *centers = cbuf[0], *old_centers = cbuf[1];
  int counters[CD_K];
  float temp[3];
  memset(cbuf, 0, sizeof cbuf);
  const double epsilon = 10.0 * 10.0; /* criteria.epsilon *= criteria.epsilon */
  const int maxCount = 100;
  double max_center_shift = DBL_MAX;
  int iter;
  for (iter = 0;;) {
    float* t = centers;
    centers = old_centers;
    old_centers = t;
    if (iter == 0) {
      generate_centers_pp(data, N, centers, K, &rng, 3);
    } else {
      for (int k = 0; k < K * 3; k++) centers[k] = 0.f;
      for (int k = 0; k < K; k++) counters[k] = 0;
      for (int i = 0; i < N; i++) {
        const float* sample = data + 3 * (size_t)i;
        int k = labels[i];
        for (int j = 0; j < 3; j++) centers[3 * k + j] += sample[j];
        counters[k]++;
      }
      max_center_shift = 0;
      for (int k = 0; k < K; k++) {
        if (counters[k] != 0) continue;
        /* empty cluster: split the farthest point off the biggest one */
        int max_k = 0;
        for (int k1 = 1; k1 < K; k1++)
          if (counters[max_k] < counters[k1]) max_k = k1;
        double max_dist = 0;
        int farthest_i = -1;
        float* new_center = centers + 3 * k;
        float* old_center = centers + 3 * max_k;
        float scale = 1.f / counters[max_k];
        for (int j = 0; j < 3; j++) temp[j] = old_center[j] * scale;
        for (int i = 0; i < N; i++) {
          if (labels[i] != max_k) continue;
          double dist = norm_l2sqr3(data + 3 * (size_t)i, temp);
          if (max_dist <= dist) {
            max_dist = dist;
            farthest_i = i;
          }
        }
        counters[max_k]--;
        counters[k]++;
        labels[farthest_i] = k;
        const float* sample = data + 3 * (size_t)farthest_i;
        for (int j = 0; j < 3; j++) {
          old_center[j] -= sample[j];
          new_center[j] += sample[j];
        }
      }
      for (int k = 0; k < K; k++) {
        float* center = centers + 3 * k;
        float scale = 1.f / counters[k];
        for (int j = 0; j < 3; j++) center[j] *= scale;
        double dist = 0;
        const float* old_center = old_centers + 3 * k;
        for (int j = 0; j < 3; j++) {
          double tt = center[j] - old_center[j];
          dist += tt * tt;
        }
        if (dist > max_center_shift) max_center_shift = dist;
      }
    }
    if (++iter == (maxCount > 2 ? maxCount : 2) || max_center_shift <= epsilon) break;
    /* assign labels (KMeansDistanceComputer) */
    for (int i = 0; i < N; i++) {
      const float* sample = data + 3 * (size_t)i;
      int k_best = 0;
      double min_dist = DBL_MAX;
      for (int k = 0; k < K; k++) {
        double dist = norm_l2sqr3(sample, centers + 3 * k);
        if (min_dist > dist) {
          min_dist = dist;
          k_best = k;
        }
      }
      labels[i] = k_best;
    }
  }
  memcpy(centers_out, centers, sizeof(float) * CD_K * 3);
  return iter;
}

/* ---- DescriptorColor (src/cvutil.h:57-97) --------------------------------------------------------------------------- */
static int clamp16(int n) {
  n &= -(n >= 0);
  return n | ((65535 - n) >> 31);
}
static void dc_set(float l_, float u_, float v_, uint16_t* l, uint16_t* u, uint16_t* v) {
  *l = (uint16_t)(clamp16((int)(65535 / 100.0f * l_)) & 0xFFFF);
  *u = (uint16_t)(clamp16((int)(65535 / 354.0f * (u_ + 134.0f))) & 0xFFFF);
  *v = (uint16_t)(clamp16((int)(65535 / 262.0f * (v_ + 140.0f))) & 0xFFFF);
}

/* ---- ColorDescriptor::create.  img: 8-bit BGR (channels 3) or BGRA (4).  desc: 258 bytes (32 x {l,u,v,w u16},
 *      numColors u8, pad).  Returns 0, or 1 when the reference returns without touching desc ("not enough colors";
 *      grey input never gets here).  stage (optional, for the tests): dims[2] = size after the resize, n_samples,
 *      iterations. ---------------------------------------------------------------------------------------------------- */
int orc_color_descriptor_create(const uint8_t* img, int w, int h, size_t stride, int channels, uint8_t* desc,
                                int* stage /* 4 ints or NULL */) {
  if ((channels != 3 && channels != 4) || w < 1 || h < 1) return -1;
  int cols, rows;
  orc_cd_resized_dims(w, h, &cols, &rows);
  if (cols < 1 || rows < 1) return -1; /* sizeLongestSide throws */
  uint8_t* rgb = (uint8_t*)malloc((size_t)cols * rows * 3);
  if (cols != w || rows != h) {
    resize_nn_bgr(img, w, h, stride, channels, cols, rows, rgb);
  } else {
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const uint8_t* p = img + (size_t)y * stride + (size_t)x * channels;
        uint8_t* q = rgb + ((size_t)y * w + x) * 3;
        q[0] = p[0], q[1] = p[1], q[2] = p[2];
      }
  }
  uint8_t* mask = (uint8_t*)malloc((size_t)cols * rows);
  orc_cd_ellipse_mask(cols, rows, mask);
  for (size_t i = 0; i < (size_t)cols * rows; ++i) {
    int alpha = mask[i];
    for (int c = 0; c < 3; ++c) rgb[3 * i + c] = (uint8_t)(((int)rgb[3 * i + c] * alpha >> 8) & 0xFF);
  }
  free(mask);
  /* rgb.convertTo(luv, CV_32FC3); luv *= 1.0 / 255.0; cvtColor(luv, luv, CV_BGR2Luv); histFilter = l > 4 */
  const float s255 = (float)(1.0 / 255.0);
  float* samples = (float*)malloc(sizeof(float) * 3 * (size_t)cols * rows);
  uint8_t* filter = (uint8_t*)malloc((size_t)cols * rows);
  int N = 0;
  for (size_t i = 0; i < (size_t)cols * rows; ++i) {
    float luv[3];
    orc_cd_bgr2luv((float)rgb[3 * i] * s255 + 0.f, (float)rgb[3 * i + 1] * s255 + 0.f, (float)rgb[3 * i + 2] * s255 + 0.f, luv);
    if (luv[0] > 4) {
      filter[i] = 1;
      memcpy(samples + 3 * (size_t)N, luv, sizeof luv);
      ++N;
    } else
      filter[i] = 0;
  }
  free(rgb);
  if (stage) stage[0] = cols, stage[1] = rows, stage[2] = N, stage[3] = 0;
  if (N < CD_K) {
    free(samples), free(filter);
    return 1;
  }
  int* labels = (int*)malloc(sizeof(int) * (size_t)N);
  float centers[CD_K * 3];
  int iters = orc_cd_kmeans(samples, N, labels, centers);
  if (stage) stage[3] = iters;
  /* frequency of each quantised centre colour, damped away from the image centre */
  uint64_t keys[CD_K];
  float freq[CD_K];
  int nkeys = 0;
  uint64_t ckey[CD_K];
  for (int k = 0; k < CD_K; ++k) {
    uint16_t l, u, v;
    dc_set(centers[3 * k], centers[3 * k + 1], centers[3 * k + 2], &l, &u, &v);
    ckey[k] = (uint64_t)l << 32 | (uint64_t)u << 16 | (uint64_t)v;
  }
  float maxDistFromCenter;
  {
    float dx = cols / 2.0f, dy = rows / 2.0f;
    maxDistFromCenter = sqrtf(dx * dx + dy * dy);
  }
  int sampleIndex = 0;
  for (int row = 0; row < rows; ++row)
    for (int col = 0; col < cols; ++col) {
      if (!filter[(size_t)row * cols + col]) continue;
      uint64_t key = ckey[labels[sampleIndex++]];
      int dx = col - cols / 2, dy = row - rows / 2;
      float dist = sqrtf((float)(dx * dx + dy * dy));
      int e = 0;
      while (e < nkeys && keys[e] != key) ++e;
      if (e == nkeys) keys[nkeys] = key, freq[nkeys] = 0.f, ++nkeys;
      freq[e] += (maxDistFromCenter - dist) / maxDistFromCenter;
    }
  float maxFreq = 0;
  for (int e = 0; e < nkeys; ++e) maxFreq = freq[e] > maxFreq ? freq[e] : maxFreq;
  /* sort on frequency, descending (ties: ascending key -- header, (2)) */
  int order[CD_K];
  for (int e = 0; e < nkeys; ++e) order[e] = e;
  for (int a = 1; a < nkeys; ++a) {
    int o = order[a], b = a;
    while (b > 0 && (freq[order[b - 1]] < freq[o] || (freq[order[b - 1]] == freq[o] && keys[order[b - 1]] > keys[o]))) {
      order[b] = order[b - 1];
      --b;
    }
    order[b] = o;
  }
  memset(desc, 0, 258);
  for (int di = 0; di < nkeys; ++di) {
    int e = order[di];
    uint16_t l = (uint16_t)((keys[e] >> 32) & 0xFFFF), u = (uint16_t)((keys[e] >> 16) & 0xFFFF), v = (uint16_t)(keys[e] & 0xFFFF);
    uint16_t wv = (uint16_t)((int)(freq[e] * 65535 / maxFreq) & 0xFFFF);
    memcpy(desc + di * 8 + 0, &l, 2);
    memcpy(desc + di * 8 + 2, &u, 2);
    memcpy(desc + di * 8 + 4, &v, 2);
    memcpy(desc + di * 8 + 6, &wv, 2);
    desc[256] = (uint8_t)di; /* desc.numColors = descIndex: the index of the last colour, as the reference writes it */
  }
  free(samples), free(filter), free(labels);
  return 0;
}
