// colordesc_create.hip -- SURVEY.md section 8 row a14: ColorDescriptor::create (/root/reference/src/cvutil.cpp:790-1099)
// for a batch of BGR / BGRA images resident in HBM: nearest resize to <= 256 px, elliptic mask, float BGR -> Luv,
// k-means++ seeding and k-means (K = 32, cv::kmeans with TermCriteria(ITER|EPS, 100, 10)), centre-weighted colour
// frequencies, the 258-byte descriptor.  Bit-exact against oracle/colordesc_oracle.c, whose header states what is and
// what cannot be pinned against the cbird binary (per-thread RNG state, tie order, the ellipse rim).
//
// Everything after the Luv conversion is ORDER-DEPENDENT arithmetic by the reference's construction: the seeding
// walks `p -= dist[i]` and sums `s += tdist2[i]` in double, centres are float sums in sample order, frequencies are
// float sums in pixel order.  None of these sums is exact, so none can be re-associated; the only parallelism that
// reproduces the reference is ACROSS IMAGES.  Hence:
//   k_cd_prepare   one workgroup per image, pixel-parallel: resize + mask + Luv (the gamma spline collapses into a
//                  256-entry table because its input is an 8-bit value; the cube-root spline table sits in LDS),
//                  ordered compaction of the samples that pass `l > 4`
//   k_cdw_*        the chain kernels (round 3, described below): the unit of parallelism is the CHAIN, not the image.
//                  (Round 2's k_cd_cluster -- one lane per image for everything -- is in the history: r05's
//                  "color_create_chains" 0.)
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "cbh_index.h"

namespace cbh {
namespace {

constexpr int kK = 32;            // ColorDescriptor::NUM_DESC_COLORS
constexpr int kTab = 1024;

struct CdImage {
  unsigned long long src_off;
  unsigned src_stride;
  int w, h;        // source
  int cols, rows;  // after the resize
  unsigned mask_off;
};

// ---- host: tables and the ellipse mask (restated separately from the oracle's copy) ----------------------------------
int cv_round(double v) { return (int)std::nearbyint(v); }
int cv_floor(double v) {
  const int i = (int)v;
  return i - (v < i);
}

float cv_cbrt(float value) {  // cvCbrt (core/src/mathfuncs.cpp)
  union {
    int i;
    float f;
  } v, m;
  v.f = value;
  const int ix = v.i & 0x7fffffff, s = v.i & 0x80000000;
  int ex = (ix >> 23) - 127;
  int shx = ex % 3;
  shx -= shx >= 0 ? 3 : 0;
  ex = (ex - shx) / 3;
  v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
  float fr = v.f;
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
void spline_build(const float* f, int n, float* tab) {
  float cn = 0;
  tab[0] = tab[1] = 0.f;
  for (int i = 1; i < n - 1; i++) {
    const float t = 3 * (f[i + 1] - 2 * f[i] + f[i - 1]);
    const float l = 1 / (4 - tab[(i - 1) * 4]);
    tab[i * 4] = l;
    tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
  }
  for (int i = n - 1; i >= 0; i--) {
    const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
    const float b = f[i + 1] - f[i] - (cn + c * 2) * (float)0.3333333333333333;
    const float d = (cn - c) * (float)0.3333333333333333;
    tab[i * 4] = f[i], tab[i * 4 + 1] = b, tab[i * 4 + 2] = c, tab[i * 4 + 3] = d;
    cn = c;
  }
}
float spline_at(float x, const float* tab, int n) {
  int ix = cv_floor((double)x);
  ix = std::min(std::max(ix, 0), n - 1);
  x -= ix;
  tab += ix * 4;
  return ((tab[3] * x + tab[2]) * x + tab[1]) * x + tab[0];
}
struct CdTables {
  float gamma_lut[256];   // linearised value of an 8-bit channel: spline(sRGBGammaTab)((p * (1/255)) * 1024)
  float cbrt_tab[kTab * 4];
};
const CdTables& tables() {
  static CdTables t;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<float> f(kTab + 1), g(kTab + 1), gtab(kTab * 4);
    float scale = 1.f / (kTab / 1.5f);
    for (int i = 0; i <= kTab; i++) {
      const float x = i * scale;
      f[i] = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cv_cbrt(x);
    }
    spline_build(f.data(), kTab, t.cbrt_tab);
    scale = 1.f / (float)kTab;
    for (int i = 0; i <= kTab; i++) {
      const float x = i * scale;
      g[i] = x <= 0.04045f ? x * (1.f / 12.92f) : (float)std::pow((double)(x + 0.055) * (1. / 1.055), 2.4);
    }
    spline_build(g.data(), kTab, gtab.data());
    const float s255 = (float)(1.0 / 255.0);
    for (int p = 0; p < 256; ++p) t.gamma_lut[p] = spline_at(((float)p * s255 + 0.f) * (float)kTab, gtab.data(), kTab);
  });
  return t;
}

// cv::ellipse(mask, RotatedRect({c/2, r/2}, {0.9 c, 0.9 r}, 0), 255, CV_FILLED): ellipse2Poly + FillConvexPoly (+ Line2)
constexpr int kXYShift = 16, kXYOne = 1 << kXYShift;
struct Pt {
  int x, y;
};
float sin_table(int deg) {  // drawing.cpp's SinTable: sin(deg) written with seven decimals
  char buf[32];
  snprintf(buf, sizeof buf, "%.7f", std::sin(deg * 3.14159265358979323846 / 180.0));
  return strtof(buf, nullptr);
}
void line2(uint8_t* img, int w, int h, Pt a, Pt b) {
  auto put = [&](int x, int y) {
    if (0 <= x && x < w && 0 <= y && y < h) img[(size_t)y * w + x] = 255;
  };
  int dx = b.x - a.x, dy = b.y - a.y;
  const int j = dx < 0 ? -1 : 0, ax = (dx ^ j) - j;
  const int i = dy < 0 ? -1 : 0, ay = (dy ^ i) - i;
  int x_step, y_step, ecount;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    a.x ^= b.x & j, b.x ^= a.x & j, a.x ^= b.x & j;
    a.y ^= b.y & j, b.y ^= a.y & j, a.y ^= b.y & j;
    x_step = kXYOne;
    y_step = (int)(((long long)dy << kXYShift) / (ax | 1));
    ecount = (b.x - a.x) >> kXYShift;
  } else {
    dx = (dx ^ i) - i;
    a.x ^= b.x & i, b.x ^= a.x & i, a.x ^= b.x & i;
    a.y ^= b.y & i, b.y ^= a.y & i, a.y ^= b.y & i;
    x_step = (int)(((long long)dx << kXYShift) / (ay | 1));
    y_step = kXYOne;
    ecount = (b.y - a.y) >> kXYShift;
  }
  a.x += kXYOne >> 1, a.y += kXYOne >> 1;
  put((b.x + (kXYOne >> 1)) >> kXYShift, (b.y + (kXYOne >> 1)) >> kXYShift);
  if (ax > ay) {
    a.x >>= kXYShift;
    for (; ecount >= 0; --ecount) {
      put(a.x, a.y >> kXYShift);
      a.x++, a.y += y_step;
    }
  } else {
    a.y >>= kXYShift;
    for (; ecount >= 0; --ecount) {
      put(a.x >> kXYShift, a.y);
      a.x += x_step, a.y++;
    }
  }
}
void fill_convex(uint8_t* img, int w, int h, const Pt* v, int npts) {
  struct {
    int idx, di, x, dx, ye;
  } edge[2];
  const int delta = 1 << (kXYShift - 1);
  int imin = 0, left = 0, right = 1, edges = npts;
  int xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
  Pt p0 = v[npts - 1];
  for (int i = 0; i < npts; i++) {
    const Pt p = v[i];
    if (p.y < ymin) ymin = p.y, imin = i;
    ymax = std::max(ymax, p.y), xmax = std::max(xmax, p.x), xmin = std::min(xmin, p.x);
    line2(img, w, h, p0, p);
    p0 = p;
  }
  xmin = (xmin + delta) >> kXYShift, xmax = (xmax + delta) >> kXYShift;
  ymin = (ymin + delta) >> kXYShift, ymax = (ymax + delta) >> kXYShift;
  if (npts < 3 || xmax < 0 || ymax < 0 || xmin >= w || ymin >= h) return;
  ymax = std::min(ymax, h - 1);
  int y = ymin;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y;
  edge[0].di = 1, edge[1].di = npts - 1;
  edge[0].x = edge[1].x = edge[0].dx = edge[1].dx = 0;
  do {
    for (int i = 0; i < 2; i++) {
      if (y >= edge[i].ye) {
        int idx = edge[i].idx, xs = 0, ty = 0;
        const int di = edge[i].di;
        for (;;) {
          ty = (v[idx].y + delta) >> kXYShift;
          if (ty > y || edges == 0) break;
          xs = v[idx].x;
          idx += di;
          if (idx >= npts) idx -= npts;
          edges--;
        }
        const int ye = ty, xe = v[idx].x;
        if (y >= ye) return;
        edge[i].ye = ye;
        edge[i].dx = ((xe - xs) * 2 + (ye - y)) / (2 * (ye - y));
        edge[i].x = xs;
        edge[i].idx = idx;
      }
    }
    if (edge[left].x > edge[right].x) left ^= 1, right ^= 1;
    int x1 = edge[left].x, x2 = edge[right].x;
    if (y >= 0) {
      int xx1 = (x1 + (kXYOne >> 1)) >> kXYShift, xx2 = (x2 + (kXYOne >> 1)) >> kXYShift;
      if (xx2 >= 0 && xx1 < w) {
        xx1 = std::max(xx1, 0), xx2 = std::min(xx2, w - 1);
        for (int x = xx1; x <= xx2; ++x) img[(size_t)y * w + x] = 255;
      }
    }
    edge[left].x = x1 + edge[left].dx;
    edge[right].x = x2 + edge[right].dx;
  } while (++y <= ymax);
}
void ellipse_mask(int cols, int rows, uint8_t* mask) {
  memset(mask, 0, (size_t)cols * rows);
  const float cxf = cols * 0.5f, cyf = rows * 0.5f, swf = cols * 0.9f, shf = rows * 0.9f;
  const Pt center = {cv_round((double)(cxf * (1 << kXYShift))), cv_round((double)(cyf * (1 << kXYShift)))};
  const int aw = std::abs(cv_round((double)(swf * (1 << (kXYShift - 1)))));
  const int ah = std::abs(cv_round((double)(shf * (1 << (kXYShift - 1)))));
  int delta = (std::max(aw, ah) + (kXYOne >> 1)) >> kXYShift;
  delta = delta < 3 ? 90 : delta < 10 ? 30 : delta < 15 ? 18 : 5;
  Pt pts[80];
  int n = 0;
  const float alpha = sin_table(450), beta = sin_table(0);
  Pt prev = {INT_MIN, INT_MIN};
  for (int i = 0; i < 360 + delta; i += delta) {
    const int angle = std::min(i, 360);
    const double x = (double)aw * sin_table(450 - angle), y = (double)ah * sin_table(angle);
    const Pt pt = {cv_round((double)center.x + x * alpha - y * beta), cv_round((double)center.y + x * beta + y * alpha)};
    if (pt.x != prev.x || pt.y != prev.y) pts[n++] = pt, prev = pt;
  }
  if (n == 1) pts[n++] = pts[0];
  fill_convex(mask, cols, rows, pts, n);
}

void resized_dims(int w, int h, int* ow, int* oh) {  // sizeLongestSide(rgb, 256, INTER_NEAREST), cvutil.cpp:811, 1932-1942
  *ow = w, *oh = h;
  if (h > 256 || w > 256) {
    const float aspect = (float)w / h;
    if (w > h) {
      *ow = 256;
      *oh = (int)(256 / aspect);
    } else {
      *oh = 256;
      *ow = (int)(aspect * 256);
    }
  }
}

// ---- device ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_i(int v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// per image three planes L[cap], U[cap], V[cap] at samples + 3 * cap * image
__global__ __launch_bounds__(256) void k_cd_prepare(const CdImage* __restrict__ images,
                                                    const unsigned char* __restrict__ imgs, int channels,
                                                    const unsigned char* __restrict__ masks,
                                                    const CdTables* __restrict__ tabs,
                                                    float* __restrict__ samples /* [image][3][cap] */,
                                                    unsigned* __restrict__ pos /* row << 16 | col */,
                                                    int* __restrict__ counts, unsigned cap /* samples per image slot */) {
  __shared__ float s_cbrt[kTab * 4];
  __shared__ float s_gamma[256];
  __shared__ int s_xofs[256], s_yofs[256];
  __shared__ int s_w[4];
  const int tid = (int)threadIdx.x;
  const unsigned img_i = blockIdx.x;
  const CdImage im = images[img_i];
  for (int i = tid; i < kTab * 4; i += 256) s_cbrt[i] = tabs->cbrt_tab[i];
  s_gamma[tid] = tabs->gamma_lut[tid];
  {
    // resizeNN: sx = min(floor(x * (1 / (dcols / w))), w - 1)   (identity when the image was not resized)
    const double ifx = 1. / ((double)im.cols / im.w), ify = 1. / ((double)im.rows / im.h);
    if (tid < im.cols) s_xofs[tid] = min((int)floor(tid * ifx), im.w - 1);
    if (tid < im.rows) s_yofs[tid] = min((int)floor(tid * ify), im.h - 1);
  }
  __syncthreads();
  // sRGB2XYZ_D65 with the R / B columns swapped for blueIdx 0; D65 white point
  const float C0 = 0.180423f, C1 = 0.357580f, C2 = 0.412453f, C3 = 0.072169f, C4 = 0.715160f, C5 = 0.212671f,
              C6 = 0.950227f, C7 = 0.119193f, C8 = 0.019334f;
  const float d0 = 1.f / (0.950456f + 1.f * 15 + 1.088754f * 3);
  const float un = 4 * 0.950456f * d0, vn = 9 * 1.f * d0;
  const float _un = 13 * un, _vn = 13 * vn;
  const float cbrt_scale = kTab / 1.5f;
  const unsigned char* __restrict__ src = imgs + im.src_off;
  const unsigned char* __restrict__ mask = masks + im.mask_off;
  const int total = im.cols * im.rows;
  int n_out = 0;
  for (int p0 = 0; p0 < total; p0 += 256) {
    const int p = p0 + tid;
    int keep = 0;
    float L = 0.f, U = 0.f, V = 0.f;
    int row = 0, col = 0;
    if (p < total) {
      row = p / im.cols, col = p - row * im.cols;
      const unsigned char* __restrict__ q = src + (size_t)s_yofs[row] * im.src_stride + (size_t)s_xofs[col] * channels;
      const int alpha = mask[p];
      // pix = (pix * alpha >> 8) & 0xFF; convertTo(CV_32F) * (1/255); gamma spline == table of the 8-bit value
      const float R = s_gamma[((int)q[0] * alpha >> 8) & 0xFF];
      const float G = s_gamma[((int)q[1] * alpha >> 8) & 0xFF];
      const float B = s_gamma[((int)q[2] * alpha >> 8) & 0xFF];
      const float X = R * C0 + G * C1 + B * C2;
      const float Y = R * C3 + G * C4 + B * C5;
      const float Z = R * C6 + G * C7 + B * C8;
      float x = Y * cbrt_scale;
      int ix = (int)floorf(x);
      ix = min(max(ix, 0), kTab - 1);
      x -= (float)ix;
      const float* __restrict__ t = s_cbrt + ix * 4;
      L = ((t[3] * x + t[2]) * x + t[1]) * x + t[0];
      L = 116.f * L - 16.f;
      const float den = X + 15 * Y + 3 * Z;
      const float d = (4 * 13) / fmaxf(den, FLT_EPSILON);
      U = L * (X * d - _un);
      V = L * ((9 * 0.25f) * Y * d - _vn);
      keep = L > 4;  // histFilter = brightFilter (cvutil.cpp:760-776)
    }
    // ordered compaction over the workgroup
    const int incl = wave_incl_scan_i(keep);
    const int wv = tid >> 6;
    __syncthreads();
    if ((tid & 63) == 63) s_w[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += s_w[i];
    const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (keep) {
      const unsigned idx = (unsigned)(n_out + base + incl - 1);
      const size_t b = (size_t)img_i * cap;
      samples[3 * b + idx] = L, samples[3 * b + cap + idx] = U, samples[3 * b + 2 * (size_t)cap + idx] = V;
      pos[b + idx] = (unsigned)row << 16 | (unsigned)col;
    }
    n_out += tot;
  }
  if (tid == 0) counts[img_i] = n_out;
}

struct Rng {
  unsigned long long state;
  __device__ __forceinline__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  __device__ __forceinline__ double real() {
    const unsigned t = next();
    return (double)(((unsigned long long)t << 32) | next()) * 5.4210108624275221700372640043497e-20;
  }
};

__device__ __forceinline__ float dist3(float a0, float a1, float a2, float b0, float b1, float b2) {
  float d = 0.f, t = a0 - b0;  // normL2Sqr_, n = 3: d += t*t in index order, floats
  d += t * t;
  t = a1 - b1;
  d += t * t;
  t = a2 - b2;
  d += t * t;
  return d;
}



// ====================================================================================================================
// Round 3: an image no longer waits on one lane's loads.
//
// What cannot change: every sum of the reference is a rounded sequential chain -- `s += tdist2[i]` and `p -= dist[i]`
// in double over all samples (k-means++ seeding: 3 candidate sums + 3 walks per round, 31 rounds), the centre sums and
// the colour frequencies in float in sample order.  Round 2 ran ALL of an image's work, the embarrassingly parallel
// distance evaluations included, on the one lane that owned its chains: 0.43 s of latency for anything up to 4096
// images.  Here the unit of parallelism is the CHAIN, not the image:
//   * a lane owns one chain and streams its inputs with 16-byte loads, four blocks ahead (the loads do not depend on the
//     sum, so the only serial latency left is the add itself);
//   * seeding: lane (image, j) evaluates candidate j's distances on the fly -- min(dist[i], |sample_i - cand_j|^2), 8
//     independent float ops that fill the issue slots the dependent double add leaves empty -- sums them, the three
//     lanes of an image compare their sums, the winner's array becomes `dist`, then the same three lanes walk it for
//     the next round's three candidates.  21 images per wave, one launch per round, nothing visits the host;
//   * k-means: labels in a sample-parallel kernel (a workgroup per image); centre sums with lane (image, cluster): all
//     32 lanes of an image read the same sample and add it or +0.0f (x + 0 is exact), so the per-cluster sums keep
//     sample order without a partition; empty clusters, new centres and the shift test by a lane per image;
//   * colour frequencies: per-sample weights in parallel, then lane (image, colour) chains like the centre sums.
// Per-image state lives in CdW (global memory) between the launches (tests/test_color_create.py: byte for byte
// oracle/colordesc_oracle.c's).
// ====================================================================================================================
struct CdW {
  unsigned long long rng;
  double sum0;
  double shift;
  int n, N;      // n = N when the image has >= 32 samples, else 0 ("not enough colors")
  int perm[4];   // distance slots: perm[0] = dist, perm[1..3] = the three candidate arrays of a round
  int ci[3];     // sample indices of the round's three candidate centres
  int iter, fin, empties;
  float c[3][kK];    // centres
  float sum[3][kK];  // centre sums of an update; later: colour frequencies in sum[0]
  int cnt[kK];       // cluster sizes; later: representative centre of a colour key
  unsigned long long key[kK];
  unsigned members[kK];  // labels whose centre compresses to colour k (bit mask), 0 when k is not a representative
};

__device__ __forceinline__ unsigned rng_next(unsigned long long& st) {
  st = (unsigned long long)(unsigned)st * 4164903690ull + (unsigned)(st >> 32);
  return (unsigned)st;
}
__device__ __forceinline__ double rng_real(unsigned long long& st) {
  const unsigned t = rng_next(st);
  return (double)(((unsigned long long)t << 32) | rng_next(st)) * 5.4210108624275221700372640043497e-20;
}

__global__ __launch_bounds__(256) void k_cdw_init(CdW* __restrict__ W, const int* __restrict__ counts, unsigned n_images,
                                                  unsigned char* __restrict__ ok, unsigned char* __restrict__ descs) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_images) return;
  CdW& w = W[i];
  const int N = counts[i];
  const bool valid = N >= kK;
  ok[i] = valid ? 1 : 0;
  w.N = N, w.n = valid ? N : 0;
  w.rng = 0xffffffffull;  // RNG(): a fresh thread's generator (oracle header, (1))
  const int c0 = valid ? (int)(rng_next(w.rng) % (unsigned)N) : 0;
  w.ci[0] = w.ci[1] = w.ci[2] = c0;
  w.perm[0] = 0, w.perm[1] = 1, w.perm[2] = 2, w.perm[3] = 3;
  w.sum0 = 0, w.shift = DBL_MAX;
  w.iter = 0, w.fin = valid ? 0 : 1, w.empties = 0;
  for (int k = 0; k < kK; ++k) w.c[0][k] = w.c[1][k] = w.c[2][k] = 0.f;
  if (!valid)  // the reference leaves the caller's (cleared) descriptor alone
    for (int b = 0; b < 258; ++b) descs[(size_t)i * 258 + b] = 0;
}

// One seeding round for G images per wave (G <= 21).  round 0: all three trials evaluate centre 0 (ci[j] = c0, no
// previous distances: old = +inf), so sum0 and dist come out of the same code; rounds 1..31: candidate j.
//   tile loop (64 samples at a time):
//     produce  all 64 lanes, lane = sample: for every image of the wave and each of its three candidates
//              t = min(dist[i], |sample_i - cand|^2) -- coalesced loads (the next tile's are in flight), the result to
//              LDS row (image, j) and, coalesced, to the candidate's distance array
//     consume  lane (image, j) adds its row to its double sum, in index order
//   then the three lanes of an image compare their sums (the first smallest wins: `s < bestSum` visits the trials in
//   order), the winner's array becomes `dist`, and -- unless this was the last round -- the same lanes walk it for the
//   next round's candidates: p = rng.real() * sum0; for (i = 0; i < N-1; i++) if ((p -= dist[i]) <= 0) break; ci = i
constexpr int kPF = 4;       // float4 blocks a lane keeps in flight per stream (k_cdw_update / k_cdw_freq)
constexpr int kTRow = 66;    // LDS row pitch in floats: 64 + 2 (rows are read two floats at a time, conflict-free)
template <int G>
__global__ __launch_bounds__(64) void k_cdw_round(CdW* __restrict__ W, unsigned n_images, const float* __restrict__ samples,
                                                  float* __restrict__ dists, unsigned cap, size_t slot_stride, int round) {
  __shared__ float s_T[3 * G][kTRow];
  __shared__ float4 s_cand[G][3];
  __shared__ unsigned long long s_off[G][4];  // element offsets of dist / the three candidate arrays inside `dists`
  __shared__ int s_n[G];
  const unsigned lane = threadIdx.x;
  const unsigned g = lane / 3u;
  const int j = (int)(lane % 3u);
  const unsigned img0 = blockIdx.x * (unsigned)G;
  const unsigned img = img0 + g;
  const bool act = g < (unsigned)G && img < n_images;
  CdW* w = W + (act ? img : 0);
  const int n = act ? w->n : 0;
  if (g < (unsigned)G) {
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) {
      const float* __restrict__ Ls = samples + (size_t)3 * cap * img;
      const int ci = w->ci[j];
      c = make_float4(Ls[ci], Ls[cap + ci], Ls[2 * (size_t)cap + ci], 0.f);
    }
    s_cand[g][j] = c;
    s_off[g][1 + j] = (size_t)cap * (act ? img : 0) + (size_t)(act ? w->perm[1 + j] : 0) * slot_stride;
    if (j == 0) {
      s_off[g][0] = (size_t)cap * (act ? img : 0) + (size_t)(act ? w->perm[0] : 0) * slot_stride;
      s_n[g] = n;
    }
  }
  __syncthreads();
  int nmax = 0;
#pragma unroll
  for (int q = 0; q < G; ++q) nmax = max(nmax, s_n[q]);
  // ---- phase S
  double sj = 0;
  {
    float pl[G], pu[G], pv[G], po[G];
    auto fetch = [&](int i0) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const unsigned im = min(img0 + (unsigned)q, n_images - 1);
        const float* __restrict__ Ls = samples + (size_t)3 * cap * im;
        const int i = min(i0 + (int)lane, (int)cap - 1);  // past an image's n: padding or stale values, never consumed
        pl[q] = Ls[i], pu[q] = Ls[cap + i], pv[q] = Ls[2 * (size_t)cap + i];
        po[q] = round ? dists[s_off[q][0] + (size_t)i] : INFINITY;
      }
    };
    fetch(0);
    for (int i0 = 0; i0 < nmax; i0 += 64) {
      float tl[G], tu[G], tv[G], to[G];
#pragma unroll
      for (int q = 0; q < G; ++q) tl[q] = pl[q], tu[q] = pu[q], tv[q] = pv[q], to[q] = po[q];
      if (i0 + 64 < nmax) fetch(i0 + 64);
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const bool in = i0 + (int)lane < s_n[q];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float4 c = s_cand[q][t];
          const float d = dist3(tl[q], tu[q], tv[q], c.x, c.y, c.z);
          const float m = to[q] < d ? to[q] : d;  // std::min(d, dist[i])
          s_T[3 * q + t][lane] = m;
          if (in) dists[s_off[q][1 + t] + (size_t)(i0 + (int)lane)] = m;
        }
      }
      __syncthreads();
      if (i0 < n) {
        // the whole row into registers first (32 independent LDS reads), then the chain of 64 adds
        const float2* __restrict__ row = reinterpret_cast<const float2*>(s_T[lane < 3u * G ? lane : 0]);
        const int m = n - i0;
#pragma unroll
        for (int h = 0; h < 32; h += 8) {  // a quarter row at a time: 8 independent LDS reads, then 16 dependent adds
          float2 x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = row[h + e];
          if (m >= 64) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sj += x[e].x, sj += x[e].y;
          } else {  // the last tile: elements past the end count as +0.0 (s + 0 is exact)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              sj += 2 * (h + e) < m ? x[e].x : 0.f, sj += 2 * (h + e) + 1 < m ? x[e].y : 0.f;
          }
        }
      }
      __syncthreads();
    }
  }
  // ---- the three lanes of an image compare
  const unsigned l0 = lane - (unsigned)j;
  const double s0 = __shfl(sj, (int)l0), s1 = __shfl(sj, (int)min(l0 + 1, 63u)), s2 = __shfl(sj, (int)min(l0 + 2, 63u));
  int best = 0;
  double bs = s0;
  if (s1 < bs) bs = s1, best = 1;
  if (s2 < bs) bs = s2, best = 2;
  if (n > 0 && j == 0) {
    const float4 c = s_cand[g][best];
    const int p0 = w->perm[0], pb = w->perm[1 + best];
    w->sum0 = bs;
    w->perm[0] = pb, w->perm[1 + best] = p0;  // std::swap(dist, tdist)
    w->c[0][round] = c.x, w->c[1][round] = c.y, w->c[2][round] = c.z;
  }
  if (round == kK - 1) return;
  // ---- phase W
  if (g < (unsigned)G && j == 0) s_off[g][0] = s_off[g][1 + best];  // the winner's array (written above, by this wave)
  unsigned long long st = n > 0 ? w->rng : 0;
  double p = 0;
  {
    unsigned long long t = st;
    for (int q = 0; q < 3; ++q) {
      const double r = rng_real(t);
      if (q == j) p = r * bs;
    }
    st = t;  // six draws on (the generator advances by two draws per trial whatever the trial finds)
  }
  __threadfence_block();
  __syncthreads();
  const int m = max(n - 1, 0);  // elements the walk may visit
  int ci = m;
  bool found = m == 0;
  {
    const int mmax = max(nmax - 1, 0);
    float pw[G];
    auto fetch = [&](int i0) {
#pragma unroll
      for (int q = 0; q < G; ++q) pw[q] = dists[s_off[q][0] + (size_t)min(i0 + (int)lane, (int)cap - 1)];
    };
    fetch(0);
    for (int i0 = 0; i0 < mmax && __any(!found); i0 += 64) {
#pragma unroll
      for (int q = 0; q < G; ++q) s_T[q][lane] = pw[q];
      if (i0 + 64 < mmax) fetch(i0 + 64);
      __syncthreads();
      if (!found && i0 < m) {
        const float2* __restrict__ row = reinterpret_cast<const float2*>(s_T[g < (unsigned)G ? g : 0]);
        const int lim = m - i0;
        // p only ever decreases, so a quarter row (16 elements) is subtracted blind and the sign looked at once; the
        // one quarter in which p crosses zero is walked again from the value it had at its start to find the element.
        // Elements past the end subtract +0.0 (which can never make a positive p non-positive).
#pragma unroll
        for (int h = 0; h < 32; h += 8) {
          float2 x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = row[h + e];
          if (lim < 64) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              x[e].x = 2 * (h + e) < lim ? x[e].x : 0.f;
              x[e].y = 2 * (h + e) + 1 < lim ? x[e].y : 0.f;
            }
          }
          const double p_in = p;
#pragma unroll
          for (int e = 0; e < 8; ++e) p -= x[e].x, p -= x[e].y;
          if (!found && p <= 0) {  // once per lane and round
            double r = p_in;
            int hit = -1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              r -= x[e].x;
              hit = (hit < 0 && r <= 0) ? 2 * (h + e) : hit;
              r -= x[e].y;
              hit = (hit < 0 && r <= 0) ? 2 * (h + e) + 1 : hit;
            }
            found = true, ci = i0 + hit;
          }
        }
      }
      __syncthreads();
    }
  }
  if (n > 0) {
    w->ci[j] = ci;
    if (j == 0) w->rng = st;
  }
}

// labels of the unfinished images (KMeansDistanceComputer): a workgroup per image, a sample per thread
__global__ __launch_bounds__(256) void k_cdw_assign(const CdW* __restrict__ W, const float* __restrict__ samples,
                                                    unsigned char* __restrict__ labels, unsigned cap) {
  const unsigned img = blockIdx.x;
  const CdW& w = W[img];
  if (w.fin) return;
  __shared__ float s_c[3][kK];
  if (threadIdx.x < 3 * kK) (&s_c[0][0])[threadIdx.x] = (&w.c[0][0])[threadIdx.x];
  __syncthreads();
  const int n = w.n;
  const float* __restrict__ Ls = samples + (size_t)3 * cap * img;
  for (int i = (int)threadIdx.x; i < n; i += 256) {
    const float a = Ls[i], b = Ls[cap + i], c = Ls[2 * (size_t)cap + i];
    int k_best = 0;
    float min_dist = dist3(a, b, c, s_c[0][0], s_c[1][0], s_c[2][0]);
#pragma unroll
    for (int k = 1; k < kK; ++k) {
      const float d = dist3(a, b, c, s_c[0][k], s_c[1][k], s_c[2][k]);
      if (min_dist > d) min_dist = d, k_best = k;  // the reference compares in double; floats order the same way
    }
    labels[(size_t)cap * img + i] = (unsigned char)k_best;
  }
}

// centre sums and sizes of the unfinished images that have been labelled: lane (image, cluster), two images per wave.
// Every lane of an image sees every sample in order and adds it or +0.0f: the per-cluster float sums are formed in
// sample order exactly as the reference's loop over the samples forms them.
__global__ __launch_bounds__(64) void k_cdw_update(CdW* __restrict__ W, unsigned n_images, const float* __restrict__ samples,
                                                   const unsigned char* __restrict__ labels, unsigned cap) {
  const unsigned lane = threadIdx.x;
  const unsigned img = blockIdx.x * 2u + (lane >> 5);
  const int k = (int)(lane & 31u);
  const bool act = img < n_images;
  CdW* w = W + (act ? img : 0);
  const int n = (act && !w->fin && w->iter > 0) ? w->n : 0;
  const float* __restrict__ Ls = samples + (size_t)3 * cap * (act ? img : 0);
  const float* __restrict__ Us = Ls + cap;
  const float* __restrict__ Vs = Us + cap;
  const unsigned char* __restrict__ lb = labels + (size_t)cap * (act ? img : 0);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  int cnt = 0;
  const int nb = (n + 3) >> 2;
  int nbmax = max(nb, __shfl_xor(nb, 32));
  float4 qL[kPF], qU[kPF], qV[kPF];
  unsigned qB[kPF];
  auto fetch = [&](int b, float4& l, float4& u, float4& v, unsigned& t) {
    const int bb = min(b, max(nb - 1, 0));
    l = reinterpret_cast<const float4*>(Ls)[bb], u = reinterpret_cast<const float4*>(Us)[bb],
    v = reinterpret_cast<const float4*>(Vs)[bb], t = reinterpret_cast<const unsigned*>(lb)[bb];
  };
#pragma unroll
  for (int u = 0; u < kPF; ++u) fetch(u, qL[u], qU[u], qV[u], qB[u]);
  for (int b0 = 0; b0 < nbmax; b0 += kPF) {
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
      const int b = b0 + u;
      const float4 l = qL[u], uu = qU[u], v = qV[u];
      const unsigned t = qB[u];
      fetch(b + kPF, qL[u], qU[u], qV[u], qB[u]);
      if (b < nb) {
        const int i = b * 4;
        const bool m0 = (int)(t & 255u) == k, m1 = i + 1 < n && (int)((t >> 8) & 255u) == k,
                   m2 = i + 2 < n && (int)((t >> 16) & 255u) == k, m3 = i + 3 < n && (int)(t >> 24) == k;
        s0 += m0 ? l.x : 0.f, s1 += m0 ? uu.x : 0.f, s2 += m0 ? v.x : 0.f;
        s0 += m1 ? l.y : 0.f, s1 += m1 ? uu.y : 0.f, s2 += m1 ? v.y : 0.f;
        s0 += m2 ? l.z : 0.f, s1 += m2 ? uu.z : 0.f, s2 += m2 ? v.z : 0.f;
        s0 += m3 ? l.w : 0.f, s1 += m3 ? uu.w : 0.f, s2 += m3 ? v.w : 0.f;
        cnt += (int)m0 + (int)m1 + (int)m2 + (int)m3;
      }
    }
  }
  if (n > 0) {
    w->sum[0][k] = s0, w->sum[1][k] = s1, w->sum[2][k] = s2;
    w->cnt[k] = cnt;
  }
}

// the rest of an iteration, a lane per image: empty clusters (an empty cluster takes the point farthest from the centre
// of the biggest one), new centres, the shift test, ++iter.  *unfinished counts the images that go on.
__global__ __launch_bounds__(64) void k_cdw_post(CdW* __restrict__ W, unsigned n_images, const float* __restrict__ samples,
                                                 unsigned char* __restrict__ labels, unsigned cap,
                                                 unsigned* __restrict__ unfinished) {
  const unsigned img = blockIdx.x * 64u + threadIdx.x;
  if (img >= n_images) return;
  CdW& w = W[img];
  if (w.fin) return;
  const int n = w.n;
  const float* __restrict__ Ls = samples + (size_t)3 * cap * img;
  unsigned char* __restrict__ lb = labels + (size_t)cap * img;
  if (w.iter > 0) {
    double max_center_shift = 0;
    for (int k = 0; k < kK; ++k) {
      if (w.cnt[k] != 0) continue;
      int max_k = 0;
      for (int k1 = 1; k1 < kK; ++k1)
        if (w.cnt[max_k] < w.cnt[k1]) max_k = k1;
      const float scale = 1.f / w.cnt[max_k];
      const float t0 = w.sum[0][max_k] * scale, t1 = w.sum[1][max_k] * scale, t2 = w.sum[2][max_k] * scale;
      double max_dist = 0;
      int farthest_i = -1;
      for (int i = 0; i < n; ++i) {
        if (lb[i] != max_k) continue;
        const double dist = dist3(Ls[i], Ls[cap + i], Ls[2 * (size_t)cap + i], t0, t1, t2);
        if (max_dist <= dist) max_dist = dist, farthest_i = i;
      }
      w.cnt[max_k]--;
      w.cnt[k]++;
      lb[farthest_i] = (unsigned char)k;
      const float b0 = Ls[farthest_i], b1 = Ls[cap + farthest_i], b2 = Ls[2 * (size_t)cap + farthest_i];
      w.sum[0][max_k] -= b0, w.sum[1][max_k] -= b1, w.sum[2][max_k] -= b2;
      w.sum[0][k] += b0, w.sum[1][k] += b1, w.sum[2][k] += b2;
    }
    for (int k = 0; k < kK; ++k) {
      const float scale = 1.f / w.cnt[k];
      const float n0 = w.sum[0][k] * scale, n1 = w.sum[1][k] * scale, n2 = w.sum[2][k] * scale;
      double dist = 0, t = n0 - w.c[0][k];
      dist += t * t;
      t = n1 - w.c[1][k];
      dist += t * t;
      t = n2 - w.c[2][k];
      dist += t * t;
      max_center_shift = fmax(max_center_shift, dist);
      w.c[0][k] = n0, w.c[1][k] = n1, w.c[2][k] = n2;
    }
    w.shift = max_center_shift;
  }
  ++w.iter;
  w.fin = (w.iter == 100 || w.shift <= 100.0) ? 1 : 0;  // ++iter == MAX(maxCount, 2) || shift <= epsilon^2
  if (!w.fin) atomicAdd(unfinished, 1u);
}

// colour keys of the centres, their representatives, and the per-sample weights (cvutil.cpp:903-989)
__global__ __launch_bounds__(256) void k_cdw_weights(CdW* __restrict__ W, const CdImage* __restrict__ images,
                                                     const unsigned* __restrict__ pos, float* __restrict__ weights,
                                                     unsigned cap) {
  const unsigned img = blockIdx.x;
  CdW& w = W[img];
  if (w.n == 0) return;
  const CdImage im = images[img];
  if (threadIdx.x < kK) {
    const int k = (int)threadIdx.x;
    auto clamp16 = [](int v) {
      v &= -(v >= 0);
      return v | ((65535 - v) >> 31);
    };
    const unsigned l = (unsigned)clamp16((int)(65535 / 100.0f * w.c[0][k])) & 0xFFFFu;
    const unsigned u = (unsigned)clamp16((int)(65535 / 354.0f * (w.c[1][k] + 134.0f))) & 0xFFFFu;
    const unsigned v = (unsigned)clamp16((int)(65535 / 262.0f * (w.c[2][k] + 140.0f))) & 0xFFFFu;
    w.key[k] = (unsigned long long)l << 32 | (unsigned long long)u << 16 | (unsigned long long)v;
  }
  __syncthreads();
  if (threadIdx.x < kK) {  // centres that compress to the same colour share one frequency (the QHash is keyed by colour)
    const int k = (int)threadIdx.x;
    int rep = k;
    for (int q = kK - 1; q >= 0; --q)
      if (q < k && w.key[q] == w.key[k]) rep = q;
    w.cnt[k] = rep;
  }
  __syncthreads();
  if (threadIdx.x < kK) {
    const int k = (int)threadIdx.x;
    unsigned m = 0;
    for (int q = 0; q < kK; ++q)
      if (w.cnt[q] == k) m |= 1u << q;
    w.members[k] = m;
  }
  const float dx0 = im.cols / 2.0f, dy0 = im.rows / 2.0f;
  const float maxDistFromCenter = sqrtf(dx0 * dx0 + dy0 * dy0);
  const int N = w.N;
  for (int i = (int)threadIdx.x; i < N; i += 256) {
    const unsigned pv = pos[(size_t)cap * img + i];
    const int dx = (int)(pv & 0xFFFFu) - im.cols / 2, dy = (int)(pv >> 16) - im.rows / 2;
    const float dist = sqrtf((float)(dx * dx + dy * dy));
    weights[(size_t)cap * img + i] = (maxDistFromCenter - dist) / maxDistFromCenter;
  }
}

// colour frequencies: lane (image, representative colour) sums the weights of its samples in pixel order
__global__ __launch_bounds__(64) void k_cdw_freq(CdW* __restrict__ W, unsigned n_images, const float* __restrict__ weights,
                                                 const unsigned char* __restrict__ labels, unsigned cap) {
  const unsigned lane = threadIdx.x;
  const unsigned img = blockIdx.x * 2u + (lane >> 5);
  const int k = (int)(lane & 31u);
  const bool act = img < n_images;
  CdW* w = W + (act ? img : 0);
  const int N = act && w->n ? w->N : 0;
  const unsigned mine = N ? w->members[k] : 0u;
  const float* __restrict__ ws = weights + (size_t)cap * (act ? img : 0);
  const unsigned char* __restrict__ lb = labels + (size_t)cap * (act ? img : 0);
  float f = 0.f;
  int cnt = 0;
  const int nb = (N + 3) >> 2;
  const int nbmax = max(nb, __shfl_xor(nb, 32));
  float4 qW[kPF];
  unsigned qB[kPF];
  auto fetch = [&](int b, float4& x, unsigned& t) {
    const int bb = min(b, max(nb - 1, 0));
    x = reinterpret_cast<const float4*>(ws)[bb], t = reinterpret_cast<const unsigned*>(lb)[bb];
  };
#pragma unroll
  for (int u = 0; u < kPF; ++u) fetch(u, qW[u], qB[u]);
  for (int b0 = 0; b0 < nbmax; b0 += kPF) {
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
      const int b = b0 + u;
      const float4 x = qW[u];
      const unsigned t = qB[u];
      fetch(b + kPF, qW[u], qB[u]);
      if (b < nb) {
        const int i = b * 4;
        const bool m0 = mine >> (t & 31u) & 1u, m1 = i + 1 < N && (mine >> ((t >> 8) & 31u) & 1u),
                   m2 = i + 2 < N && (mine >> ((t >> 16) & 31u) & 1u), m3 = i + 3 < N && (mine >> ((t >> 24) & 31u) & 1u);
        f += m0 ? x.x : 0.f;
        f += m1 ? x.y : 0.f;
        f += m2 ? x.z : 0.f;
        f += m3 ? x.w : 0.f;
        cnt += (int)m0 + (int)m1 + (int)m2 + (int)m3;
      }
    }
  }
  if (N) {
    w->sum[0][k] = f;
    w->sum[1][k] = cnt ? 1.f : 0.f;  // a colour exists in the reference's hash only if some sample carried it
  }
}

// the descriptor (cvutil.cpp:1016-1060): colours by descending frequency, ties by key
__global__ __launch_bounds__(64) void k_cdw_finish(const CdW* __restrict__ W, unsigned n_images,
                                                   unsigned char* __restrict__ descs) {
  const unsigned img = blockIdx.x * 64u + threadIdx.x;
  if (img >= n_images) return;
  const CdW& w = W[img];
  if (w.n == 0) return;
  unsigned present = 0;
  float maxFreq = 0;
  for (int k = 0; k < kK; ++k)
    if (w.sum[1][k] != 0.f) present |= 1u << k, maxFreq = fmaxf(maxFreq, w.sum[0][k]);
  unsigned char* __restrict__ out = descs + (size_t)img * 258;
  for (int b = 0; b < 258; ++b) out[b] = 0;
  unsigned left = present;
  int di = 0;
  while (left) {
    int best = -1;
    for (int k = 0; k < kK; ++k) {
      if (!(left >> k & 1u)) continue;
      if (best < 0) {
        best = k;
        continue;
      }
      const float fk = w.sum[0][k], fb = w.sum[0][best];
      if (fk > fb || (fk == fb && w.key[k] < w.key[best])) best = k;
    }
    left &= ~(1u << best);
    const unsigned long long kk = w.key[best];
    const unsigned short l = (unsigned short)((kk >> 32) & 0xFFFF), u = (unsigned short)((kk >> 16) & 0xFFFF),
                         v = (unsigned short)(kk & 0xFFFF);
    const unsigned short wv = (unsigned short)((int)(w.sum[0][best] * 65535 / maxFreq) & 0xFFFF);
    unsigned short* __restrict__ o16 = reinterpret_cast<unsigned short*>(out + di * 8);
    o16[0] = l, o16[1] = u, o16[2] = v, o16[3] = wv;
    out[256] = (unsigned char)di;  // desc.numColors = descIndex, the index of the last colour (cvutil.cpp:1035)
    ++di;
  }
}

// "color_create_chunk_mb": scratch one launch may take.  The sample / distance / position / label planes cost 33 bytes per
// sample slot = 1.6 MB per 256 x 192 image, so a batch is worked off in chunks of that many images (19 000 at the default
// of 32 GiB; the rate is flat from ~16 000 images on) and every chunk reuses the blocks of the one before: 100 000 images
// took 150 GB of scratch in one piece -- above any budget the arena keeps cached, so every call mapped it afresh
// (4-6 s per call against 0.9 s in chunks, tools/ab/color_create_pool.py).
int g_cd_chunk_mb = 32768;

}  // namespace

static int launch_color_descriptors_chunk(const uint8_t* d_imgs, size_t n, const uint64_t* img_off,
                                          const uint32_t* img_w, const uint32_t* img_h, const uint32_t* img_row_stride,
                                          int channels, uint8_t* d_descs, uint8_t* d_ok, hipStream_t s) {
  std::vector<CdImage> images(n);
  std::map<std::pair<int, int>, unsigned> mask_of;
  std::vector<uint8_t> masks;
  for (size_t i = 0; i < n; ++i) {
    CdImage& im = images[i];
    im.src_off = img_off[i];
    im.src_stride = img_row_stride[i];
    im.w = (int)img_w[i], im.h = (int)img_h[i];
    resized_dims(im.w, im.h, &im.cols, &im.rows);
    if (im.cols < 1 || im.rows < 1) return CBH_E_INVAL;  // sizeLongestSide throws
    const auto keyv = std::make_pair(im.cols, im.rows);
    auto it = mask_of.find(keyv);
    if (it == mask_of.end()) {
      const unsigned off = (unsigned)masks.size();
      masks.resize(masks.size() + (size_t)im.cols * im.rows);
      ellipse_mask(im.cols, im.rows, masks.data() + off);
      it = mask_of.emplace(keyv, off).first;
    }
    im.mask_off = it->second;
  }
  unsigned cap = 64;
  for (const CdImage& im : images) cap = std::max(cap, (unsigned)(im.cols * im.rows));  // <= kMaxSamples
  cap = (cap + 63u) & ~63u;  // (planes start on 16-byte boundaries and can be read four samples at a time)
  const size_t slots = n * (size_t)cap;  // sample slots in every array
  CdImage* d_images = nullptr;
  uint8_t *d_masks = nullptr, *d_labels = nullptr;
  CdTables* d_tabs = nullptr;
  float *d_samples = nullptr, *d_dists = nullptr;
  unsigned* d_pos = nullptr;
  int* d_counts = nullptr;
  CdW* d_w = nullptr;
  unsigned* d_unfinished = nullptr;
  unsigned* h_unfinished = nullptr;
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = cbh::malloc_async(p, std::max<size_t>(bytes, 256), s);
  };
  alloc((void**)&d_images, n * sizeof(CdImage));
  alloc((void**)&d_masks, masks.size());
  alloc((void**)&d_tabs, sizeof(CdTables));
  alloc((void**)&d_samples, slots * 3 * sizeof(float));
  alloc((void**)&d_dists, slots * 4 * sizeof(float));
  alloc((void**)&d_pos, slots * sizeof(unsigned));
  alloc((void**)&d_labels, slots);
  alloc((void**)&d_counts, n * sizeof(int));
  alloc((void**)&d_w, n * sizeof(CdW));
  alloc((void**)&d_unfinished, sizeof(unsigned));
  if (e == hipSuccess) e = hipHostMalloc(&h_unfinished, sizeof(unsigned));
  int rc = CBH_OK;
  if (e == hipSuccess) e = hipMemcpyAsync(d_images, images.data(), n * sizeof(CdImage), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_masks, masks.data(), masks.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_tabs, &tables(), sizeof(CdTables), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const unsigned ni = (unsigned)n;
    const size_t slot_stride = slots;  // floats between two distance slots
    hipLaunchKernelGGL(k_cd_prepare, dim3(ni), dim3(256), 0, s, d_images, d_imgs, channels, d_masks, d_tabs,
                       d_samples, d_pos, d_counts, cap);
    hipLaunchKernelGGL(k_cdw_init, dim3((ni + 255) / 256), dim3(256), 0, s, d_w, d_counts, ni, d_ok, d_descs);
    // images per wave: two waves per SIMD while the batch allows it (measured at 4096 images: 1 / 2 / 4 / 8 / 16 images
    // per wave = 76 / 60 / 68 / 88 / 131 ms -- the chain phases of one wave hide behind the produce phase of the
    // other), at most 21 (63 chain lanes); a large batch is bound by the HBM traffic of the distance arrays instead
    const unsigned want = (ni + 2047) / 2048;
    const int G = want <= 1 ? 1 : want <= 2 ? 2 : want <= 4 ? 4 : want <= 8 ? 8 : want <= 16 ? 16 : 21;
    for (int round = 0; round < kK; ++round) {  // generateCentersPP: centre 0, then 31 rounds of three trials
#define CBH_ROUND(GG)                                                                                                 \
  hipLaunchKernelGGL(k_cdw_round<GG>, dim3((ni + GG - 1) / GG), dim3(64), 0, s, d_w, ni, d_samples, d_dists, cap, \
                     slot_stride, round)
      switch (G) {
        case 1: CBH_ROUND(1); break;
        case 2: CBH_ROUND(2); break;
        case 4: CBH_ROUND(4); break;
        case 8: CBH_ROUND(8); break;
        case 16: CBH_ROUND(16); break;
        default: CBH_ROUND(21); break;
      }
#undef CBH_ROUND
    }
    e = hipGetLastError();
    // the k-means loop: every image stops on its own (shift <= epsilon^2 or 100 iterations); the host only learns how
    // many are still running (4 bytes per iteration)
    for (int it = 0; it <= 100 && e == hipSuccess; ++it) {
      if (it > 0)
        hipLaunchKernelGGL(k_cdw_update, dim3((ni + 1) / 2), dim3(64), 0, s, d_w, ni, d_samples, d_labels, cap);
      e = hipMemsetAsync(d_unfinished, 0, sizeof(unsigned), s);
      if (e != hipSuccess) break;
      hipLaunchKernelGGL(k_cdw_post, dim3((ni + 63) / 64), dim3(64), 0, s, d_w, ni, d_samples, d_labels, cap, d_unfinished);
      e = hipMemcpyAsync(h_unfinished, d_unfinished, sizeof(unsigned), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess || *h_unfinished == 0) break;
      hipLaunchKernelGGL(k_cdw_assign, dim3(ni), dim3(256), 0, s, d_w, d_samples, d_labels, cap);
    }
    if (e == hipSuccess) {
      float* d_weights = d_dists;  // the distance slots are free now
      hipLaunchKernelGGL(k_cdw_weights, dim3(ni), dim3(256), 0, s, d_w, d_images, d_pos, d_weights, cap);
      hipLaunchKernelGGL(k_cdw_freq, dim3((ni + 1) / 2), dim3(64), 0, s, d_w, ni, d_weights, d_labels, cap);
      hipLaunchKernelGGL(k_cdw_finish, dim3((ni + 63) / 64), dim3(64), 0, s, d_w, ni, d_descs);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // the host tables above must outlive the copies
  if (e != hipSuccess) {
    set_last_error("color descriptors", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  for (void* p : {(void*)d_images, (void*)d_masks, (void*)d_tabs, (void*)d_samples, (void*)d_dists, (void*)d_pos,
                  (void*)d_labels, (void*)d_counts, (void*)d_w, (void*)d_unfinished})
    if (p) (void)cbh::free_async(p, s);
  if (h_unfinished) (void)hipHostFree(h_unfinished);
  return rc;
}

int launch_color_descriptors(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                             const uint32_t* img_h, const uint32_t* img_row_stride, int channels, uint8_t* d_descs,
                             uint8_t* d_ok, hipStream_t s) {
  // images are independent (a chain per image): chunks of at most `per` images, sized by the largest image's slots
  size_t cap = 64;
  for (size_t i = 0; i < n; ++i) {
    int cols = 0, rows = 0;
    resized_dims((int)img_w[i], (int)img_h[i], &cols, &rows);
    if (cols < 1 || rows < 1) return CBH_E_INVAL;
    cap = std::max(cap, (size_t)cols * (size_t)rows);
  }
  cap = (cap + 63) & ~(size_t)63;
  const size_t budget = (size_t)std::max(g_cd_chunk_mb, 64) << 20;
  size_t per = std::max<size_t>(budget / (33 * cap), 64);
  if (per >= 4096) per &= ~(size_t)2047;  // whole multiples of the launch shapes' image groups
  for (size_t i0 = 0; i0 < n; i0 += per) {
    const size_t m = std::min(per, n - i0);
    const int rc = launch_color_descriptors_chunk(d_imgs, m, img_off + i0, img_w + i0, img_h + i0, img_row_stride + i0,
                                                  channels, d_descs + i0 * 258, d_ok + i0, s);
    if (rc != CBH_OK) return rc;
  }
  return CBH_OK;
}

void set_cd_chunk_mb(int v) {
  if (v > 0) g_cd_chunk_mb = v;
}


void color_ellipse_mask(int cols, int rows, uint8_t* mask) { ellipse_mask(cols, rows, mask); }

}  // namespace cbh

extern "C" {

void cbh_color_descriptor_dims(int w, int h, int* cols, int* rows) { cbh::resized_dims(w, h, cols, rows); }

int cbh_color_ellipse_mask(int cols, int rows, uint8_t* mask) {
  if (cols < 1 || rows < 1 || cols > 8192 || rows > 8192 || !mask) return CBH_E_INVAL;
  cbh::color_ellipse_mask(cols, rows, mask);
  return CBH_OK;
}

int cbh_color_descriptors_dev(const void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                              const uint32_t* img_h, const uint32_t* img_row_stride, int channels, void* d_descs,
                              void* d_ok, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!d_imgs || !img_off || !img_w || !img_h || !img_row_stride || !d_descs || !d_ok || (channels != 3 && channels != 4) ||
      n > (1u << 24))
    return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (img_w[i] == 0 || img_h[i] == 0 || img_w[i] > 65535 || img_h[i] > 65535 ||
        img_row_stride[i] < img_w[i] * (uint32_t)channels)
      return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  return cbh::launch_color_descriptors((const uint8_t*)d_imgs, n, img_off, img_w, img_h, img_row_stride, channels,
                                       (uint8_t*)d_descs, (uint8_t*)d_ok, (hipStream_t)stream);
}

int cbh_color_descriptors(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                          const uint32_t* img_h, const uint32_t* img_row_stride, int channels, uint8_t* descs, uint8_t* ok,
                          int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!imgs || !descs || !ok || !img_off || !img_w || !img_h || !img_row_stride || (channels != 3 && channels != 4))
    return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (img_w[i] == 0 || img_h[i] == 0 ||
        img_off[i] + (uint64_t)(img_h[i] - 1) * img_row_stride[i] + (uint64_t)img_w[i] * channels > imgs_bytes)
      return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = nullptr;
  uint8_t *d_imgs = nullptr, *d_descs = nullptr, *d_ok = nullptr;
  hipError_t e;
  int rc = CBH_OK;
  if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(&d_imgs, imgs_bytes)) != hipSuccess || (e = hipMalloc(&d_descs, n * 258)) != hipSuccess ||
      (e = hipMalloc(&d_ok, n)) != hipSuccess ||
      (e = hipMemcpyAsync(d_imgs, imgs, imgs_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) {
    cbh::set_last_error("color descriptors setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  if (rc == CBH_OK)
    rc = cbh_color_descriptors_dev(d_imgs, n, img_off, img_w, img_h, img_row_stride, channels, d_descs, d_ok, device, s);
  if (rc == CBH_OK) {
    if ((e = hipMemcpyAsync(descs, d_descs, n * 258, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      cbh::set_last_error("color descriptors fetch", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) (void)hipStreamSynchronize(s);
  for (void* p : {(void*)d_imgs, (void*)d_descs, (void*)d_ok})
    if (p) (void)hipFree(p);
  if (s) cbh::stream_destroy(s);
  return rc;
}

}  // extern "C"
