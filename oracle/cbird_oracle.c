/* oracle/cbird_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of cbird's hash-build + Hamming-find hot path.  It is the
 * parity checker for the HIP kernels: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product (cbird_amd/, libcbird_hip.so)
 * never links, imports or falls back to anything in oracle/.
 *
 * Pinning status (see DESIGN.md "(c) Oracle", NOTES.md section 4):
 *   - search semantics (orc_scan64*, orc_hamm64): PINNED against the real reference
 *     VP-tree compiled in place (oracle/_ref, tests/test_oracle_ref.py) and against the
 *     committed golden vectors generated from it (tests/golden/vptree_*.json).
 *   - orc_dcthash64: "PARITY UNPINNED" versus the cbird binary.  The arithmetic below
 *     the call sites (cv::blur, cv::resize, cv::dct, cv::sum) lives in OpenCV 2.4.13.7
 *     (pinned by cbird.pri:148-152) which is not vendored in /root/reference and not
 *     installed here, and no reference test pins a hash value (unit/testcvutil.cpp:354
 *     has the check commented out).  Integer stages follow OpenCV 2.4 semantics as
 *     recalled in SURVEY.md section 8(a1).  The DCT and the sum exist in two evaluations
 *     behind orc_set_hash_variant(): 1 (default) = cv::dct's factorised float algorithm and
 *     cv::sum's grouping as recalled (oracle/cv_dct32.c, one labelled unit), 0 = the canonical
 *     separable f32 matrix form of NOTES.md section 3 (fixed fmaf order).  The GPU reproduces either
 *     bit for bit (knob "hash_dct"); tools/hash_at_risk.py bounds how many bits the choice,
 *     or any float evaluation of the same transform, can move; tools/gen_golden_opencv.cpp
 *     + tests/test_opencv_golden.py pin it the day someone runs it against the real library.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -mpopcnt).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_UNSUPPORTED (-2)
#define ORC_E_INVAL (-1)

/* ---- hamm64: src/hamm.h:24-26 ------------------------------------------------------ */
int orc_hamm64(uint64_t a, uint64_t b) { return __builtin_popcountll(a ^ b); }

/* ---- brute-force threshold scan: the semantics of DctHashIndex::find ---------------
 * src/dcthashindex.cpp:193-220.  The live code path is the exact VP-tree
 * (`_tree->search(target, p.dctThresh)`, :208); the compiled-out loop at :210-217 states
 * the same predicate directly: every entry i with hamm64(target, hash[i]) < thresh,
 * skipping mediaId 0 (slots nulled by remove(), :183-187).  target == 0 -> no results
 * (:196-200).  Output here is in haystack order; orc_find64 below sorts it.
 * Returns the full match count, writes at most cap entries. */
long long orc_scan64(const uint64_t* hashes, const uint32_t* ids, size_t n, uint64_t target,
                     int thresh, uint32_t* out_ids, int32_t* out_dist, size_t cap) {
  long long m = 0;
  if (target == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    int d = __builtin_popcountll(target ^ hashes[i]);
    if (d < thresh) {
      uint32_t id = ids[i];
      if (id != 0) {
        if ((size_t)m < cap) {
          out_ids[m] = id;
          out_dist[m] = d;
        }
        ++m;
      }
    }
  }
  return m;
}

typedef struct {
  int32_t dist;
  uint32_t id;
} orc_pair;

static int cmp_pair(const void* a, const void* b) {
  const orc_pair* x = (const orc_pair*)a;
  const orc_pair* y = (const orc_pair*)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}

/* find = scan + total order (score, mediaId).  The reference returns ascending distance
 * with heap-order ties (vptree.h:50-69) and searchIndex re-sorts by score with an
 * unstable std::sort (database.cpp:1729, index.h:284), so tie order is unspecified
 * there; the build fixes it to (score, mediaId) -- SURVEY.md section 7 hard part 2. */
long long orc_find64(const uint64_t* hashes, const uint32_t* ids, size_t n, uint64_t target,
                     int thresh, uint32_t* out_ids, int32_t* out_dist, size_t cap) {
  long long m = orc_scan64(hashes, ids, n, target, thresh, NULL, NULL, 0);
  if (m <= 0) return m;
  uint32_t* tid = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)m);
  int32_t* td = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
  orc_pair* p = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)m);
  orc_scan64(hashes, ids, n, target, thresh, tid, td, (size_t)m);
  for (long long i = 0; i < m; ++i) {
    p[i].dist = td[i];
    p[i].id = tid[i];
  }
  qsort(p, (size_t)m, sizeof(orc_pair), cmp_pair);
  for (long long i = 0; i < m && (size_t)i < cap; ++i) {
    out_ids[i] = p[i].id;
    out_dist[i] = p[i].dist;
  }
  free(p);
  free(tid);
  free(td);
  return m;
}

/* Batched find with a per-query cap: for each query the first min(count, k) matches in
 * (score, mediaId) order go to out_*[q*k ..], counts[q] = full count.  This is the
 * reference's find() followed by the sort + truncate of Database::searchIndex
 * (database.cpp:1729-1735) without the self filter. */
void orc_find64_batch(const uint64_t* hashes, const uint32_t* ids, size_t n, const uint64_t* q,
                      size_t nq, int thresh, int k, uint32_t* out_ids, int32_t* out_dist,
                      uint32_t* counts) {
  size_t cap = n ? n : 1;
  uint32_t* tid = (uint32_t*)malloc(sizeof(uint32_t) * cap);
  int32_t* td = (int32_t*)malloc(sizeof(int32_t) * cap);
  for (size_t i = 0; i < nq; ++i) {
    long long m = orc_find64(hashes, ids, n, q[i], thresh, tid, td, cap);
    counts[i] = (uint32_t)m;
    for (int j = 0; j < k; ++j) {
      if (j < m) {
        out_ids[i * (size_t)k + j] = tid[j];
        out_dist[i * (size_t)k + j] = td[j];
      } else {
        out_ids[i * (size_t)k + j] = 0;
        out_dist[i * (size_t)k + j] = 0;
      }
    }
  }
  free(tid);
  free(td);
}

/* Count-only all-pairs brute force (used for CPU "port" baseline timing and for
 * full-size property checks).  Returns sum over queries of match counts. */
long long orc_count64_pairs(const uint64_t* hashes, const uint32_t* ids, size_t n,
                            const uint64_t* q, size_t nq, int thresh) {
  long long total = 0;
  for (size_t j = 0; j < nq; ++j) {
    uint64_t t = q[j];
    if (t == 0) continue;
    for (size_t i = 0; i < n; ++i) {
      int d = __builtin_popcountll(t ^ hashes[i]);
      total += (d < thresh) & (ids[i] != 0);
    }
  }
  return total;
}

/* ---- dctHash64: src/cvutil.cpp:435-545 --------------------------------------------- */

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) (OpenCV 2.4 imgproc/filter.cpp). */
static int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0)
      p = -p;
    else
      p = 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

/* zig-zag order of the 9x9 low-frequency block, first step downwards; equals the
 * zigZag[81] table at cvutil.cpp:491-495 (checked in tests/test_oracle.py against the
 * (row,col) list in SURVEY.md 8(a1)). out[i] = row*9+col */
void orc_zigzag81(int* out) {
  int k = 0;
  for (int s = 0; s <= 16; ++s) {
    if (s & 1) {
      for (int r = (s < 8 ? s : 8); r >= 0 && s - r <= 8; --r) out[k++] = r * 9 + (s - r);
    } else {
      for (int r = (s > 8 ? s - 8 : 0); r <= 8 && r <= s; ++r) out[k++] = r * 9 + (s - r);
    }
  }
}

/* Orthonormal DCT-II basis rows 0..8 for N=32, evaluated in double and rounded once to
 * f32: C[k][j] = sqrt((k?2:1)/32) * cos(pi*(2j+1)*k/64).  (cv::dct computes the same
 * transform through a factorised float algorithm whose rounding cannot be reproduced
 * without its source: "parity unpinned".) */
void orc_dct9_table(float* c /* [9*32] */) {
  for (int k = 0; k < 9; ++k)
    for (int j = 0; j < 32; ++j) {
      double a = sqrt((k ? 2.0 : 1.0) / 32.0);
      c[k * 32 + j] = (float)(a * cos(M_PI * (2 * j + 1) * k / 64.0));
    }
}

/* kernel size rule, cvutil.cpp:446-455 */
int orc_blur_ksize(int w, int h) {
  long long area = (long long)w * h;
  if (area <= 32 * 32) return 0;
  if (area <= 64 * 64) return 3;
  if (area <= 128 * 128) return 5;
  return 7;
}

/* cv::blur(src, dst, Size(k,k)) on 8UC1: normalised box filter, centre anchor,
 * BORDER_REFLECT_101, integer window sum then saturate_cast<uchar>(sum * (1.0/k^2))
 * (double scale, cvRound).  k is odd so sum/k^2 never hits .5 -> plain nearest. */
static void box_blur_u8(const uint8_t* src, int w, int h, size_t stride, int k, uint8_t* dst) {
  /* separable (row sums, then sliding column sums) like OpenCV's RowSum/ColumnSum pair;
   * pure integer, so the result equals the direct k*k window sum */
  int r = k / 2;
  int area = k * k;
  uint16_t* rs = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * (size_t)h);
  int* xi = (int*)malloc(sizeof(int) * (size_t)(w + 2 * r));
  for (int x = -r; x < w + r; ++x) xi[x + r] = reflect101(x, w);
  for (int y = 0; y < h; ++y) {
    const uint8_t* row = src + (size_t)y * stride;
    uint16_t* o = rs + (size_t)y * w;
    int s = 0;
    for (int t = 0; t < k; ++t) s += row[xi[t]];
    o[0] = (uint16_t)s;
    for (int x = 1; x < w; ++x) {
      s += row[xi[x + k - 1]] - row[xi[x - 1]];
      o[x] = (uint16_t)s;
    }
  }
  int* cs = (int*)calloc((size_t)w, sizeof(int));
  for (int dy = -r; dy <= r; ++dy) {
    const uint16_t* o = rs + (size_t)reflect101(dy, h) * w;
    for (int x = 0; x < w; ++x) cs[x] += o[x];
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* d = dst + (size_t)y * w;
    for (int x = 0; x < w; ++x) d[x] = (uint8_t)((2 * cs[x] + area) / (2 * area));
    if (y + 1 < h) {
      const uint16_t* add = rs + (size_t)reflect101(y + 1 + r, h) * w;
      const uint16_t* sub = rs + (size_t)reflect101(y - r, h) * w;
      for (int x = 0; x < w; ++x) cs[x] += add[x] - sub[x];
    }
  }
  free(cs);
  free(xi);
  free(rs);
}

/* direct k*k window form, kept as an independent cross-check of the separable version
 * (tests/test_oracle.py) */
void orc_box_blur_direct(const uint8_t* src, int w, int h, size_t stride, int k, uint8_t* dst) {
  int r = k / 2;
  int area = k * k;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int dy = -r; dy <= r; ++dy) {
        const uint8_t* row = src + (size_t)reflect101(y + dy, h) * stride;
        for (int dx = -r; dx <= r; ++dx) s += row[reflect101(x + dx, w)];
      }
      dst[(size_t)y * w + x] = (uint8_t)((2 * s + area) / (2 * area));
    }
}

void orc_box_blur(const uint8_t* src, int w, int h, size_t stride, int k, uint8_t* dst) {
  box_blur_u8(src, w, h, stride, k, dst);
}

/* cv::resize(src, dst, Size(32,32), 0, 0, INTER_AREA) for w,h >= 32 (OpenCV 2.4 imgproc/imgwarp.cpp, as
 * recalled -- "parity unpinned"):
 *   both ratios integer ("area fast", resizeAreaFast_): block sum; 2x2 -> (a+b+c+d+2)>>2, otherwise
 *     saturate_cast<uchar>(sum * (1.f/area)) in float with cvRound (nearest-even);
 *   otherwise (resizeArea_): per axis a table of (src index, dst index, alpha) from
 *     computeResizeAreaTab -- a partially covered leading pixel with weight (sx1 - fsx1)/cellWidth when that
 *     exceeds 1e-3, fully covered pixels with 1/cellWidth, a trailing one with
 *     min(min(fsx2 - sx2, 1), cellWidth)/cellWidth when fsx2 - sx2 > 1e-3 (double arithmetic, alpha stored
 *     as float) -- then for every source row a float row buffer buf[dx] += S[sx]*alpha in table order, and
 *     per destination row sum[dx] = beta*buf[dx] for its first source row, += for the following ones,
 *     finally saturate_cast<uchar>(sum[dx]) (cvRound). */
typedef struct {
  int si, di;
  float alpha;
} orc_dalpha;

/* cv::resize derives the source/destination ratio as  inv_scale = (double)dsize/ssize;  scale = 1./inv_scale
 * -- two roundings, not ssize/dsize -- and takes the integer "area fast" path only when
 * |scale - saturate_cast<int>(scale)| < DBL_EPSILON on both axes.  The double rounding matters: for
 * ssize = 32*m with m = 49, 93, 98, 99, 103, ... the reciprocal of the reciprocal misses m by one ulp, so those
 * sizes go through the weighted tables; and for 3 widths up to 8192 (3885, 6734, 7770) a table weight changes. */
static double cv_resize_scale(int ssize, int dsize) {
  double inv_scale = (double)dsize / ssize;
  return 1. / inv_scale;
}
int orc_resize_area_fast(int w, int h) {
  double sx = cv_resize_scale(w, 32), sy = cv_resize_scale(h, 32);
  return w >= 32 && h >= 32 && fabs(sx - (double)lrint(sx)) < DBL_EPSILON && fabs(sy - (double)lrint(sy)) < DBL_EPSILON;
}

static int resize_area_tab(int ssize, int dsize, double scale, orc_dalpha* tab) {
  int k = 0;
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    if (sx1 - fsx1 > 1e-3) {
      tab[k].di = dx;
      tab[k].si = sx1 - 1;
      tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
    }
    for (int sx = sx1; sx < sx2; sx++) {
      tab[k].di = dx;
      tab[k].si = sx;
      tab[k++].alpha = (float)(1.0 / cellWidth);
    }
    if (fsx2 - sx2 > 1e-3) {
      double a = fsx2 - sx2;
      a = a < 1.0 ? a : 1.0;
      a = a < cellWidth ? a : cellWidth;
      tab[k].di = dx;
      tab[k].si = sx2;
      tab[k++].alpha = (float)(a / cellWidth);
    }
  }
  return k;
}

/* exported for the stage tests: the tables themselves */
int orc_resize_area_tab(int ssize, int dsize, int* si, int* di, float* alpha) {
  orc_dalpha* tab = (orc_dalpha*)malloc(sizeof(orc_dalpha) * (size_t)(ssize + 2 * dsize + 2));
  int k = resize_area_tab(ssize, dsize, cv_resize_scale(ssize, dsize), tab);
  for (int i = 0; i < k; ++i) {
    si[i] = tab[i].si;
    di[i] = tab[i].di;
    alpha[i] = tab[i].alpha;
  }
  free(tab);
  return k;
}

/* cv::resize(..., INTER_AREA) when an axis ENLARGES (scale < 1 on either axis; here: a side shorter than 32, e.g.
 * the 31-pixel keypoint rectangles of Media::makeKeyPointHashes).  "true area interpolation is only implemented
 * for the case (scale_x >= 1 && scale_y >= 1); in other cases it is emulated using some variant of bilinear
 * interpolation" (OpenCV 2.4 imgproc/imgwarp.cpp, as recalled -- "parity unpinned"): both axes then run the
 * 2-tap fixed-point resizer with area-mode coefficients
 *     s = floor(d*scale);  f = (float)((d+1) - (s+1)*inv_scale);  f = f <= 0 ? 0 : f - floor(f);
 *     x only: if (s + 1 >= ssize) f = 0, s = ssize-1            (y: the two source rows are clipped to the image)
 *     coefficients saturate_cast<short>((1-f)*2048), saturate_cast<short>(f*2048)            (cvRound: half-even)
 *   horizontal: D = S[s]*a0 + S[s+1]*a1 (int)
 *   vertical:   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2 */
typedef struct {
  int ofs[32];
  short c0[32], c1[32];
} orc_lin_tab;

static short sat_short_round(float v) {
  long r = lrintf(v);
  return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

static void resize_linear_tab(int ssize, int is_x, orc_lin_tab* t) {
  double inv_scale = (double)32 / ssize;
  double scale = 1. / inv_scale;
  for (int d = 0; d < 32; ++d) {
    int s = (int)floor(d * scale);
    float f = (float)((d + 1) - (s + 1) * inv_scale);
    f = f <= 0 ? 0.f : f - floorf(f);
    if (is_x && s + 1 >= ssize) {
      f = 0.f;
      s = ssize - 1;
    }
    t->ofs[d] = s;
    t->c0[d] = sat_short_round((1.f - f) * 2048.f);
    t->c1[d] = sat_short_round(f * 2048.f);
  }
}

/* exported for the stage tests */
void orc_resize_linear_tab(int ssize, int is_x, int* ofs, short* c0, short* c1) {
  orc_lin_tab t;
  resize_linear_tab(ssize, is_x, &t);
  for (int d = 0; d < 32; ++d) {
    ofs[d] = t.ofs[d];
    c0[d] = t.c0[d];
    c1[d] = t.c1[d];
  }
}

static void resize_linear_area32_u8(const uint8_t* src, int w, int h, size_t stride, uint8_t* dst /*32*32*/) {
  orc_lin_tab xt, yt;
  resize_linear_tab(w, 1, &xt);
  resize_linear_tab(h, 0, &yt);
  for (int dy = 0; dy < 32; ++dy) {
    int sy0 = yt.ofs[dy], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : sy0 > h - 1 ? h - 1 : sy0;
    sy1 = sy1 < 0 ? 0 : sy1 > h - 1 ? h - 1 : sy1;
    const uint8_t* S0 = src + (size_t)sy0 * stride;
    const uint8_t* S1 = src + (size_t)sy1 * stride;
    for (int dx = 0; dx < 32; ++dx) {
      int sx = xt.ofs[dx];
      int sx1 = sx + 1 < w ? sx + 1 : w - 1; /* its coefficient is 0 there */
      int D0 = S0[sx] * xt.c0[dx] + S0[sx1] * xt.c1[dx];
      int D1 = S1[sx] * xt.c0[dx] + S1[sx1] * xt.c1[dx];
      int v = (((yt.c0[dy] * (D0 >> 4)) >> 16) + ((yt.c1[dy] * (D1 >> 4)) >> 16) + 2) >> 2;
      dst[dy * 32 + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
}

static int area_resize32_u8(const uint8_t* src, int w, int h, uint8_t* dst /*32*32*/) {
  if (w == 32 && h == 32) {
    memcpy(dst, src, 1024);
    return ORC_OK;
  }
  if (w < 32 || h < 32) { /* an axis enlarges: the bilinear emulation, both axes */
    resize_linear_area32_u8(src, w, h, (size_t)w, dst);
    return ORC_OK;
  }
  if (orc_resize_area_fast(w, h)) {
    int sx = w / 32, sy = h / 32;
    float scale = 1.f / (float)(sx * sy);
    for (int y = 0; y < 32; ++y)
      for (int x = 0; x < 32; ++x) {
        int s = 0;
        for (int dy = 0; dy < sy; ++dy)
          for (int dx = 0; dx < sx; ++dx) s += src[(size_t)(y * sy + dy) * w + (x * sx + dx)];
        int v;
        if (sx == 2 && sy == 2)
          v = (s + 2) >> 2;
        else
          v = (int)lrintf((float)s * scale); /* default rounding mode = nearest-even */
        dst[y * 32 + x] = (uint8_t)(v > 255 ? 255 : v);
      }
    return ORC_OK;
  }
  orc_dalpha* xtab = (orc_dalpha*)malloc(sizeof(orc_dalpha) * (size_t)(w + 66));
  orc_dalpha* ytab = (orc_dalpha*)malloc(sizeof(orc_dalpha) * (size_t)(h + 66));
  int xn = resize_area_tab(w, 32, cv_resize_scale(w, 32), xtab);
  int yn = resize_area_tab(h, 32, cv_resize_scale(h, 32), ytab);
  float buf[32], sum[32];
  int prev_dy = ytab[0].di;
  for (int dx = 0; dx < 32; ++dx) sum[dx] = 0.f;
  for (int j = 0; j < yn; ++j) {
    float beta = ytab[j].alpha;
    int dy = ytab[j].di, sy = ytab[j].si;
    const uint8_t* S = src + (size_t)sy * w;
    for (int dx = 0; dx < 32; ++dx) buf[dx] = 0.f;
    for (int k = 0; k < xn; ++k) buf[xtab[k].di] += (float)S[xtab[k].si] * xtab[k].alpha;
    if (dy != prev_dy) {
      for (int dx = 0; dx < 32; ++dx) {
        long v = lrintf(sum[dx]);
        dst[prev_dy * 32 + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        sum[dx] = beta * buf[dx];
      }
      prev_dy = dy;
    } else {
      for (int dx = 0; dx < 32; ++dx) sum[dx] += beta * buf[dx];
    }
  }
  for (int dx = 0; dx < 32; ++dx) {
    long v = lrintf(sum[dx]);
    dst[prev_dy * 32 + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
  free(xtab);
  free(ytab);
  return ORC_OK;
}

/* Stages 3-6 on a 32x32 u8 tile (cvutil.cpp:475-545): DCT, zig-zag select 64, mean threshold, bits 1..63, 0 -> 1.
 * Two evaluations of the float arithmetic OpenCV owns (stage 3 cv::dct, stage 5 cv::sum), chosen by
 * orc_set_hash_variant():
 *   1 (default)  cv::dct / cv::sum as OpenCV 2.4.13.7 evaluates them, restated in the labelled unit oracle/cv_dct32.c
 *                (factorised float DCT via a 16-point complex FFT; sum in float groups of four, accumulated in double)
 *   0            canonical separable 9x32 matrix form with a fixed fmaf order, sequential double sum (NOTES.md 3)
 * The HIP library has the same switch (tuning knob "hash_dct").  Neither is pinned against the real library here
 * ("parity unpinned"); tools/hash_at_risk.py measures how often they -- and the float64 evaluation below -- disagree.
 * Also returns the 64 selected coefficients and the threshold when coefs != NULL (for at-risk-bit statistics). */
void orc_cv_dct32x32(float* m);
double orc_cv_sum_f32(const float* src, int len);

static int g_hash_variant = 1;
void orc_set_hash_variant(int v) { g_hash_variant = v ? 1 : 0; }
int orc_get_hash_variant(void) { return g_hash_variant; }

static void hash_tables(const float** C, const int** zz) {
  static float sC[9 * 32];
  static int szz[81];
  static int init = 0;
  if (!init) {
    orc_dct9_table(sC);
    orc_zigzag81(szz);
    init = 1;
  }
  *C = sC;
  *zz = szz;
}

uint64_t orc_hash_from_tile32_v(const uint8_t* tile, int variant, float* coefs /*64 or NULL*/, float* thr_out) {
  const float* C;
  const int* zz;
  hash_tables(&C, &zz);
  float Y[81];
  float sel[64];
  float thresh;
  if (variant) {
    float m[1024]; /* gray.convertTo(freq, CV_32F); cv::dct(freq, freq) */
    for (int i = 0; i < 1024; ++i) m[i] = (float)tile[i];
    orc_cv_dct32x32(m);
    for (int u = 0; u < 9; ++u)
      for (int k = 0; k < 9; ++k) Y[u * 9 + k] = m[u * 32 + k];
    for (int i = 0; i < 64; ++i) sel[i] = Y[zz[6 + i]];
    float sum = (float)orc_cv_sum_f32(sel, 64); /* float sum = float(cv::sum(freq)[0]) */
    thresh = sum / 64;
  } else {
    float T[32][9]; /* row pass: T[r][k] = sum_j X[r][j] * C[k][j], j ascending */
    for (int r = 0; r < 32; ++r)
      for (int k = 0; k < 9; ++k) {
        float acc = 0.f;
        for (int j = 0; j < 32; ++j) acc = fmaf((float)tile[r * 32 + j], C[k * 32 + j], acc);
        T[r][k] = acc;
      }
    /* col pass: Y[u][k] = sum_r C[u][r] * T[r][k], r ascending */
    for (int u = 0; u < 9; ++u)
      for (int k = 0; k < 9; ++k) {
        float acc = 0.f;
        for (int r = 0; r < 32; ++r) acc = fmaf(C[u * 32 + r], T[r][k], acc);
        Y[u * 9 + k] = acc;
      }
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) {
      sel[i] = Y[zz[6 + i]];
      sum += (double)sel[i];
    }
    thresh = (float)sum / 64;
  }
  uint64_t hash = 0;
  for (int i = 1; i < 64; ++i)
    if (sel[i] > thresh) hash |= 1ULL << i;
  if (hash == 0) hash = 1;
  if (coefs) memcpy(coefs, sel, sizeof(sel));
  if (thr_out) *thr_out = thresh;
  return hash;
}

uint64_t orc_hash_from_tile32(const uint8_t* tile, float* coefs /*64 or NULL*/, float* thr_out) {
  return orc_hash_from_tile32_v(tile, g_hash_variant, coefs, thr_out);
}

/* The same stages in float64 with the exact basis (cos in double, products and sums in double, no float rounding
 * anywhere): the yardstick of tools/hash_at_risk.py -- a bit is "at risk" when its coefficient is so close to the
 * threshold that float evaluations of the transform may put it on either side. */
uint64_t orc_hash_from_tile32_f64(const uint8_t* tile, double* coefs /*64 or NULL*/, double* thr_out) {
  static double Cd[9 * 32];
  static int init = 0;
  const float* Cf;
  const int* zz;
  hash_tables(&Cf, &zz);
  if (!init) {
    for (int k = 0; k < 9; ++k)
      for (int j = 0; j < 32; ++j) Cd[k * 32 + j] = sqrt((k ? 2.0 : 1.0) / 32.0) * cos(M_PI * (2 * j + 1) * k / 64.0);
    init = 1;
  }
  double T[32][9], Y[81], sel[64], sum = 0.0;
  for (int r = 0; r < 32; ++r)
    for (int k = 0; k < 9; ++k) {
      double acc = 0;
      for (int j = 0; j < 32; ++j) acc += (double)tile[r * 32 + j] * Cd[k * 32 + j];
      T[r][k] = acc;
    }
  for (int u = 0; u < 9; ++u)
    for (int k = 0; k < 9; ++k) {
      double acc = 0;
      for (int r = 0; r < 32; ++r) acc += Cd[u * 32 + r] * T[r][k];
      Y[u * 9 + k] = acc;
    }
  for (int i = 0; i < 64; ++i) {
    sel[i] = Y[zz[6 + i]];
    sum += sel[i];
  }
  const double thresh = sum / 64;
  uint64_t hash = 0;
  for (int i = 1; i < 64; ++i)
    if (sel[i] > thresh) hash |= 1ULL << i;
  if (hash == 0) hash = 1;
  if (coefs) memcpy(coefs, sel, sizeof(sel));
  if (thr_out) *thr_out = thresh;
  return hash;
}

/* batch helper for the statistics tool: n tiles -> hashes under variant v (0, 1) or the float64 evaluation (2), plus
 * per tile the smallest |coef - thresh| over bits 1..63 (in the evaluation's own arithmetic) */
void orc_hash_tiles_stats(const uint8_t* tiles, size_t n, int v, uint64_t* hashes, double* min_margin) {
  for (size_t t = 0; t < n; ++t) {
    double mm = INFINITY;
    if (v == 2) {
      double c[64], th;
      hashes[t] = orc_hash_from_tile32_f64(tiles + t * 1024, c, &th);
      for (int i = 1; i < 64; ++i) mm = fmin(mm, fabs(c[i] - th));
    } else {
      float c[64], th;
      hashes[t] = orc_hash_from_tile32_v(tiles + t * 1024, v, c, &th);
      for (int i = 1; i < 64; ++i) mm = fmin(mm, fabs((double)c[i] - (double)th));
    }
    if (min_margin) min_margin[t] = mm;
  }
}

/* tools/hash_at_risk.py: all three evaluations of one tile side by side.  Per tile: the three hashes; the smallest
 * float64 margin |coef - thresh| over bits 1..63 and its bit; the largest deviation of a float evaluation's
 * (coef - thresh) from the float64 one, per variant.  A bit can only come out differently in ANY float evaluation of
 * the same transform whose error stays below eps if its float64 margin is below eps: that is the at-risk set. */
void orc_hash_tiles_risk(const uint8_t* tiles, size_t n, uint64_t* h0, uint64_t* h1, uint64_t* h2,
                         double* min_margin, int32_t* min_bit, double* err0, double* err1, double* thr64) {
  for (size_t t = 0; t < n; ++t) {
    float c0[64], c1[64], t0, t1;
    double c2[64], t2;
    h0[t] = orc_hash_from_tile32_v(tiles + t * 1024, 0, c0, &t0);
    h1[t] = orc_hash_from_tile32_v(tiles + t * 1024, 1, c1, &t1);
    h2[t] = orc_hash_from_tile32_f64(tiles + t * 1024, c2, &t2);
    double mm = INFINITY, e0 = 0, e1 = 0;
    int mb = 0;
    for (int i = 1; i < 64; ++i) {
      const double d2 = c2[i] - t2;
      if (fabs(d2) < mm) mm = fabs(d2), mb = i;
      e0 = fmax(e0, fabs(((double)c0[i] - (double)t0) - d2));
      e1 = fmax(e1, fabs(((double)c1[i] - (double)t1) - d2));
    }
    min_margin[t] = mm, min_bit[t] = mb, err0[t] = e0, err1[t] = e1, thr64[t] = t2;
  }
}

/* Full pipeline for one 8UC1 image.  Returns ORC_OK / ORC_E_*; hash in *out. */
int orc_dcthash64(const uint8_t* img, int w, int h, size_t stride, uint64_t* out) {
  if (!img || w <= 0 || h <= 0 || stride < (size_t)w) return ORC_E_INVAL;
  int k = orc_blur_ksize(w, h);
  uint8_t* blur = (uint8_t*)malloc((size_t)w * h);
  if (k)
    box_blur_u8(img, w, h, stride, k, blur);
  else
    for (int y = 0; y < h; ++y) memcpy(blur + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  uint8_t tile[1024];
  int rc = area_resize32_u8(blur, w, h, tile);
  free(blur);
  if (rc != ORC_OK) return rc;
  *out = orc_hash_from_tile32(tile, NULL, NULL);
  return ORC_OK;
}

/* exposes the intermediate 32x32 tile (blur + area resize) for stage-level tests */
int orc_dcthash_tile32(const uint8_t* img, int w, int h, size_t stride, uint8_t* tile) {
  if (!img || w <= 0 || h <= 0 || stride < (size_t)w) return ORC_E_INVAL;
  int k = orc_blur_ksize(w, h);
  uint8_t* blur = (uint8_t*)malloc((size_t)w * h);
  if (k)
    box_blur_u8(img, w, h, stride, k, blur);
  else
    for (int y = 0; y < h; ++y) memcpy(blur + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
  int rc = area_resize32_u8(blur, w, h, tile);
  free(blur);
  return rc;
}

int orc_dcthash64_batch(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride,
                        size_t img_stride, uint64_t* out) {
  for (size_t i = 0; i < n; ++i) {
    int rc = orc_dcthash64(imgs + i * img_stride, w, h, row_stride, out + i);
    if (rc != ORC_OK) return rc;
  }
  return ORC_OK;
}

/* ---- sizeLongestSide: src/cvutil.cpp:1932-1950 (called at scanner.cpp:876 before ORB) -------------------------
 * Target size from the float aspect ratio (:1934-1942), then cv::resize(..., INTER_LANCZOS4), the default filter
 * (cvutil.h:251).  cv::resize's 8-bit Lanczos path (OpenCV 2.4 imgproc/imgwarp.cpp, as recalled -- "parity
 * unpinned"): per destination column  fx = (float)((dx+0.5)*scale_x - 0.5); sx = floor(fx); fx -= sx;  eight taps
 * sx-3..sx+4 with interpolateLanczos4(fx) weights normalised in float, converted to fixed point
 * saturate_cast<short>(w * 2048); tap positions outside the image are clamped to the edge; rows likewise.
 * Horizontal pass in int (sum S*alpha), vertical pass in int (sum D*beta), result (v + 2^21) >> 22 saturated --
 * all integer, so any evaluation order gives the same bytes.  There is no anti-aliasing: a 10x reduction still reads
 * 8x8 source pixels per output pixel. */
void orc_longest_side_dims(int w, int h, int size, int* ow, int* oh) {
  float aspect = (float)w / h;
  if (w > h) {
    *ow = size;
    *oh = (int)(size / aspect);
  } else {
    *oh = size;
    *ow = (int)(aspect * size);
  }
}

static void lanczos4_coeffs(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) coeffs[i] = 0;
    coeffs[3] = 1;
    return;
  }
  float sum = 0;
  double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; i++) {
    double y = -(x + 3 - i) * M_PI * 0.25;
    coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += coeffs[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

/* per destination index: first source index (the tap "3" position) and 8 fixed-point weights */
void orc_lanczos4_tab(int ssize, int dsize, int* ofs, short* coef /* dsize*8 */) {
  double inv_scale = (double)dsize / ssize;
  double scale = 1. / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    float c[8];
    lanczos4_coeffs(fx, c);
    ofs[d] = sx;
    for (int k = 0; k < 8; ++k) coef[d * 8 + k] = sat_short_round(c[k] * 2048.f);
  }
}

int orc_resize_lanczos4_u8(const uint8_t* src, int w, int h, size_t stride, int dw, int dh, uint8_t* dst) {
  if (!src || !dst || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || stride < (size_t)w) return ORC_E_INVAL;
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  int* yofs = (int*)malloc(sizeof(int) * (size_t)dh);
  short* xa = (short*)malloc(sizeof(short) * 8 * (size_t)dw);
  short* yb = (short*)malloc(sizeof(short) * 8 * (size_t)dh);
  orc_lanczos4_tab(w, dw, xofs, xa);
  orc_lanczos4_tab(h, dh, yofs, yb);
  for (int dy = 0; dy < dh; ++dy)
    for (int dx = 0; dx < dw; ++dx) {
      int v = 0;
      for (int k = 0; k < 8; ++k) {
        int sy = yofs[dy] - 3 + k;
        sy = sy < 0 ? 0 : sy > h - 1 ? h - 1 : sy;
        const uint8_t* S = src + (size_t)sy * stride;
        int D = 0;
        for (int j = 0; j < 8; ++j) {
          int sx = xofs[dx] - 3 + j;
          sx = sx < 0 ? 0 : sx > w - 1 ? w - 1 : sx;
          D += S[sx] * xa[dx * 8 + j];
        }
        v += D * yb[dy * 8 + k];
      }
      v = (v + (1 << 21)) >> 22;
      dst[(size_t)dy * dw + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(xofs);
  free(yofs);
  free(xa);
  free(yb);
  return ORC_OK;
}

/* ---- Media::makeKeyPointHashes: src/media.cpp:874-923 -----------------------------------------------
 * Rectangles: keypoints with size >= 31 whose square (pt, pt + size) lies inside (0, cols-2) x (0, rows-2)
 * (float comparisons, :887-894) become Rect(floor x, floor y, ceil size, ceil size) -- anchored at the keypoint,
 * not centred on it (:896-900).  kp = (x, y, size) triples.  Returns the number of rectangles; rects = x,y,s. */
int orc_keypoint_rects(int cols, int rows, const float* kp, int nkp, int* rects) {
  int n = 0;
  for (int i = 0; i < nkp; ++i) {
    float size = kp[3 * i + 2];
    if (size < 31) continue;
    float x0 = kp[3 * i], y0 = kp[3 * i + 1];
    float x1 = x0 + size, y1 = y0 + size;
    if (x0 > 0 && y0 > 0 && x1 < cols - 2 && y1 < rows - 2) {
      rects[3 * n] = (int)floorf(x0);
      rects[3 * n + 1] = (int)floorf(y0);
      rects[3 * n + 2] = (int)ceilf(size);
      ++n;
    }
  }
  return n;
}

/* cv::blur on a sub-rectangle VIEW of a larger image: the filter engine is not "isolated", it takes the pixels
 * around the rectangle from the parent image (FilterEngine::start -> Mat::locateROI, OpenCV 2.4 filter.cpp, as
 * recalled -- "parity unpinned") and applies BORDER_REFLECT_101 only at the parent's own edges. */
static void box_blur_roi_u8(const uint8_t* parent, int W, int H, size_t stride, int x, int y, int rw, int rh, int k,
                            uint8_t* dst /* rw*rh */) {
  int r = k / 2, area = k * k;
  for (int i = 0; i < rh; ++i)
    for (int j = 0; j < rw; ++j) {
      int s = 0;
      for (int dy = -r; dy <= r; ++dy) {
        const uint8_t* row = parent + (size_t)reflect101(y + i + dy, H) * stride;
        for (int dx = -r; dx <= r; ++dx) s += row[reflect101(x + j + dx, W)];
      }
      dst[(size_t)i * rw + j] = (uint8_t)((2 * s + area) / (2 * area));
    }
}

/* dctHash64(sub, inPlace = true) for one rectangle of `img` (cvutil.cpp:435-545 with :457-463: 8UC1 input is not
 * copied, so cv::blur writes its result back into the caller's image -- the engine buffers source rows, the result
 * is the blur of the pixels as they were before the call).  img is MODIFIED when the rectangle is blurred. */
static int dcthash64_rect(uint8_t* img, int W, int H, size_t stride, int x, int y, int rw, int rh, int write_back,
                          uint64_t* out) {
  if (!img || x < 0 || y < 0 || rw <= 0 || rh <= 0 || x + rw > W || y + rh > H) return ORC_E_INVAL;
  int k = orc_blur_ksize(rw, rh);
  uint8_t* sub = (uint8_t*)malloc((size_t)rw * rh);
  if (k) {
    box_blur_roi_u8(img, W, H, stride, x, y, rw, rh, k, sub);
    if (write_back)
      for (int i = 0; i < rh; ++i) memcpy(img + (size_t)(y + i) * stride + x, sub + (size_t)i * rw, (size_t)rw);
  } else {
    for (int i = 0; i < rh; ++i) memcpy(sub + (size_t)i * rw, img + (size_t)(y + i) * stride + x, (size_t)rw);
  }
  uint8_t tile[1024];
  int rc = area_resize32_u8(sub, rw, rh, tile);
  free(sub);
  if (rc != ORC_OK) return rc;
  *out = orc_hash_from_tile32(tile, NULL, NULL);
  return ORC_OK;
}
int orc_dcthash64_rect_inplace(uint8_t* img, int W, int H, size_t stride, int x, int y, int rw, int rh,
                               uint64_t* out) {
  return dcthash64_rect(img, W, H, stride, x, y, rw, rh, 1, out);
}
/* dctHash64(view) with inPlace = false: the same view semantics (blur border from the parent), image untouched.
 * This is what Scanner::processImage computes after autocrop(), which narrows cvGray to a colRange/rowRange VIEW of
 * the full image (cvutil.cpp:1397-1401) -- the cropped-away margins still feed the blur at the view's edges. */
int orc_dcthash64_view(const uint8_t* img, int W, int H, size_t stride, int x, int y, int rw, int rh, uint64_t* out) {
  return dcthash64_rect((uint8_t*)img, W, H, stride, x, y, rw, rh, 0, out);
}

/* the whole of makeKeyPointHashes for one image: rectangles in keypoint order, each hashed in place, so a later
 * rectangle sees the blurred pixels an earlier, overlapping one left behind (media.cpp:904-910).  Returns the
 * number of hashes written (<= nkp) or a negative error. */
int orc_keypoint_hashes(uint8_t* img, int W, int H, size_t stride, const float* kp, int nkp, uint64_t* out) {
  if (!img || W <= 0 || H <= 0 || stride < (size_t)W || nkp < 0) return ORC_E_INVAL;
  int* rects = (int*)malloc(sizeof(int) * 3 * (size_t)(nkp > 0 ? nkp : 1));
  int n = orc_keypoint_rects(W, H, kp, nkp, rects);
  for (int i = 0; i < n; ++i) {
    int rc = orc_dcthash64_rect_inplace(img, W, H, stride, rects[3 * i], rects[3 * i + 1], rects[3 * i + 2],
                                        rects[3 * i + 2], out + i);
    if (rc != ORC_OK) {
      free(rects);
      return rc;
    }
  }
  free(rects);
  return n;
}

/* ---- DctFeaturesIndex::find: src/dctfeaturesindex.cpp:260-358 ---------------------------------
 * entries = the HammingTree values (mediaId, hash), several per media; removed entries keep their
 * hash with id 0 (hammingtree.h:349-361).  Candidates are taken by exact brute force (the reference
 * tree only visits the needle's own leaf, hammingtree.h:248-252, so it returns a subset once the tree
 * has split; while it is a single leaf of <= 8192 entries it returns exactly these).  Per needle
 * hash: candidates sorted by distance -- the reference's std::sort leaves equal distances in
 * unspecified order, fixed here to (distance, mediaId) -- first 10 kept (:301-303), id 0 skipped
 * AFTER the cut (:308), votes and distance sums per mediaId (:314-323), maxMatches over ids other than
 * the needle (:325), results in ascending mediaId (QMap) with the score rule of :334-355.
 * Returns the number of results. */
typedef struct {
  int32_t dist;
  uint32_t id;
} orc_cand;

static int cmp_cand(const void* a, const void* b) {
  const orc_cand* x = (const orc_cand*)a;
  const orc_cand* y = (const orc_cand*)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}

static int cmp_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : x > y;
}

long long orc_fdct_find_masked(const uint64_t* hashes, const uint32_t* ids, size_t n, const uint64_t* nhash,
                               const uint64_t* nmask, size_t nn, uint32_t needle_id, int thresh,
                               uint32_t* out_ids, int32_t* out_scores, size_t cap);

/* ---- HammingTree leaf of a needle: src/tree/hammingtree.h --------------------------------------------
 * insert() (:366-425): a leaf at depth d that would hold more than CLUSTER_SIZE / sizeof(hash) = 8192 values
 * becomes internal and splits on bit d (getBit(depth) = depth, :243; partition tests hash & (1 << bit), :234-241);
 * its children are filled by the same rule.  Counts only grow (remove() keeps the hashes, :347-364), so node
 * (d, prefix) is internal iff more than 8192 stored hashes have that low-d-bit prefix.  search() (:244-252)
 * follows the needle's bits to one leaf and scans only it.
 * out_masks[j] = (1 << depth of needle j's leaf) - 1: the entries a search can see are those with
 * ((hash ^ needle) & mask) == 0.  Deliberately the dumbest possible statement: recount per level. */
void orc_htree_leaf_masks(const uint64_t* hashes, size_t n, const uint64_t* q, size_t nq, uint64_t* out_masks) {
  for (size_t j = 0; j < nq; ++j) {
    int d = 0;
    for (; d < 58; ++d) {
      const uint64_t m = (d == 0) ? 0 : ((1ull << d) - 1);
      size_t cnt = 0;
      for (size_t i = 0; i < n; ++i) cnt += ((hashes[i] ^ q[j]) & m) == 0;
      if (cnt <= 8192) break; /* this node is a leaf */
    }
    out_masks[j] = (d == 0) ? 0 : ((1ull << d) - 1);
  }
}

long long orc_fdct_find(const uint64_t* hashes, const uint32_t* ids, size_t n, const uint64_t* nhash,
                        size_t nn, uint32_t needle_id, int thresh, uint32_t* out_ids,
                        int32_t* out_scores, size_t cap) {
  return orc_fdct_find_masked(hashes, ids, n, nhash, 0, nn, needle_id, thresh, out_ids, out_scores, cap);
}

/* nmask (optional): the equal-bits mask of every needle hash (tree-compatible candidates) */
long long orc_fdct_find_masked(const uint64_t* hashes, const uint32_t* ids, size_t n, const uint64_t* nhash,
                               const uint64_t* nmask, size_t nn, uint32_t needle_id, int thresh,
                               uint32_t* out_ids, int32_t* out_scores, size_t cap) {
  /* votes: at most 10 per needle hash */
  size_t maxv = nn * 10 + 1;
  uint32_t* vid = (uint32_t*)malloc(sizeof(uint32_t) * maxv);
  int32_t* vdist = (int32_t*)malloc(sizeof(int32_t) * maxv);
  orc_cand* cand = (orc_cand*)malloc(sizeof(orc_cand) * (n ? n : 1));
  size_t nv = 0;
  for (size_t j = 0; j < nn; ++j) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
      int d = __builtin_popcountll(nhash[j] ^ hashes[i]);
      if (d < thresh && (!nmask || ((nhash[j] ^ hashes[i]) & nmask[j]) == 0)) {
        cand[m].dist = d;
        cand[m].id = ids[i];
        ++m;
      }
    }
    qsort(cand, m, sizeof(orc_cand), cmp_cand);
    size_t len = m < 10 ? m : 10;
    for (size_t k = 0; k < len; ++k) {
      if (cand[k].id == 0) continue; /* "zero index means deleted" */
      vid[nv] = cand[k].id;
      vdist[nv] = cand[k].dist;
      ++nv;
    }
  }
  /* unique ids ascending */
  uint32_t* uid = (uint32_t*)malloc(sizeof(uint32_t) * (nv ? nv : 1));
  memcpy(uid, vid, sizeof(uint32_t) * nv);
  qsort(uid, nv, sizeof(uint32_t), cmp_u32);
  size_t nu = 0;
  for (size_t i = 0; i < nv; ++i)
    if (i == 0 || uid[i] != uid[i - 1]) uid[nu++] = uid[i];
  uint32_t* cnt = (uint32_t*)calloc(nu ? nu : 1, sizeof(uint32_t));
  int* sum = (int*)calloc(nu ? nu : 1, sizeof(int));
  for (size_t i = 0; i < nv; ++i) {
    uint32_t* p = (uint32_t*)bsearch(&vid[i], uid, nu, sizeof(uint32_t), cmp_u32);
    size_t u = (size_t)(p - uid);
    cnt[u] += 1;
    sum[u] += vdist[i];
  }
  uint32_t maxMatches = 0;
  for (size_t u = 0; u < nu; ++u)
    if (uid[u] != needle_id && cnt[u] > maxMatches) maxMatches = cnt[u];
  long long r = 0;
  for (size_t u = 0; u < nu; ++u) {
    int score;
    float avgScore = (float)sum[u] / (float)cnt[u];
    if (uid[u] == needle_id)
      score = -1;
    else if (maxMatches == 1)
      score = (int)(10 * avgScore);
    else
      score = (int)(maxMatches - cnt[u]);
    if ((size_t)r < cap) {
      out_ids[r] = uid[u];
      out_scores[r] = score;
    }
    ++r;
  }
  free(sum);
  free(cnt);
  free(uid);
  free(cand);
  free(vdist);
  free(vid);
  return r;
}

/* ---- VideoIndex .vdx v2 codec: src/videoindex.cpp:271-429 -------------------------------------
 * header "cbird video index:<cbird version>:2:<byte order>:1:8:<numFrames>:\n" (byte order =
 * QSysInfo::ByteOrder, 1 on little endian), u32 packedLen, packed frame deltas (first byte = frame 0;
 * each delta as 7-bit groups LSB first, bit 7 set on every group but the last), zero padding so that the
 * hashes start 8-byte aligned relative to the file start, u64 hashes[numFrames], trailer "cbir".
 * Returns the encoded size (0 on error: first frame must be 0, frames strictly increasing); writes at
 * most cap bytes. */
size_t orc_vdx_encode(const int32_t* frames, const uint64_t* hashes, size_t n, const char* version,
                      uint8_t* out, size_t cap) {
  char header[256];
  int hl = snprintf(header, sizeof header, "cbird video index:%s:%d:%d:%d:%d:%zu:\n", version, 2, 1, 1, 8, n);
  size_t pos = 0;
#define PUT(ptr, len)                                              \
  do {                                                             \
    if (pos + (len) <= cap) memcpy(out + pos, (ptr), (len));       \
    pos += (len);                                                  \
  } while (0)
  PUT(header, (size_t)hl);
  if (n == 0) return pos;
  if (frames[0] != 0) return 0;
  uint8_t* packed = (uint8_t*)malloc(n * 5 + 16);
  size_t pl = 0;
  int prev = frames[0];
  int nextByte = prev;
  for (size_t i = 1; i < n; ++i) {
    int offset = frames[i] - prev;
    prev = frames[i];
    if (offset < 1) {
      free(packed);
      return 0;
    }
    while (offset > 0) {
      packed[pl++] = (uint8_t)nextByte;
      int lsb = offset & 0x7F;
      offset >>= 7;
      nextByte = lsb | (offset == 0 ? 0x00 : 0x80);
    }
  }
  packed[pl++] = (uint8_t)nextByte;
  uint32_t len = (uint32_t)pl;
  PUT(&len, 4);
  size_t here = (size_t)hl + 4 + pl;
  size_t pad = 8 - (here % 8);
  if (pad == 8) pad = 0;
  memset(packed + pl, 0, pad);
  PUT(packed, pl + pad);
  PUT(hashes, n * 8);
  PUT("cbir", 4);
#undef PUT
  free(packed);
  return pos;
}

/* load_v2 (:349-429) + the trailer check of verify_v2 (:260-268).  Returns the number of frames or a
 * negative error: -1 bad header, -2 truncated/corrupt, -3 missing trailer, -4 capacity. */
long long orc_vdx_decode(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap) {
  size_t nl = 0;
  while (nl < len && nl < 255 && buf[nl] != '\n') ++nl;
  if (nl >= len || buf[nl] != '\n') return -1;
  /* split on ':' -> 8 fields, last one is "\n" */
  const char* f[8];
  size_t fl[8];
  int nf = 0;
  size_t start = 0;
  for (size_t i = 0; i <= nl && nf < 8; ++i)
    if (i == nl || buf[i] == ':') {
      f[nf] = (const char*)buf + start;
      fl[nf] = i - start;
      ++nf;
      start = i + 1;
    }
  if (nf != 7 + 0 && nf != 8) return -1;
  if (nf == 7) return -1; /* "a:b:c:d:e:f:g:\n" splits into 8 with the final empty/newline field */
  if (fl[0] != 17 || memcmp(f[0], "cbird video index", 17) != 0) return -1;
  if (atoi(f[2]) != 2 || atoi(f[4]) != 1 || atoi(f[5]) != 8 || atoi(f[3]) != 1) return -1;
  unsigned long numFrames = strtoul(f[6], NULL, 10);
  size_t hdr = nl + 1;
  if (numFrames == 0) return 0;
  int reduced = 0;
  if (numFrames > (1u << 24)) numFrames = 1u << 24, reduced = 1; /* MAX_FRAMES_PER_VIDEO (:366-370) */
  if (numFrames > cap) return -4;
  if (hdr + 4 > len) return -2;
  uint32_t packedLen;
  memcpy(&packedLen, buf + hdr, 4);
  if (packedLen < numFrames || hdr + 4 + packedLen > len) return -2;
  const uint8_t* packed = buf + hdr + 4;
  int frame = 0, jump = 0, shift = 0;
  size_t nfr = 0;
  for (uint32_t i = 0; i < packedLen; ++i) {
    uint8_t byte = packed[i];
    if (0 == (byte & 0x80)) {
      frame += jump | (byte << shift);
      jump = 0;
      shift = 0;
      if (nfr < numFrames) frames[nfr] = frame;
      ++nfr;
      if (reduced && nfr == numFrames) break; /* videoindex.cpp:395 */
    } else {
      jump |= (byte & 0x7F) << shift;
      shift += 7;
    }
  }
  if (jump) return -2;
  if (nfr != numFrames) return -2;
  size_t here = hdr + 4 + packedLen;
  size_t pad = 8 - (here % 8);
  if (pad == 8) pad = 0;
  if (here + pad + numFrames * 8 > len) return -2;
  memcpy(hashes, buf + here + pad, numFrames * 8);
  /* load_v2 (:350-429) stops here: the "cbir" trailer is verify_v2's business (:248-269), see orc_vdx_verify */
  return (long long)numFrames;
}

/* VideoIndex::verify_v2 (:248-269) = isValid(): header as load_v2 checks it, then -- unless the file stores 0 frames --
 * the last four bytes must be "cbir".  1 = valid. */
int orc_vdx_verify(const uint8_t* buf, size_t len) {
  size_t nl = 0;
  while (nl < len && nl < 255 && buf[nl] != '\n') ++nl;
  if (nl >= len) return 0;
  int colons = 0;
  for (size_t i = 0; i < nl; ++i) colons += buf[i] == ':';
  if (colons != 7 || nl < 18 || memcmp(buf, "cbird video index:", 18) != 0) return 0;
  const char* p = (const char*)buf + 18;
  const char* f[6];
  int nf = 0;
  f[nf++] = p;
  for (size_t i = 18; i < nl && nf < 6; ++i)
    if (buf[i] == ':') f[nf++] = (const char*)buf + i + 1;
  if (nf < 6) return 0;
  if (atoi(f[1]) != 2 || atoi(f[2]) != 1 || atoi(f[3]) != 1 || atoi(f[4]) != 8) return 0;
  if (strtoul(f[5], NULL, 10) == 0) return 1;
  return len >= nl + 5 && memcmp(buf + len - 4, "cbir", 4) == 0;
}

/* ---- DctVideoIndex: src/dctvideoindex.cpp ---------------------------------------------------------
 * insertHashes filter (:61-111): drop hashes with < 5 ones or < 5 zeros (:89); when skip != 0 and
 * lastFrame/2 > skip drop frames < skip or > lastFrame - skip (:93-95).  keep[i] = 1 if entry i enters the
 * search tree.  Returns the number kept. */
size_t orc_video_insert_filter(const int32_t* frames, const uint64_t* hashes, size_t n, int skip,
                               uint8_t* keep) {
  size_t m = 0;
  if (n == 0) return 0;
  int lastFrame = frames[n - 1];
  for (size_t j = 0; j < n; ++j) {
    keep[j] = 0;
    int pc = __builtin_popcountll(hashes[j]);
    if (pc < 5 || 64 - pc < 5) continue;
    if (skip && lastFrame / 2 > skip) {
      if (frames[j] < skip || frames[j] > lastFrame - skip) continue;
    }
    keep[j] = 1;
    ++m;
  }
  return m;
}

/* RadixMap::indexOf (src/tree/radix.h:135-141) */
static size_t radix_index_of(uint64_t hash, unsigned radix) {
  uint64_t mask = radix >= 64 ? ~0ull : ((1ull << radix) - 1);
  return (size_t)((hash >> 1) & mask);
}

typedef struct {
  uint32_t id;
  int32_t score, src_in, dst_in, len;
} orc_vmatch;

/* findFrame (:291-387): entries = tree values in insertion order (video index ascending, file order
 * inside a video): evidx[], eframe[], ehash[]; mediaIds[] = _mediaId.  radix 0 = exact scan, > 0 = only
 * the needle's bucket (reference approximation).  One result per matched video = its closest frame, first
 * in scan order among equals (strict <, :351); results ascending video index (QMap). */
long long orc_video_find_frame(const uint32_t* evidx, const int32_t* eframe, const uint64_t* ehash, size_t ne,
                               const uint32_t* mediaIds, size_t nvid, unsigned radix, uint64_t hash,
                               int thresh, int src_in, orc_vmatch* out, size_t cap) {
  if (hash == 0) return 0;
  int32_t* best = (int32_t*)malloc(sizeof(int32_t) * (nvid ? nvid : 1));
  int32_t* bestf = (int32_t*)malloc(sizeof(int32_t) * (nvid ? nvid : 1));
  for (size_t v = 0; v < nvid; ++v) best[v] = -1;
  size_t bucket = radix_index_of(hash, radix);
  for (size_t i = 0; i < ne; ++i) {
    if (radix && radix_index_of(ehash[i], radix) != bucket) continue;
    int d = __builtin_popcountll(hash ^ ehash[i]);
    if (d < thresh) {
      uint32_t v = evidx[i];
      if (best[v] < 0 || d < best[v]) {
        best[v] = d;
        bestf[v] = eframe[i];
      }
    }
  }
  long long r = 0;
  if (src_in < 0) src_in = 0;
  for (size_t v = 0; v < nvid; ++v)
    if (best[v] >= 0) {
      if ((size_t)r < cap) {
        out[r].id = mediaIds[v];
        out[r].score = best[v];
        out[r].src_in = src_in;
        out[r].dst_in = bestf[v];
        out[r].len = 1;
      }
      ++r;
    }
  free(best);
  free(bestf);
  return r;
}

typedef struct {
  int32_t src, dst;
} orc_range;
static int cmp_range(const void* a, const void* b) {
  int x = ((const orc_range*)a)->src, y = ((const orc_range*)b)->src;
  return x < y ? -1 : x > y;
}

/* findVideo (:399-657).  Needle frames outside [skip, lastFrame - skip] are dropped unconditionally
 * (:431); per needle frame and matched media the closest entry (first in scan order among equals, :499-502)
 * contributes MatchRange(needle frame, entry frame); per media: ranges sorted by needle frame, numAdjacent
 * counts |dstIn - previous dstIn| < 15 starting from 0 (:606-613), percentNear = numAdjacent*100/num,
 * rejected when num < minFramesMatched or percentNear < minFramesNear, score = 100 - percentNear, range =
 * first pair, len = max(src span, dst span).  Results ascending mediaId (QMap). */
long long orc_video_find_video(const uint32_t* evidx, const int32_t* eframe, const uint64_t* ehash, size_t ne,
                               const uint32_t* mediaIds, size_t nvid, unsigned radix, const int32_t* nframes,
                               const uint64_t* nhashes, size_t nn, uint32_t needle_id, int thresh, int skip,
                               int min_frames_matched, int min_frames_near, int filter_self, orc_vmatch* out,
                               size_t cap) {
  if (nn == 0) return 0;
  /* per video index: list of ranges */
  orc_range** rg = (orc_range**)calloc(nvid ? nvid : 1, sizeof(orc_range*));
  size_t* rn = (size_t*)calloc(nvid ? nvid : 1, sizeof(size_t));
  int32_t* best = (int32_t*)malloc(sizeof(int32_t) * (nvid ? nvid : 1));
  int32_t* bestf = (int32_t*)malloc(sizeof(int32_t) * (nvid ? nvid : 1));
  int lastFrame = nframes[nn - 1];
  for (size_t q = 0; q < nn; ++q) {
    int srcFrame = nframes[q];
    if (srcFrame < skip || srcFrame > lastFrame - skip) continue;
    for (size_t v = 0; v < nvid; ++v) best[v] = -1;
    size_t bucket = radix_index_of(nhashes[q], radix);
    for (size_t i = 0; i < ne; ++i) {
      if (radix && radix_index_of(ehash[i], radix) != bucket) continue;
      int d = __builtin_popcountll(nhashes[q] ^ ehash[i]);
      if (d < thresh) {
        uint32_t v = evidx[i];
        if (filter_self && mediaIds[v] == needle_id) continue;
        if (best[v] < 0 || d < best[v]) {
          best[v] = d;
          bestf[v] = eframe[i];
        }
      }
    }
    for (size_t v = 0; v < nvid; ++v)
      if (best[v] >= 0) {
        if (!rg[v]) rg[v] = (orc_range*)malloc(sizeof(orc_range) * nn);
        rg[v][rn[v]].src = srcFrame;
        rg[v][rn[v]].dst = bestf[v];
        rn[v]++;
      }
  }
  /* results keyed by media id ascending: order video indices by media id (ids are unique per video) */
  size_t* order = (size_t*)malloc(sizeof(size_t) * (nvid ? nvid : 1));
  for (size_t v = 0; v < nvid; ++v) order[v] = v;
  for (size_t a = 1; a < nvid; ++a) { /* insertion sort: _mediaId is ascending in practice (:186) */
    size_t x = order[a];
    size_t b = a;
    while (b > 0 && mediaIds[order[b - 1]] > mediaIds[x]) {
      order[b] = order[b - 1];
      --b;
    }
    order[b] = x;
  }
  long long r = 0;
  for (size_t k = 0; k < nvid; ++k) {
    size_t v = order[k];
    if (!rn[v]) continue;
    qsort(rg[v], rn[v], sizeof(orc_range), cmp_range);
    int numAdjacent = 0, last = 0;
    for (size_t i = 0; i < rn[v]; ++i) {
      int frame = rg[v][i].dst;
      if (abs(frame - last) < 15) numAdjacent++;
      last = frame;
    }
    int num = (int)rn[v];
    int percentNear = numAdjacent * 100 / num;
    if (num < min_frames_matched) continue;
    if (percentNear < min_frames_near) continue;
    if ((size_t)r < cap) {
      out[r].id = mediaIds[v];
      out[r].score = 100 - percentNear;
      out[r].src_in = rg[v][0].src;
      out[r].dst_in = rg[v][0].dst;
      int srcLen = rg[v][rn[v] - 1].src - rg[v][0].src;
      int dstLen = rg[v][rn[v] - 1].dst - rg[v][0].dst;
      out[r].len = srcLen > dstLen ? srcLen : dstLen;
    }
    ++r;
  }
  for (size_t v = 0; v < nvid; ++v) free(rg[v]);
  free(order);
  free(rg);
  free(rn);
  free(best);
  free(bestf);
  return r;
}

/* Media::makeVideoIndex frame de-dup (src/media.cpp:958-1024).  The first frame is always stored and
 * does NOT enter the window (:958-965).  Every later frame is compared with the window of hashes seen
 * since the last stored frame: it is stored (and the window cleared) when at least one window hash is
 * >= threshold away (`close != window.size()`, :1003-1007) -- so with an empty window (right after the
 * first frame) it is never stored -- and it is appended to the window in both cases (:1011).  The last
 * frame is always stored (:1018-1024).  threshold <= 0 stores every frame.  keep[i] = 1 if frame i is
 * stored; returns the number stored. */
size_t orc_video_dedup(const uint64_t* hashes, size_t n, int threshold, uint8_t* keep) {
  if (n == 0) return 0;
  uint64_t* window = (uint64_t*)malloc(sizeof(uint64_t) * n);
  size_t wlen = 0, kept = 1;
  keep[0] = 1;
  for (size_t i = 1; i < n; ++i) {
    keep[i] = 0;
    if (threshold > 0) {
      size_t close = 0;
      for (size_t k = 0; k < wlen; ++k)
        if (__builtin_popcountll(window[k] ^ hashes[i]) < threshold) close++;
      if (close != wlen) {
        wlen = 0;
        keep[i] = 1;
      }
      window[wlen++] = hashes[i];
    } else {
      keep[i] = 1;
    }
    kept += keep[i];
  }
  if (!keep[n - 1]) {
    keep[n - 1] = 1;
    ++kept;
  }
  free(window);
  return kept;
}

/* Media::makeVideoIndex (src/media.cpp:925-1037) as a whole, given the hash of every decoded frame in decode order
 * (hash = dctHash64 of the autocropped grey frame, :961-962 / :987-992 -- orc_process_image).  io_frames / io_hashes
 * hold n_resume entries of an earlier index on entry (:929-936: decoding restarts at io_frames[n_resume-1] + 1; 0 =
 * fresh run) and the finished index on return (room for cap entries; returns the count, or -1 when cap is too small).
 * The first decoded frame is stored unconditionally and does not enter the window (:958-968); every later one goes
 * through the near-frame filter (:994-1011); decoding stops when frameNumber reaches max_frames (:1013-1016,
 * MAX_FRAMES_PER_VIDEO = 1 << 24); the last frame is appended if it was not stored (:1018-1024). */
long long orc_make_video_index(const uint64_t* frame_hashes, size_t n, int threshold, int max_frames,
                               int32_t* io_frames, uint64_t* io_hashes, size_t n_resume, size_t cap) {
  size_t cnt = n_resume;
  int frame_number = n_resume ? io_frames[n_resume - 1] + 1 : 0;
  uint64_t* window = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
  size_t wlen = 0, i = 0;
  long long rc = 0;
#define ORC_STORE(hv, fv)                 \
  do {                                    \
    if (cnt >= cap) {                     \
      rc = -1;                            \
      goto done;                          \
    }                                     \
    io_hashes[cnt] = (hv);                \
    io_frames[cnt] = (fv);                \
    ++cnt;                                \
  } while (0)
  if (i < n) {
    ORC_STORE(frame_hashes[i], frame_number);
    frame_number++;
    ++i;
  }
  for (; i < n; ++i) {
    const uint64_t hash = frame_hashes[i];
    if (threshold > 0) {
      size_t close = 0;
      for (size_t k = 0; k < wlen; ++k)
        if (__builtin_popcountll(window[k] ^ hash) < threshold) close++;
      if (close != wlen) {
        wlen = 0;
        ORC_STORE(hash, frame_number);
      }
      window[wlen++] = hash;
    } else {
      ORC_STORE(hash, frame_number);
    }
    frame_number++;
    if (frame_number == max_frames) break;
  }
  frame_number--;
  if (cnt > 0 && io_frames[cnt - 1] != frame_number) ORC_STORE(window[wlen - 1], frame_number);
  rc = (long long)cnt;
done:
#undef ORC_STORE
  free(window);
  return rc;
}

/* ---- CvFeaturesIndex: src/cvfeaturesindex.cpp:438-604 ----------------------------------------------
 * rows: N x 32 bytes (the cv::Mat of all ORB descriptors, cvfeaturesindex.h:73).  Exact brute-force
 * statement of `_index->knnSearch(descriptors, indices, dists, 10)` followed by `distance < cvThresh`
 * (:497-508): per needle row the k rows of smallest Hamming distance among those under thresh, ordered
 * (distance, row) -- FLANN's LSH is approximate and its tie order unspecified.  counts[q] = rows under
 * thresh. */
static int hamm256(const uint8_t* a, const uint8_t* b) {
  uint64_t x[4], y[4];
  memcpy(x, a, 32);
  memcpy(y, b, 32);
  return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
         __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

typedef struct {
  int32_t dist;
  uint32_t row;
} orc_nn;
static int cmp_nn(const void* a, const void* b) {
  const orc_nn* x = (const orc_nn*)a;
  const orc_nn* y = (const orc_nn*)b;
  if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
  return x->row < y->row ? -1 : x->row > y->row;
}

void orc_knn256(const uint8_t* rows, size_t n, const uint8_t* needles, size_t nq, int k, int thresh,
                uint32_t* out_row, int32_t* out_dist, uint32_t* counts) {
  orc_nn* c = (orc_nn*)malloc(sizeof(orc_nn) * (n ? n : 1));
  for (size_t q = 0; q < nq; ++q) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
      int d = hamm256(needles + q * 32, rows + i * 32);
      if (d < thresh) {
        c[m].dist = d;
        c[m].row = (uint32_t)i;
        ++m;
      }
    }
    qsort(c, m, sizeof(orc_nn), cmp_nn);
    counts[q] = (uint32_t)m;
    for (int j = 0; j < k; ++j) {
      out_row[q * (size_t)k + j] = (size_t)j < m ? c[j].row : 0;
      out_dist[q * (size_t)k + j] = (size_t)j < m ? c[j].dist : 0;
    }
  }
  free(c);
}

static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return x < y ? -1 : x > y;
}

/* find(): first_row[nm] ascending / media_id[nm] = the _indexMap without its sentinel (media_id 0 =
 * removed, :152-165).  Per needle row the knn above; row -> media by upper_bound - 1 (:514-516); votes per
 * media; score = median(distances) * 1000 / votes with the integer rules of :579-592; results ascending
 * mediaId.  Returns the number of results. */
long long orc_cvfeatures_find(const uint8_t* rows, size_t n, const uint32_t* first_row, const uint32_t* media_id,
                              size_t nm, const uint8_t* needles, size_t nq, int k, int thresh, uint32_t* out_ids,
                              int32_t* out_scores, size_t cap) {
  uint32_t* row = (uint32_t*)malloc(sizeof(uint32_t) * (nq * (size_t)k + 1));
  int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * (nq * (size_t)k + 1));
  uint32_t* cnt = (uint32_t*)malloc(sizeof(uint32_t) * (nq + 1));
  orc_knn256(rows, n, needles, nq, k, thresh, row, dist, cnt);
  size_t maxv = nq * (size_t)k + 1;
  uint32_t* vid = (uint32_t*)malloc(sizeof(uint32_t) * maxv);
  int* vd = (int*)malloc(sizeof(int) * maxv);
  size_t nv = 0;
  for (size_t q = 0; q < nq; ++q) {
    size_t len = cnt[q] < (uint32_t)k ? cnt[q] : (size_t)k;
    for (size_t t = 0; t < len; ++t) {
      uint32_t r = row[q * (size_t)k + t];
      size_t lo = 0, hi = nm; /* upper_bound(r) */
      while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (first_row[mid] <= r)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (lo == 0) continue;
      uint32_t id = media_id[lo - 1];
      if (!id) continue;
      vid[nv] = id;
      vd[nv] = dist[q * (size_t)k + t];
      ++nv;
    }
  }
  uint32_t* uid = (uint32_t*)malloc(sizeof(uint32_t) * (nv ? nv : 1));
  memcpy(uid, vid, sizeof(uint32_t) * nv);
  qsort(uid, nv, sizeof(uint32_t), cmp_u32);
  size_t nu = 0;
  for (size_t i = 0; i < nv; ++i)
    if (i == 0 || uid[i] != uid[i - 1]) uid[nu++] = uid[i];
  long long r = 0;
  int* sc = (int*)malloc(sizeof(int) * (nv ? nv : 1));
  for (size_t u = 0; u < nu; ++u) {
    size_t m = 0;
    for (size_t i = 0; i < nv; ++i)
      if (vid[i] == uid[u]) sc[m++] = vd[i];
    qsort(sc, m, sizeof(int), cmp_int);
    int score;
    size_t middle = m / 2;
    if (m < 2)
      score = sc[0];
    else if (m % 2 == 0)
      score = (sc[middle - 1] + sc[middle]) / 2;
    else
      score = sc[middle];
    score = score * 1000 / (int)m;
    if ((size_t)r < cap) {
      out_ids[r] = uid[u];
      out_scores[r] = score;
    }
    ++r;
  }
  free(sc);
  free(uid);
  free(vd);
  free(vid);
  free(cnt);
  free(dist);
  free(row);
  return r;
}

/* ---- ColorDescriptor::distance + ColorDescIndex::find: src/cvutil.cpp:682-749, src/colordescindex.cpp:250-278
 * desc: 258 bytes = 32 x {l,u,v,w : u16} + numColors : u8 + 1 pad (src/cvutil.h:57-113).  Compiled with
 * -ffp-contract=off: the reference's release build has no -march flag (cbird.pri:198-217), so its float
 * expressions are evaluated without FMA, left to right. */
static void color_get(const uint8_t* desc, int i, float* l_, float* u_, float* v_) {
  uint16_t l, u, v;
  memcpy(&l, desc + i * 8 + 0, 2);
  memcpy(&u, desc + i * 8 + 2, 2);
  memcpy(&v, desc + i * 8 + 4, 2);
  *l_ = l * 100.0f / 65535;
  *u_ = u * 354.0f / 65535 - 134.0f;
  *v_ = v * 262.0f / 65535 - 140.0f;
}

float orc_color_distance(const uint8_t* a_, const uint8_t* b_) {
  int na = a_[256], nb = b_[256];
  if (na == 0 || nb == 0 || abs(na - nb) > 2) return 3.402823466e+38F; /* FLT_MAX */
  const uint8_t *a, *b;
  if (na < nb) {
    a = b_;
    b = a_;
  } else {
    a = a_;
    b = b_;
  }
  const int numA = a[256], numB = b[256];
  float minDist[32];
  for (int i = 0; i < numA; i++) {
    minDist[i] = 3.402823466e+38F;
    float l1, u1, v1;
    color_get(a, i, &l1, &u1, &v1);
    for (int j = 0; j < numB; j++) {
      float l2, u2, v2;
      color_get(b, j, &l2, &u2, &v2);
      float dl = l1 - l2;
      float du = u1 - u2;
      float dv = v1 - v2;
      float dist = sqrtf(dl * dl + du * du + dv * dv);
      if (dist < minDist[i]) minDist[i] = dist;
    }
  }
  float score = 1;
  for (int i = 0; i < numA; i++) score += minDist[i];
  return score;
}

/* find(): Match(id, int(distance)) for every entry with finite distance and id != 0, index order */
long long orc_color_find(const uint8_t* descs, const uint32_t* ids, size_t n, const uint8_t* target,
                         uint32_t* out_ids, int32_t* out_scores, size_t cap) {
  long long m = 0;
  if (target[256] == 0) return 0;
  for (size_t i = 0; i < n; ++i) {
    float d = orc_color_distance(target, descs + i * 258);
    if (d < 3.402823466e+38F) {
      if (ids[i] != 0) {
        if ((size_t)m < cap) {
          out_ids[m] = ids[i];
          out_scores[m] = (int)d;
        }
        ++m;
      }
    }
  }
  return m;
}

/* ---- pre-stages of Scanner::processImage (src/scanner.cpp:852-862) --------------------------------
 * grayscale(): cv::cvtColor(BGR2GRAY / BGRA2GRAY) on 8-bit data (src/cvutil.cpp:1265-1283).  OpenCV 2.4
 * uses 14-bit fixed point: (B*1868 + G*9617 + R*4899 + 8192) >> 14 (as recalled: "parity unpinned"). */
void orc_bgr2gray(const uint8_t* src, int w, int h, size_t stride, int channels, uint8_t* dst) {
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + (size_t)y * stride + (size_t)x * channels;
      dst[(size_t)y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
    }
}

/* autocrop(): src/cvutil.cpp:1285-1402, restated statement by statement.  rect = {left, top, right,
 * bottom} of the kept region (the full image when nothing is cropped).  Returns 1 if a crop happens. */
int orc_autocrop(const uint8_t* img, int cols, int rows, size_t stride, int range, int* rect) {
  rect[0] = 0;
  rect[1] = 0;
  rect[2] = cols;
  rect[3] = rows;
  if (rows == 0 || cols == 0) return 0;
#define PX(y, x) ((int)img[(size_t)(y)*stride + (x)])
  const int color = PX(0, 0);
  const int minWidthCovered = (int)(cols * 0.66f);
  const int minHeightCovered = (int)(rows * 0.66f);
  const int maxHMarginDifference = (int)(cols * 0.05f);
  const int maxVMarginDifference = (int)(rows * 0.05f);
  int top;
  for (top = rows / 2; top >= 0; top--) {
    int left, right;
    for (left = 0; left < cols; left++)
      if (abs(PX(top, left) - color) > range) break;
    for (right = cols - 1; right >= 0; right--)
      if (abs(PX(top, right) - color) > range) break;
    right++;
    if (left > 0 && right < cols && left + cols - right > minWidthCovered) break;
  }
  top++;
  int bottom;
  for (bottom = rows / 2 + 1; bottom < rows; bottom++) {
    int left, right;
    for (left = 0; left < cols; left++)
      if (abs(PX(bottom, left) - color) > range) break;
    for (right = cols - 1; right >= 0; right--)
      if (abs(PX(bottom, right) - color) > range) break;
    right++;
    if (left + cols - right > minWidthCovered) break;
  }
  int left;
  for (left = cols / 2; left >= 0; left--) {
    int t, b;
    for (t = 0; t < rows; t++)
      if (abs(PX(t, left) - color) > range) break;
    for (b = rows - 1; b >= 0; b--)
      if (abs(PX(b, left) - color) > range) break;
    b++;
    if (t > 0 && b < rows && t + rows - b > minHeightCovered) break;
  }
  left++;
  int right;
  for (right = cols / 2 + 1; right < cols; right++) {
    int t, b;
    for (t = 0; t < rows; t++)
      if (abs(PX(t, right) - color) > range) break;
    for (b = rows - 1; b >= 0; b--)
      if (abs(PX(b, right) - color) > range) break;
    b++;
    if (t > 0 && b < rows && t + rows - b > minHeightCovered) break;
  }
#undef PX
  int bmargin = rows - bottom;
  if (abs(top - bmargin) > maxVMarginDifference) {
    if (top > bmargin)
      top = bmargin;
    else
      bottom = rows - top;
  }
  int rmargin = cols - right;
  if (abs(left - rmargin) > maxHMarginDifference) {
    if (left > rmargin)
      left = rmargin;
    else
      right = cols - left;
  }
  if ((left != 0 && right != cols) || (top != 0 && bottom != rows))
    if (left < right && top < bottom && (right - left) / (float)cols > 0.65f &&
        (bottom - top) / (float)rows > 0.65f) {
      rect[0] = left;
      rect[1] = top;
      rect[2] = right;
      rect[3] = bottom;
      return 1;
    }
  return 0;
}

/* processImage's hash: grayscale -> autocrop(gray, 20) when enabled -> dctHash64 (scanner.cpp:852-862).
 * channels 1, 3 (BGR) or 4 (BGRA). */
int orc_process_image(const uint8_t* img, int w, int h, size_t stride, int channels, int autocrop_range,
                      uint64_t* out, int* rect) {
  uint8_t* gray = NULL;
  const uint8_t* g = img;
  size_t gs = stride;
  if (channels != 1) {
    gray = (uint8_t*)malloc((size_t)w * h);
    orc_bgr2gray(img, w, h, stride, channels, gray);
    g = gray;
    gs = (size_t)w;
  }
  int r[4] = {0, 0, w, h};
  if (autocrop_range >= 0) orc_autocrop(g, w, h, gs, autocrop_range, r);
  if (rect) memcpy(rect, r, sizeof r);
  int rc = orc_dcthash64_view(g, w, h, gs, r[0], r[1], r[2] - r[0], r[3] - r[1], out);
  free(gray);
  return rc;
}

/* TemplateMatcher::match's score for one candidate (src/templatematcher.cpp:331-374), given the candidate patch as
 * warpAffine left it (template-sized, the candidate's channel count; undefined pixels are 0) and the template image:
 *   tmplMasked = tmplImg.clone(); grayscale(cand) (:334-337); then per pixel (:343-364) the candidate's grey value is
 *   "the mask indicator": where it is 0 the template's pixel is zeroed too (srcChannels < 4), and for a BGRA template
 *   the colour is premultiplied by its alpha, alpha set to 255 and the candidate's grey value scaled by the same alpha;
 *   candHash = dctHash64(cand), tmplHash = dctHash64(tmplMasked) (dctHash64 greys colour input itself), score =
 *   hamm64 (:366-371).  Channels 1, 3 (BGR) or 4 (BGRA).  cand_gray / tmpl_gray (optional, w*h bytes) receive the two
 *   images that were hashed (the template one after dctHash64's own grayscale).  Returns the distance, < 0 on error. */
int orc_template_score(const uint8_t* cand, int cand_channels, size_t cand_stride, const uint8_t* tmpl,
                       int tmpl_channels, size_t tmpl_stride, int w, int h, uint64_t* cand_hash, uint64_t* tmpl_hash,
                       uint8_t* cand_gray, uint8_t* tmpl_gray) {
  if (w <= 0 || h <= 0 || (cand_channels != 1 && cand_channels != 3 && cand_channels != 4) ||
      (tmpl_channels != 1 && tmpl_channels != 3 && tmpl_channels != 4))
    return -1;
  uint8_t* img = (uint8_t*)malloc((size_t)w * h);                      /* grayscale(img, img) */
  uint8_t* masked = (uint8_t*)malloc((size_t)w * h * tmpl_channels);   /* tmplImg.clone() */
  uint8_t* mg = (uint8_t*)malloc((size_t)w * h);
  if (cand_channels == 1)
    for (int y = 0; y < h; ++y) memcpy(img + (size_t)y * w, cand + (size_t)y * cand_stride, (size_t)w);
  else
    orc_bgr2gray(cand, w, h, cand_stride, cand_channels, img);
  for (int y = 0; y < h; ++y)
    memcpy(masked + (size_t)y * w * tmpl_channels, tmpl + (size_t)y * tmpl_stride, (size_t)w * tmpl_channels);
  const int srcChannels = tmpl_channels;
  for (int y = 0; y < h; ++y) {
    uint8_t* src = masked + (size_t)y * w * srcChannels;
    uint8_t* dst = img + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      uint8_t* dp = dst + x;
      uint8_t* sp = src + x * srcChannels;
      const uint8_t dstPixel = *dp;
      const uint8_t dstMask = dstPixel != 0 ? 255 : 0;
      if (srcChannels < 4) {
        for (int j = 0; j < srcChannels; ++j) *sp++ &= dstMask;
      } else {
        const int srcAlpha = sp[3];
        sp[0] = ((sp[0] * srcAlpha) >> 8) & dstMask;
        sp[1] = ((sp[1] * srcAlpha) >> 8) & dstMask;
        sp[2] = ((sp[2] * srcAlpha) >> 8) & dstMask;
        sp[3] = 255;
        *dp = (uint8_t)((dstPixel * srcAlpha) >> 8);
      }
    }
  }
  if (srcChannels == 1)
    memcpy(mg, masked, (size_t)w * h);
  else
    orc_bgr2gray(masked, w, h, (size_t)w * srcChannels, srcChannels, mg);
  uint64_t ch = 0, th = 0;
  int rc = orc_dcthash64(img, w, h, (size_t)w, &ch);
  if (rc == 0) rc = orc_dcthash64(mg, w, h, (size_t)w, &th);
  if (cand_gray) memcpy(cand_gray, img, (size_t)w * h);
  if (tmpl_gray) memcpy(tmpl_gray, mg, (size_t)w * h);
  free(img);
  free(masked);
  free(mg);
  if (rc) return -2;
  if (cand_hash) *cand_hash = ch;
  if (tmpl_hash) *tmpl_hash = th;
  return __builtin_popcountll(ch ^ th);
}

/* The version-1 .vdx file (src/videoindex.cpp:41-68 getVersion, :431-446 verify_v1, :448-476 save_v1, :478-541 load_v1):
 *   u16 numFrames | numFrames x u16 frame number | numFrames x u64 hash
 * orc_vdx_any_decode = VideoIndex::load (:70-90): "cbird" magic -> load_v2 (orc_vdx_decode), else load_v1 with its two
 * repairs; orc_vdx_any_verify = VideoIndex::isValid (:92-103).  Negative = the loader fails. */
long long orc_vdx_any_decode(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap) {
  if (len >= 5 && memcmp(buf, "cbird", 5) == 0) return orc_vdx_decode(buf, len, frames, hashes, cap);
  size_t pos = 0;
  uint16_t numFrames = 0;
  if (pos + 2 > len) return -1; /* io.read(&numFrames, 1, "header") */
  memcpy(&numFrames, buf + pos, 2);
  pos += 2;
  if (numFrames == 0) return 0;
  if (pos + 2 * (size_t)numFrames > len) return -2; /* "frame numbers" */
  const uint8_t* fr = buf + pos;
  pos += 2 * (size_t)numFrames;
  if ((size_t)numFrames > cap) return -4;
  uint16_t last = 0;
  int count = numFrames; /* the vector sizes, shrunk by the wrap repair */
  for (int i = 0; i < numFrames; ++i) {
    uint16_t frame;
    memcpy(&frame, fr + 2 * i, 2);
    if (frame < last) {
      if (last > 65000) {
        if (last != UINT16_MAX) {
          frames[i] = UINT16_MAX;
          i++;
        }
        count = i;
        break;
      } else {
        return -2; /* non-sequential frame number (corrupt file?) */
      }
    }
    last = frame;
    frames[i] = frame;
  }
  if (pos + 8 * (size_t)count > len) return -2; /* "hashes" */
  memcpy(hashes, buf + pos, 8 * (size_t)count);
  if (count && frames[0] != 0) { /* "fixing non-zero first frame bug" */
    if ((size_t)count + 1 > cap) return -4;
    memmove(frames + 1, frames, sizeof(int32_t) * (size_t)count);
    memmove(hashes + 1, hashes, sizeof(uint64_t) * (size_t)count);
    frames[0] = 0;
    hashes[0] = 0;
    ++count;
  }
  return count;
}

int orc_vdx_any_verify(const uint8_t* buf, size_t len) {
  if (len >= 5 && memcmp(buf, "cbird", 5) == 0) return orc_vdx_verify(buf, len);
  uint16_t numFrames = 0;
  if (len < 2) return 0;
  memcpy(&numFrames, buf, 2);
  const size_t size = sizeof(uint16_t) + sizeof(uint16_t) * numFrames + sizeof(uint64_t) * numFrames;
  return len == size;
}

/* save_v1 (:448-476) */
size_t orc_vdx_encode_v1(const int32_t* frames, const uint64_t* hashes, size_t n, uint8_t* out, size_t cap) {
  uint16_t numFrames = (uint16_t)(n < (size_t)INT16_MAX ? n : (size_t)INT16_MAX);
  uint16_t* f16 = (uint16_t*)malloc(sizeof(uint16_t) * ((size_t)numFrames + 1));
  size_t m = 0;
  for (int i = 0; i < numFrames; ++i) {
    if (frames[i] > UINT16_MAX) {
      numFrames = (uint16_t)m;
      break;
    }
    f16[m++] = (uint16_t)frames[i];
  }
  const size_t size = 2 + 10 * (size_t)numFrames;
  if (out && size <= cap) {
    memcpy(out, &numFrames, 2);
    memcpy(out + 2, f16, 2 * (size_t)numFrames);
    memcpy(out + 2 + 2 * (size_t)numFrames, hashes, 8 * (size_t)numFrames);
  }
  free(f16);
  return size;
}
