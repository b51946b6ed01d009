// prestage.hip -- the steps of Scanner::processImage in front of dctHash64 (src/scanner.cpp:852-862):
//   grayscale()  cv::cvtColor BGR/BGRA -> gray, 14-bit fixed point (src/cvutil.cpp:1265-1283)
//   autocrop()   de-letterboxing by scanning outwards from the centre (src/cvutil.cpp:1285-1402)
// and cbh_process_images, which chains gray -> autocrop -> hash for a batch of decoded images of one size.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "cbh_index.h"

namespace {

// cv::resize(..., INTER_LANCZOS4) on 8-bit grey images (sizeLongestSide, src/cvutil.cpp:1932-1950): one thread per
// destination pixel, 8 x 8 clamped taps with the per-column / per-row fixed-point weights; integer arithmetic
// throughout (horizontal sums, vertical sum, (v + 2^21) >> 22), so the bytes do not depend on evaluation order.
// No anti-aliasing -- a reduction reads 64 source pixels per output pixel however large the ratio, so the kernel is
// bound by the (sparse) source reads, a fraction of the image.
__global__ __launch_bounds__(256) void k_resize_lanczos4(const unsigned char* __restrict__ src, int w, int h,
                                                         size_t row_stride, size_t img_stride, int dw, int dh,
                                                         const int* __restrict__ xofs, const short* __restrict__ xa,
                                                         const int* __restrict__ yofs, const short* __restrict__ yb,
                                                         unsigned char* __restrict__ dst /* n*dw*dh */) {
  const unsigned char* img = src + (size_t)blockIdx.y * img_stride;
  unsigned char* out = dst + (size_t)blockIdx.y * (size_t)dw * dh;
  const int total = dw * dh;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int dy = i / dw, dx = i - dy * dw;
    const int sx0 = xofs[dx] - 3, sy0 = yofs[dy] - 3;
    int a[8], cx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = xa[dx * 8 + j];
      cx[j] = min(max(sx0 + j, 0), w - 1);
    }
    int v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned char* S = img + (size_t)min(max(sy0 + k, 0), h - 1) * row_stride;
      int D = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) D += (int)S[cx[j]] * a[j];
      v += D * (int)yb[dy * 8 + k];
    }
    v = (v + (1 << 21)) >> 22;
    out[i] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
  }
}

// interpolateLanczos4 + the fixed-point conversion of cv::resize (see oracle/cbird_oracle.c orc_lanczos4_tab)
void lanczos4_tab(int ssize, int dsize, std::vector<int>* ofs, std::vector<short>* coef) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1. / inv_scale;
  ofs->resize((size_t)dsize);
  coef->resize((size_t)dsize * 8);
  for (int d = 0; d < dsize; ++d) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    const int sx = (int)std::floor(fx);
    fx -= sx;
    float c[8];
    if (fx < FLT_EPSILON) {
      for (int i = 0; i < 8; i++) c[i] = 0;
      c[3] = 1;
    } else {
      float sum = 0;
      const double y0 = -(fx + 3) * M_PI * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
      for (int i = 0; i < 8; i++) {
        const double y = -(fx + 3 - i) * M_PI * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
      }
      sum = 1.f / sum;
      for (int i = 0; i < 8; i++) c[i] *= sum;
    }
    (*ofs)[(size_t)d] = sx;
    for (int k = 0; k < 8; ++k) {
      const long r = std::lrintf(c[k] * 2048.f);
      (*coef)[(size_t)d * 8 + k] = (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
    }
  }
}

__global__ __launch_bounds__(256) void k_bgr2gray(const unsigned char* __restrict__ src, int w, int h,
                                                  size_t row_stride, size_t img_stride, int channels,
                                                  unsigned char* __restrict__ dst /* n*w*h */) {
  const size_t img = blockIdx.y;
  const unsigned char* s = src + img * img_stride;
  unsigned char* d = dst + img * (size_t)w * h;
  const size_t total = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
    const unsigned char* p = s + (size_t)y * row_stride + (size_t)x * channels;
    d[i] = (unsigned char)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
  }
}

// autocrop(gray, range), src/cvutil.cpp:1285-1402, in two kernels.
//
// k_autocrop_runs computes what the reference's four scan loops look at -- for every row the length of the border-
// coloured run from its left end (rowL) and the end of the run from its right end (rowR), for every column the same
// from the top (colT) and the bottom (colB); a superset of the rows/columns the early-exit loops visit -- with one
// WAVE per task and the reference's own early exits inside each task:
//   * a row task takes eight rows and reads each in 256-byte segments from the left until the first content pixel,
//     then from the right (a content row costs its two end segments, a bar row is read once); the end segments of
//     all eight rows are requested together;
//   * a column task (second launch) owns 256 adjacent columns and walks down from the top (or up from the bottom) eight
//     rows per step until every one of its columns has met content, skipping the rows the row tasks found all border
//     and the columns outside every row's content span (pillarbox bars: nothing to walk).
// So a frame without bars costs ~1/4 of its bytes, a letterboxed one its bars plus the edges.  Every output has one
// owner: no atomics.  k_autocrop_decide (one workgroup per image, a wave per search) then replays the selection and centring rules, each of
// the four searches as a ballot over 64 candidates at a time.
typedef unsigned ac_u32_any_align __attribute__((aligned(1)));
constexpr int kAcColChunk = 256;  // columns per column task (4 per lane)
constexpr int kAcRowsPerStep = 8;  // rows a column task requests per step
constexpr int kAcRowsPerWave = 8;  // rows per row task

// bit j = pixel x + j is content (differs from the border colour by more than range); pixels at x >= cols are not
__device__ __forceinline__ unsigned ac_content4(const unsigned char* __restrict__ row, int x, int cols, int color,
                                                int range) {
  if (x >= cols) return 0u;
  unsigned v;
  if (x + 4 <= cols) {
    v = *reinterpret_cast<const ac_u32_any_align*>(row + x);
  } else {
    v = row[x];
    if (x + 1 < cols) v |= (unsigned)row[x + 1] << 8;
    if (x + 2 < cols) v |= (unsigned)row[x + 2] << 16;
  }
  const int n = min(4, cols - x);
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (int)((v >> (8 * j)) & 0xffu);
    m |= (unsigned)(j < n && abs(p - color) > range) << j;
  }
  return m;
}

__global__ __launch_bounds__(256) void k_autocrop_runs(const unsigned char* __restrict__ imgs, int cols, int rows,
                                                       size_t row_stride, size_t img_stride, int range,
                                                       int* __restrict__ scratch /* n*2*(rows+cols) */,
                                                       int phase /* 0: row tasks, 1: column tasks */) {
  const unsigned char* img = imgs + (size_t)blockIdx.y * img_stride;
  int* rowL = scratch + (size_t)blockIdx.y * 2 * (size_t)(rows + cols);
  int* rowR = rowL + rows;
  int* colT = rowR + rows;
  int* colB = colT + cols;
  const int lane = (int)threadIdx.x & 63;
  const int nchunks = (cols + kAcColChunk - 1) / kAcColChunk;
  const int row_tasks = (rows + kAcRowsPerWave - 1) / kAcRowsPerWave;
  // the column tasks run in a second launch: they skip the rows the row tasks found to be all border (a letterbox is
  // then read once, not twice)
  const int task = (phase ? row_tasks : 0) + (int)blockIdx.x * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
  if (task >= (phase ? row_tasks + 2 * nchunks : row_tasks)) return;
  const int color = img[0];
  if (task < row_tasks) {
    const int y0 = task * kAcRowsPerWave;
    const int xe = (cols - 1) & ~3;  // segment grid anchored at 0; the right scan walks it from the last dword down
    const int xl = 4 * lane;
    // The reference's scans of one row, a wave per row, in 256-byte segments: from the left until the first content pixel
    // (cols: none), from the right until the last (its index + 1).
    auto scan_left = [&](const unsigned char* row) {
      for (int x0 = 0; x0 < cols; x0 += 256) {
        const unsigned m = ac_content4(row, x0 + xl, cols, color, range);
        const unsigned long long b = __ballot(m != 0u);
        if (b) {
          const int fl = __builtin_ctzll(b);
          return x0 + 4 * fl + __builtin_ctz(__shfl(m, fl));
        }
      }
      return cols;
    };
    auto scan_right = [&](const unsigned char* row) {  // (content exists: the scan ends before x0 runs out)
      for (int x0 = xe;; x0 -= 256) {
        const int x = x0 - 4 * lane;
        const unsigned m = x >= 0 ? ac_content4(row, x, cols, color, range) : 0u;
        const unsigned long long b = __ballot(m != 0u);
        if (b) {
          const int fl = __builtin_ctzll(b);  // lowest lane = rightmost dword
          return x0 - 4 * fl + (31 - __builtin_clz(__shfl(m, fl))) + 1;
        }
      }
    };
    // First look, all eight rows at once: lane = (row r, end e, chunk j) reads 16 bytes of the row's first (e = 0) or last
    // (e = 1) 64 -- a content row (nearly every row of a frame without bars, every row between the bars of one with) is
    // settled by that one load and ~100 instructions for the eight of them; the row-by-row scans above are left to the rows
    // whose first / last 64 pixels are all border.  pend: bit 8 r + 4 e = that end of row r still wants its scan.
    unsigned long long pend = 0x1111111111111111ull;
    const unsigned long long img_bytes = (unsigned long long)(rows - 1) * row_stride + (unsigned)cols;
    if (cols >= 64 && img_bytes < (1ull << 32)) {
      const int r = lane >> 3, e = (lane >> 2) & 1, j = lane & 3;
      const int y = min(y0 + r, rows - 1);
      const int xq = (e ? cols - 64 : 0) + 16 * j;  // [xq, xq + 16) lies inside the row
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(img), 0, (int)(unsigned)img_bytes, 0x27000);
      typedef unsigned ac_v4u __attribute__((ext_vector_type(4)));
      const ac_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((unsigned)y * (unsigned)row_stride + (unsigned)xq), 0, 0);
      unsigned m = 0;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int px = (int)((v[d] >> (8 * k)) & 0xffu);
          m |= (unsigned)(abs(px - color) > range) << (4 * d + k);
        }
      // the group's answer: leftmost content column (e = 0: cols if none) / rightmost + 1 (e = 1: 0 if none)
      int cand = e ? (m ? xq + 32 - __builtin_clz(m) : 0) : (m ? xq + __builtin_ctz(m) : cols);
#pragma unroll
      for (int dlt = 1; dlt <= 2; dlt <<= 1) {
        const int o = __shfl_xor(cand, dlt);
        cand = e ? max(cand, o) : min(cand, o);
      }
      const bool settled = e ? cand > 0 : cand < cols;
      const bool owner = j == 0 && y0 + r < rows;
      if (owner && settled) (e ? rowR : rowL)[y0 + r] = cand;
      pend = __ballot(owner && !settled);
    } else {
      if (y0 + kAcRowsPerWave > rows) pend &= (1ull << (8 * (rows - y0))) - 1ull;
    }
    while (pend) {  // (uniform)
      const int r = __builtin_ctzll(pend) >> 3;
      const bool need_l = (pend >> (8 * r)) & 1ull, need_r = (pend >> (8 * r + 4)) & 1ull;
      pend &= ~(0xffull << (8 * r));
      const int y = y0 + r;
      const unsigned char* row = img + (size_t)y * row_stride;
      int left = 0;  // (need_r alone: the first look found content from the left)
      if (need_l) {
        left = scan_left(row);
        if (lane == 0) rowL[y] = left;
      }
      if (left >= cols) {  // no content at all: left == cols, right + 1 == 0
        if (lane == 0) rowR[y] = 0;
      } else if (need_r) {
        const int right = scan_right(row);
        if (lane == 0) rowR[y] = right;
      }
    }
    return;
  }
  const int ct = task - row_tasks, chunk = ct >> 1;
  const bool from_top = !(ct & 1);
  const int x = chunk * kAcColChunk + 4 * lane;
  // f0..f3: the first content row met from this end (top walker: its index; bottom walker: index + 1)
  const int none = from_top ? rows : 0;
  int f0 = none, f1 = none, f2 = none, f3 = none;
  unsigned live = x < cols ? (1u << min(4, cols - x)) - 1u : 0u;  // columns that have not met content yet
  // columns left of every row's first content pixel or right of every row's last one (the bars of a pillarboxed frame)
  // have none: settled from the row tasks' results, without walking them down the whole frame
  {
    int min_l = cols, max_r = 0;
    for (int y0 = 0; y0 < rows; y0 += 512) {  // (sixteen loads in flight per lane: one round trip per 512 rows)
      int l[8], r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int y = y0 + 64 * i + lane;
        l[i] = y < rows ? rowL[y] : cols;
        r[i] = y < rows ? rowR[y] : 0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (l[i] < cols) min_l = min(min_l, l[i]), max_r = max(max_r, r[i]);
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) {
      min_l = min(min_l, __shfl_xor(min_l, d));
      max_r = max(max_r, __shfl_xor(max_r, d));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < min_l || x + j >= max_r) live &= ~(1u << j);
    if (__ballot(live != 0u) == 0ull) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (x + j < cols) (from_top ? colT : colB)[x + j] = none;
      return;
    }
  }
  for (int s0 = 0; s0 < rows; s0 += kAcRowsPerStep) {
    // c[j]: bit u = column x + j has content in the u-th row of this step
    unsigned c[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < kAcRowsPerStep; ++u) {
      const int step = s0 + u;
      const int y = from_top ? step : rows - 1 - step;
      const bool look = step < rows && live != 0u && rowL[y] < cols;  // (an all-border row has no content)
      const unsigned m = look ? ac_content4(img + (size_t)y * row_stride, x, cols, color, range) : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] |= ((m >> j) & 1u) << u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((live >> j & 1u) && c[j]) {
        const int step = s0 + __builtin_ctz(c[j]);
        const int v = from_top ? step : rows - step;
        if (j == 0) f0 = v;
        if (j == 1) f1 = v;
        if (j == 2) f2 = v;
        if (j == 3) f3 = v;
        live &= ~(1u << j);
      }
    }
    if (__ballot(live != 0u) == 0ull) break;
  }
  const int found[4] = {f0, f1, f2, f3};
  int* out = from_top ? colT : colB;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (x + j < cols) out[x + j] = found[j];
}

// first candidate c in [from, to] (step +1) or [to, from] walking down (step -1) for which pred holds, else `miss`
template <class Pred>
__device__ __forceinline__ int ac_first(int from, int to, int dir, int miss, Pred pred) {
  const int lane = (int)threadIdx.x & 63;
  for (int c0 = from; dir > 0 ? c0 <= to : c0 >= to; c0 += 64 * dir) {
    const int c = c0 + dir * lane;
    const bool ok = (dir > 0 ? c <= to : c >= to) && pred(c);
    const unsigned long long b = __ballot(ok);
    if (b) return c0 + dir * __builtin_ctzll(b);
  }
  return miss;
}

__global__ __launch_bounds__(256) void k_autocrop_decide(int cols, int rows, const int* __restrict__ scratch,
                                                         int* __restrict__ rects /* n*4 */) {
  const int* rowL = scratch + (size_t)blockIdx.x * 2 * (size_t)(rows + cols);
  const int* rowR = rowL + rows;
  const int* colT = rowR + rows;
  const int* colB = colT + cols;
  const int minWidthCovered = (int)(cols * 0.66f);
  const int minHeightCovered = (int)(rows * 0.66f);
  const int maxHMarginDifference = (int)(cols * 0.05f);
  const int maxVMarginDifference = (int)(rows * 0.05f);
  __shared__ int s_edge[4];
  auto col_pred = [&](int x) {
    return colT[x] > 0 && colB[x] < rows && colT[x] + rows - colB[x] > minHeightCovered;
  };
  const int wave = (int)threadIdx.x >> 6;  // one of the four searches each
  int e;
  if (wave == 0)
    e = ac_first(rows / 2, 0, -1, -1, [&](int y) {
          return rowL[y] > 0 && rowR[y] < cols && rowL[y] + cols - rowR[y] > minWidthCovered;
        }) + 1;
  else if (wave == 1)
    e = ac_first(rows / 2 + 1, rows - 1, 1, rows, [&](int y) { return rowL[y] + cols - rowR[y] > minWidthCovered; });
  else if (wave == 2)
    e = ac_first(cols / 2, 0, -1, -1, col_pred) + 1;
  else
    e = ac_first(cols / 2 + 1, cols - 1, 1, cols, col_pred);
  if ((threadIdx.x & 63) == 0) s_edge[wave] = e;
  __syncthreads();
  if (threadIdx.x != 0) return;
  int top = s_edge[0], bottom = s_edge[1], left = s_edge[2], right = s_edge[3];
  const int bmargin = rows - bottom;
  if (abs(top - bmargin) > maxVMarginDifference) {
    if (top > bmargin)
      top = bmargin;
    else
      bottom = rows - top;
  }
  const int rmargin = cols - right;
  if (abs(left - rmargin) > maxHMarginDifference) {
    if (left > rmargin)
      left = rmargin;
    else
      right = cols - left;
  }
  int* r = rects + (size_t)blockIdx.x * 4;
  r[0] = 0;
  r[1] = 0;
  r[2] = cols;
  r[3] = rows;
  if ((left != 0 && right != cols) || (top != 0 && bottom != rows))
    if (left < right && top < bottom && (right - left) / (float)cols > 0.65f &&
        (bottom - top) / (float)rows > 0.65f) {
      r[0] = left;
      r[1] = top;
      r[2] = right;
      r[3] = bottom;
    }
}

// TemplateMatcher::match's masking step (src/templatematcher.cpp:334-364) for n candidate patches against one template:
// the candidate's grey value is the mask -- where it is 0 (outside the warped patch) the template's pixel is zeroed as
// well; a BGRA template is premultiplied by its alpha and scales the candidate's grey value by it.  Writes the two grey
// images dctHash64 is then taken of (dctHash64 greys the masked colour template itself, :366-367).
__device__ __forceinline__ unsigned tm_gray(unsigned b, unsigned g, unsigned r) {
  return (b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14;
}
__global__ __launch_bounds__(256) void k_tm_mask(const unsigned char* __restrict__ cand, int w, int h, size_t crs,
                                                 size_t cis, int cc, const unsigned char* __restrict__ tmpl, size_t trs,
                                                 int tc, unsigned char* __restrict__ cand_gray /* n*w*h */,
                                                 unsigned char* __restrict__ tmpl_gray /* n*w*h */) {
  const size_t img = blockIdx.y;
  const unsigned char* c = cand + img * cis;
  const size_t total = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
    const unsigned char* p = c + (size_t)y * crs + (size_t)x * cc;
    const unsigned char* t = tmpl + (size_t)y * trs + (size_t)x * tc;
    unsigned g = cc == 1 ? p[0] : tm_gray(p[0], p[1], p[2]);
    const unsigned m = g != 0u ? 255u : 0u;
    unsigned tg;
    if (tc == 1) {
      tg = t[0] & m;
    } else if (tc == 3) {
      tg = tm_gray(t[0] & m, t[1] & m, t[2] & m);
    } else {
      const unsigned a = t[3];
      tg = tm_gray(((t[0] * a) >> 8) & m, ((t[1] * a) >> 8) & m, ((t[2] * a) >> 8) & m);
      g = (g * a) >> 8;
    }
    cand_gray[img * total + i] = (unsigned char)g;
    tmpl_gray[img * total + i] = (unsigned char)tg;
  }
}

}  // namespace

extern "C" {

int cbh_bgr2gray_dev(const void* d_src, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                     int channels, void* d_gray, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!d_src || !d_gray || w <= 0 || h <= 0 || (channels != 3 && channels != 4)) return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned bx = (unsigned)std::min<size_t>(1024, ((size_t)w * h + 255) / 256);
  for (size_t i0 = 0; i0 < n; i0 += 65535) {
    const size_t m = std::min<size_t>(65535, n - i0);
    hipLaunchKernelGGL(k_bgr2gray, dim3(bx, (unsigned)m), dim3(256), 0, s,
                       (const unsigned char*)d_src + i0 * img_stride, w, h, row_stride, img_stride, channels,
                       (unsigned char*)d_gray + i0 * (size_t)w * h);
  }
  CBH_HIP(hipGetLastError());
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

int cbh_autocrop_dev(const void* d_gray, size_t n, int w, int h, size_t row_stride, size_t img_stride, int range,
                     void* d_rects, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!d_gray || !d_rects || w <= 0 || h <= 0) return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int* scratch = nullptr;
  CBH_HIP(cbh::malloc_async((void**)&scratch, n * 2 * (size_t)(w + h) * sizeof(int), s));
  const int row_tasks = (h + kAcRowsPerWave - 1) / kAcRowsPerWave, col_tasks = 2 * ((w + kAcColChunk - 1) / kAcColChunk);
  for (size_t i0 = 0; i0 < n; i0 += 65535) {
    const size_t m = std::min<size_t>(65535, n - i0);
    int* sc = scratch + i0 * 2 * (size_t)(w + h);
    hipLaunchKernelGGL(k_autocrop_runs, dim3((unsigned)((row_tasks + 3) / 4), (unsigned)m), dim3(256), 0, s,
                       (const unsigned char*)d_gray + i0 * img_stride, w, h, row_stride, img_stride, range, sc, 0);
    hipLaunchKernelGGL(k_autocrop_runs, dim3((unsigned)((col_tasks + 3) / 4), (unsigned)m), dim3(256), 0, s,
                       (const unsigned char*)d_gray + i0 * img_stride, w, h, row_stride, img_stride, range, sc, 1);
    hipLaunchKernelGGL(k_autocrop_decide, dim3((unsigned)m), dim3(256), 0, s, w, h, (const int*)sc,
                       (int*)d_rects + i0 * 4);
  }
  CBH_HIP(hipGetLastError());
  CBH_HIP(cbh::free_async(scratch, s));
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

/* Scanner::processImage's hash for n decoded images of one size (host buffers): grayscale (channels 3 = BGR,
 * 4 = BGRA, 1 = already gray) -> autocrop(gray, autocrop_range) when autocrop_range >= 0 -> dctHash64.
 * rects (optional, n*4 ints) receives the kept region {left, top, right, bottom} of every image. */
int cbh_process_images(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                       int channels, int autocrop_range, uint64_t* out, int32_t* rects, int device) {
  return cbh_process_images_ex(imgs, n, w, h, row_stride, img_stride, channels, autocrop_range, out, rects, 0, nullptr,
                               nullptr, device);
}

/* ... and, from the same upload, the image Scanner::processImage hands to ORB: sizeLongestSide(cvGray, size) of the
 * (autocropped) grey image (src/scanner.cpp:876).  resized: n slots of size*size bytes, image i packed at the start
 * of slot i with the dimensions resized_dims[2i], [2i+1] (w, h; 0, 0 where the reference would throw). */
int cbh_process_images_ex(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                          int channels, int autocrop_range, uint64_t* out, int32_t* rects, int resize_size,
                          uint8_t* resized, int32_t* resized_dims, int device) {
  if (resize_size < 0 || resize_size > 8192 || (resize_size > 0 && n && (!resized || !resized_dims))) return CBH_E_INVAL;
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!imgs || !out || w <= 0 || h <= 0 || (channels != 1 && channels != 3 && channels != 4) ||
      row_stride < (size_t)w * channels)
    return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  const size_t span1 = (size_t)(h - 1) * row_stride + (size_t)w * channels;
  size_t per_chunk = std::max<size_t>(1, ((size_t)256 << 20) / std::max(img_stride, span1));
  per_chunk = std::min(per_chunk, n);
  uint8_t *d_src = nullptr, *d_gray = nullptr, *d_res = nullptr;
  uint64_t* d_out = nullptr;
  int* d_rects = nullptr;
  hipStream_t s = nullptr;
  int rc = CBH_OK;
  const size_t slot = (size_t)resize_size * (size_t)resize_size;
  auto cleanup = [&]() {
    if (s) cbh::stream_destroy(s);
    for (void* p : {(void*)d_src, (void*)d_gray, (void*)d_out, (void*)d_rects, (void*)d_res})
      if (p) (void)hipFree(p);
  };
#define CBH_TRY(call)                       \
  do {                                      \
    hipError_t e_ = (call);                 \
    if (e_ != hipSuccess) {                 \
      cbh::set_last_error(#call, e_);       \
      cleanup();                            \
      return e_ == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP; \
    }                                       \
  } while (0)
  CBH_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CBH_TRY(hipMalloc(&d_src, (per_chunk - 1) * img_stride + span1));
  if (channels != 1) CBH_TRY(hipMalloc(&d_gray, per_chunk * (size_t)w * h));
  CBH_TRY(hipMalloc(&d_out, per_chunk * sizeof(uint64_t)));
  CBH_TRY(hipMalloc(&d_rects, per_chunk * 4 * sizeof(int)));
  if (resize_size > 0) CBH_TRY(hipMalloc(&d_res, per_chunk * slot));
  std::vector<int> hr(per_chunk * 4);
  for (size_t i0 = 0; rc == CBH_OK && i0 < n; i0 += per_chunk) {
    const size_t m = std::min(per_chunk, n - i0);
    CBH_TRY(hipMemcpyAsync(d_src, imgs + i0 * img_stride, (m - 1) * img_stride + span1, hipMemcpyHostToDevice, s));
    const uint8_t* gray = d_src;
    size_t gs = row_stride, gi = img_stride;
    if (channels != 1) {
      rc = cbh_bgr2gray_dev(d_src, m, w, h, row_stride, img_stride, channels, d_gray, device, s);
      if (rc) break;
      gray = d_gray;
      gs = (size_t)w;
      gi = (size_t)w * h;
    }
    bool cropped = false;
    if (autocrop_range >= 0) {
      rc = cbh_autocrop_dev(gray, m, w, h, gs, gi, autocrop_range, d_rects, device, s);
      if (rc) break;
      CBH_TRY(hipMemcpyAsync(hr.data(), d_rects, m * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
      CBH_TRY(hipStreamSynchronize(s));
      for (size_t i = 0; i < m; ++i)
        cropped |= hr[i * 4] != 0 || hr[i * 4 + 1] != 0 || hr[i * 4 + 2] != w || hr[i * 4 + 3] != h;
    } else {
      for (size_t i = 0; i < m; ++i) {
        hr[i * 4] = hr[i * 4 + 1] = 0;
        hr[i * 4 + 2] = w;
        hr[i * 4 + 3] = h;
      }
    }
    if (!cropped) {
      rc = cbh::launch_dcthash(gray, m, w, h, gs, gi, d_out, s);
    } else {
      // cropped images have their own geometry: one launch per run of images with the same kept region
      // (frames of one letterboxed video all share it)
      for (size_t i = 0, run = 1; i < m && rc == CBH_OK; i += run) {
        const int* r = &hr[i * 4];
        run = 1;
        while (i + run < m && !memcmp(r, &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
        // autocrop() narrows cvGray to a VIEW of the full image (cvutil.cpp:1397-1401): dctHash64's blur still
        // sees the cropped-away margins at the view's edges
        const cbh::HashView view{w, h, r[0], r[1]};
        rc = cbh::launch_dcthash(gray + i * gi, run, r[2] - r[0], r[3] - r[1], gs, gi, d_out + i, s, nullptr, &view);
      }
    }
    if (rc) break;
    if (resize_size > 0) {
      // sizeLongestSide of each kept region (a view: cv::resize does not look outside it), one launch per run of
      // images with the same kept region
      for (size_t i = 0, run = 1; i < m && rc == CBH_OK; i += run) {
        const int* r = &hr[i * 4];
        run = 1;
        while (i + run < m && !memcmp(r, &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
        int dw = 0, dh = 0;
        cbh_longest_side_dims(r[2] - r[0], r[3] - r[1], resize_size, &dw, &dh);
        if (dw <= 0 || dh <= 0 || dw > resize_size || dh > resize_size) dw = dh = 0;
        for (size_t t = 0; t < run; ++t) {
          resized_dims[2 * (i0 + i + t)] = dw;
          resized_dims[2 * (i0 + i + t) + 1] = dh;
        }
        if (dw == 0) continue;
        rc = cbh_resize_lanczos4_dev(gray + i * gi + (size_t)r[1] * gs + r[0], run, r[2] - r[0], r[3] - r[1], gs, gi,
                                     dw, dh, d_res + i * slot, device, s);
        if (rc) break;
        CBH_TRY(hipMemcpy2DAsync(resized + (i0 + i) * slot, slot, d_res + i * slot, (size_t)dw * dh, (size_t)dw * dh,
                                 run, hipMemcpyDeviceToHost, s));
      }
      if (rc) break;
    }
    CBH_TRY(hipMemcpyAsync(out + i0, d_out, m * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    CBH_TRY(hipStreamSynchronize(s));
    if (rects) memcpy(rects + i0 * 4, hr.data(), m * 4 * sizeof(int));
  }
#undef CBH_TRY
  cleanup();
  return rc;
}

/* TemplateMatcher::match's score for n candidate patches of one template (src/templatematcher.cpp:331-374): mask (above),
 * candHash = dctHash64(cand), tmplHash = dctHash64(tmplMasked), hamm64.  Device buffers; the hashes stay on the device. */
int cbh_template_hashes_dev(const void* d_cands, size_t n, int w, int h, size_t cand_row_stride, size_t cand_img_stride,
                            int cand_channels, const void* d_tmpl, size_t tmpl_row_stride, int tmpl_channels,
                            void* d_cand_hashes, void* d_tmpl_hashes, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  auto ch_ok = [](int c) { return c == 1 || c == 3 || c == 4; };
  if (!d_cands || !d_tmpl || !d_cand_hashes || !d_tmpl_hashes || w <= 0 || h <= 0 || !ch_ok(cand_channels) ||
      !ch_ok(tmpl_channels) || cand_row_stride < (size_t)w * cand_channels || tmpl_row_stride < (size_t)w * tmpl_channels)
    return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  const size_t px = (size_t)w * h;
  const unsigned bx = (unsigned)std::min<size_t>(1024, (px + 255) / 256);
  const size_t per = std::min<size_t>(n, std::max<size_t>(1, std::min<size_t>(65535, ((size_t)1 << 30) / px)));
  unsigned char *cg = nullptr, *tg = nullptr;
  cbh::Scratch scratch(s);
  CBH_HIP(scratch.get(&cg, per * px));
  CBH_HIP(scratch.get(&tg, per * px));
  int rc = CBH_OK;
  for (size_t i0 = 0; i0 < n && rc == CBH_OK; i0 += per) {
    const size_t m = std::min(per, n - i0);
    hipLaunchKernelGGL(k_tm_mask, dim3(bx, (unsigned)m), dim3(256), 0, s,
                       (const unsigned char*)d_cands + i0 * cand_img_stride, w, h, cand_row_stride, cand_img_stride,
                       cand_channels, (const unsigned char*)d_tmpl, tmpl_row_stride, tmpl_channels, cg, tg);
    rc = cbh::launch_dcthash(cg, m, w, h, (size_t)w, px, (uint64_t*)d_cand_hashes + i0, s);
    if (rc == CBH_OK) rc = cbh::launch_dcthash(tg, m, w, h, (size_t)w, px, (uint64_t*)d_tmpl_hashes + i0, s);
  }
  hipError_t e = hipGetLastError();
  if (rc) return rc;
  CBH_HIP(e);
  if (!stream) CBH_HIP(hipStreamSynchronize(s));
  return CBH_OK;
}

/* ... the same from host buffers, with the scores (hamm64 of the two hashes; the caller compares with tmThresh,
 * :373-376).  cand_hashes / tmpl_hashes may be NULL. */
int cbh_template_scores(const uint8_t* cands, size_t n, int w, int h, size_t cand_row_stride, size_t cand_img_stride,
                        int cand_channels, const uint8_t* tmpl, size_t tmpl_row_stride, int tmpl_channels,
                        uint64_t* cand_hashes, uint64_t* tmpl_hashes, int32_t* scores, int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!cands || !tmpl || !scores || w <= 0 || h <= 0 || cand_channels < 1 || cand_channels > 4 || tmpl_channels < 1 ||
      tmpl_channels > 4 || cand_row_stride < (size_t)w * cand_channels || tmpl_row_stride < (size_t)w * tmpl_channels)
    return CBH_E_INVAL;
  const size_t cspan = (size_t)(h - 1) * cand_row_stride + (size_t)w * cand_channels;
  if (n > 1 && cand_img_stride < cspan) return CBH_E_INVAL;
  const size_t cis = n > 1 ? cand_img_stride : cspan;
  const size_t tspan = (size_t)(h - 1) * tmpl_row_stride + (size_t)w * tmpl_channels;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = nullptr;
  unsigned char *d_c = nullptr, *d_t = nullptr;
  uint64_t* d_h = nullptr;
  auto cleanup = [&]() {
    if (s) cbh::stream_destroy(s);
    for (void* p : {(void*)d_c, (void*)d_t, (void*)d_h})
      if (p) (void)hipFree(p);
  };
#define CBH_TRY(call)                       \
  do {                                      \
    hipError_t e_ = (call);                 \
    if (e_ != hipSuccess) {                 \
      cbh::set_last_error(#call, e_);       \
      cleanup();                            \
      return e_ == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP; \
    }                                       \
  } while (0)
  CBH_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t per = std::min(n, std::max<size_t>(1, ((size_t)256 << 20) / std::max(cis, cspan)));
  CBH_TRY(hipMalloc(&d_c, (per - 1) * cis + cspan));
  CBH_TRY(hipMalloc(&d_t, tspan));
  CBH_TRY(hipMalloc(&d_h, 2 * per * sizeof(uint64_t)));
  CBH_TRY(hipMemcpyAsync(d_t, tmpl, tspan, hipMemcpyHostToDevice, s));
  std::vector<uint64_t> hh(2 * per);
  int rc = CBH_OK;
  for (size_t i0 = 0; i0 < n; i0 += per) {
    const size_t m = std::min(per, n - i0);
    CBH_TRY(hipMemcpyAsync(d_c, cands + i0 * cis, (m - 1) * cis + cspan, hipMemcpyHostToDevice, s));
    rc = cbh_template_hashes_dev(d_c, m, w, h, cand_row_stride, cis, cand_channels, d_t, tmpl_row_stride, tmpl_channels,
                                 d_h, d_h + per, device, s);
    if (rc) break;
    CBH_TRY(hipMemcpyAsync(hh.data(), d_h, 2 * per * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    CBH_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < m; ++i) {
      scores[i0 + i] = __builtin_popcountll(hh[i] ^ hh[per + i]);
      if (cand_hashes) cand_hashes[i0 + i] = hh[i];
      if (tmpl_hashes) tmpl_hashes[i0 + i] = hh[per + i];
    }
  }
#undef CBH_TRY
  cleanup();
  return rc;
}

/* sizeLongestSide's target size (src/cvutil.cpp:1933-1942): float aspect ratio, truncation */
void cbh_longest_side_dims(int w, int h, int size, int* out_w, int* out_h) {
  const float aspect = (float)w / (float)h;
  if (w > h) {
    *out_w = size;
    *out_h = (int)((float)size / aspect);
  } else {
    *out_h = size;
    *out_w = (int)(aspect * (float)size);
  }
}

int cbh_resize_lanczos4_dev(const void* d_src, size_t n, int w, int h, size_t row_stride, size_t img_stride, int dw,
                            int dh, void* d_dst, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!d_src || !d_dst || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || row_stride < (size_t)w || n > 65535 ||
      (long long)dw * dh > 0x7fffffffLL)
    return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  std::vector<int> xofs, yofs;
  std::vector<short> xa, yb;
  lanczos4_tab(w, dw, &xofs, &xa);
  lanczos4_tab(h, dh, &yofs, &yb);
  int *d_xofs = nullptr, *d_yofs = nullptr;
  short *d_xa = nullptr, *d_yb = nullptr;
  hipError_t e = hipSuccess;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (e != hipSuccess) return;
    if ((e = cbh::malloc_async(dst, bytes, s)) != hipSuccess) return;
    e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s);
  };
  up((void**)&d_xofs, xofs.data(), xofs.size() * sizeof(int));
  up((void**)&d_yofs, yofs.data(), yofs.size() * sizeof(int));
  up((void**)&d_xa, xa.data(), xa.size() * sizeof(short));
  up((void**)&d_yb, yb.data(), yb.size() * sizeof(short));
  if (e == hipSuccess) {
    const unsigned gx = (unsigned)std::min<long long>(((long long)dw * dh + 255) / 256, 4096);
    hipLaunchKernelGGL(k_resize_lanczos4, dim3(gx, (unsigned)n), dim3(256), 0, s, (const unsigned char*)d_src, w, h,
                       row_stride, img_stride, dw, dh, d_xofs, d_xa, d_yofs, d_yb, (unsigned char*)d_dst);
    e = hipGetLastError();
  }
  for (void* p : {(void*)d_xofs, (void*)d_yofs, (void*)d_xa, (void*)d_yb})
    if (p) (void)cbh::free_async(p, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // the tables came from pageable host vectors
  if (e != hipSuccess) {
    cbh::set_last_error("resize_lanczos4", e);
    return e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  return CBH_OK;
}

/* sizeLongestSide(img, size) for n grey images of one geometry in host memory; out receives n packed
 * out_w x out_h images. */
int cbh_size_longest_side(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride, int size,
                          uint8_t* out, int* out_w, int* out_h, int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (w <= 0 || h <= 0 || size <= 0 || !out_w || !out_h) return CBH_E_INVAL;
  cbh_longest_side_dims(w, h, size, out_w, out_h);
  if (*out_w <= 0 || *out_h <= 0) return CBH_E_INVAL;  // "computed width or height is 0, probably bad input"
  if (n == 0) return CBH_OK;
  if (!imgs || !out || row_stride < (size_t)w) return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  const int dw = *out_w, dh = *out_h;
  const size_t span1 = (size_t)(h - 1) * row_stride + (size_t)w;
  size_t per_chunk = std::max<size_t>(1, ((size_t)256 << 20) / std::max(img_stride, span1));
  per_chunk = std::min<size_t>(std::min(per_chunk, n), 65535);
  uint8_t *d_src = nullptr, *d_dst = nullptr;
  hipStream_t s = nullptr;
  int rc = CBH_OK;
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(&d_src, (per_chunk - 1) * img_stride + span1)) != hipSuccess ||
      (e = hipMalloc(&d_dst, per_chunk * (size_t)dw * dh)) != hipSuccess) {
    cbh::set_last_error("size_longest_side setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  for (size_t i0 = 0; rc == CBH_OK && i0 < n; i0 += per_chunk) {
    const size_t m = std::min(per_chunk, n - i0);
    if ((e = hipMemcpyAsync(d_src, imgs + i0 * img_stride, (m - 1) * img_stride + span1, hipMemcpyHostToDevice, s)) !=
        hipSuccess) {
      cbh::set_last_error("size_longest_side H2D", e);
      rc = CBH_E_HIP;
      break;
    }
    rc = cbh_resize_lanczos4_dev(d_src, m, w, h, row_stride, img_stride, dw, dh, d_dst, device, s);
    if (rc) break;
    if ((e = hipMemcpyAsync(out + i0 * (size_t)dw * dh, d_dst, m * (size_t)dw * dh, hipMemcpyDeviceToHost, s)) !=
            hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      cbh::set_last_error("size_longest_side D2H", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) cbh::stream_destroy(s);
  if (d_src) (void)hipFree(d_src);
  if (d_dst) (void)hipFree(d_dst);
  return rc;
}


}  // extern "C"
