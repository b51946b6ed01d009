// orb.hip -- SURVEY.md section 8 row a11: ORB keypoints and rBRIEF descriptors as cbird configures OpenCV 2.4's ORB
//   Media::makeKeyPoints            /root/reference/src/media.cpp:859-866   OrbFeatureDetector(n, 1.2f, 12, 31, 0, 2, HARRIS_SCORE, 31)
//   Media::makeKeyPointDescriptors  /root/reference/src/media.cpp:868-872   OrbDescriptorExtractor() (256 bits, WTA_K 2)
// for a batch of grey images of any sizes resident in HBM (cbird feeds <= 400 px on the longest side,
// /root/reference/src/scanner.cpp:876).  Bit-exact against oracle/orb_oracle.c, which carries the statement of what
// is and is not pinned against the cbird binary (the learned rBRIEF pattern is an INPUT: cbh_orb_set_pattern;
// retainBest leaves the survivors as libstdc++'s nth_element + partition do, or -- orb_retain_order = 0 -- keeps every
// tie in raster order).
//
// Decomposition (one launch each, all images of the batch at once; nothing visits the host between them):
//   k_orb_level0                    level 0 = the caller's image copied into the pyramid buffer WITH its reflect-101
//                                   border (4 columns, 3 rows): every later kernel reads aligned dwords without edge cases
//   k_orb_resize   x (levels - 1)   level l from l-1: cv::resize INTER_LINEAR's 8-bit fixed-point path.  The x / y
//                                   coefficient tables of a workgroup's rows are formed in LDS (double arithmetic as the
//                                   library's, IEEE on the device); border pixels are the resized pixels at the reflected
//                                   coordinates (= copyMakeBorder of the resized level).  Only the levels that can hold a
//                                   keypoint (both sides > 62) are built.
//   k_orb_fast                      FAST-9/16 score + 3x3 non-maximum suppression per 64x32 tile staged in LDS by dword
//                                   loads: a compass-point test rejects most pixels after 4 reads, 16-bit brighter /
//                                   darker masks, "9 contiguous" by four shift-ands, the score (largest threshold that
//                                   keeps the corner) only for the rare corners.  Writes the non-zero entries of the
//                                   suppressed score map (the map is cleared by a memset).
//   k_orb_select                    one workgroup per (image, level): 16-byte scans of the score map (histogram ->
//                                   n-th best score: retainBest(2N) on 8-bit keys is a counting select; ordered
//                                   compaction) -> Harris response per candidate -> radix select of the N-th best float
//                                   -> ordered compaction -> intensity-centroid orientation (one wave per keypoint).
//   k_orb_blur                      7x7 sigma-2 Gaussian in OpenCV's 8-bit fixed point: a lane owns 4 columns and walks
//                                   down a strip of 29 rows; per row three aligned dword loads, the horizontal taps as
//                                   ten v_dot4_u32_u8 against constant weight masks, the last 7 row sums in registers.
//   k_orb_describe                  one wave per keypoint: lane t evaluates tests t, t+64, t+128, t+192 of the
//                                   rotated pattern, four ballots are the 256 bits; also scales the keypoints to
//                                   image coordinates and writes them level by level per image.
// All of it is byte / small-integer work on L2-resident pyramids (a 400x300 image: 130 KB + 290 KB of levels); the
// bound is instruction issue and launch geometry, not HBM.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "cbh_index.h"

namespace cbh {
namespace {

constexpr int kLevels = 12;       // media.cpp:861
constexpr int kEdge = 31;         // edgeThreshold
constexpr int kFastT = 20;        // orb.cpp: FastFeatureDetector fd(20, true)
constexpr float kHarrisK = 0.04f;
constexpr int kBorderX = 4, kBorderY = 3;  // reflect-101 border stored around every pyramid level

struct OrbImage {
  unsigned long long src_off;  // the caller's image
  unsigned src_stride;
  int nlev;                    // levels that can hold keypoints (both sides > 2 * kEdge); they form a prefix
  int w[kLevels], h[kLevels];
  unsigned pitch[kLevels];               // = round_up(w, 4) + 2 * kBorderX
  unsigned long long poff[kLevels];      // byte offset of pixel (0, 0) of level l in the pyramid buffer (4-byte aligned)
  unsigned long long soff[kLevels];      // byte offset of level l in the score / blurred buffer (16-byte aligned)
  unsigned spitch[kLevels];              // = round_up(w, 16)
  unsigned long long coff[kLevels];      // first candidate slot of level l
  int nfeat[kLevels];
  unsigned tile_first[kLevels + 1];      // FAST tiles (keypoint region): level l owns [tile_first[l], tile_first[l+1])
  unsigned bwg_first[kLevels + 1];       // blur workgroups: level l owns [bwg_first[l], bwg_first[l+1])
  float scale[kLevels];                  // getScale(level): (float)pow((double)1.2f, level), from the host's libm
};

struct OrbPattern {
  signed char v[1024];
};

struct OrbCand {  // one candidate / keypoint of a level, in level coordinates
  unsigned short x, y;
  float response;
  float angle;
  float ca, sa;  // (float)cos, (float)sin of the angle in radians, as computeOrbDescriptor forms them
};

__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }  // cvRound: half to even
__device__ __forceinline__ short sat_short_rn(float v) {
  const float r = rintf(v);
  return (short)(r < -32768.f ? -32768.f : r > 32767.f ? 32767.f : r);
}
__device__ __forceinline__ int reflect101(int p, int len) {  // one reflection: |overhang| < len
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return p;
}

// ---- level 0: copy + border ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_orb_level0(const OrbImage* __restrict__ images,
                                                    const unsigned char* __restrict__ imgs,
                                                    unsigned char* __restrict__ pyr) {
  const OrbImage& im = images[blockIdx.y];
  if (im.nlev < 1) return;
  const int w = im.w[0], h = im.h[0];
  const int P = (int)im.pitch[0], nd = P >> 2;  // dwords per padded row
  const int rows = h + 2 * kBorderY;
  const unsigned char* __restrict__ src = imgs + im.src_off;
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(pyr + im.poff[0] - (size_t)kBorderY * P - kBorderX);
  for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < rows * nd; i += (int)(gridDim.x * 256)) {
    const int r = i / nd, c = i - r * nd;
    const unsigned char* __restrict__ S = src + (size_t)reflect101(r - kBorderY, h) * im.src_stride;
    const int x0 = 4 * c - kBorderX;
    unsigned v = 0;
    if (x0 >= 0 && x0 + 3 < w) {
      v = *reinterpret_cast<const unsigned*>(S + x0);  // any alignment
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v |= (unsigned)S[reflect101(min(x0 + j, w + kBorderX + 2), w)] << (8 * j);
    }
    dst[(size_t)r * nd + c] = v;
  }
}

// ---- pyramid: cv::resize INTER_LINEAR, 8UC1 (oracle: orc_resize_linear_u8_cv) ---------------------------------------
constexpr int kResizeRows = 16;
__global__ __launch_bounds__(256) void k_orb_resize(const OrbImage* __restrict__ images, int level,
                                                    unsigned char* __restrict__ pyr) {
  extern __shared__ __attribute__((aligned(8))) unsigned char s_res[];
  __shared__ int s_yofs[kResizeRows];
  __shared__ int s_yc[kResizeRows];
  const OrbImage& im = images[blockIdx.y];
  if (level >= im.nlev) return;
  const int dw = im.w[level], dh = im.h[level], sw = im.w[level - 1], sh = im.h[level - 1];
  const int P = (int)im.pitch[level], nd = P >> 2;
  const int rows = dh + 2 * kBorderY;
  const int r0 = (int)blockIdx.x * kResizeRows;
  if (r0 >= rows) return;
  int* __restrict__ xofs = reinterpret_cast<int*>(s_res);        // per padded column: source column
  int* __restrict__ xc = reinterpret_cast<int*>(s_res) + P;      // (c1 << 16) | (c0 & 0xffff)
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  for (int cp = threadIdx.x; cp < P; cp += 256) {
    const int d = reflect101(min(cp - kBorderX, dw + kBorderX + 2), dw);
    float f = (float)((d + 0.5) * scale_x - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) f = 0.f, s = 0;
    if (s >= sw - 1) f = 0.f, s = sw - 1;
    xofs[cp] = s;
    xc[cp] = ((int)sat_short_rn(f * 2048.f) << 16) | ((int)sat_short_rn((1.f - f) * 2048.f) & 0xffff);
  }
  if (threadIdx.x < kResizeRows) {
    const int dy = reflect101(min(r0 + (int)threadIdx.x, rows - 1) - kBorderY, dh);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    // both source rows clipped to the image (the library's invoker); the coefficients stay as they are
    s_yofs[threadIdx.x] = (min(max(sy, 0), sh - 1) << 16) | min(max(sy + 1, 0), sh - 1);
    s_yc[threadIdx.x] = ((int)sat_short_rn(fy * 2048.f) << 16) | ((int)sat_short_rn((1.f - fy) * 2048.f) & 0xffff);
  }
  __syncthreads();
  const unsigned sp = im.pitch[level - 1];
  const unsigned char* __restrict__ src = pyr + im.poff[level - 1];
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(pyr + im.poff[level] - (size_t)kBorderY * P - kBorderX);
  const int nr = min(kResizeRows, rows - r0);
  // (staging the ~21 source rows of a workgroup in LDS was tried: 2.6x SLOWER -- the four byte loads of neighbouring
  // pixels already coalesce in the vector L1, the staging only adds LDS traffic and a barrier)
  for (int i = threadIdx.x; i < nr * nd; i += 256) {
    const int r = i / nd, c = i - r * nd;
    const int yo = s_yofs[r], yc = s_yc[r];
    const int b0 = (short)(yc & 0xffff), b1 = yc >> 16;
    const unsigned char* __restrict__ S0 = src + (size_t)(yo >> 16) * sp;
    const unsigned char* __restrict__ S1 = src + (size_t)(yo & 0xffff) * sp;
    unsigned out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sx = xofs[4 * c + j], cc = xc[4 * c + j];
      const int a0 = (short)(cc & 0xffff), a1 = cc >> 16;
      // sx + 1 <= sw - 1 + 1: the source level carries a border, and a1 == 0 there
      const int D0 = S0[sx] * a0 + S0[sx + 1] * a1;
      const int D1 = S1[sx] * a0 + S1[sx + 1] * a1;
      const int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
      out |= (unsigned)min(max(v, 0), 255) << (8 * j);
    }
    dst[(size_t)(r0 + r) * nd + c] = out;
  }
}

// ---- FAST-9/16 + non-maximum suppression (oracle: orc_fast_nms_scores restricted to the keypoint region) -----------
constexpr int kTileW = 64, kTileH = 32;
constexpr int kPxW = 76, kPxH = kTileH + 8;  // staged pixels: x from ox0 - 7 (4-byte aligned) to ox0 + 68, 19 dwords
constexpr int kRawW = kTileW + 2, kRawH = kTileH + 2;

// circle offsets in the order of fast.cpp's offsets16
__device__ __forceinline__ void fast_circle(const unsigned char* __restrict__ c, int* d) {
  constexpr int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int oy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const int v = c[0];
#pragma unroll
  for (int k = 0; k < 16; ++k) d[k] = v - (int)c[oy[k] * kPxW + ox[k]];
}
// stage 1: a run of 9 of the 16 contains pixel 0 or pixel 8 (they are opposite), so a corner has one of the two
// outside [v - t, v + t] -- fast.cpp's own first test; two reads reject most pixels
__device__ __forceinline__ bool fast_compass(const unsigned char* __restrict__ c) {
  const int v = c[0];
  const int d0 = v - (int)c[3 * kPxW], d8 = v - (int)c[-3 * kPxW];
  return (unsigned)(d0 + kFastT) > 2u * kFastT || (unsigned)(d8 + kFastT) > 2u * kFastT;
}
// stage 2: the segment test proper on 16-bit brighter / darker masks
__device__ __forceinline__ bool fast_is_corner(const unsigned char* __restrict__ c) {
  int d[16];
  fast_circle(c, d);
  unsigned dark = 0, bright = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    dark |= (unsigned)(d[k] > kFastT) << k;     // x < v - t
    bright |= (unsigned)(d[k] < -kFastT) << k;  // x > v + t
  }
  auto run9 = [](unsigned m) {
    m |= m << 16;
    unsigned t = m & (m >> 1);  // runs of 2
    t &= t >> 2;                // 4
    t &= t >> 4;                // 8
    t &= m >> 8;                // 9
    return t != 0;
  };
  return run9(dark) || run9(bright);
}
// stage 3: cornerScore<16> = max(threshold, max over the 16 arcs of min(d), max over the arcs of min(-d)) - 1
__device__ __forceinline__ int fast_corner_score(const unsigned char* __restrict__ c) {
  int d[16];
  fast_circle(c, d);
  // minima / maxima of the 16 arcs of 9 by doubling: arcs of 2, 4, 8, then 8 + 1
  int mn[16], mx[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) mn[k] = min(d[k], d[(k + 1) & 15]), mx[k] = max(d[k], d[(k + 1) & 15]);
  int mn4[16], mx4[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) mn4[k] = min(mn[k], mn[(k + 2) & 15]), mx4[k] = max(mx[k], mx[(k + 2) & 15]);
  int best = kFastT;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int m9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
    const int M9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
    best = max(best, max(m9, -M9));
  }
  return best - 1;
}
// append v to an LDS queue, one atomic per wave
__device__ __forceinline__ void queue_push(bool pred, unsigned short v, unsigned short* q, int* n) {
  const unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  const int lane = (int)(threadIdx.x & 63);
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(n, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (pred) q[base + __popcll(m & ((1ull << lane) - 1ull))] = v;
}

__device__ __forceinline__ bool find_tile(const OrbImage& im, unsigned t, int* level, int* tx, int* ty) {
  if (t >= im.tile_first[im.nlev]) return false;
  int l = 0;
  while (t >= im.tile_first[l + 1]) ++l;
  const int iw = im.w[l] - 2 * kEdge;
  const int ntx = (iw + kTileW - 1) / kTileW;
  const unsigned k = t - im.tile_first[l];
  *level = l;
  *ty = (int)(k / (unsigned)ntx);
  *tx = (int)(k - (unsigned)*ty * (unsigned)ntx);
  return true;
}

__global__ __launch_bounds__(256) void k_orb_fast(const OrbImage* __restrict__ images,
                                                  const unsigned char* __restrict__ pyr,
                                                  unsigned char* __restrict__ scores /* cleared */) {
  __shared__ __attribute__((aligned(16))) unsigned char s_px[kPxH * kPxW];
  __shared__ __attribute__((aligned(4))) unsigned char s_raw[kRawH * kRawW];
  static_assert(kRawH * kRawW % 4 == 0, "s_raw is cleared by dwords");
  __shared__ unsigned short s_q1[kRawH * kRawW], s_q2[kRawH * kRawW];
  __shared__ int s_n[2];
  const OrbImage& im = images[blockIdx.y];
  int l, tx, ty;
  if (!find_tile(im, blockIdx.x, &l, &tx, &ty)) return;
  const int w = im.w[l], h = im.h[l];
  const unsigned char* __restrict__ src = pyr + im.poff[l];
  const unsigned sp = im.pitch[l];
  const int ox0 = kEdge + tx * kTileW, oy0 = kEdge + ty * kTileH;  // first output pixel of the tile
  const int px0 = ox0 - 7, py0 = oy0 - 4;                            // px0 = 24 + 64 tx: dword aligned
  for (int i = threadIdx.x; i < kPxH * (kPxW / 4); i += 256) {
    const int r = i / (kPxW / 4), c = i - r * (kPxW / 4);
    const int y = min(py0 + r, h + kBorderY - 1);                 // overhanging tiles: clamped into the buffer,
    const int x = min(px0 + 4 * c, (int)sp - 2 * kBorderX);       // those values are never used
    reinterpret_cast<unsigned*>(s_px)[i] = *reinterpret_cast<const unsigned*>(src + (ptrdiff_t)y * sp + x);
  }
  for (int i = threadIdx.x; i < kRawH * kRawW / 4; i += 256) reinterpret_cast<unsigned*>(s_raw)[i] = 0u;
  if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
  __syncthreads();
  // three stages with the survivors compacted in between, so that the expensive ones run on full waves
  auto centre = [&](int i) {
    const int r = i / kRawW, c = i - r * kRawW;
    return s_px + (r + 3) * kPxW + (c + 6);
  };
  for (int i0 = 0; i0 < kRawH * kRawW; i0 += 256) {
    const int i = i0 + (int)threadIdx.x;
    queue_push(i < kRawH * kRawW && fast_compass(centre(min(i, kRawH * kRawW - 1))), (unsigned short)i, s_q1, &s_n[0]);
  }
  __syncthreads();
  const int n1 = s_n[0];
  for (int k0 = 0; k0 < n1; k0 += 256) {
    const int k = k0 + (int)threadIdx.x;
    const int i = s_q1[min(k, n1 - 1)];
    queue_push(k < n1 && fast_is_corner(centre(i)), (unsigned short)i, s_q2, &s_n[1]);
  }
  __syncthreads();
  const int n2 = s_n[1];
  for (int k = threadIdx.x; k < n2; k += 256) {
    const int i = s_q2[k];
    s_raw[i] = (unsigned char)fast_corner_score(centre(i));
  }
  __syncthreads();
  unsigned char* __restrict__ out = scores + im.soff[l];
  const unsigned op = im.spitch[l];
  for (int i = threadIdx.x; i < kTileH * kTileW; i += 256) {
    const int r = i / kTileW, c = i - r * kTileW;
    const unsigned char* __restrict__ q = s_raw + (r + 1) * kRawW + (c + 1);
    const int s = q[0];
    if (s == 0) continue;
    const int x = ox0 + c, y = oy0 + r;
    if (x >= w - kEdge || y >= h - kEdge) continue;
    if (s > q[1] && s > q[-1] && s > q[-kRawW - 1] && s > q[-kRawW] && s > q[-kRawW + 1] && s > q[kRawW - 1] &&
        s > q[kRawW] && s > q[kRawW + 1])
      out[(size_t)y * op + x] = (unsigned char)s;
  }
}

// ---- per (image, level): retainBest(2N) on the FAST score, Harris, retainBest(N), orientation ----------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
// exclusive prefix of v over the 256 threads of the workgroup; *total = sum.  s_w: 4 ints of LDS.  Two barriers.
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int* total) {
  const int incl = wave_incl_scan(v);
  const int wv = (int)(threadIdx.x >> 6);
  __syncthreads();  // s_w free again
  if ((threadIdx.x & 63) == 63) s_w[wv] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) base += i < wv ? s_w[i] : 0;
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return base + incl - v;
}

__device__ __forceinline__ float harris_at(const unsigned char* __restrict__ img, unsigned pitch, int x, int y) {
  // orb.cpp HarrisResponses, blockSize 7: Sobel-like Ix, Iy on the 7x7 block around (x, y)
  float scale = (1 << 2) * 7 * 255.0f;
  scale = 1.0f / scale;
  const float scale_sq_sq = scale * scale * scale * scale;
  int a = 0, b = 0, c = 0;
  const unsigned char* __restrict__ p0 = img + (ptrdiff_t)(y - 4) * pitch + (x - 4);
  // three rows of the 9x9 patch in registers at a time
  int r0[9], r1[9], r2[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) r0[j] = p0[j], r1[j] = p0[pitch + j];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const unsigned char* __restrict__ pr = p0 + (size_t)(i + 2) * pitch;
#pragma unroll
    for (int j = 0; j < 9; ++j) r2[j] = pr[j];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int Ix = (r1[j + 2] - r1[j]) * 2 + (r0[j + 2] - r0[j]) + (r2[j + 2] - r2[j]);
      const int Iy = (r2[j + 1] - r0[j + 1]) * 2 + (r2[j] - r0[j]) + (r2[j + 2] - r0[j + 2]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) r0[j] = r1[j], r1[j] = r2[j];
  }
  return ((float)a * (float)b - (float)c * (float)c - kHarrisK * ((float)a + (float)b) * ((float)a + (float)b)) *
         scale_sq_sq;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // cv::fastAtan2 (2.4), oracle: orc_fast_atan2
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
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

__device__ __forceinline__ unsigned float_key(float f) {  // order-preserving: larger float <-> larger key
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : b | 0x80000000u;
}

// ---- KeyPointsFilter::retainBest in the order libstdc++ leaves the survivors (tuning key orb_retain_order = 1) -----
// OpenCV 2.4's retainBest is std::nth_element(begin, begin + n, end, response greater), then std::partition of the tail
// on `response >= keypoints[n - 1].response` (oracle/retain_stl.cpp quotes it).  Which ties survive and in which order
// therefore follow from the C++ library.  This is libstdc++'s introselect restated for one workgroup: the Hoare
// partition step is a PAIRING -- the k-th element from the left that is not greater than the pivot is exchanged with
// the k-th element from the right that the pivot is not greater than, for as long as the first lies left of the second
// -- so the two stop lists are built with one prefix scan per 256 elements, the exchanges happen in parallel and the cut
// is where the lists cross.  The O(1) steps (median of three, the final insertion sort of <= 3 elements) and the
// depth-limit fallback (heap select, which real inputs do not reach) run on thread 0.  Keys and the raster positions
// they came from travel together; equal keys are never reordered except as the library would.
struct RetainScratch {
  float* key;
  unsigned* ref;
  unsigned* lp;  // stop positions from the left, ascending
  unsigned* ra;  // stop positions from the right, ASCENDING (read backwards)
};
__device__ __forceinline__ void kr_swap(float* key, unsigned* ref, int a, int b) {
  const float k = key[a];
  key[a] = key[b], key[b] = k;
  const unsigned r = ref[a];
  ref[a] = ref[b], ref[b] = r;
}
// __move_median_to_first(result, a, b, c) under comp(x, y) = key[x] > key[y]
__device__ void stl_median_to_first(float* key, unsigned* ref, int result, int a, int b, int c) {
  const float ka = key[a], kb = key[b], kc = key[c];
  int pick;
  if (ka > kb)
    pick = kb > kc ? b : (ka > kc ? c : a);
  else
    pick = ka > kc ? a : (kb > kc ? c : b);
  kr_swap(key, ref, result, pick);
}
__device__ void stl_insertion_sort(float* key, unsigned* ref, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const float vk = key[i];
    const unsigned vr = ref[i];
    if (vk > key[first]) {  // move_backward(first, i, i + 1)
      for (int j = i; j > first; --j) key[j] = key[j - 1], ref[j] = ref[j - 1];
      key[first] = vk, ref[first] = vr;
    } else {  // __unguarded_linear_insert
      int hole = i, next = i - 1;
      while (vk > key[next]) {
        key[hole] = key[next], ref[hole] = ref[next];
        hole = next, --next;
      }
      key[hole] = vk, ref[hole] = vr;
    }
  }
}
__device__ void stl_push_heap(float* key, unsigned* ref, int first, int hole, int top, float vk, unsigned vr) {
  int parent = (hole - 1) / 2;
  while (hole > top && key[first + parent] > vk) {
    key[first + hole] = key[first + parent], ref[first + hole] = ref[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  key[first + hole] = vk, ref[first + hole] = vr;
}
__device__ void stl_adjust_heap(float* key, unsigned* ref, int first, int hole, int len, float vk, unsigned vr) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (key[first + child] > key[first + child - 1]) child--;
    key[first + hole] = key[first + child], ref[first + hole] = ref[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    key[first + hole] = key[first + child - 1], ref[first + hole] = ref[first + child - 1];
    hole = child - 1;
  }
  stl_push_heap(key, ref, first, hole, top, vk, vr);
}
__device__ void stl_heap_select(float* key, unsigned* ref, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2)  // __make_heap
    for (int parent = (len - 2) / 2;; --parent) {
      stl_adjust_heap(key, ref, first, parent, len, key[first + parent], ref[first + parent]);
      if (parent == 0) break;
    }
  for (int i = middle; i < last; ++i)
    if (key[i] > key[first]) {  // __pop_heap(first, middle, i)
      const float vk = key[i];
      const unsigned vr = ref[i];
      key[i] = key[first], ref[i] = ref[first];
      stl_adjust_heap(key, ref, first, 0, len, vk, vr);
    }
}
// __unguarded_partition(lo, hi, pivot value p) by the whole workgroup; returns the cut (uniform)
__device__ int stl_partition(float* key, unsigned* ref, int lo, int hi, float p, const RetainScratch& rs, int* s_w) {
  const int tid = (int)threadIdx.x;
  int nL = 0, nR = 0;
  for (int c = lo; c < hi; c += 256) {
    const int i = c + tid;
    int isL = 0, isR = 0;
    if (i < hi) {
      const float k = key[i];
      isL = !(k > p), isR = !(p > k);
    }
    int tot;
    const int ex = block_excl_scan(isL | (isR << 16), s_w, &tot);
    if (isL) rs.lp[nL + (ex & 0xffff)] = (unsigned)i;
    if (isR) rs.ra[nR + (ex >> 16)] = (unsigned)i;
    nL += tot & 0xffff, nR += tot >> 16;
  }
  __syncthreads();
  const int m = min(nL, nR);
  int mine = 0;
  for (int k = tid; k < m; k += 256) mine += rs.lp[k] < rs.ra[nR - 1 - k];
  int K;
  (void)block_excl_scan(mine, s_w, &K);  // the condition holds for a prefix of k: K exchanges
  for (int k = tid; k < K; k += 256) kr_swap(key, ref, (int)rs.lp[k], (int)rs.ra[nR - 1 - k]);
  const int rprev = K > 0 ? (int)rs.ra[nR - K] : hi;
  const int cut = (K < nL && (int)rs.lp[K] < rprev) ? (int)rs.lp[K] : rprev;
  __syncthreads();
  return cut;
}
// retainBest(n_points) on key[0, cnt) / ref[0, cnt); returns the new size (uniform).  depth_limit < 0: 2 * lg(cnt)
__device__ int stl_retain_best(float* key, unsigned* ref, int cnt, int n_points, int depth_limit,
                               const RetainScratch& rs, int* s_w) {
  if (n_points < 0 || cnt <= n_points) return cnt;
  if (n_points == 0) return 0;
  const int tid = (int)threadIdx.x;
  int first = 0, last = cnt;
  int depth = depth_limit >= 0 ? depth_limit : 2 * (31 - __clz(cnt));
  bool heap = false;
  while (last - first > 3) {
    if (depth == 0) {
      if (tid == 0) {
        stl_heap_select(key, ref, first, n_points + 1, last);
        kr_swap(key, ref, first, n_points);
      }
      heap = true;
      break;
    }
    --depth;
    if (tid == 0) stl_median_to_first(key, ref, first, first + 1, first + (last - first) / 2, last - 1);
    __syncthreads();
    const int cut = stl_partition(key, ref, first + 1, last, key[first], rs, s_w);
    if (cut <= n_points)
      first = cut;
    else
      last = cut;
  }
  if (!heap && tid == 0) stl_insertion_sort(key, ref, first, last);
  __syncthreads();
  // std::partition(begin + n, end, response >= keypoints[n - 1].response): the k-th failing element from the left
  // changes places with the k-th passing one from the right; T passing elements end up in front
  const float thr = key[n_points - 1];
  int mine = 0;
  for (int i = n_points + tid; i < cnt; i += 256) mine += key[i] >= thr;
  int T;
  (void)block_excl_scan(mine, s_w, &T);
  const int mid = n_points + T;
  int nA = 0, nB = 0;
  for (int c = n_points; c < cnt; c += 256) {
    const int i = c + tid;
    int isA = 0, isB = 0;
    if (i < cnt) {
      const bool pass = key[i] >= thr;
      isA = i < mid && !pass, isB = i >= mid && pass;
    }
    int tot;
    const int ex = block_excl_scan(isA | (isB << 16), s_w, &tot);
    if (isA) rs.lp[nA + (ex & 0xffff)] = (unsigned)i;
    if (isB) rs.ra[nB + (ex >> 16)] = (unsigned)i;
    nA += tot & 0xffff, nB += tot >> 16;
  }
  __syncthreads();
  for (int k = tid; k < nA; k += 256) kr_swap(key, ref, (int)rs.lp[k], (int)rs.ra[nB - 1 - k]);  // nA == nB
  __syncthreads();
  return mid;
}

__global__ __launch_bounds__(256) void k_retain_best(const float* __restrict__ resp, int cnt, int n_points,
                                                     int depth_limit, RetainScratch rs, unsigned* __restrict__ order,
                                                     unsigned* __restrict__ out_count) {
  __shared__ int s_w[4];
  const int tid = (int)threadIdx.x;
  for (int i = tid; i < cnt; i += 256) rs.key[i] = resp[i], rs.ref[i] = (unsigned)i;
  __syncthreads();
  const int k = stl_retain_best(rs.key, rs.ref, cnt, n_points, depth_limit, rs, s_w);
  for (int i = tid; i < k; i += 256) order[i] = rs.ref[i];
  if (tid == 0) *out_count = (unsigned)k;
}

__global__ __launch_bounds__(256) void k_orb_select(const OrbImage* __restrict__ images,
                                                    const unsigned char* __restrict__ pyr,
                                                    const unsigned char* __restrict__ scores,
                                                    OrbCand* __restrict__ cand, unsigned* __restrict__ level_counts,
                                                    int retain_order, RetainScratch rs_all,
                                                    OrbCand* __restrict__ cand_tmp) {
  __shared__ int s_hist[256];
  __shared__ int s_w[4];
  __shared__ int s_pick[2];
  const int tid = (int)threadIdx.x;
  const OrbImage& im = images[blockIdx.y];
  const int l = (int)blockIdx.x;
  unsigned* __restrict__ out_count = level_counts + (size_t)blockIdx.y * kLevels + l;
  if (l >= im.nlev) {
    if (tid == 0) *out_count = 0;
    return;
  }
  const int h = im.h[l];
  const int ih = h - 2 * kEdge;
  const unsigned char* __restrict__ sc = scores + im.soff[l];
  const unsigned sp = im.spitch[l];
  const int G = (int)(sp >> 4);      // 16-byte groups per score row
  const int groups = G * ih;         // the rows [31, h-31) in raster order; only keypoint columns are ever non-zero
  const uint4* __restrict__ sc16 = reinterpret_cast<const uint4*>(sc + (size_t)kEdge * sp);
  const int N = im.nfeat[l];
  OrbCand* __restrict__ cd = cand + im.coff[l];
  // -- (a) one scan of the score map: every suppressed corner of the keypoint region into the candidate list in
  //        raster order (response = its FAST score for now) + the histogram of the scores
  s_hist[tid] = 0;
  __syncthreads();
  int c0 = 0;
  for (int g0 = 0; g0 < groups; g0 += 256) {
    const int g = g0 + tid;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (g < groups) q = sc16[g];
    const unsigned qq[4] = {q.x, q.y, q.z, q.w};
    int nz = 0;
    if ((q.x | q.y | q.z | q.w) != 0u) {
#pragma unroll
      for (int j = 0; j < 16; ++j) nz += ((qq[j >> 2] >> (8 * (j & 3))) & 255u) != 0u;
    }
    int tot;
    int pos = c0 + block_excl_scan(nz, s_w, &tot);
    if (nz) {
      const int gy = g / G, gx = g - gy * G;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned sv = (qq[j >> 2] >> (8 * (j & 3))) & 255u;
        if (sv) {
          OrbCand c;
          c.x = (unsigned short)(16 * gx + j), c.y = (unsigned short)(gy + kEdge);
          c.response = (float)sv, c.angle = 0.f, c.ca = 0.f, c.sa = 0.f;
          cd[pos++] = c;
          atomicAdd(&s_hist[sv], 1);
        }
      }
    }
    c0 += tot;
  }
  __syncthreads();
  const unsigned char* __restrict__ img = pyr + im.poff[l];
  const unsigned ip = im.pitch[l];
  int c2 = 0;
  if (retain_order == 1) {
    // -- (b'-e') the same three steps with the survivors in libstdc++'s order: retainBest(2N) on the FAST score,
    //    HarrisResponses, retainBest(N); the list is then gathered through cand_tmp
    const RetainScratch rs = {rs_all.key + im.coff[l], rs_all.ref + im.coff[l], rs_all.lp + im.coff[l],
                              rs_all.ra + im.coff[l]};
    OrbCand* __restrict__ tmp = cand_tmp + im.coff[l];
    for (int i = tid; i < c0; i += 256) rs.key[i] = cd[i].response, rs.ref[i] = (unsigned)i;
    __syncthreads();
    const int k1 = stl_retain_best(rs.key, rs.ref, c0, 2 * N, -1, rs, s_w);
    for (int i = tid; i < k1; i += 256) {
      const unsigned r = rs.ref[i];
      rs.key[i] = harris_at(img, ip, cd[r].x, cd[r].y);
    }
    __syncthreads();
    c2 = stl_retain_best(rs.key, rs.ref, k1, N, -1, rs, s_w);
    for (int i = tid; i < c2; i += 256) {
      OrbCand c = cd[rs.ref[i]];
      c.response = rs.key[i];
      tmp[i] = c;
    }
    __syncthreads();
    for (int i = tid; i < c2; i += 256) cd[i] = tmp[i];
    __syncthreads();
  } else {
  // -- (b) retainBest(2N): every score >= the 2N-th best survives
  {
    // bins in descending order: thread t holds bin 255 - t (bin 0 is never counted)
    const int v = s_hist[255 - tid];
    int total;
    const int excl = block_excl_scan(v, s_w, &total);
    const int want = 2 * N;
    if (tid == 0) s_pick[0] = want == 0 ? 256 : 1;  // nothing / everything
    __syncthreads();
    if (want > 0 && total > want && excl < want && want <= excl + v) s_pick[0] = 255 - tid;  // the 2N-th best score
    __syncthreads();
  }
  const float thr = (float)s_pick[0];
  // -- (c) ordered compaction of the list in place
  int c1 = 0;
  for (int i0 = 0; i0 < c0; i0 += 256) {
    const int i = i0 + tid;
    OrbCand c;
    int keep = 0;
    if (i < c0) {
      c = cd[i];
      keep = c.response >= thr;
    }
    int tot;
    const int pos = block_excl_scan(keep, s_w, &tot);
    if (keep) cd[c1 + pos] = c;
    c1 += tot;
  }
  __syncthreads();  // the candidate list is visible to the whole workgroup (global memory, same workgroup)
  for (int i = tid; i < c1; i += 256) cd[i].response = harris_at(img, ip, cd[i].x, cd[i].y);
  __syncthreads();
  // -- (d) retainBest(N) on the Harris response: radix select of the N-th best key, MSB first
  unsigned key_thr = 0;  // keep key >= key_thr
  if (N == 0) {
    c1 = 0;
  } else if (c1 > N) {
    unsigned prefix = 0, mask = 0;
    int want = N;  // the want-th largest among the keys matching the prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
      s_hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < c1; i += 256) {
        const unsigned k = float_key(cd[i].response);
        if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
      }
      __syncthreads();
      {
        const int v = s_hist[255 - tid];
        int total;
        const int excl = block_excl_scan(v, s_w, &total);
        if (excl < want && want <= excl + v) s_pick[0] = 255 - tid, s_pick[1] = want - excl;
        __syncthreads();
      }
      prefix |= (unsigned)s_pick[0] << shift;
      mask |= 255u << shift;
      want = s_pick[1];
      __syncthreads();
    }
    key_thr = prefix;
  }
  // -- (e) ordered compaction in place (a write never passes the reads of its own or a later chunk)
  for (int i0 = 0; i0 < c1; i0 += 256) {
    const int i = i0 + tid;
    OrbCand c;
    int keep = 0;
    if (i < c1) {
      c = cd[i];
      keep = float_key(c.response) >= key_thr;
    }
    int tot;
    const int pos = block_excl_scan(keep, s_w, &tot);  // its barriers order this chunk's reads before its writes
    if (keep) cd[c2 + pos] = c;
    c2 += tot;
  }
  __syncthreads();
  }  // retain_order
  if (tid == 0) *out_count = (unsigned)c2;
  // -- (f) orientation: IC_Angle over the circular patch of radius 15, HALF a wave per keypoint, no loop: lane =
  //        (row pair +-v, v = (lane & 31) >> 1; 16-column segment u0 = -16 + 16 * (lane & 1)) -- eight unaligned dword
  //        loads per lane, all in flight together, then a reduction over the 32 lanes
  const int lane = tid & 63, hw = tid >> 5;  // 8 half-waves
  {
    const int v = (lane & 31) >> 1, u0 = -16 + 16 * (lane & 1);
    // u_max of orb.cpp for half patch 15 (15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3), one nibble per row
    const int um = (int)((0x3689ABCDDEEEFFFFull >> (4 * v)) & 15ull);
    for (int i0 = 0; i0 < c2; i0 += 8) {
      const int i = i0 + hw;
      const bool live = i < c2;
      const int x = live ? cd[i].x : kEdge, y = live ? cd[i].y : kEdge;
      const unsigned char* __restrict__ pp = img + (ptrdiff_t)(y + v) * ip + (x + u0);
      const unsigned char* __restrict__ pm = img + (ptrdiff_t)(y - v) * ip + (x + u0);
      unsigned p[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = *reinterpret_cast<const unsigned*>(pp + 4 * k), q[k] = *reinterpret_cast<const unsigned*>(pm + 4 * k);
      int m10 = 0, v_sum = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int u = u0 + j;
        const int vp = (int)((p[j >> 2] >> (8 * (j & 3))) & 255u);
        const int vm = v == 0 ? 0 : (int)((q[j >> 2] >> (8 * (j & 3))) & 255u);  // the centre row counts once
        const bool in = u >= -um && u <= um;
        m10 += in ? u * (vp + vm) : 0;
        v_sum += in ? vp - vm : 0;
      }
      int m01 = v * v_sum;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        m01 += __shfl_xor(m01, d);
        m10 += __shfl_xor(m10, d);
      }
      if (live && (lane & 31) == 0) cd[i].angle = fast_atan2_deg((float)m01, (float)m10);
    }
  }
  __syncthreads();
  // the rotation of the descriptor pattern, once per keypoint (double-precision libm, one lane each)
  for (int i = tid; i < c2; i += 256) {
    float angle = cd[i].angle;
    angle *= (float)(3.14159265358979323846 / 180.f);
    cd[i].ca = (float)cos((double)angle);
    cd[i].sa = (float)sin((double)angle);
  }
}

// ---- GaussianBlur(7x7, sigma 2) in the library's 8-bit fixed point (oracle: orc_gauss7_blur_u8) ---------------------
struct GaussK {
  int k[7];
};
constexpr int kBlurRows = 29;  // output rows per strip: 29 + 6 halo rows = 5 x 7 row steps (ring slot = step % 7, static)
__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) {
  return __builtin_amdgcn_udot4(a, b, c, false);
}
__global__ __launch_bounds__(256) void k_orb_blur(const OrbImage* __restrict__ images,
                                                  const unsigned char* __restrict__ pyr,
                                                  const unsigned* __restrict__ level_counts, GaussK g,
                                                  unsigned char* __restrict__ blurred) {
  const OrbImage& im = images[blockIdx.y];
  const unsigned t = blockIdx.x;
  if (t >= im.bwg_first[im.nlev]) return;
  int l = 0;
  while (t >= im.bwg_first[l + 1]) ++l;
  if (level_counts[(size_t)blockIdx.y * kLevels + l] == 0) return;
  const int w = im.w[l], h = im.h[l];
  const int nc = (w + 3) >> 2;  // dword columns
  const int nstrips = (h + kBlurRows - 1) / kBlurRows;
  const int item = (int)(t - im.bwg_first[l]) * 256 + (int)threadIdx.x;
  if (item >= nc * nstrips) return;
  const int strip = item / nc, cx = item - strip * nc;
  const int P = (int)im.pitch[l];
  const unsigned char* __restrict__ col = pyr + im.poff[l] + 4 * cx;  // pixel (4 cx, 0)
  unsigned char* __restrict__ out = blurred + im.soff[l] + 4 * cx;
  const unsigned op = im.spitch[l];
  const int y0 = strip * kBlurRows;
  // constant weight masks: pixel j of the lane's four takes bytes j+1 .. j+7 of the 12-byte window (left | own | right)
  const unsigned k0 = (unsigned)g.k[0], k1 = (unsigned)g.k[1], k2 = (unsigned)g.k[2], k3 = (unsigned)g.k[3];
  const unsigned wL0 = k0 << 8 | k1 << 16 | k2 << 24, wO0 = k3 | k2 << 8 | k1 << 16 | k0 << 24;
  const unsigned wL1 = k0 << 16 | k1 << 24, wO1 = k2 | k3 << 8 | k2 << 16 | k1 << 24, wR1 = k0;
  const unsigned wL2 = k0 << 24, wO2 = k1 | k2 << 8 | k3 << 16 | k2 << 24, wR2 = k1 | k0 << 8;
  const unsigned wO3 = k0 | k1 << 8 | k2 << 16 | k3 << 24, wR3 = k2 | k1 << 8 | k0 << 16;
  unsigned ring[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) ring[i][j] = 0u;
  for (int s0 = 0; s0 < kBlurRows + 6; s0 += 7) {
#pragma unroll
    for (int ss = 0; ss < 7; ++ss) {
      const int step = s0 + ss;
      const int y = min(y0 - 3 + step, h + kBorderY - 1);  // strips overhanging the level: clamped into the buffer
      const unsigned char* __restrict__ rp = col + (ptrdiff_t)y * P;
      const unsigned L = *reinterpret_cast<const unsigned*>(rp - 4);
      const unsigned O = *reinterpret_cast<const unsigned*>(rp);
      const unsigned R = *reinterpret_cast<const unsigned*>(rp + 4);
      ring[ss][0] = udot4(L, wL0, udot4(O, wO0, 0u));
      ring[ss][1] = udot4(L, wL1, udot4(O, wO1, udot4(R, wR1, 0u)));
      ring[ss][2] = udot4(L, wL2, udot4(O, wO2, udot4(R, wR2, 0u)));
      ring[ss][3] = udot4(O, wO3, udot4(R, wR3, 0u));
      if (step >= 6) {
        const int oy = y0 + step - 6;
        if (oy < h) {
          unsigned pk = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // rows oy-3 .. oy+3 sit in ring slots (ss+1)%7 .. ss; the kernel is symmetric
            const unsigned a = ring[(ss + 1) % 7][j] + ring[ss][j];
            const unsigned b = ring[(ss + 2) % 7][j] + ring[(ss + 6) % 7][j];
            const unsigned c = ring[(ss + 3) % 7][j] + ring[(ss + 5) % 7][j];
            const unsigned d = ring[(ss + 4) % 7][j];
            const unsigned sum = k0 * a + k1 * b + k2 * c + k3 * d + (1u << 15);
            pk |= min(sum >> 16, 255u) << (8 * j);
          }
          *reinterpret_cast<unsigned*>(out + (size_t)oy * op) = pk;
        }
      }
    }
  }
}

// ---- descriptors + output (oracle: orc_orb_descriptor; ORB::operator() scaling of the keypoints) -------------------
__global__ __launch_bounds__(256) void k_orb_describe(const OrbImage* __restrict__ images,
                                                      const unsigned char* __restrict__ blurred,
                                                      const OrbCand* __restrict__ cand,
                                                      const unsigned* __restrict__ level_counts,
                                                      const int* __restrict__ pat /* 256 x (x0,y0,x1,y1) int8 */,
                                                      int kp_cap, cbh_keypoint* __restrict__ out_kp,
                                                      float* __restrict__ out_after /* 2 per slot, or null */,
                                                      unsigned char* __restrict__ out_desc /* or null */,
                                                      unsigned* __restrict__ out_counts) {
  const int lane = (int)(threadIdx.x & 63);
  const unsigned slot = blockIdx.x * 4u + (threadIdx.x >> 6);
  const OrbImage& im = images[blockIdx.y];
  const unsigned* __restrict__ cnt = level_counts + (size_t)blockIdx.y * kLevels;
  unsigned total = 0, first = 0;
  int l = -1;
#pragma unroll
  for (int i = 0; i < kLevels; ++i) {
    const unsigned c = cnt[i];
    if (l < 0 && slot < total + c) l = i, first = total;
    total += c;
  }
  if (slot == 0 && lane == 0) out_counts[blockIdx.y] = total;
  if (l < 0 || slot >= (unsigned)kp_cap) return;
  const OrbCand c = cand[im.coff[l] + (slot - first)];
  const float sf = im.scale[l];  // getScale(level, 0, 1.2f)
  const size_t o = (size_t)blockIdx.y * (size_t)kp_cap + slot;
  float fx = (float)c.x, fy = (float)c.y;
  if (l != 0) fx *= sf, fy *= sf;  // what detect() returns
  if (lane == 0) {
    cbh_keypoint k;
    k.x = fx, k.y = fy;
    k.size = 31 * sf;
    k.angle = c.angle;
    k.response = c.response;
    k.octave = l;
    out_kp[o] = k;
  }
  // compute(): pt *= 1/scale; descriptors at cvRound(pt); pt *= scale
  float lx = fx, ly = fy;
  if (l != 0) {
    const float inv = 1 / sf;
    lx *= inv, ly *= inv;
  }
  if (out_after && lane == 0) {
    float ax = lx, ay = ly;
    if (l != 0) ax *= sf, ay *= sf;
    out_after[2 * o] = ax, out_after[2 * o + 1] = ay;
  }
  if (!out_desc) return;
  const int cx = cv_round_f(lx), cy = cv_round_f(ly);
  const float a = c.ca, b = c.sa;
  const unsigned char* __restrict__ ctr = blurred + im.soff[l] + (size_t)cy * im.spitch[l] + cx;
  const int step = (int)im.spitch[l];
  unsigned long long bits[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int pw = pat[64 * m + lane];
    const float x0 = (float)(signed char)(pw & 255), y0 = (float)(signed char)((pw >> 8) & 255);
    const float x1 = (float)(signed char)((pw >> 16) & 255), y1 = (float)(signed char)((pw >> 24) & 255);
    const int iy0 = cv_round_f(x0 * b + y0 * a), ix0 = cv_round_f(x0 * a - y0 * b);
    const int iy1 = cv_round_f(x1 * b + y1 * a), ix1 = cv_round_f(x1 * a - y1 * b);
    const int t0 = ctr[iy0 * step + ix0], t1 = ctr[iy1 * step + ix1];
    bits[m] = __ballot(t0 < t1);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(out_desc + o * 32)[lane] = bits[lane];
}

// the extractor alone on keypoints the caller provides (ORB::operator() with useProvidedKeypoints = true): the host has
// filtered and grouped them by octave; pt *= 1/scale, descriptor at cvRound(pt), pt *= scale
struct OrbGivenKp {
  unsigned img;
  int octave;
  float x, y, angle;
};
__global__ __launch_bounds__(256) void k_orb_describe_given(const OrbImage* __restrict__ images,
                                                            const unsigned char* __restrict__ blurred,
                                                            const OrbGivenKp* __restrict__ kps, unsigned nk,
                                                            const int* __restrict__ pat,
                                                            float* __restrict__ out_xy /* 2 per keypoint */,
                                                            unsigned char* __restrict__ out_desc) {
  const int lane = (int)(threadIdx.x & 63);
  const unsigned k = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (k >= nk) return;
  const OrbGivenKp kp = kps[k];
  const OrbImage& im = images[kp.img];
  const int l = kp.octave;
  const float sf = im.scale[l];
  float lx = kp.x, ly = kp.y;
  if (l != 0) {
    const float inv = 1 / sf;
    lx *= inv, ly *= inv;
  }
  const int cx = cv_round_f(lx), cy = cv_round_f(ly);
  if (lane == 0) {
    float ax = lx, ay = ly;
    if (l != 0) ax *= sf, ay *= sf;
    out_xy[2 * k] = ax, out_xy[2 * k + 1] = ay;
  }
  float angle = kp.angle;
  angle *= (float)(3.14159265358979323846 / 180.f);
  const float a = (float)cos((double)angle), b = (float)sin((double)angle);
  const unsigned char* __restrict__ ctr = blurred + im.soff[l] + (size_t)cy * im.spitch[l] + cx;
  const int step = (int)im.spitch[l];
  unsigned long long bits[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int pw = pat[64 * m + lane];
    const float x0 = (float)(signed char)(pw & 255), y0 = (float)(signed char)((pw >> 8) & 255);
    const float x1 = (float)(signed char)((pw >> 16) & 255), y1 = (float)(signed char)((pw >> 24) & 255);
    const int iy0 = cv_round_f(x0 * b + y0 * a), ix0 = cv_round_f(x0 * a - y0 * b);
    const int iy1 = cv_round_f(x1 * b + y1 * a), ix1 = cv_round_f(x1 * a - y1 * b);
    const int t0 = ctr[iy0 * step + ix0], t1 = ctr[iy1 * step + ix1];
    bits[m] = __ballot(t0 < t1);
  }
  if (lane < 4) reinterpret_cast<unsigned long long*>(out_desc + (size_t)k * 32)[lane] = bits[lane];
}

int g_retain_order = 1;  // "orb_retain_order": 1 (default) what libstdc++'s nth_element + partition leave -- cbird's Linux
                         // builds; 0 canonical (every tie kept, raster order)
std::mutex g_pat_mu;
OrbPattern g_pattern;
bool g_have_pattern = false;

void gauss7_kernel(int* k) {  // getGaussianKernel(7, 2.0, CV_32F) -> convertTo(CV_32S, 256)
  const int n = 7;
  const double sigma = 2.0;
  float cf[7];
  const double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    cf[i] = (float)std::exp(scale2X * x * x);
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = (int)std::nearbyintf(cf[i] * 256.f);
  }
}

float get_scale(int level) { return (float)std::pow((double)1.2f, (double)level); }

void features_per_level(int nfeatures, int* out) {  // orb.cpp computeKeyPoints
  const float factor = (float)(1.0 / (double)1.2f);
  float ndesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kLevels));
  int sum = 0;
  for (int level = 0; level < kLevels - 1; ++level) {
    out[level] = (int)std::nearbyint((double)ndesired);
    sum += out[level];
    ndesired *= factor;
  }
  out[kLevels - 1] = std::max(nfeatures - sum, 0);
}

struct OrbPlan {
  std::vector<OrbImage> images;
  unsigned long long pyr_bytes = 0, sc_bytes = 0, cands = 0;
  unsigned max_tiles = 0, max_bwg = 0, max_pitch = 4;
  int max_lev = 0;
  int max_h[kLevels] = {0};
};
// levels, buffer offsets and work lists of every image; level_limit[i] (optional) = levels image i needs at most
OrbPlan make_plan(size_t n, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
                  const uint32_t* img_row_stride, const int* nper, const int* level_limit) {
  OrbPlan pl;
  pl.images.resize(n);
  auto& images = pl.images;
  unsigned long long &pyr_bytes = pl.pyr_bytes, &sc_bytes = pl.sc_bytes, &cands = pl.cands;
  unsigned &max_tiles = pl.max_tiles, &max_bwg = pl.max_bwg, &max_pitch = pl.max_pitch;
  int& max_lev = pl.max_lev;
  int* max_h = pl.max_h;
  for (size_t i = 0; i < n; ++i) {
    OrbImage& im = images[i];
    memset(&im, 0, sizeof im);
    im.src_off = img_off[i];
    im.src_stride = img_row_stride[i];
    const int w = (int)img_w[i], h = (int)img_h[i];
    int nl = 0;
    unsigned tiles = 0, bwg = 0;
    for (int l = 0; l < kLevels; ++l) {
      const float scale = 1 / get_scale(l);
      const int lw = (int)std::nearbyint((double)(w * scale)), lh = (int)std::nearbyint((double)(h * scale));
      if (lw <= 2 * kEdge || lh <= 2 * kEdge || (level_limit && l >= level_limit[i])) break;
      im.w[l] = lw, im.h[l] = lh;
      im.nfeat[l] = nper[l];
      const unsigned P = (unsigned)((lw + 3) & ~3) + 2 * kBorderX;
      im.pitch[l] = P;
      im.poff[l] = pyr_bytes + (unsigned long long)kBorderY * P + kBorderX;
      pyr_bytes += ((unsigned long long)P * (lh + 2 * kBorderY) + 15) & ~15ull;
      max_pitch = std::max(max_pitch, P);
      max_h[l] = std::max(max_h[l], lh);
      im.spitch[l] = (unsigned)((lw + 15) & ~15);
      im.soff[l] = sc_bytes;
      sc_bytes += (unsigned long long)im.spitch[l] * lh;
      const int iw = lw - 2 * kEdge, ih = lh - 2 * kEdge;
      im.coff[l] = cands;
      cands += (unsigned long long)((iw + 1) / 2) * ((ih + 1) / 2);  // strict 3x3 maxima cannot be neighbours
      im.tile_first[l] = tiles;
      tiles += (unsigned)((iw + kTileW - 1) / kTileW) * (unsigned)((ih + kTileH - 1) / kTileH);
      im.bwg_first[l] = bwg;
      bwg += (unsigned)((((lw + 3) >> 2) * ((lh + kBlurRows - 1) / kBlurRows) + 255) / 256);
      im.scale[l] = get_scale(l);
      nl = l + 1;
    }
    for (int l = nl; l <= kLevels; ++l) im.tile_first[l] = tiles, im.bwg_first[l] = bwg;
    im.nlev = nl;
    max_tiles = std::max(max_tiles, tiles);
    max_bwg = std::max(max_bwg, bwg);
    max_lev = std::max(max_lev, nl);
  }
  return pl;
}

}  // namespace

void set_orb_retain_order(int v) {
  if (v == 0 || v == 1) g_retain_order = v;
}

int launch_retain_best(const float* d_resp, uint32_t cnt, int n_points, int depth_limit, uint32_t* d_order,
                       uint32_t* d_count, hipStream_t s) {
  RetainScratch rs = {nullptr, nullptr, nullptr, nullptr};
  void* block = nullptr;
  const size_t slots = (size_t)cnt + 1;
  hipError_t e = cbh::malloc_async(&block, slots * 16, s);
  if (e == hipSuccess) {
    rs.key = (float*)block, rs.ref = (unsigned*)block + slots, rs.lp = (unsigned*)block + 2 * slots,
    rs.ra = (unsigned*)block + 3 * slots;
    hipLaunchKernelGGL(k_retain_best, dim3(1), dim3(256), 0, s, d_resp, (int)cnt, n_points, depth_limit, rs, d_order,
                       d_count);
    e = hipGetLastError();
  }
  if (block) (void)cbh::free_async(block, s);
  if (e != hipSuccess) {
    set_last_error("orb retain_best", e);
    return e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  return CBH_OK;
}

int orb_set_pattern(const int8_t* xy) {
  if (!xy) return CBH_E_INVAL;
  for (int i = 0; i < 1024; ++i)
    if (xy[i] < -15 || xy[i] > 15) return CBH_E_INVAL;
  std::lock_guard<std::mutex> lk(g_pat_mu);
  memcpy(g_pattern.v, xy, 1024);
  g_have_pattern = true;
  return CBH_OK;
}

int launch_orb(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
               const uint32_t* img_row_stride, int nfeatures, int kp_cap, cbh_keypoint* d_kp, float* d_kp_after,
               uint8_t* d_desc, uint32_t* d_counts, hipStream_t s) {
  OrbPattern pat;
  if (d_desc) {
    std::lock_guard<std::mutex> lk(g_pat_mu);
    if (!g_have_pattern) return CBH_E_INVAL;  // no built-in pattern: it is OpenCV's learned table, an input
    pat = g_pattern;
  } else {
    memset(&pat, 0, sizeof pat);
  }
  int nper[kLevels];
  features_per_level(nfeatures, nper);
  OrbPlan pl = make_plan(n, img_off, img_w, img_h, img_row_stride, nper, nullptr);
  const unsigned long long pyr_bytes = pl.pyr_bytes, sc_bytes = pl.sc_bytes, cands = pl.cands;
  const unsigned max_tiles = pl.max_tiles, max_bwg = pl.max_bwg, max_pitch = pl.max_pitch;
  const int max_lev = pl.max_lev;
  const int* max_h = pl.max_h;
  const std::vector<OrbImage>& images = pl.images;
  OrbImage* d_images = nullptr;
  unsigned char *d_pyr = nullptr, *d_sc = nullptr;
  OrbCand *d_cand = nullptr, *d_cand_tmp = nullptr;
  unsigned* d_lc = nullptr;
  int* d_pat = nullptr;
  void* d_retain = nullptr;
  const int retain_order = g_retain_order;
  RetainScratch rs = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = cbh::malloc_async(p, std::max<size_t>(bytes, 256), s);
  };
  alloc((void**)&d_images, n * sizeof(OrbImage));
  alloc((void**)&d_pyr, pyr_bytes + 64);
  alloc((void**)&d_sc, sc_bytes + 64);
  alloc((void**)&d_cand, (cands + 1) * sizeof(OrbCand));
  if (retain_order == 1) {  // keys, raster positions and the two stop lists of every level + the gathered list
    alloc((void**)&d_cand_tmp, (cands + 1) * sizeof(OrbCand));
    alloc(&d_retain, (cands + 1) * 16);
    if (e == hipSuccess)
      rs.key = (float*)d_retain, rs.ref = (unsigned*)d_retain + (cands + 1), rs.lp = (unsigned*)d_retain + 2 * (cands + 1),
      rs.ra = (unsigned*)d_retain + 3 * (cands + 1);
  }
  alloc((void**)&d_lc, n * kLevels * sizeof(unsigned));
  alloc((void**)&d_pat, sizeof pat);
  int rc = CBH_OK;
  // (pageable sources: hipMemcpyAsync has staged them when it returns)
  if (e == hipSuccess) e = hipMemcpyAsync(d_images, images.data(), n * sizeof(OrbImage), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pat, &pat, sizeof pat, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(d_sc, 0, sc_bytes + 64, s);  // k_orb_fast writes the non-zero scores only
  if (e == hipSuccess) {
    const unsigned ny = (unsigned)n;
    if (max_lev >= 1) hipLaunchKernelGGL(k_orb_level0, dim3(32, ny), dim3(256), 0, s, d_images, d_imgs, d_pyr);
    if ((size_t)max_pitch * 8 > 64 * 1024)  // images wider than 8184 pixels: the x tables pass the default LDS limit
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_orb_resize), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(max_pitch * 8));
    for (int l = 1; l < max_lev; ++l)
      hipLaunchKernelGGL(k_orb_resize,
                         dim3((unsigned)((max_h[l] + 2 * kBorderY + kResizeRows - 1) / kResizeRows), ny), dim3(256),
                         (size_t)max_pitch * 8, s, d_images, l, d_pyr);
    if (max_tiles) hipLaunchKernelGGL(k_orb_fast, dim3(max_tiles, ny), dim3(256), 0, s, d_images, d_pyr, d_sc);
    hipLaunchKernelGGL(k_orb_select, dim3(kLevels, ny), dim3(256), 0, s, d_images, d_pyr, d_sc, d_cand, d_lc,
                       retain_order, rs, d_cand_tmp);
    if (d_desc && max_bwg) {
      GaussK g;
      gauss7_kernel(g.k);
      hipLaunchKernelGGL(k_orb_blur, dim3(max_bwg, ny), dim3(256), 0, s, d_images, d_pyr, d_lc, g, d_sc);
    }
    hipLaunchKernelGGL(k_orb_describe, dim3((unsigned)((std::max(kp_cap, 1) + 3) / 4), ny), dim3(256), 0, s, d_images,
                       d_sc, d_cand, d_lc, d_pat, kp_cap, d_kp, d_kp_after, d_desc, d_counts);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    set_last_error("orb", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  for (void* p : {(void*)d_images, (void*)d_pyr, (void*)d_sc, (void*)d_cand, (void*)d_lc, (void*)d_pat,
                  (void*)d_cand_tmp, d_retain})
    if (p) (void)cbh::free_async(p, s);
  return rc;
}

// makeKeyPointDescriptors on provided keypoints (host arrays; d_imgs on the device).  out_kp: the keypoints as
// compute() leaves them (border filter, grouped by octave, pt round trip), image i owns [out_first[i], out_first[i+1])
int launch_orb_describe(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                        const uint32_t* img_h, const uint32_t* img_row_stride, const cbh_keypoint* kp,
                        const uint32_t* kp_first, cbh_keypoint* out_kp, uint8_t* out_desc, uint32_t* out_first,
                        hipStream_t s) {
  OrbPattern pat;
  {
    std::lock_guard<std::mutex> lk(g_pat_mu);
    if (!g_have_pattern) return CBH_E_INVAL;
    pat = g_pattern;
  }
  // ORB::operator(): runByImageBorder(keypoints, image size, edgeThreshold) -- Rect::contains on the rounded point --
  // then one list per octave
  std::vector<OrbGivenKp> flat;
  std::vector<int> limit(n, 0);
  std::vector<unsigned> lc(n * kLevels, 0u);
  out_first[0] = 0;
  for (size_t i = 0; i < n; ++i) {
    const int w = (int)img_w[i], h = (int)img_h[i];
    std::vector<const cbh_keypoint*> kept;
    if (w > 2 * kEdge && h > 2 * kEdge)
      for (uint32_t j = kp_first[i]; j < kp_first[i + 1]; ++j) {
        const int px = (int)std::nearbyint((double)kp[j].x), py = (int)std::nearbyint((double)kp[j].y);
        if (px >= kEdge && px < w - kEdge && py >= kEdge && py < h - kEdge) kept.push_back(&kp[j]);
      }
    int levels = 0;
    for (const cbh_keypoint* k : kept) levels = std::max(levels, std::max(k->octave, 0) + 1);
    if (levels > kLevels) return CBH_E_INVAL;
    for (int l = 0; l < levels; ++l)
      for (const cbh_keypoint* k : kept)
        if (k->octave == l) {
          // its level must exist and hold the point: ORB's own keypoints always do
          const float scale = 1 / get_scale(l);
          const int lw = (int)std::nearbyint((double)(w * scale)), lh = (int)std::nearbyint((double)(h * scale));
          const int cx = (int)std::nearbyint((double)(k->x * scale)), cy = (int)std::nearbyint((double)(k->y * scale));
          if (lw <= 2 * kEdge || lh <= 2 * kEdge || cx < 22 || cy < 22 || cx >= lw - 22 || cy >= lh - 22)
            return CBH_E_INVAL;
          flat.push_back(OrbGivenKp{(unsigned)i, l, k->x, k->y, k->angle});
          out_kp[flat.size() - 1] = *k;
          lc[i * kLevels + l]++;
        }
    limit[i] = levels;
    out_first[i + 1] = (uint32_t)flat.size();
  }
  const size_t nk = flat.size();
  if (nk == 0) return CBH_OK;
  int nper[kLevels] = {0};
  OrbPlan pl = make_plan(n, img_off, img_w, img_h, img_row_stride, nper, limit.data());
  OrbImage* d_images = nullptr;
  unsigned char *d_pyr = nullptr, *d_sc = nullptr, *d_desc = nullptr;
  unsigned* d_lc = nullptr;
  int* d_pat = nullptr;
  OrbGivenKp* d_kps = nullptr;
  float* d_xy = nullptr;
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = cbh::malloc_async(p, std::max<size_t>(bytes, 256), s);
  };
  alloc((void**)&d_images, n * sizeof(OrbImage));
  alloc((void**)&d_pyr, pl.pyr_bytes + 64);
  alloc((void**)&d_sc, pl.sc_bytes + 64);
  alloc((void**)&d_lc, n * kLevels * sizeof(unsigned));
  alloc((void**)&d_pat, sizeof pat);
  alloc((void**)&d_kps, nk * sizeof(OrbGivenKp));
  alloc((void**)&d_xy, nk * 2 * sizeof(float));
  alloc((void**)&d_desc, nk * 32);
  std::vector<float> xy(nk * 2);
  int rc = CBH_OK;
  if (e == hipSuccess) e = hipMemcpyAsync(d_images, pl.images.data(), n * sizeof(OrbImage), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pat, &pat, sizeof pat, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_lc, lc.data(), lc.size() * sizeof(unsigned), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_kps, flat.data(), nk * sizeof(OrbGivenKp), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const unsigned ny = (unsigned)n;
    if (pl.max_lev >= 1) hipLaunchKernelGGL(k_orb_level0, dim3(32, ny), dim3(256), 0, s, d_images, d_imgs, d_pyr);
    if ((size_t)pl.max_pitch * 8 > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_orb_resize), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(pl.max_pitch * 8));
    for (int l = 1; l < pl.max_lev; ++l)
      hipLaunchKernelGGL(k_orb_resize,
                         dim3((unsigned)((pl.max_h[l] + 2 * kBorderY + kResizeRows - 1) / kResizeRows), ny), dim3(256),
                         (size_t)pl.max_pitch * 8, s, d_images, l, d_pyr);
    GaussK g;
    gauss7_kernel(g.k);
    if (pl.max_bwg) hipLaunchKernelGGL(k_orb_blur, dim3(pl.max_bwg, ny), dim3(256), 0, s, d_images, d_pyr, d_lc, g, d_sc);
    hipLaunchKernelGGL(k_orb_describe_given, dim3((unsigned)((nk + 3) / 4)), dim3(256), 0, s, d_images, d_sc, d_kps,
                       (unsigned)nk, d_pat, d_xy, d_desc);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(xy.data(), d_xy, nk * 2 * sizeof(float), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(out_desc, d_desc, nk * 32, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    set_last_error("orb describe", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  } else {
    for (size_t k = 0; k < nk; ++k) out_kp[k].x = xy[2 * k], out_kp[k].y = xy[2 * k + 1];
  }
  for (void* p : {(void*)d_images, (void*)d_pyr, (void*)d_sc, (void*)d_lc, (void*)d_pat, (void*)d_kps, (void*)d_xy,
                  (void*)d_desc})
    if (p) (void)cbh::free_async(p, s);
  return rc;
}

}  // namespace cbh

extern "C" {

int cbh_orb_set_pattern(const int8_t* xy) { return cbh::orb_set_pattern(xy); }

int cbh_orb_retain_best_dev(const void* d_responses, uint32_t count, int n_points, int depth_limit, void* d_order,
                            void* d_count, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (!d_order || !d_count || (count && !d_responses) || count > (1u << 30)) return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = cbh::launch_retain_best((const float*)d_responses, count, n_points, depth_limit, (uint32_t*)d_order,
                                   (uint32_t*)d_count, s);
  if (rc == CBH_OK && !s) {
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      cbh::set_last_error("orb retain_best sync", e);
      rc = CBH_E_HIP;
    }
  }
  return rc;
}

int cbh_orb_dev(const void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
                const uint32_t* img_row_stride, int nfeatures, int kp_cap, void* d_kp, void* d_kp_after, void* d_desc,
                void* d_counts, int device, void* stream) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!d_imgs || !img_off || !img_w || !img_h || !img_row_stride || !d_kp || !d_counts || nfeatures < 0 ||
      nfeatures > 100000 || kp_cap < 1 || n > 65535)
    return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (img_w[i] == 0 || img_h[i] == 0 || img_w[i] > 8192 || img_h[i] > 8192 || img_row_stride[i] < img_w[i])
      return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = (hipStream_t)stream;
  int rc = cbh::launch_orb((const uint8_t*)d_imgs, n, img_off, img_w, img_h, img_row_stride, nfeatures, kp_cap,
                           (cbh_keypoint*)d_kp, (float*)d_kp_after, (uint8_t*)d_desc, (uint32_t*)d_counts, s);
  if (rc == CBH_OK && !s) {  // the NULL stream is synchronous by contract
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      cbh::set_last_error("orb sync", e);
      rc = CBH_E_HIP;
    }
  }
  return rc;
}

int cbh_orb(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
            const uint32_t* img_h, const uint32_t* img_row_stride, int nfeatures, int kp_cap, cbh_keypoint* kp,
            float* kp_after, uint8_t* desc, uint32_t* counts, int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (n == 0) return CBH_OK;
  if (!imgs || !kp || !counts || kp_cap < 1) return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (!img_off || !img_w || !img_h || !img_row_stride ||
        img_off[i] + (uint64_t)(img_h[i] - 1) * img_row_stride[i] + img_w[i] > imgs_bytes)
      return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = nullptr;
  uint8_t *d_imgs = nullptr, *d_desc = nullptr;
  cbh_keypoint* d_kp = nullptr;
  float* d_after = nullptr;
  uint32_t* d_counts = nullptr;
  const size_t slots = n * (size_t)kp_cap;
  hipError_t e;
  int rc = CBH_OK;
  if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(&d_imgs, imgs_bytes)) != hipSuccess || (e = hipMalloc(&d_kp, slots * sizeof(cbh_keypoint))) != hipSuccess ||
      (kp_after && (e = hipMalloc(&d_after, slots * 2 * sizeof(float))) != hipSuccess) ||
      (desc && (e = hipMalloc(&d_desc, slots * 32)) != hipSuccess) ||
      (e = hipMalloc(&d_counts, n * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMemcpyAsync(d_imgs, imgs, imgs_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) {
    cbh::set_last_error("orb setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  if (rc == CBH_OK)
    rc = cbh_orb_dev(d_imgs, n, img_off, img_w, img_h, img_row_stride, nfeatures, kp_cap, d_kp, d_after, d_desc, d_counts,
                     device, s);
  if (rc == CBH_OK) {
    if ((e = hipMemcpyAsync(kp, d_kp, slots * sizeof(cbh_keypoint), hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (kp_after && (e = hipMemcpyAsync(kp_after, d_after, slots * 2 * sizeof(float), hipMemcpyDeviceToHost, s)) != hipSuccess) ||
        (desc && (e = hipMemcpyAsync(desc, d_desc, slots * 32, hipMemcpyDeviceToHost, s)) != hipSuccess) ||
        (e = hipMemcpyAsync(counts, d_counts, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess) {
      cbh::set_last_error("orb fetch", e);
      rc = CBH_E_HIP;
    }
  }
  if (s) (void)hipStreamSynchronize(s);
  for (void* p : {(void*)d_imgs, (void*)d_kp, (void*)d_after, (void*)d_desc, (void*)d_counts})
    if (p) (void)hipFree(p);
  if (s) cbh::stream_destroy(s);
  return rc;
}

int cbh_orb_describe(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                     const uint32_t* img_h, const uint32_t* img_row_stride, const cbh_keypoint* kp,
                     const uint32_t* kp_first, cbh_keypoint* out_kp, uint8_t* out_desc, uint32_t* out_first, int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (!out_first) return CBH_E_INVAL;
  out_first[0] = 0;
  if (n == 0) return CBH_OK;
  if (!imgs || !img_off || !img_w || !img_h || !img_row_stride || !kp_first || n > 65535) return CBH_E_INVAL;
  if (kp_first[n] && (!kp || !out_kp || !out_desc)) return CBH_E_INVAL;
  for (size_t i = 0; i < n; ++i)
    if (kp_first[i + 1] < kp_first[i] || img_w[i] == 0 || img_h[i] == 0 || img_w[i] > 8192 || img_h[i] > 8192 ||
        img_row_stride[i] < img_w[i] || img_off[i] + (uint64_t)(img_h[i] - 1) * img_row_stride[i] + img_w[i] > imgs_bytes)
      return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  hipStream_t s = nullptr;
  uint8_t* d_imgs = nullptr;
  hipError_t e;
  int rc = CBH_OK;
  if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(&d_imgs, imgs_bytes)) != hipSuccess ||
      (e = hipMemcpyAsync(d_imgs, imgs, imgs_bytes, hipMemcpyHostToDevice, s)) != hipSuccess) {
    cbh::set_last_error("orb describe setup", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  if (rc == CBH_OK)
    rc = cbh::launch_orb_describe(d_imgs, n, img_off, img_w, img_h, img_row_stride, kp, kp_first, out_kp, out_desc,
                                  out_first, s);
  if (s) (void)hipStreamSynchronize(s);
  if (d_imgs) (void)hipFree(d_imgs);
  if (s) cbh::stream_destroy(s);
  return rc;
}

}  // extern "C"
