// orb.hip -- SURVEY.md section 8 row a11: ORB keypoints and rBRIEF descriptors as cbird configures OpenCV 2.4's ORB
//   Media::makeKeyPoints            /root/reference/src/media.cpp:859-866   OrbFeatureDetector(n, 1.2f, 12, 31, 0, 2, HARRIS_SCORE, 31)
//   Media::makeKeyPointDescriptors  /root/reference/src/media.cpp:868-872   OrbDescriptorExtractor() (256 bits, WTA_K 2)
// for a batch of grey images of any sizes resident in HBM (cbird feeds <= 400 px on the longest side,
// /root/reference/src/scanner.cpp:876).  Bit-exact against oracle/orb_oracle.c, which carries the statement of what
// is and is not pinned against the cbird binary (the learned rBRIEF pattern is an INPUT: cbh_orb_set_pattern;
// retainBest's tie rule is canonical: ties kept, raster order).
//
// Decomposition (one launch each, all images of the batch at once; nothing visits the host between them):
//   k_orb_resize   x (levels - 1)   pyramid level l from l-1: cv::resize INTER_LINEAR's 8-bit fixed-point path; the
//                                   coefficient tables of a workgroup's rows are formed in LDS (double arithmetic
//                                   as the library's, IEEE on the device).  Level 0 is the caller's image, in place.
//                                   Only the levels that can hold a keypoint (both sides > 62) are built.
//   k_orb_fast                      FAST-9/16 score + 3x3 non-maximum suppression per 64x16 tile staged in LDS:
//                                   16-bit brighter / darker masks, "9 contiguous" by four shift-ands, the score
//                                   (largest threshold that keeps the corner) only for the rare corners.  Writes the
//                                   suppressed score map of the keypoint region [31, w-31) x [31, h-31).
//   k_orb_select                    one workgroup per (image, level): histogram of the scores -> n-th best score
//                                   (retainBest(2N) on 8-bit keys = counting select) -> ordered compaction ->
//                                   Harris response per candidate -> radix select of the N-th best float ->
//                                   ordered compaction -> intensity-centroid orientation (one wave per keypoint).
//   k_orb_blur                      7x7 sigma-2 Gaussian in OpenCV's 8-bit fixed point per tile (levels with keypoints)
//   k_orb_describe                  one wave per keypoint: lane t evaluates tests t, t+64, t+128, t+192 of the
//                                   rotated pattern, four ballots are the 256 bits; also scales the keypoints to
//                                   image coordinates and writes them level by level per image.
// All of it is byte / small-integer work on L2-resident pyramids (a 400x300 image: 120 KB + 270 KB of levels); the
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
constexpr int kHalfPatch = 15;    // patchSize 31
constexpr int kFastT = 20;        // orb.cpp: FastFeatureDetector fd(20, true)
constexpr float kHarrisK = 0.04f;

struct OrbImage {
  unsigned long long src_off;  // level 0 = the caller's image
  unsigned src_stride;
  int nlev;                    // levels that can hold keypoints (both sides > 2 * kEdge); they form a prefix
  int w[kLevels], h[kLevels];
  unsigned pitch[kLevels];               // pitch[0] = src_stride
  unsigned long long poff[kLevels];      // byte offset of level l >= 1 in the pyramid buffer
  unsigned long long soff[kLevels];      // byte offset of level l in the score / blurred buffer (pitch = w rounded up to 4)
  unsigned spitch[kLevels];
  unsigned long long coff[kLevels];      // first candidate slot of level l
  int nfeat[kLevels];
  unsigned tile_first[kLevels + 1];      // FAST tiles (keypoint region): level l owns [tile_first[l], tile_first[l+1])
  unsigned btile_first[kLevels + 1];     // blur tiles (whole level)
  float scale[kLevels];                  // getScale(level): (float)pow((double)1.2f, level), from the host's libm
};

struct OrbPattern {
  signed char v[1024];
};

struct OrbCand {  // one candidate / keypoint of a level, in level coordinates
  unsigned short x, y;
  float response;
  float angle;
};

__device__ __forceinline__ const unsigned char* level_ptr(const OrbImage& im, int l, const unsigned char* imgs,
                                                          const unsigned char* pyr) {
  return l == 0 ? imgs + im.src_off : pyr + im.poff[l];
}

__device__ __forceinline__ int cv_round_f(float v) { return (int)rintf(v); }  // cvRound: half to even
__device__ __forceinline__ short sat_short_rn(float v) {
  const float r = rintf(v);
  return (short)(r < -32768.f ? -32768.f : r > 32767.f ? 32767.f : r);
}

// ---- pyramid: cv::resize INTER_LINEAR, 8UC1 (oracle: orc_resize_linear_u8_cv) ---------------------------------------
constexpr int kResizeRows = 32;
__global__ __launch_bounds__(256) void k_orb_resize(const OrbImage* __restrict__ images, int level,
                                                    const unsigned char* __restrict__ imgs,
                                                    unsigned char* __restrict__ pyr) {
  extern __shared__ __attribute__((aligned(8))) unsigned char s_res[];
  const OrbImage& im = images[blockIdx.y];
  if (level >= im.nlev) return;
  const int dw = im.w[level], dh = im.h[level], sw = im.w[level - 1], sh = im.h[level - 1];
  const int y0 = (int)blockIdx.x * kResizeRows;
  if (y0 >= dh) return;
  int* __restrict__ xofs = reinterpret_cast<int*>(s_res);
  short* __restrict__ xc = reinterpret_cast<short*>(s_res + (size_t)dw * 4);  // (c0, c1) pairs
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  for (int d = threadIdx.x; d < dw; d += 256) {
    float f = (float)((d + 0.5) * scale_x - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) f = 0.f, s = 0;
    if (s >= sw - 1) f = 0.f, s = sw - 1;
    xofs[d] = s;
    xc[2 * d] = sat_short_rn((1.f - f) * 2048.f);
    xc[2 * d + 1] = sat_short_rn(f * 2048.f);
  }
  __syncthreads();
  const unsigned char* __restrict__ src = level_ptr(im, level - 1, imgs, pyr);
  const unsigned sp = im.pitch[level - 1], dp = im.pitch[level];
  unsigned char* __restrict__ dst = pyr + im.poff[level];
  const int y1 = min(dh, y0 + kResizeRows);
  for (int dy = y0; dy < y1; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = sat_short_rn((1.f - fy) * 2048.f), b1 = sat_short_rn(fy * 2048.f);
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    const unsigned char* __restrict__ S0 = src + (size_t)r0 * sp;
    const unsigned char* __restrict__ S1 = src + (size_t)r1 * sp;
    for (int dx = threadIdx.x; dx < dw; dx += 256) {
      const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
      const int a0 = xc[2 * dx], a1 = xc[2 * dx + 1];
      const int D0 = S0[sx] * a0 + S0[sx1] * a1;
      const int D1 = S1[sx] * a0 + S1[sx1] * a1;
      const int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
      dst[(size_t)dy * dp + dx] = (unsigned char)min(max(v, 0), 255);
    }
  }
}

// ---- FAST-9/16 + non-maximum suppression (oracle: orc_fast_nms_scores restricted to the keypoint region) -----------
constexpr int kTileW = 64, kTileH = 16;
constexpr int kPxW = kTileW + 8, kPxH = kTileH + 8;  // pixels: outputs + 1 (NMS) + 3 (circle) on every side
constexpr int kRawW = kTileW + 2, kRawH = kTileH + 2;

__device__ __forceinline__ int fast_score_at(const unsigned char* __restrict__ c /* centre in the LDS tile */) {
  // circle offsets in the order of fast.cpp's offsets16
  constexpr int ox[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int oy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const int v = c[0];
  int d[16];
  unsigned dark = 0, bright = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    d[k] = v - (int)c[oy[k] * kPxW + ox[k]];
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
  if (!run9(dark) && !run9(bright)) return 0;
  // cornerScore<16>: max(threshold, max over the 16 arcs of min(d), max over the arcs of min(-d)) - 1
  int best = kFastT;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int mn = d[s], mx = d[s];
#pragma unroll
    for (int j = 1; j < 9; ++j) {
      mn = min(mn, d[(s + j) & 15]);
      mx = max(mx, d[(s + j) & 15]);
    }
    best = max(best, max(mn, -mx));
  }
  return best - 1;
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
                                                  const unsigned char* __restrict__ imgs,
                                                  const unsigned char* __restrict__ pyr,
                                                  unsigned char* __restrict__ scores) {
  __shared__ unsigned char s_px[kPxH * kPxW];
  __shared__ unsigned char s_raw[kRawH * kRawW];
  const OrbImage& im = images[blockIdx.y];
  int l, tx, ty;
  if (!find_tile(im, blockIdx.x, &l, &tx, &ty)) return;
  const int w = im.w[l], h = im.h[l];
  const unsigned char* __restrict__ src = level_ptr(im, l, imgs, pyr);
  const unsigned sp = im.pitch[l];
  const int ox0 = kEdge + tx * kTileW, oy0 = kEdge + ty * kTileH;  // first output pixel of the tile
  const int px0 = ox0 - 4, py0 = oy0 - 4;
  for (int i = threadIdx.x; i < kPxH * kPxW; i += 256) {
    const int r = i / kPxW, c = i - r * kPxW;
    const int y = min(py0 + r, h - 1), x = min(px0 + c, w - 1);  // overhanging tiles: clamped (never used)
    s_px[i] = src[(size_t)y * sp + x];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRawH * kRawW; i += 256) {
    const int r = i / kRawW, c = i - r * kRawW;
    s_raw[i] = (unsigned char)fast_score_at(s_px + (r + 3) * kPxW + (c + 3));
  }
  __syncthreads();
  unsigned char* __restrict__ out = scores + im.soff[l];
  const unsigned op = im.spitch[l];
  for (int i = threadIdx.x; i < kTileH * kTileW; i += 256) {
    const int r = i / kTileW, c = i - r * kTileW;
    const int x = ox0 + c, y = oy0 + r;
    if (x >= w - kEdge || y >= h - kEdge) continue;
    const unsigned char* __restrict__ q = s_raw + (r + 1) * kRawW + (c + 1);
    const int s = q[0];
    const bool keep = s != 0 && s > q[1] && s > q[-1] && s > q[-kRawW - 1] && s > q[-kRawW] && s > q[-kRawW + 1] &&
                      s > q[kRawW - 1] && s > q[kRawW] && s > q[kRawW + 1];
    out[(size_t)y * op + x] = keep ? (unsigned char)s : (unsigned char)0;
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
  const unsigned char* __restrict__ p0 = img + (size_t)(y - 4) * pitch + (x - 4);
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

__global__ __launch_bounds__(256) void k_orb_select(const OrbImage* __restrict__ images,
                                                    const unsigned char* __restrict__ imgs,
                                                    const unsigned char* __restrict__ pyr,
                                                    const unsigned char* __restrict__ scores,
                                                    OrbCand* __restrict__ cand, unsigned* __restrict__ level_counts) {
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
  const int w = im.w[l], h = im.h[l];
  const int iw = w - 2 * kEdge, ih = h - 2 * kEdge;
  const int area = iw * ih;
  const unsigned char* __restrict__ sc = scores + im.soff[l];
  const unsigned sp = im.spitch[l];
  const int N = im.nfeat[l];
  OrbCand* __restrict__ cd = cand + im.coff[l];
  // -- (a) histogram of the suppressed scores
  s_hist[tid] = 0;
  __syncthreads();
  for (int p = tid; p < area; p += 256) {
    const int y = p / iw, x = p - y * iw;
    const int s = sc[(size_t)(y + kEdge) * sp + (x + kEdge)];
    if (s) atomicAdd(&s_hist[s], 1);
  }
  __syncthreads();
  // -- (b) retainBest(2N): every score >= the 2N-th best survives
  if (tid == 0) {
    int total = 0;
    for (int s = 255; s >= 1; --s) total += s_hist[s];
    int thr = 1;
    if (2 * N == 0) {
      thr = 256;
    } else if (total > 2 * N) {
      int acc = 0;
      for (int s = 255; s >= 1; --s) {
        acc += s_hist[s];
        if (acc >= 2 * N) {
          thr = s;
          break;
        }
      }
    }
    s_pick[0] = thr;
  }
  __syncthreads();
  const int thr = s_pick[0];
  // -- (c) ordered compaction (raster order) + Harris response
  int c1 = 0;
  for (int p0 = 0; p0 < area; p0 += 256) {
    const int p = p0 + tid;
    int x = 0, y = 0, keep = 0;
    if (p < area) {
      y = p / iw, x = p - y * iw;
      x += kEdge, y += kEdge;
      keep = sc[(size_t)y * sp + x] >= thr;
    }
    int tot;
    const int pos = block_excl_scan(keep, s_w, &tot);
    if (keep) {
      OrbCand c;
      c.x = (unsigned short)x, c.y = (unsigned short)y;
      c.response = 0.f, c.angle = 0.f;
      cd[c1 + pos] = c;
    }
    c1 += tot;
  }
  __syncthreads();  // the candidate list is visible to the whole workgroup (global memory, same workgroup)
  const unsigned char* __restrict__ img = level_ptr(im, l, imgs, pyr);
  const unsigned ip = im.pitch[l];
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
      if (tid == 0) {
        int acc = 0, b = 255;
        for (; b > 0; --b) {
          if (acc + s_hist[b] >= want) break;
          acc += s_hist[b];
        }
        s_pick[0] = b;
        s_pick[1] = want - acc;
      }
      __syncthreads();
      prefix |= (unsigned)s_pick[0] << shift;
      mask |= 255u << shift;
      want = s_pick[1];
      __syncthreads();
    }
    key_thr = prefix;
  }
  // -- (e) ordered compaction in place (a write never passes the reads of its own or a later chunk)
  int c2 = 0;
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
  if (tid == 0) *out_count = (unsigned)c2;
  // -- (f) orientation: IC_Angle over the circular patch of radius 15, one wave per keypoint, lane = row v
  //        (lane 0: the centre row; lanes 1..15: the row pair +-v)
  const int lane = tid & 63, wv = tid >> 6;
  for (int i = wv; i < c2; i += 4) {
    const int x = cd[i].x, y = cd[i].y;
    const unsigned char* __restrict__ ctr = img + (size_t)y * ip + x;
    int m01 = 0, m10 = 0;
    if (lane == 0) {
      for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * (int)ctr[u];
    } else if (lane <= kHalfPatch) {
      // u_max of orb.cpp for half patch 15 (15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3), one nibble per row
      const int v = lane;
      const int um = (int)((0x3689ABCDDEEEFFFFull >> (4 * v)) & 15ull);
      int v_sum = 0;
      const unsigned char* __restrict__ pp = ctr + (size_t)v * ip;
      const unsigned char* __restrict__ pm = ctr - (size_t)v * ip;
      for (int u = -um; u <= um; ++u) {
        const int vp = pp[u], vm = pm[u];
        v_sum += vp - vm;
        m10 += u * (vp + vm);
      }
      m01 = v * v_sum;
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      m01 += __shfl_xor(m01, d);
      m10 += __shfl_xor(m10, d);
    }
    if (lane == 0) cd[i].angle = fast_atan2_deg((float)m01, (float)m10);
  }
}

// ---- GaussianBlur(7x7, sigma 2) in the library's 8-bit fixed point (oracle: orc_gauss7_blur_u8) ---------------------
struct GaussK {
  int k[7];
};
__device__ __forceinline__ int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return p;
}
__global__ __launch_bounds__(256) void k_orb_blur(const OrbImage* __restrict__ images,
                                                  const unsigned char* __restrict__ imgs,
                                                  const unsigned char* __restrict__ pyr,
                                                  const unsigned* __restrict__ level_counts, GaussK g,
                                                  unsigned char* __restrict__ blurred) {
  constexpr int BW = kTileW, BH = kTileH;
  __shared__ unsigned char s_px[(BH + 6) * (BW + 6)];
  __shared__ int s_row[(BH + 6) * BW];
  const OrbImage& im = images[blockIdx.y];
  const unsigned t = blockIdx.x;
  if (t >= im.btile_first[im.nlev]) return;
  int l = 0;
  while (t >= im.btile_first[l + 1]) ++l;
  if (level_counts[(size_t)blockIdx.y * kLevels + l] == 0) return;
  const int w = im.w[l], h = im.h[l];
  const int ntx = (w + BW - 1) / BW;
  const int ty = (int)((t - im.btile_first[l]) / (unsigned)ntx), tx = (int)(t - im.btile_first[l]) - ty * ntx;
  const int bx = tx * BW, by = ty * BH;
  const unsigned char* __restrict__ src = level_ptr(im, l, imgs, pyr);
  const unsigned sp = im.pitch[l];
  unsigned char* __restrict__ out = blurred + im.soff[l];
  const unsigned op = im.spitch[l];
  for (int i = threadIdx.x; i < (BH + 6) * (BW + 6); i += 256) {
    const int r = i / (BW + 6), c = i - r * (BW + 6);
    const int y = reflect101(min(by + r - 3, h + 2), h), x = reflect101(min(bx + c - 3, w + 2), w);  // overhang: clamped
    s_px[i] = src[(size_t)y * sp + x];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (BH + 6) * BW; i += 256) {
    const int r = i / BW, c = i - r * BW;
    const unsigned char* __restrict__ q = s_px + r * (BW + 6) + c;
    int s = 0;
#pragma unroll
    for (int t = 0; t < 7; ++t) s += g.k[t] * (int)q[t];
    s_row[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BH * BW; i += 256) {
    const int r = i / BW, c = i - r * BW;
    const int x = bx + c, y = by + r;
    if (x >= w || y >= h) continue;
    int s = 0;
#pragma unroll
    for (int t = 0; t < 7; ++t) s += g.k[t] * s_row[(r + t) * BW + c];
    const int v = (s + (1 << 15)) >> 16;
    out[(size_t)y * op + x] = (unsigned char)min(max(v, 0), 255);
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
  float angle = c.angle;
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
  if (lane < 4) reinterpret_cast<unsigned long long*>(out_desc + o * 32)[lane] = bits[lane];
}

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

}  // namespace

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
  std::vector<OrbImage> images(n);
  unsigned long long pyr_bytes = 0, sc_bytes = 0, cands = 0;
  unsigned max_tiles = 0, max_btiles = 0;
  int max_lev = 0, max_w = 1, max_h1 = 1;
  for (size_t i = 0; i < n; ++i) {
    OrbImage& im = images[i];
    memset(&im, 0, sizeof im);
    im.src_off = img_off[i];
    im.src_stride = img_row_stride[i];
    const int w = (int)img_w[i], h = (int)img_h[i];
    int nl = 0;
    unsigned tiles = 0, btiles = 0;
    for (int l = 0; l < kLevels; ++l) {
      const float scale = 1 / get_scale(l);
      const int lw = (int)std::nearbyint((double)(w * scale)), lh = (int)std::nearbyint((double)(h * scale));
      if (lw <= 2 * kEdge || lh <= 2 * kEdge) break;
      im.w[l] = lw, im.h[l] = lh;
      im.nfeat[l] = nper[l];
      if (l == 0) {
        im.pitch[0] = im.src_stride;
      } else {
        im.pitch[l] = (unsigned)((lw + 3) & ~3);
        im.poff[l] = pyr_bytes;
        pyr_bytes += ((unsigned long long)im.pitch[l] * lh + 15) & ~15ull;
        max_w = std::max(max_w, lw);
        max_h1 = std::max(max_h1, lh);
      }
      im.spitch[l] = (unsigned)((lw + 3) & ~3);
      im.soff[l] = sc_bytes;
      sc_bytes += ((unsigned long long)im.spitch[l] * lh + 15) & ~15ull;
      const int iw = lw - 2 * kEdge, ih = lh - 2 * kEdge;
      im.coff[l] = cands;
      cands += (unsigned long long)((iw + 1) / 2) * ((ih + 1) / 2);  // strict 3x3 maxima cannot be neighbours
      im.tile_first[l] = tiles;
      tiles += (unsigned)((iw + kTileW - 1) / kTileW) * (unsigned)((ih + kTileH - 1) / kTileH);
      im.btile_first[l] = btiles;
      btiles += (unsigned)((lw + kTileW - 1) / kTileW) * (unsigned)((lh + kTileH - 1) / kTileH);
      im.scale[l] = get_scale(l);
      nl = l + 1;
    }
    for (int l = nl; l <= kLevels; ++l) im.tile_first[l] = tiles, im.btile_first[l] = btiles;
    im.nlev = nl;
    max_tiles = std::max(max_tiles, tiles);
    max_btiles = std::max(max_btiles, btiles);
    max_lev = std::max(max_lev, nl);
  }
  OrbImage* d_images = nullptr;
  unsigned char *d_pyr = nullptr, *d_sc = nullptr;
  OrbCand* d_cand = nullptr;
  unsigned* d_lc = nullptr;
  int* d_pat = nullptr;
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMallocAsync(p, std::max<size_t>(bytes, 256), s);
  };
  alloc((void**)&d_images, n * sizeof(OrbImage));
  alloc((void**)&d_pyr, pyr_bytes + 64);
  alloc((void**)&d_sc, sc_bytes + 64);
  alloc((void**)&d_cand, (cands + 1) * sizeof(OrbCand));
  alloc((void**)&d_lc, n * kLevels * sizeof(unsigned));
  alloc((void**)&d_pat, sizeof pat);
  int rc = CBH_OK;
  // (pageable sources: hipMemcpyAsync has staged them when it returns)
  if (e == hipSuccess) e = hipMemcpyAsync(d_images, images.data(), n * sizeof(OrbImage), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pat, &pat, sizeof pat, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    const unsigned ny = (unsigned)n;
    for (int l = 1; l < max_lev; ++l) {
      const size_t smem = (size_t)max_w * 8;
      hipLaunchKernelGGL(k_orb_resize, dim3((unsigned)((max_h1 + kResizeRows - 1) / kResizeRows), ny), dim3(256), smem,
                         s, d_images, l, d_imgs, d_pyr);
    }
    if (max_tiles) {
      hipLaunchKernelGGL(k_orb_fast, dim3(max_tiles, ny), dim3(256), 0, s, d_images, d_imgs, d_pyr, d_sc);
    }
    hipLaunchKernelGGL(k_orb_select, dim3(kLevels, ny), dim3(256), 0, s, d_images, d_imgs, d_pyr, d_sc, d_cand, d_lc);
    if (d_desc && max_btiles) {
      GaussK g;
      gauss7_kernel(g.k);
      hipLaunchKernelGGL(k_orb_blur, dim3(max_btiles, ny), dim3(256), 0, s, d_images, d_imgs, d_pyr, d_lc, g, d_sc);
    }
    hipLaunchKernelGGL(k_orb_describe, dim3((unsigned)((std::max(kp_cap, 1) + 3) / 4), ny), dim3(256), 0, s, d_images,
                       d_sc, d_cand, d_lc, d_pat, kp_cap, d_kp, d_kp_after, d_desc, d_counts);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    set_last_error("orb", e);
    rc = e == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP;
  }
  for (void* p : {(void*)d_images, (void*)d_pyr, (void*)d_sc, (void*)d_cand, (void*)d_lc, (void*)d_pat})
    if (p) (void)hipFreeAsync(p, s);
  return rc;
}

}  // namespace cbh

extern "C" {

int cbh_orb_set_pattern(const int8_t* xy) { return cbh::orb_set_pattern(xy); }

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
  if (s) (void)hipStreamDestroy(s);
  return rc;
}

}  // extern "C"
