// kphash.hip -- SURVEY.md section 8 row a12 and the small-image corner of a1: dctHash64 of RECTANGLES of an image --
// Media::makeKeyPointHashes (src/media.cpp:874-923: up to 400 keypoint squares per image, hashed one after the other in
// place), TemplateMatcher's two rectangles, and whole images with a side < 32 (cv::resize enlarges) or of exactly 32 x 32
// (no blur, no resize).  The whole-image kernels live in dcthash.hip; the shared arithmetic in dcthash_common.h.
#include "dcthash_common.h"

namespace cbh {
namespace {

// ---------------------------------------------------------------------------------------------
// Rectangles of an image, hashed one after the other IN PLACE: Media::makeKeyPointHashes (src/media.cpp:874-923)
// calls dctHash64(sub, inPlace = true) on up to 400 keypoint squares of the same grey image, and cv::blur writes
// each blurred square back before the next (overlapping) one is read -- an order dependence inside one image that
// cannot be broken.  The parallelism is across images: one workgroup per image walks its rectangles in order;
// inside a rectangle all 256 lanes work (blur: a lane per column and row segment, sliding K-row sums; resize: four
// of the 1024 tile pixels per lane; stages 3-6: hash_from_tile).  The blur takes the pixels around the rectangle
// from the parent image (cv::FilterEngine is not "isolated" on a view) and reflects only at the parent's edges.
// The same kernel serves whole images with a side < 32 (one rectangle = the image, nothing written back): there
// cv::resize(INTER_AREA) enlarges and runs its 2-tap fixed-point bilinear emulation (mode 3).
struct RectJob {
  int x, y, w, h;
  int mode;    // 0: already 32x32; 1: integer block means; 2: weighted area tables; 3: bilinear emulation (a side < 32)
  int xt, yt;  // mode 1: block width / height; modes 2, 3: axis table ids
};
struct RectImage {
  unsigned long long off;  // first byte of the image in the batch buffer
  int w, h;
  unsigned row_stride;
  unsigned first, count;   // its jobs: [first, first + count)
};
struct AxisTab {
  int tab_off, first_off;  // mode 2: AreaTab pool offset, `first` (33 ints) offset in the int pool
  int lin_off;             // mode 3: 96 ints (32 x source offset, 32 x c0, 32 x c1) in the int pool
};

template <int K>
__device__ __forceinline__ void blur_rect(const unsigned char* __restrict__ img, int W, int H, size_t stride, int x,
                                          int y, int rw, int rh, unsigned char* __restrict__ scr) {
  constexpr int R = K / 2;
  const int nseg = rw >= kThreads ? 1 : min(rh, kThreads / rw);  // row segments, so that narrow rectangles use all lanes
  const int rps = (rh + nseg - 1) / nseg;
  const int items = rw * nseg;
  for (int it = (int)threadIdx.x; it < items; it += kThreads) {
    const int g = it / rw, j = it - g * rw;
    const int r0 = g * rps, r1 = min(rh, r0 + rps);
    if (r0 >= r1) continue;
    int pc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) pc[t] = reflect101(x + j + t - R, W);
    unsigned ring[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const unsigned char* row = img + (size_t)reflect101(y + r0 - R + t, H) * stride;
      unsigned s = 0;
#pragma unroll
      for (int u = 0; u < K; ++u) s += row[pc[u]];
      ring[t] = s;
    }
    for (int i = r0; i < r1; ++i) {
      unsigned S = 0;
#pragma unroll
      for (int t = 0; t < K; ++t) S += ring[t];
      scr[(size_t)i * rw + j] = (unsigned char)((2u * S + (unsigned)(K * K)) / (2u * (unsigned)(K * K)));
      if (i + 1 < r1) {
        const unsigned char* row = img + (size_t)reflect101(y + i + 1 + R, H) * stride;
        unsigned s = 0;
#pragma unroll
        for (int u = 0; u < K; ++u) s += row[pc[u]];
#pragma unroll
        for (int t = 0; t + 1 < K; ++t) ring[t] = ring[t + 1];
        ring[K - 1] = s;
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_rect_hashes(unsigned char* __restrict__ base,
                                                          const RectImage* __restrict__ images, unsigned n_images,
                                                          const RectJob* __restrict__ jobs,
                                                          const AxisTab* __restrict__ axes,
                                                          const AreaTab* __restrict__ apool,
                                                          const int* __restrict__ ipool,
                                                          unsigned char* __restrict__ scratch, size_t scratch_per_wg,
                                                          const DctTables* __restrict__ tabs, int write_back,
                                                          uint64_t* __restrict__ out,
                                                          unsigned char* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) float sT[288], sY[84];
  __shared__ __attribute__((aligned(16))) unsigned char tile[1024], sZ[64];
  const int tid = threadIdx.x;
  if (tid < 64) sZ[tid] = tabs->zz[tid];
  unsigned char* __restrict__ scr = scratch + (size_t)blockIdx.x * scratch_per_wg;
  for (unsigned im = blockIdx.x; im < n_images; im += gridDim.x) {
    const RectImage I = images[im];
    unsigned char* img = base + I.off;
    for (unsigned r = 0; r < I.count; ++r) {
      const RectJob J = jobs[I.first + r];
      const long long area = (long long)J.w * J.h;
      const int K = area <= 32 * 32 ? 0 : area <= 64 * 64 ? 3 : area <= 128 * 128 ? 5 : 7;
      const unsigned char* src = img + (size_t)J.y * I.row_stride + J.x;
      size_t sp = I.row_stride;
      if (K) {
        if (K == 3) blur_rect<3>(img, I.w, I.h, I.row_stride, J.x, J.y, J.w, J.h, scr);
        else if (K == 5) blur_rect<5>(img, I.w, I.h, I.row_stride, J.x, J.y, J.w, J.h, scr);
        else blur_rect<7>(img, I.w, I.h, I.row_stride, J.x, J.y, J.w, J.h, scr);
        __syncthreads();  // the whole rectangle is blurred (from the old pixels) before any of it is replaced
        if (write_back)
          for (long long i = tid; i < area; i += kThreads) {
            const int yy = (int)(i / J.w), xx = (int)(i - (long long)yy * J.w);
            img[(size_t)(J.y + yy) * I.row_stride + J.x + xx] = scr[i];
          }
        src = scr;
        sp = (size_t)J.w;
      }
      for (int o = tid; o < 1024; o += kThreads) {
        const int dy = o >> 5, dx = o & 31;
        if (J.mode == 0) {
          tile[o] = src[(size_t)dy * sp + dx];
        } else if (J.mode == 1) {  // resizeAreaFast_: block sum, 2x2 -> (s+2)>>2, else rint(s * (1.f/area))
          const int isx = J.xt, isy = J.yt;
          unsigned int s = 0;
          for (int yy = 0; yy < isy; ++yy)
            for (int xx = 0; xx < isx; ++xx) s += src[(size_t)(dy * isy + yy) * sp + (dx * isx + xx)];
          const unsigned int v = (isx == 2 && isy == 2)
                                     ? (s + 2u) >> 2
                                     : (unsigned int)__builtin_rintf((float)s * (1.f / (float)(isx * isy)));
          tile[o] = (unsigned char)(v > 255u ? 255u : v);
        } else if (J.mode == 2) {  // resizeArea_: the float accumulation order is part of the result
          const AxisTab ax = axes[J.xt], ay = axes[J.yt];
          const AreaTab* __restrict__ xtab = apool + ax.tab_off;
          const AreaTab* __restrict__ ytab = apool + ay.tab_off;
          const int* __restrict__ xfirst = ipool + ax.first_off;
          const int* __restrict__ yfirst = ipool + ay.first_off;
          float sum = 0.f;
          for (int j = yfirst[dy]; j < yfirst[dy + 1]; ++j) {
            const unsigned char* S = src + (size_t)ytab[j].si * sp;
            float buf = 0.f;
            for (int k = xfirst[dx]; k < xfirst[dx + 1]; ++k) buf += (float)S[xtab[k].si] * xtab[k].alpha;
            const float t = ytab[j].alpha * buf;
            sum = (j == yfirst[dy]) ? t : sum + t;
          }
          const float rr = __builtin_rintf(sum);
          tile[o] = (unsigned char)(rr < 0.f ? 0.f : rr > 255.f ? 255.f : rr);
        } else {  // 2-tap fixed-point resizer with area-mode coefficients (HResizeLinear / VResizeLinear, 8u)
          const int* __restrict__ xl = ipool + axes[J.xt].lin_off;
          const int* __restrict__ yl = ipool + axes[J.yt].lin_off;
          const int sy0 = min(max(yl[dy], 0), J.h - 1), sy1 = min(max(yl[dy] + 1, 0), J.h - 1);
          const int sx = xl[dx], sx1 = min(sx + 1, J.w - 1);
          const int a0 = xl[32 + dx], a1 = xl[64 + dx], b0 = yl[32 + dy], b1 = yl[64 + dy];
          const unsigned char* S0 = src + (size_t)sy0 * sp;
          const unsigned char* S1 = src + (size_t)sy1 * sp;
          const int D0 = (int)S0[sx] * a0 + (int)S0[sx1] * a1;
          const int D1 = (int)S1[sx] * a0 + (int)S1[sx1] * a1;
          const int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
          tile[o] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
      }
      __syncthreads();
      if (tiles)
        for (int i = tid; i < 1024; i += kThreads) tiles[(size_t)(I.first + r) * 1024 + i] = tile[i];
      hash_from_tile(tile, sZ, sT, sY, out + I.first + r, tabs);
      __syncthreads();  // tile / sT / sY are reused, and the written-back pixels are in place for the next rectangle
    }
  }
}

// ---------------------------------------------------------------------------------------------
// makeKeyPointHashes proper: the rectangle rule of media.cpp:880-901 evaluated on the device from the keypoints
// themselves (no per-rectangle descriptors cross PCIe), squares up to `lds_side` pixels staged in LDS:
//   A  the square plus its blur halo is copied from the image into LDS once (REFLECT_101 at the image's edges
//      resolved while loading); the resize table of this side length is cached in LDS until the side changes
//   B  K x K box blur from LDS to LDS (a lane per column and row segment, sliding K-row sums)
//   C  the blurred square goes back to the image (stores only) while the 32x32 tile is formed from LDS
//   D  hash_from_tile
// One global round trip per rectangle instead of one per dependent table / pixel access; what remains serial is what
// the reference makes serial (a rectangle sees the blurred pixels of the ones before it).  Larger squares take the
// global-memory routine of k_rect_hashes.  Hashes of image i land at out[kp_first[i] + 0, 1, ...]; counts[i] = how many.
struct KpImage {
  unsigned long long off;
  int w, h;
  unsigned row_stride;
  unsigned kp_first, kp_count;
};
struct SizeInfo {  // per side length s (index s): how a s x s square is reduced to 32 x 32
  int mode;        // 0 copy, 1 integer blocks (s / 32), 2 area table, 3 bilinear emulation, -1: no table uploaded
  int tab_off, n;  // mode 2: AreaTab pool offset and entry count
  int aux_off;     // mode 2: first[33] in the int pool; mode 3: x table (96 ints) followed by the y table (96 ints)
};

template <int K>
__device__ __forceinline__ void blur_lds8(const unsigned char* __restrict__ reg, int P, int s,
                                          unsigned char* __restrict__ dst, int Pb) {
  // reg: (s + K - 1) rows, pitch P (a multiple of 8, >= 8 * ceil(s / 8) + 8); LDS column c holds the square's column
  // c - 4, halo included.  Like k_blur_rows: a lane owns 8 adjacent columns, reads a 16-byte window per row, forms the
  // K-tap sums with v_dot4_u32_u8 against byte masks and slides packed-u16 column sums down its row segment.
  // dst: s rows, pitch Pb = 8 * ceil(s / 8).
  constexpr int R = K / 2;
  const int L = (s + 7) >> 3;
  const int nseg = min(s, max(1, kThreads / L));
  const int rps = (s + nseg - 1) / nseg;
  for (int it = (int)threadIdx.x; it < L * nseg; it += kThreads) {
    const int g = it / L, l = it - g * L;
    const int r0 = g * rps, r1 = min(s, r0 + rps);
    if (r0 >= r1) continue;
    unsigned ring[K][4];
    unsigned S[4];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) ring[j][c] = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) S[c] = BlurK<K>::add | (BlurK<K>::add << 16);
    const unsigned char* __restrict__ win = reg + 8 * l;
    const int rend = r1 + 2 * R;  // region rows [r0, rend) feed output rows [r0, r1)
    for (int base = r0; base < rend; base += K) {
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int rr = base + j;
        if (rr < rend) {
          const uint2 a = *reinterpret_cast<const uint2*>(win + rr * P);
          const uint2 b = *reinterpret_cast<const uint2*>(win + rr * P + 8);
          const unsigned W[4] = {a.x, a.y, b.x, b.y};
          unsigned Pk[4];
        hsum_pairs<R>(W, Pk);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            S[c] = (S[c] - ring[j][c]) + Pk[c];
            ring[j][c] = Pk[c];
          }
          if (rr - r0 >= 2 * R) {
            *reinterpret_cast<uint2*>(dst + (rr - 2 * R) * Pb + 8 * l) = blur_quotients<K>(S);
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_kp_hashes(unsigned char* __restrict__ base,
                                                        const KpImage* __restrict__ images, unsigned n_images,
                                                        const float* __restrict__ kp,
                                                        const SizeInfo* __restrict__ sizes,
                                                        const AreaTab* __restrict__ apool,
                                                        const int* __restrict__ ipool, int lds_side, int blur_side,
                                                        int tab_cap,
                                                        unsigned char* __restrict__ scratch, size_t scratch_per_wg,
                                                        const DctTables* __restrict__ tabs,
                                                        uint64_t* __restrict__ out, unsigned* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) float sT[288], sY[84];
  __shared__ __attribute__((aligned(16))) unsigned char tile[1024], sZ[64];
  __shared__ int sFirst[36];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  // dynamic LDS: [table: tab_cap x (int si, float alpha)] [region of the current square: (s + K - 1) rows x
  // (8*ceil(s/8) + 8), s <= lds_side] [its blurred copy when s <= blur_side: s rows x 8*ceil(s/8)]; sized by the
  // host for the larger of region(lds_side) and region + blurred(blur_side).  tab_cap is even: 16-byte alignment holds.
  int* sTabSi = reinterpret_cast<int*>(dyn);
  float* sTabA = reinterpret_cast<float*>(dyn + (size_t)tab_cap * 4);
  unsigned char* sReg = dyn + (size_t)tab_cap * 8;
  const int tid = threadIdx.x;
  if (tid < 64) sZ[tid] = tabs->zz[tid];
  unsigned char* __restrict__ scr = scratch + (size_t)blockIdx.x * scratch_per_wg;
  int cached = -1;  // side length whose table is in LDS (uniform)
  for (unsigned im = blockIdx.x; im < n_images; im += gridDim.x) {
    const KpImage I = images[im];
    unsigned char* img = base + I.off;
    unsigned cnt = 0;
    for (unsigned q = 0; q < I.kp_count; ++q) {
      const float* k3 = kp + 3 * (size_t)(I.kp_first + q);
      const float x0 = k3[0], y0 = k3[1], size = k3[2];
      if (!(size >= 31.f)) continue;
      const float x1 = x0 + size, y1 = y0 + size;
      if (!(x0 > 0 && y0 > 0 && x1 < (float)(I.w - 2) && y1 < (float)(I.h - 2))) continue;
      const int x = (int)__builtin_floorf(x0), y = (int)__builtin_floorf(y0), s = (int)__builtin_ceilf(size);
      const SizeInfo si = sizes[s];
      const int K = s * s <= 32 * 32 ? 0 : s * s <= 64 * 64 ? 3 : s * s <= 128 * 128 ? 5 : 7;
      uint64_t* dst = out + I.kp_first + cnt;
      ++cnt;
      if (s > lds_side || (si.mode == 2 && si.n > tab_cap)) {
        // ---- global-memory routine (large squares) ----
        const unsigned char* src = img + (size_t)y * I.row_stride + x;
        size_t sp = I.row_stride;
        if (K) {
          if (K == 3) blur_rect<3>(img, I.w, I.h, I.row_stride, x, y, s, s, scr);
          else if (K == 5) blur_rect<5>(img, I.w, I.h, I.row_stride, x, y, s, s, scr);
          else blur_rect<7>(img, I.w, I.h, I.row_stride, x, y, s, s, scr);
          __syncthreads();
          for (int i = tid; i < s * s; i += kThreads) {
            const int yy = i / s, xx = i - yy * s;
            img[(size_t)(y + yy) * I.row_stride + x + xx] = scr[i];
          }
          src = scr;
          sp = (size_t)s;
        }
        for (int o = tid; o < 1024; o += kThreads) {
          const int dy = o >> 5, dx = o & 31;
          if (si.mode == 1) {
            const int b = s / 32;
            unsigned int sum = 0;
            for (int yy = 0; yy < b; ++yy)
              for (int xx = 0; xx < b; ++xx) sum += src[(size_t)(dy * b + yy) * sp + (dx * b + xx)];
            const unsigned int v = b == 2 ? (sum + 2u) >> 2
                                          : (unsigned int)__builtin_rintf((float)sum * (1.f / (float)(b * b)));
            tile[o] = (unsigned char)(v > 255u ? 255u : v);
          } else {  // mode 2 (sides above 32 that are not exact multiples)
            const AreaTab* __restrict__ tab = apool + si.tab_off;
            const int* __restrict__ first = ipool + si.aux_off;
            float sum = 0.f;
            for (int j = first[dy]; j < first[dy + 1]; ++j) {
              const unsigned char* S = src + (size_t)tab[j].si * sp;
              float buf = 0.f;
              for (int k = first[dx]; k < first[dx + 1]; ++k) buf += (float)S[tab[k].si] * tab[k].alpha;
              const float t = tab[j].alpha * buf;
              sum = (j == first[dy]) ? t : sum + t;
            }
            const float rr = __builtin_rintf(sum);
            tile[o] = (unsigned char)(rr < 0.f ? 0.f : rr > 255.f ? 255.f : rr);
          }
        }
        __syncthreads();
        hash_from_tile(tile, sZ, sT, sY, dst, tabs);
        __syncthreads();
        continue;
      }
      // ---- A: table (when the side changed) and region -> LDS ----
      const int R = K / 2, P = ((s + 7) & ~7) + 8, Pb = (s + 7) & ~7, rows = s + 2 * R;
      if (cached != s) {
        if (si.mode == 2) {
          const AreaTab* __restrict__ tab = apool + si.tab_off;
          for (int i = tid; i < si.n; i += kThreads) {
            sTabSi[i] = tab[i].si;
            sTabA[i] = tab[i].alpha;
          }
          if (tid < 33) sFirst[tid] = ipool[si.aux_off + tid];
        } else if (si.mode == 3) {
          for (int i = tid; i < 192; i += kThreads) sTabSi[i] = ipool[si.aux_off + i];
        }
        cached = s;
      }
      {
        // LDS column c <-> image column x - 4 + c; dwords [dw0, dw1) cover the columns the blur reads
        const int dw0 = (4 - R) >> 2, dw1 = (4 + s + R + 3) >> 2, ndw = dw1 - dw0;
        for (int i = tid; i < rows * ndw; i += kThreads) {
          const int rr = i / ndw, dwi = dw0 + (i - rr * ndw);
          const int gx = x - 4 + 4 * dwi;
          const unsigned char* row = img + (size_t)reflect101(y - R + rr, I.h) * I.row_stride;
          unsigned v;
          if (gx >= 0 && gx + 3 < I.w) {
            v = *reinterpret_cast<const u32_any_align*>(row + gx);
          } else {
            v = (unsigned)row[reflect101(gx, I.w)] | ((unsigned)row[reflect101(gx + 1, I.w)] << 8) |
                ((unsigned)row[reflect101(gx + 2, I.w)] << 16) | ((unsigned)row[reflect101(gx + 3, I.w)] << 24);
          }
          *reinterpret_cast<unsigned*>(sReg + rr * P + 4 * dwi) = v;
        }
      }
      __syncthreads();
      // ---- B, C: blur, write-back, 32x32 tile.  The blurred square lives in LDS behind the region when it fits
      // (side <= blur_side), else in this workgroup's global scratch: the same code at two call sites so that each
      // keeps its address space.
      auto finish = [&](const unsigned char* __restrict__ src, int sp, const unsigned char* __restrict__ blurred) {
        if (blurred) {
          // ---- C1: back into the image ----
          const int ndw = (s + 3) >> 2;
          for (int i = tid; i < s * ndw; i += kThreads) {
            const int yy = i / ndw, d = i - yy * ndw;
            const unsigned v = *reinterpret_cast<const unsigned*>(blurred + yy * Pb + 4 * d);
            unsigned char* o = img + (size_t)(y + yy) * I.row_stride + x + 4 * d;
            if (4 * d + 3 < s) {
              *reinterpret_cast<u32_any_align*>(o) = v;
            } else {
              for (int b = 0; 4 * d + b < s; ++b) o[b] = (unsigned char)(v >> (8 * b));
            }
          }
        }
        // ---- C2: 32x32 tile ----
  #pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
          const int o = tid + o4 * kThreads;
          const int dy = o >> 5, dx = o & 31;
          if (si.mode == 0) {
            tile[o] = src[dy * sp + dx];
          } else if (si.mode == 1) {
            const int b = s / 32;
            unsigned int sum = 0;
            for (int yy = 0; yy < b; ++yy)
              for (int xx = 0; xx < b; ++xx) sum += src[(dy * b + yy) * sp + (dx * b + xx)];
            const unsigned int v = b == 2 ? (sum + 2u) >> 2
                                          : (unsigned int)__builtin_rintf((float)sum * (1.f / (float)(b * b)));
            tile[o] = (unsigned char)(v > 255u ? 255u : v);
          } else if (si.mode == 2) {
            float sum = 0.f;
            const int j0 = sFirst[dy], j1 = sFirst[dy + 1], k0 = sFirst[dx], k1 = sFirst[dx + 1];
            for (int j = j0; j < j1; ++j) {
              const unsigned char* S = src + sTabSi[j] * sp;
              float buf = 0.f;
              for (int k = k0; k < k1; ++k) buf += (float)S[sTabSi[k]] * sTabA[k];
              const float t = sTabA[j] * buf;
              sum = (j == j0) ? t : sum + t;
            }
            const float rr = __builtin_rintf(sum);
            tile[o] = (unsigned char)(rr < 0.f ? 0.f : rr > 255.f ? 255.f : rr);
          } else {
            const int* xl = sTabSi;
            const int* yl = sTabSi + 96;
            const int sy0 = min(max(yl[dy], 0), s - 1), sy1 = min(max(yl[dy] + 1, 0), s - 1);
            const int sx = xl[dx], sx1 = min(sx + 1, s - 1);
            const int a0 = xl[32 + dx], a1 = xl[64 + dx], b0 = yl[32 + dy], b1 = yl[64 + dy];
            const unsigned char* S0 = src + sy0 * sp;
            const unsigned char* S1 = src + sy1 * sp;
            const int D0 = (int)S0[sx] * a0 + (int)S0[sx1] * a1;
            const int D1 = (int)S1[sx] * a0 + (int)S1[sx1] * a1;
            const int v = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2;
            tile[o] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
          }
        }
      };
      if (!K) {
        finish(sReg + 4, P, nullptr);  // the square itself, pitch P
      } else if (s <= blur_side) {
        unsigned char* __restrict__ sBlur = sReg + (((size_t)rows * P + 15) & ~(size_t)15);
        if (K == 3) blur_lds8<3>(sReg, P, s, sBlur, Pb);
        else if (K == 5) blur_lds8<5>(sReg, P, s, sBlur, Pb);
        else blur_lds8<7>(sReg, P, s, sBlur, Pb);
        __syncthreads();
        finish(sBlur, Pb, sBlur);
      } else {
        if (K == 3) blur_lds8<3>(sReg, P, s, scr, Pb);
        else if (K == 5) blur_lds8<5>(sReg, P, s, scr, Pb);
        else blur_lds8<7>(sReg, P, s, scr, Pb);
        __syncthreads();
        finish(scr, Pb, scr);
      }
      __syncthreads();
      // ---- D ----
      hash_from_tile(tile, sZ, sT, sY, dst, tabs);
      __syncthreads();
    }
    if (tid == 0) counts[im] = cnt;
  }
}

// out_dense[out_first[i] + t] = out_slots[kp_first[i] + t], t < counts[i]
__global__ __launch_bounds__(kThreads) void k_kp_compact(const uint64_t* __restrict__ slots,
                                                         const KpImage* __restrict__ images,
                                                         const unsigned* __restrict__ out_first,
                                                         uint64_t* __restrict__ dense, unsigned n_images) {
  for (unsigned im = blockIdx.x; im < n_images; im += gridDim.x) {
    const unsigned a = out_first[im], m = out_first[im + 1] - a, src = images[im].kp_first;
    for (unsigned t = threadIdx.x; t < m; t += kThreads) dense[a + t] = slots[src + t];
  }
}

}  // namespace

int g_kp_blur_side = 112;  // ... and up to this side their blurred copy stays in LDS as well (larger: global scratch)
void set_kp_blur_side(int v) {
  if (v >= 32 && v <= 200) g_kp_blur_side = v;
}
int g_kp_lds_side = 134;  // keypoint squares up to this side are processed in LDS (k_kp_hashes)
void set_kp_lds_side(int v) {
  if (v >= 32 && v <= 200) g_kp_lds_side = v;
}

namespace {

// cv::resize(INTER_AREA) with an enlarging axis: coefficient tables of the 2-tap resizer (see oracle/cbird_oracle.c
// resize_linear_tab for the prose).  96 ints: source offsets, c0, c1.
void make_linear_tab(int ssize, bool is_x, std::vector<int>* pool) {
  const double inv_scale = (double)32 / ssize;
  const double scale = 1. / inv_scale;
  int ofs[32], c0[32], c1[32];
  for (int d = 0; d < 32; ++d) {
    int sx = (int)std::floor(d * scale);
    float f = (float)((d + 1) - (sx + 1) * inv_scale);
    f = f <= 0 ? 0.f : f - std::floor(f);
    if (is_x && sx + 1 >= ssize) {
      f = 0.f;
      sx = ssize - 1;
    }
    ofs[d] = sx;
    c0[d] = (int)std::min<long>(32767, std::max<long>(-32768, std::lrintf((1.f - f) * 2048.f)));
    c1[d] = (int)std::min<long>(32767, std::max<long>(-32768, std::lrintf(f * 2048.f)));
  }
  pool->insert(pool->end(), ofs, ofs + 32);
  pool->insert(pool->end(), c0, c0 + 32);
  pool->insert(pool->end(), c1, c1 + 32);
}

struct RectTables {  // axis tables of one launch, deduplicated by (size, kind)
  std::vector<AxisTab> axes;
  std::vector<AreaTab> apool;
  std::vector<int> ipool;
  std::map<std::tuple<int, int>, int> ids;  // (size, kind: 0 area, 1 linear x, 2 linear y)
  int get(int size, int kind) {
    auto key = std::make_tuple(size, kind);
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    AxisTab a{0, 0, 0};
    if (kind == 0) {
      std::vector<int> first;
      std::vector<AreaTab> t = make_area_tab(size, 32, &first);
      a.tab_off = (int)apool.size();
      a.first_off = (int)ipool.size();
      apool.insert(apool.end(), t.begin(), t.end());
      ipool.insert(ipool.end(), first.begin(), first.end());
    } else {
      a.lin_off = (int)ipool.size();
      make_linear_tab(size, kind == 1, &ipool);
    }
    axes.push_back(a);
    return ids[key] = (int)axes.size() - 1;
  }
};

}  // namespace

// rects: x, y, w, h per job; images[i].first/count index them.  d_base is written (the blurred rectangles) iff
// write_back.  Synchronises `stream` before returning (the descriptor uploads come from pageable host vectors).
int launch_rect_hashes(uint8_t* d_base, const std::vector<RectImageDesc>& images, const std::vector<int>& rects,
                       int write_back, uint64_t* d_out, hipStream_t stream, uint8_t* d_tiles) {
  const size_t nj = rects.size() / 4;
  if (images.empty() || nj == 0) return CBH_OK;
  const DctTables* tabs = nullptr;
  int rc = get_tables(&tabs);
  if (rc) return rc;
  RectTables rt;
  std::vector<RectJob> jobs(nj);
  size_t max_blur = 16;
  for (size_t i = 0; i < nj; ++i) {
    RectJob& J = jobs[i];
    J.x = rects[4 * i], J.y = rects[4 * i + 1], J.w = rects[4 * i + 2], J.h = rects[4 * i + 3];
    if (J.w <= 0 || J.h <= 0 || J.w > 8192 || J.h > 8192) return CBH_E_INVAL;
    if ((long long)J.w * J.h > 32 * 32) max_blur = std::max(max_blur, (size_t)J.w * (size_t)J.h);
    if (J.w == 32 && J.h == 32) {
      J.mode = 0, J.xt = J.yt = 0;
    } else if (J.w < 32 || J.h < 32) {
      J.mode = 3, J.xt = rt.get(J.w, 1), J.yt = rt.get(J.h, 2);
    } else if (area_fast(J.w, J.h)) {
      J.mode = 1, J.xt = J.w / 32, J.yt = J.h / 32;
    } else {
      J.mode = 2, J.xt = rt.get(J.w, 0), J.yt = rt.get(J.h, 0);
    }
  }
  std::vector<RectImage> imgs(images.size());
  for (size_t i = 0; i < images.size(); ++i) {
    const RectImageDesc& d = images[i];
    if (d.w <= 0 || d.h <= 0 || d.row_stride < (unsigned)d.w || (size_t)d.first + d.count > nj) return CBH_E_INVAL;
    for (unsigned r = d.first; r < d.first + d.count; ++r)
      if (jobs[r].x < 0 || jobs[r].y < 0 || jobs[r].x + jobs[r].w > d.w || jobs[r].y + jobs[r].h > d.h)
        return CBH_E_INVAL;
    imgs[i] = RectImage{d.off, d.w, d.h, d.row_stride, d.first, d.count};
  }
  if (rt.axes.empty()) rt.axes.push_back(AxisTab{0, 0, 0});
  if (rt.apool.empty()) rt.apool.push_back(AreaTab{0, 0, 0.f});
  if (rt.ipool.empty()) rt.ipool.push_back(0);
  max_blur = (max_blur + 255) / 256 * 256;
  const unsigned grid = (unsigned)std::min<size_t>(images.size(), 2048);
  RectImage* d_images = nullptr;
  RectJob* d_jobs = nullptr;
  AxisTab* d_axes = nullptr;
  AreaTab* d_apool = nullptr;
  int* d_ipool = nullptr;
  unsigned char* d_scr = nullptr;
  hipError_t e = hipSuccess;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (e != hipSuccess) return;
    if ((e = cbh::malloc_async(dst, bytes, stream)) != hipSuccess) return;
    e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, stream);
  };
  up((void**)&d_images, imgs.data(), imgs.size() * sizeof(RectImage));
  up((void**)&d_jobs, jobs.data(), jobs.size() * sizeof(RectJob));
  up((void**)&d_axes, rt.axes.data(), rt.axes.size() * sizeof(AxisTab));
  up((void**)&d_apool, rt.apool.data(), rt.apool.size() * sizeof(AreaTab));
  up((void**)&d_ipool, rt.ipool.data(), rt.ipool.size() * sizeof(int));
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_scr, (size_t)grid * max_blur, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_rect_hashes, dim3(grid), dim3(kThreads), 0, stream, d_base, d_images, (unsigned)imgs.size(),
                       d_jobs, d_axes, d_apool, d_ipool, d_scr, max_blur, tabs, write_back, d_out, d_tiles);
    e = hipGetLastError();
  }
  for (void* p : {(void*)d_images, (void*)d_jobs, (void*)d_axes, (void*)d_apool, (void*)d_ipool, (void*)d_scr})
    if (p) (void)cbh::free_async(p, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  CBH_HIP(e);
  return CBH_OK;
}

extern int g_kp_lds_side, g_kp_blur_side;
// Media::makeKeyPointHashes for a batch, keypoints evaluated on the device.  kp / kp_first / descriptors are host
// arrays; d_out receives the hashes densely (image i at out_first[i]); out_first has n + 1 entries.
int launch_keypoint_hashes(uint8_t* d_base, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                           const uint32_t* img_h, const uint32_t* img_row_stride, const float* kp,
                           const uint32_t* kp_first, uint64_t* d_out, uint32_t* out_first, hipStream_t stream) {
  const size_t nkp = kp_first[n];
  out_first[0] = 0;
  if (nkp == 0) {
    for (size_t i = 0; i <= n; ++i) out_first[i] = 0;
    return CBH_OK;
  }
  const DctTables* tabs = nullptr;
  int rc = get_tables(&tabs);
  if (rc) return rc;
  // side lengths present (candidates only) -> tables
  std::vector<unsigned char> present(8194, 0);
  int max_side = 0;
  for (size_t i = 0; i < nkp; ++i) {
    const float size = kp[3 * i + 2];
    if (!(size >= 31.f) || size > 8192.f) continue;
    const int sd = (int)std::ceil(size);
    present[(size_t)sd] = 1;
    max_side = std::max(max_side, sd);
  }
  std::vector<SizeInfo> sizes((size_t)std::max(max_side, 32) + 1, SizeInfo{-1, 0, 0, 0});
  std::vector<AreaTab> apool(1, AreaTab{0, 0, 0.f});
  std::vector<int> ipool(1, 0);
  int tab_cap = 192;  // the bilinear tables (2 x 96 ints) share the table area
  int lds_side = 32;
  const int kLdsSideMax = g_kp_lds_side;  // default 134 = ORB level 8 (31 * 1.2^8 = 133.3): region + blurred square = 39 KB
  for (int sd = 31; sd <= max_side; ++sd) {
    if (!present[(size_t)sd]) continue;
    SizeInfo& si = sizes[(size_t)sd];
    if (sd == 32) {
      si.mode = 0;
    } else if (sd < 32) {
      si.mode = 3;
      si.aux_off = (int)ipool.size();
      make_linear_tab(sd, true, &ipool);
      make_linear_tab(sd, false, &ipool);
    } else if (area_fast(sd, sd)) {
      si.mode = 1;
    } else {
      si.mode = 2;
      std::vector<int> first;
      std::vector<AreaTab> t = make_area_tab(sd, 32, &first);
      si.tab_off = (int)apool.size();
      si.n = (int)t.size();
      si.aux_off = (int)ipool.size();
      apool.insert(apool.end(), t.begin(), t.end());
      ipool.insert(ipool.end(), first.begin(), first.end());
      if (sd <= kLdsSideMax) tab_cap = std::max(tab_cap, si.n);
    }
    if (sd <= kLdsSideMax) lds_side = std::max(lds_side, sd);
  }
  std::vector<KpImage> imgs(n);
  for (size_t i = 0; i < n; ++i)
    imgs[i] = KpImage{img_off[i], (int)img_w[i], (int)img_h[i], img_row_stride[i], kp_first[i],
                      kp_first[i + 1] - kp_first[i]};
  // squares up to blur_side keep their blurred copy in LDS too; larger ones (still <= lds_side) put it in global
  // scratch, which keeps the LDS footprint at ~27 KB = 6 workgroups per CU instead of 3
  const int blur_side = std::min(lds_side, g_kp_blur_side);
  auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
  auto region_bytes = [&](size_t sd) { return ((sd + 6) * (up8(sd) + 8) + 15) & ~(size_t)15; };
  tab_cap = (tab_cap + 1) & ~1;
  const size_t smem = (size_t)tab_cap * 8 +
                      std::max(region_bytes((size_t)lds_side), region_bytes((size_t)blur_side) + (size_t)blur_side * up8((size_t)blur_side));
  const size_t scratch_per_wg = max_side > blur_side ? (((size_t)max_side * up8((size_t)max_side) + 255) / 256 * 256) : 256;
  const unsigned grid = (unsigned)std::min<size_t>(n, 2048);
  KpImage* d_images = nullptr;
  float* d_kp = nullptr;
  SizeInfo* d_sizes = nullptr;
  AreaTab* d_apool = nullptr;
  int* d_ipool = nullptr;
  unsigned char* d_scr = nullptr;
  uint64_t* d_slots = nullptr;
  unsigned *d_counts = nullptr, *d_first = nullptr;
  std::vector<unsigned> counts(n);
  hipError_t e = hipSuccess;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (e != hipSuccess) return;
    if ((e = cbh::malloc_async(dst, bytes, stream)) != hipSuccess) return;
    e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, stream);
  };
  up((void**)&d_images, imgs.data(), imgs.size() * sizeof(KpImage));
  up((void**)&d_kp, kp, nkp * 3 * sizeof(float));
  up((void**)&d_sizes, sizes.data(), sizes.size() * sizeof(SizeInfo));
  up((void**)&d_apool, apool.data(), apool.size() * sizeof(AreaTab));
  up((void**)&d_ipool, ipool.data(), ipool.size() * sizeof(int));
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_scr, (size_t)grid * scratch_per_wg, stream);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_slots, nkp * sizeof(uint64_t), stream);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_counts, n * sizeof(unsigned), stream);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_first, (n + 1) * sizeof(unsigned), stream);
  if (e == hipSuccess && smem > 48 * 1024)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_kp_hashes), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_kp_hashes, dim3(grid), dim3(kThreads), smem, stream, d_base, d_images, (unsigned)n, d_kp,
                       d_sizes, d_apool, d_ipool, lds_side, blur_side, tab_cap, d_scr, scratch_per_wg, tabs, d_slots,
                       d_counts);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d_counts, n * sizeof(unsigned), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) {
    for (size_t i = 0; i < n; ++i) out_first[i + 1] = out_first[i] + counts[i];
    e = hipMemcpyAsync(d_first, out_first, (n + 1) * sizeof(unsigned), hipMemcpyHostToDevice, stream);
  }
  if (e == hipSuccess && out_first[n]) {
    hipLaunchKernelGGL(k_kp_compact, dim3(grid), dim3(kThreads), 0, stream, d_slots, d_images, d_first, d_out,
                       (unsigned)n);
    e = hipGetLastError();
  }
  for (void* p : {(void*)d_images, (void*)d_kp, (void*)d_sizes, (void*)d_apool, (void*)d_ipool, (void*)d_scr,
                  (void*)d_slots, (void*)d_counts, (void*)d_first})
    if (p) (void)cbh::free_async(p, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  CBH_HIP(e);
  return CBH_OK;
}

}  // namespace cbh
