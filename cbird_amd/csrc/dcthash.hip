// dcthash.hip -- K1+K2: batched 64-bit DCT perceptual hash for gfx950 (CDNA4).
//
// Replaces dctHash64(const cv::Mat&, bool) (src/cvutil.cpp:435-545) for 8UC1 input:
//   stage 1  box blur, kernel 0/3/5/7 chosen by input area (:446-455), cv::blur semantics:
//            centre anchor, BORDER_REFLECT_101, u8 result = nearest(sum / k^2)        (:463)
//   stage 2  cv::resize(->32x32, INTER_AREA), integer-ratio path: block sum, 2x2 -> (s+2)>>2,
//            otherwise rint_even(float(s) * (1.f/area))                                  (:471)
//   stage 3  f32 32x32 DCT-II, only the 9x9 low-frequency block is needed                (:475-482)
//   stage 4  zig-zag order, keep positions 6..69                                         (:491-513)
//   stage 5  threshold = float(sum in double of the 64 coefficients) / 64               (:528-529)
//   stage 6  bit i (1..63) = coef[i] > threshold; hash 0 -> 1                            (:537-542)
//
// Bit-exact contract with oracle/cbird_oracle.c: stages 1-2 are integer (plus one exactly
// specified f32 multiply + round-to-nearest-even), stage 3 is the canonical separable form
//   T[r][k] = sum_j fmaf(X[r][j], C[k][j], .)  (j ascending, start 0.0f)
//   Y[u][k] = sum_r fmaf(C[u][r], T[r][k], .)  (r ascending, start 0.0f)
// with C[k][j] = f32(sqrt((k?2:1)/32) * cos(pi*(2j+1)*k/64)) supplied by the host, stage 5 is a
// sequential f64 sum in zig-zag order.  Every output element is produced by one thread in that
// fixed order, so GPU == CPU restatement bit for bit.
//
// k_dcthash_generic: one 256-thread workgroup per image, any w,h in {32} or multiples of 32 up
// to 1024; the image is consumed as 32 horizontal bands (one per output row) staged in LDS.
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "cbh_internal.h"

namespace cbh {
namespace {

constexpr int kThreads = 256;

struct DctTables {
  float C[9 * 32];
  unsigned char zz[64];  // zig-zag positions 6..69 -> index into the 9x9 block (row*9+col)
};

__device__ __forceinline__ int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    p = p < 0 ? -p : 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// stages 3-6 from a 32x32 u8 tile in LDS; all 256 threads of the workgroup participate.
__device__ __forceinline__ void hash_from_tile(const unsigned char* __restrict__ tile /*LDS*/,
                                               const float* __restrict__ sC /*LDS 288*/,
                                               const unsigned char* __restrict__ sZ /*LDS 64*/,
                                               float* __restrict__ sT /*LDS 288*/,
                                               float* __restrict__ sY /*LDS 81*/,
                                               float* __restrict__ sThr /*LDS 1*/,
                                               uint64_t* __restrict__ out) {
  const int tid = threadIdx.x;
  // row pass: 288 outputs (r,k)
  for (int o = tid; o < 288; o += kThreads) {
    const int r = o / 9, k = o - r * 9;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc = __builtin_fmaf((float)tile[r * 32 + j], sC[k * 32 + j], acc);
    sT[r * 9 + k] = acc;
  }
  __syncthreads();
  if (tid < 81) {
    const int u = tid / 9, k = tid - u * 9;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) acc = __builtin_fmaf(sC[u * 32 + r], sT[r * 9 + k], acc);
    sY[tid] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += (double)sY[sZ[i]];
    *sThr = (float)sum / 64;
  }
  __syncthreads();
  if (tid < 64) {
    const float c = sY[sZ[tid]];
    const unsigned long long b = __ballot(tid >= 1 && c > *sThr);
    if (tid == 0) *out = b ? b : 1ull;
  }
}

template <int K>
__global__ __launch_bounds__(kThreads) void k_dcthash_generic(
    const unsigned char* __restrict__ imgs, int w, int h, size_t row_stride, size_t img_stride,
    const DctTables* __restrict__ tabs, uint64_t* __restrict__ out,
    unsigned char* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = K / 2;
  const int sx = w / 32, sy = h / 32;
  const int band = sy + 2 * R;
  // LDS carve-up
  float* sC = reinterpret_cast<float*>(smem);                 // 288 f32
  float* sT = sC + 288;                                        // 288 f32
  float* sY = sT + 288;                                        // 81 f32 (+3 pad)
  float* sThr = sY + 84;                                       // 1 f32 (+3 pad)
  unsigned int* colsum = reinterpret_cast<unsigned int*>(sThr + 4);  // w u32
  unsigned char* tile = reinterpret_cast<unsigned char*>(colsum + w);  // 1024 u8
  unsigned char* sZ = tile + 1024;                                      // 64 u8
  unsigned short* hs = reinterpret_cast<unsigned short*>(sZ + 64);     // band*w u16
  unsigned char* raw = reinterpret_cast<unsigned char*>(hs + (size_t)band * w);  // band*w u8

  const int tid = threadIdx.x;
  const unsigned char* img = imgs + (size_t)blockIdx.x * img_stride;
  for (int i = tid; i < 288; i += kThreads) sC[i] = tabs->C[i];
  if (tid < 64) sZ[tid] = tabs->zz[tid];

  if constexpr (K == 0) {
    // area <= 32*32 with w,h multiples of 32 means exactly 32x32: no blur, resize is a no-op
    for (int i = tid; i < 1024; i += kThreads) tile[i] = img[(size_t)(i >> 5) * row_stride + (i & 31)];
    __syncthreads();
  } else {
    const bool two = (sx == 2 && sy == 2);
    const float scale = 1.f / (float)(sx * sy);
    for (int oy = 0; oy < 32; ++oy) {
      __syncthreads();
      // P0: raw band rows -> LDS
      for (int i = tid; i < band * w; i += kThreads) {
        const int b = i / w, x = i - b * w;
        const int ry = reflect101(oy * sy - R + b, h);
        raw[i] = img[(size_t)ry * row_stride + x];
      }
      __syncthreads();
      // P1: horizontal K-sums
      for (int i = tid; i < band * w; i += kThreads) {
        const int b = i / w, x = i - b * w;
        unsigned int s = 0;
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) s += raw[b * w + reflect101(x + dx, w)];
        hs[i] = (unsigned short)s;
      }
      __syncthreads();
      // P2: vertical K-sums, divide, accumulate the sy blurred rows of this band per column
      for (int x = tid; x < w; x += kThreads) {
        unsigned int s = 0;
#pragma unroll
        for (int t = 0; t < K; ++t) s += hs[t * w + x];
        unsigned int acc = 0;
        for (int j = 0; j < sy; ++j) {
          acc += (2u * s + (unsigned)(K * K)) / (2u * (unsigned)(K * K));
          if (j + 1 < sy) s += (unsigned)hs[(j + K) * w + x] - (unsigned)hs[j * w + x];
        }
        colsum[x] = acc;
      }
      __syncthreads();
      // P3: sx adjacent columns -> one output pixel, INTER_AREA rounding
      if (tid < 32) {
        unsigned int s = 0;
        for (int dx = 0; dx < sx; ++dx) s += colsum[tid * sx + dx];
        unsigned int v = two ? (s + 2u) >> 2 : (unsigned int)__builtin_rintf((float)s * scale);
        tile[oy * 32 + tid] = (unsigned char)(v > 255u ? 255u : v);
      }
    }
    __syncthreads();
  }
  if (tiles)
    for (int i = tid; i < 1024; i += kThreads) tiles[(size_t)blockIdx.x * 1024 + i] = tile[i];
  hash_from_tile(tile, sC, sZ, sT, sY, sThr, out + blockIdx.x);
}


// ---------------------------------------------------------------------------------------------
// k_dcthash_256: the BASELINE configuration (256x256 tiles: 7x7 blur, 8x8 area mean).
//
// Work split: a 32-lane half-wave owns one image; lane l owns pixels [8l, 8l+8) of every row
// (one 8-pixel-wide output column), so a 256-thread workgroup hashes 8 images and every row
// load is a fully coalesced 256-B segment per half-wave (global_load_dwordx2 per lane).
// Per row and lane (44 VALU ops for 8 pixels, all integer, exact):
//   neighbours  2x ds_bpermute (LDS crossbar, no VALU) + 2x v_perm_b32 with a per-lane selector
//               that turns the two border lanes' halo into the REFLECT_101 mirror of their own
//               pixels;
//   horizontal  7-tap sums with v_dot4_u32_u8 against 0/1 byte masks (14 ops for 8 outputs);
//   vertical    7-row sliding sum on u16 pairs packed in u32 (no field ever borrows/overflows:
//               7*7*255 + 24 < 2^16): S += H(new) - H(7 rows ago), ring of 7 rows in VGPRs;
//   divide      nearest(S/49) = ((S+24) * 342393) >> 24 exactly for S <= 12495 (the +24 lives in
//               the running sum), accumulated over the 8 rows of an output cell; 8 columns of the
//               cell are this lane's 8 pixels, so the lane ends up with the 8x8 block sum and
//               rounds it half-to-even (/64) into the 32x32 tile in LDS.
// The 262 "virtual" rows -3..258 are mapped through REFLECT_101 so top/bottom borders need no
// special code; row loads run 7 rows ahead of use.
// Stages 3-6 then run per half-wave on its own tile (row pass: lane = tile row, basis values as
// wave-uniform SGPR operands; column pass and threshold through LDS) in the oracle's fma order.
__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) {
  return __builtin_amdgcn_udot4(a, b, c, false);
}

// acc + (p >> 24) in one VALU op (SDWA byte select); hipcc otherwise emits shift + add
__device__ __forceinline__ unsigned add_byte3(unsigned acc, unsigned p) {
  unsigned r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
      : "=v"(r)
      : "v"(p), "v"(acc));
  return r;
}

template <bool DUMP>
__global__ __launch_bounds__(kThreads) void k_dcthash_256(
    const unsigned char* __restrict__ imgs, unsigned n, unsigned row_stride, unsigned img_stride,
    const DctTables* __restrict__ tabs, uint64_t* __restrict__ out,
    unsigned char* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char sTile[8][1024];
  __shared__ float sT[8][288];
  __shared__ float sY[8][84];
  __shared__ float sC[9 * 33];
  __shared__ float sThr[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l32 = lane & 31;
  const int slot = tid >> 5;  // image within the workgroup, 0..7
  for (int i = tid; i < 288; i += kThreads) sC[(i >> 5) * 33 + (i & 31)] = tabs->C[i];

  const unsigned first = blockIdx.x * 8u;
  unsigned img = first + (unsigned)slot;
  const bool valid = img < n;
  if (!valid) img = n - 1;
  const unsigned char* __restrict__ base = imgs + (size_t)first * img_stride;  // wave-uniform
  const unsigned voff = (img - first) * img_stride + (unsigned)l32 * 8u;       // per lane

  // halo exchange: left neighbour's pixels 4..7, right neighbour's pixels 0..3
  const int addrL = ((l32 == 0 ? lane : lane - 1)) << 2;
  const int addrR = ((l32 == 31 ? lane : lane + 1)) << 2;
  // v_perm_b32(S0,S1,sel): selector 4..7 -> S0 byte 0..3, 0..3 -> S1 byte 0..3
  const unsigned selL = l32 == 0 ? 0x01020300u : 0x07060504u;   // lane 0: (x,px3,px2,px1)
  const unsigned selR = l32 == 31 ? 0x00000102u : 0x07060504u;  // lane 31: (px254,px253,px252,x)

  uint2 raw[7];
  unsigned ring[7][4];
  unsigned S[4];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
#pragma unroll
    for (int c = 0; c < 4; ++c) ring[j][c] = 0u;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) S[c] = 24u | (24u << 16);
  unsigned acc = 0;

  // virtual row s-3 -> REFLECT_101 source row; steps past the image (s > 261) re-read row 252,
  // their sums never reach the tile
  auto row_off = [&](int s) -> unsigned {
    int v = s - 3;
    v = v < 0 ? -v : v;
    v = v > 255 ? 510 - v : v;
    v = v < 0 ? 0 : v;
    return (unsigned)v * row_stride + voff;  // 32-bit lane offset from the uniform base
  };
  auto halo = [&](const uint2 d, unsigned& DL, unsigned& DR) {
    const unsigned bl = (unsigned)__builtin_amdgcn_ds_bpermute(addrL, (int)d.y);
    const unsigned br = (unsigned)__builtin_amdgcn_ds_bpermute(addrR, (int)d.x);
    DL = __builtin_amdgcn_perm(bl, d.x, selL);  // bytes 1..3 = px -3,-2,-1
    DR = __builtin_amdgcn_perm(br, d.y, selR);  // bytes 0..2 = px 8,9,10
  };
#pragma unroll
  for (int j = 0; j < 7; ++j) raw[j] = *reinterpret_cast<const uint2*>(base + row_off(j));
  unsigned DLn, DRn;  // halo of the row consumed by the next step (exchange runs one row ahead)
  halo(raw[0], DLn, DRn);

  for (int s0 = 0; s0 < 266; s0 += 7) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int s = s0 + j;
      const unsigned D0 = raw[j].x, D1 = raw[j].y;
      const unsigned DL = DLn, DR = DRn;
      raw[j] = *reinterpret_cast<const uint2*>(base + row_off(s + 7));
      halo(raw[(j + 1) % 7], DLn, DRn);
      const unsigned T0 = udot4(D0, 0x01010101u, 0u);
      const unsigned T1 = udot4(D1, 0x01010101u, 0u);
      const unsigned H0 = udot4(DL, 0x01010100u, T0);
      const unsigned H1 = udot4(DL, 0x01010000u, udot4(D1, 0x00000001u, T0));
      const unsigned H2 = udot4(DL, 0x01000000u, udot4(D1, 0x00000101u, T0));
      const unsigned H3 = udot4(D1, 0x00010101u, T0);
      const unsigned H4 = udot4(D0, 0x01010100u, T1);
      const unsigned H5 = udot4(D0, 0x01010000u, udot4(DR, 0x00000001u, T1));
      const unsigned H6 = udot4(D0, 0x01000000u, udot4(DR, 0x00000101u, T1));
      const unsigned H7 = udot4(DR, 0x00010101u, T1);
      const unsigned P[4] = {H0 | (H1 << 16), H2 | (H3 << 16), H4 | (H5 << 16), H6 | (H7 << 16)};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        S[c] = (S[c] - ring[j][c]) + P[c];
        ring[j][c] = P[c];
      }
      // output row y = s - 6 (garbage for s < 6: acc is reset before the first real row)
      if (s == 6) acc = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc = add_byte3(acc, (S[c] & 0xffffu) * 342393u);  // operands < 2^24 -> v_mul_u32_u24
        acc = add_byte3(acc, (S[c] >> 16) * 342393u);
      }
      const int y = s - 6;
      if (y >= 0 && (y & 7) == 7) {
        const unsigned t = (acc + 31u + ((acc >> 6) & 1u)) >> 6;  // /64, half to even
        if (y < 256) sTile[slot][(y >> 3) * 32 + l32] = (unsigned char)t;
        acc = 0;
      }
    }
  }
  __syncthreads();
  if (DUMP) {  // stage-level parity aid: the 32x32 tile after blur + area resize
    if (valid)
      for (int i = l32; i < 256; i += 32)
        reinterpret_cast<unsigned*>(tiles + (size_t)img * 1024)[i] =
            reinterpret_cast<const unsigned*>(sTile[slot])[i];
  }

  // ---- stage 3, row pass: lane = tile row r; T[r][k] = sum_j fmaf(X[r][j], C[k][j], .)
  {
    float x[32];
    const uint4* trow = reinterpret_cast<const uint4*>(&sTile[slot][l32 * 32]);
    const uint4 a = trow[0], b = trow[1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[4 * i + 0] = (float)(w[i] & 0xffu);
      x[4 * i + 1] = (float)((w[i] >> 8) & 0xffu);
      x[4 * i + 2] = (float)((w[i] >> 16) & 0xffu);
      x[4 * i + 3] = (float)(w[i] >> 24);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) t = __builtin_fmaf(x[j], tabs->C[k * 32 + j], t);
      sT[slot][l32 * 9 + k] = t;
    }
  }
  __syncthreads();
  // ---- column pass: Y[u][k] = sum_r fmaf(C[u][r], T[r][k], .), 81 outputs over 32 lanes
#pragma unroll
  for (int rep = 0; rep < 3; ++rep) {
    const int o = l32 + 32 * rep;
    if (o < 81) {
      const int u = o / 9, k = o - u * 9;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) t = __builtin_fmaf(sC[u * 33 + r], sT[slot][r * 9 + k], t);
      sY[slot][o] = t;
    }
  }
  __syncthreads();
  if (l32 == 0) {
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += (double)sY[slot][tabs->zz[i]];
    sThr[slot] = (float)sum / 64;
  }
  __syncthreads();
  {
    const float thr = sThr[slot];
    const float c0 = sY[slot][tabs->zz[l32]];
    const float c1 = sY[slot][tabs->zz[l32 + 32]];
    const unsigned long long b0 = __ballot(c0 > thr);
    const unsigned long long b1 = __ballot(c1 > thr);
    const int sh = (lane >> 5) * 32;
    unsigned long long hv = ((b0 >> sh) & 0xffffffffull) | (((b1 >> sh) & 0xffffffffull) << 32);
    hv &= ~1ull;  // bit 0 is never encoded (cvutil.cpp:537)
    if (hv == 0) hv = 1;
    if (l32 == 0 && valid) out[img] = hv;
  }
}

// ---------------------------------------------------------------------------------------------
// Any other size (w,h >= 32, not both multiples of 32): cv::resize's general INTER_AREA path
// (resizeArea_) with fractional cell weights.  Two launches: k_blur_u8 writes the blurred u8 image to a
// scratch buffer (bands of rows staged in LDS, separable box sums), k_area_hash resamples it with the
// reference's float accumulation order -- per source row buf += S[sx]*alpha over the x table, per output
// row sum = beta*buf then += beta*buf over the y table, round-half-even -- and hashes the tile.
struct AreaTab {
  int si, di;
  float alpha;
};

template <int K>
__global__ __launch_bounds__(kThreads) void k_blur_u8(const unsigned char* __restrict__ imgs, int w, int h,
                                                      size_t row_stride, size_t img_stride, int band_rows,
                                                      unsigned char* __restrict__ blur /* n*w*h */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = K / 2;
  const int rows = band_rows + 2 * R;
  unsigned short* hs = reinterpret_cast<unsigned short*>(smem);                  // rows*w u16
  unsigned char* raw = reinterpret_cast<unsigned char*>(hs + (size_t)rows * w);  // rows*w u8
  const int tid = threadIdx.x;
  const unsigned char* img = imgs + (size_t)blockIdx.x * img_stride;
  unsigned char* dst = blur + (size_t)blockIdx.x * (size_t)w * h;
  const int y0 = blockIdx.y * band_rows;
  const int nrows = min(band_rows, h - y0);
  if (K == 0) {
    for (int i = tid; i < nrows * w; i += kThreads) {
      const int b = i / w, x = i - b * w;
      dst[(size_t)(y0 + b) * w + x] = img[(size_t)(y0 + b) * row_stride + x];
    }
    return;
  }
  for (int i = tid; i < rows * w; i += kThreads) {
    const int b = i / w, x = i - b * w;
    raw[i] = img[(size_t)reflect101(y0 - R + b, h) * row_stride + x];
  }
  __syncthreads();
  for (int i = tid; i < rows * w; i += kThreads) {
    const int b = i / w, x = i - b * w;
    unsigned int s = 0;
#pragma unroll
    for (int dx = -R; dx <= R; ++dx) s += raw[b * w + reflect101(x + dx, w)];
    hs[i] = (unsigned short)s;
  }
  __syncthreads();
  for (int x = tid; x < w; x += kThreads) {
    unsigned int s = 0;
#pragma unroll
    for (int t = 0; t < (K ? K : 1); ++t) s += hs[t * w + x];
    for (int j = 0; j < nrows; ++j) {
      dst[(size_t)(y0 + j) * w + x] = (unsigned char)((2u * s + (unsigned)(K * K)) / (2u * (unsigned)(K * K) + (K == 0)));
      if (j + 1 < nrows) s += (unsigned)hs[(j + K) * w + x] - (unsigned)hs[j * w + x];
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_area_hash(const unsigned char* __restrict__ blur, int w, int h,
                                                        const AreaTab* __restrict__ xtab, int xn,
                                                        const AreaTab* __restrict__ ytab, int yn,
                                                        const int* __restrict__ xfirst /*33*/,
                                                        const int* __restrict__ yfirst /*33*/,
                                                        int isx, int isy /* integer ratios or 0 */,
                                                        const DctTables* __restrict__ tabs,
                                                        uint64_t* __restrict__ out,
                                                        unsigned char* __restrict__ tiles) {
  __shared__ float sC[288], sT[288], sY[84], sThr[4];
  __shared__ unsigned char tile[1024], sZ[64];
  const int tid = threadIdx.x;
  const unsigned char* src = blur + (size_t)blockIdx.x * (size_t)w * h;
  for (int i = tid; i < 288; i += kThreads) sC[i] = tabs->C[i];
  if (tid < 64) sZ[tid] = tabs->zz[tid];
  for (int o = tid; o < 1024; o += kThreads) {
    const int dy = o >> 5, dx = o & 31;
    if (isx) {  // both ratios integer: resizeAreaFast_ (block sum, 2x2 -> (s+2)>>2, else rint(s * (1.f/area)))
      unsigned int s = 0;
      for (int yy = 0; yy < isy; ++yy)
        for (int xx = 0; xx < isx; ++xx) s += src[(size_t)(dy * isy + yy) * w + (dx * isx + xx)];
      const unsigned int v = (isx == 2 && isy == 2)
                                 ? (s + 2u) >> 2
                                 : (unsigned int)__builtin_rintf((float)s * (1.f / (float)(isx * isy)));
      tile[o] = (unsigned char)(v > 255u ? 255u : v);
      continue;
    }
    float sum = 0.f;
    for (int j = yfirst[dy]; j < yfirst[dy + 1]; ++j) {
      const unsigned char* S = src + (size_t)ytab[j].si * w;
      float buf = 0.f;
      for (int k = xfirst[dx]; k < xfirst[dx + 1]; ++k) buf += (float)S[xtab[k].si] * xtab[k].alpha;
      const float t = ytab[j].alpha * buf;
      sum = (j == yfirst[dy]) ? t : sum + t;
    }
    const float r = __builtin_rintf(sum);
    tile[o] = (unsigned char)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
  }
  __syncthreads();
  if (tiles)
    for (int i = tid; i < 1024; i += kThreads) tiles[(size_t)blockIdx.x * 1024 + i] = tile[i];
  hash_from_tile(tile, sC, sZ, sT, sY, sThr, out + blockIdx.x);
}

size_t generic_smem_bytes(int w, int h, int K) {
  const int band = h / 32 + 2 * (K / 2);
  return (288 + 288 + 84 + 4) * 4 + (size_t)w * 4 + 1024 + 64 + (size_t)band * w * 3;
}

struct TableCache {
  std::mutex mu;
  DctTables* d[16] = {};
} g_tabs;

}  // namespace

namespace {

// computeResizeAreaTab (OpenCV 2.4 imgwarp.cpp, as recalled): see oracle/cbird_oracle.c for the prose
std::vector<AreaTab> make_area_tab(int ssize, int dsize, std::vector<int>* first) {
  std::vector<AreaTab> tab;
  const double scale = (double)ssize / dsize;
  first->assign((size_t)dsize + 1, 0);
  for (int dx = 0; dx < dsize; dx++) {
    (*first)[(size_t)dx] = (int)tab.size();
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cellWidth = std::min(scale, ssize - fsx1);
    int sx1 = (int)__builtin_ceil(fsx1), sx2 = (int)__builtin_floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back(AreaTab{sx1 - 1, dx, (float)((sx1 - fsx1) / cellWidth)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back(AreaTab{sx, dx, (float)(1.0 / cellWidth)});
    if (fsx2 - sx2 > 1e-3)
      tab.push_back(AreaTab{sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
  }
  (*first)[(size_t)dsize] = (int)tab.size();
  return tab;
}

struct AreaTabsDev {
  AreaTab *x = nullptr, *y = nullptr;
  int *xfirst = nullptr, *yfirst = nullptr;
  int xn = 0, yn = 0;
};
std::mutex g_area_mu;
std::map<std::tuple<int, int, int>, AreaTabsDev> g_area;  // (device, w, h)

int get_area_tabs(int w, int h, AreaTabsDev* out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_area_mu);
  auto key = std::make_tuple(dev, w, h);
  auto it = g_area.find(key);
  if (it == g_area.end()) {
    std::vector<int> xf, yf;
    std::vector<AreaTab> xt = make_area_tab(w, 32, &xf), yt = make_area_tab(h, 32, &yf);
    AreaTabsDev d;
    d.xn = (int)xt.size();
    d.yn = (int)yt.size();
    CBH_HIP(hipMalloc(&d.x, xt.size() * sizeof(AreaTab)));
    CBH_HIP(hipMalloc(&d.y, yt.size() * sizeof(AreaTab)));
    CBH_HIP(hipMalloc(&d.xfirst, 33 * sizeof(int)));
    CBH_HIP(hipMalloc(&d.yfirst, 33 * sizeof(int)));
    CBH_HIP(hipMemcpy(d.x, xt.data(), xt.size() * sizeof(AreaTab), hipMemcpyHostToDevice));
    CBH_HIP(hipMemcpy(d.y, yt.data(), yt.size() * sizeof(AreaTab), hipMemcpyHostToDevice));
    CBH_HIP(hipMemcpy(d.xfirst, xf.data(), 33 * sizeof(int), hipMemcpyHostToDevice));
    CBH_HIP(hipMemcpy(d.yfirst, yf.data(), 33 * sizeof(int), hipMemcpyHostToDevice));
    it = g_area.emplace(key, d).first;
  }
  *out = it->second;
  return CBH_OK;
}

}  // namespace

// Host-side table construction (same closed forms as the oracle, computed independently here).
static void make_tables(DctTables* t) {
  for (int k = 0; k < 9; ++k)
    for (int j = 0; j < 32; ++j) {
      const double a = __builtin_sqrt((k ? 2.0 : 1.0) / 32.0);
      t->C[k * 32 + j] = (float)(a * __builtin_cos(3.14159265358979323846 * (2 * j + 1) * k / 64.0));
    }
  // 9x9 zig-zag, first step downwards (equals the table at cvutil.cpp:491-495); keep 6..69
  int zz[81], n = 0;
  for (int s = 0; s <= 16; ++s) {
    if (s & 1) {
      for (int r = (s < 8 ? s : 8); r >= 0 && s - r <= 8; --r) zz[n++] = r * 9 + (s - r);
    } else {
      for (int r = (s > 8 ? s - 8 : 0); r <= 8 && r <= s; ++r) zz[n++] = r * 9 + (s - r);
    }
  }
  for (int i = 0; i < 64; ++i) t->zz[i] = (unsigned char)zz[6 + i];
}

static int get_tables(const DctTables** out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return CBH_E_INVAL;
  std::lock_guard<std::mutex> lk(g_tabs.mu);
  if (!g_tabs.d[dev]) {
    DctTables host;
    make_tables(&host);
    DctTables* d = nullptr;
    CBH_HIP(hipMalloc(&d, sizeof(DctTables)));
    CBH_HIP(hipMemcpy(d, &host, sizeof(DctTables), hipMemcpyHostToDevice));
    g_tabs.d[dev] = d;
  }
  *out = g_tabs.d[dev];
  return CBH_OK;
}

int launch_dcthash(const uint8_t* d_imgs, size_t n, int w, int h, size_t row_stride,
                   size_t img_stride, uint64_t* d_out, hipStream_t stream, uint8_t* d_tiles) {
  if (n == 0) return CBH_OK;
  if (w <= 0 || h <= 0 || row_stride < (size_t)w) return CBH_E_INVAL;
  if (w < 32 || h < 32 || w > 8192 || h > 8192) return CBH_E_UNSUPPORTED;  // < 32: bilinear upscale path
  if (n > 0x7fffffffull) return CBH_E_INVAL;
  const DctTables* tabs = nullptr;
  int rc = get_tables(&tabs);
  if (rc) return rc;
  if (w % 32 || h % 32 || w > 1024 || h > 1024) {
    // general INTER_AREA path: blur to scratch, then weighted resample + hash
    const long long area_ = (long long)w * h;
    const int K_ = area_ <= 32 * 32 ? 0 : area_ <= 64 * 64 ? 3 : area_ <= 128 * 128 ? 5 : 7;
    AreaTabsDev at;
    if ((rc = get_area_tabs(w, h, &at))) return rc;
    int band = (int)std::min<long long>(h, std::max<long long>(1, (96 * 1024) / (3LL * w) - 2 * (K_ / 2)));
    const size_t smem = (size_t)(band + 2 * (K_ / 2)) * (size_t)w * 3;
    const size_t per_chunk = std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)w * h));
    unsigned char* d_blur = nullptr;
    CBH_HIP(hipMallocAsync((void**)&d_blur, std::min(per_chunk, n) * (size_t)w * h, stream));
    for (size_t i0 = 0; i0 < n; i0 += per_chunk) {
      const size_t m = std::min(per_chunk, n - i0);
      dim3 g1((unsigned)m, (unsigned)((h + band - 1) / band)), block(kThreads);
      const unsigned char* src = d_imgs + i0 * img_stride;
#define CBH_BLUR(KK)                                                                                    \
  do {                                                                                                  \
    if (smem > 64 * 1024)                                                                               \
      CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_u8<KK>),                         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));              \
    hipLaunchKernelGGL(k_blur_u8<KK>, g1, block, smem, stream, src, w, h, row_stride, img_stride, band, \
                       d_blur);                                                                         \
  } while (0)
      switch (K_) {
        case 0: CBH_BLUR(0); break;
        case 3: CBH_BLUR(3); break;
        case 5: CBH_BLUR(5); break;
        default: CBH_BLUR(7); break;
      }
#undef CBH_BLUR
      hipLaunchKernelGGL(k_area_hash, dim3((unsigned)m), block, 0, stream, d_blur, w, h, at.x, at.xn, at.y,
                         at.yn, at.xfirst, at.yfirst, (w % 32 || h % 32) ? 0 : w / 32, (w % 32 || h % 32) ? 0 : h / 32,
                         tabs, d_out + i0, d_tiles ? d_tiles + i0 * 1024 : nullptr);
    }
    CBH_HIP(hipGetLastError());
    CBH_HIP(hipFreeAsync(d_blur, stream));
    return CBH_OK;
  }
  if (w == 256 && h == 256 && ((uintptr_t)d_imgs % 8) == 0 && row_stride % 8 == 0 &&
      img_stride % 8 == 0 && row_stride * 256 < (1u << 24) && img_stride < (1u << 28)) {
    dim3 grid((unsigned)((n + 7) / 8)), block(kThreads);
    if (d_tiles)
      hipLaunchKernelGGL(k_dcthash_256<true>, grid, block, 0, stream, d_imgs, (unsigned)n,
                         (unsigned)row_stride, (unsigned)img_stride, tabs, d_out, d_tiles);
    else
      hipLaunchKernelGGL(k_dcthash_256<false>, grid, block, 0, stream, d_imgs, (unsigned)n,
                         (unsigned)row_stride, (unsigned)img_stride, tabs, d_out, d_tiles);
    CBH_HIP(hipGetLastError());
    return CBH_OK;
  }
  const long long area = (long long)w * h;
  const int K = area <= 32 * 32 ? 0 : area <= 64 * 64 ? 3 : area <= 128 * 128 ? 5 : 7;
  const size_t smem = generic_smem_bytes(w, h, K);
  if (smem > 160 * 1024) return CBH_E_UNSUPPORTED;
  dim3 grid((unsigned)n), block(kThreads);
#define CBH_LAUNCH_GENERIC(KK)                                                              \
  do {                                                                                      \
    if (smem > 64 * 1024)                                                                   \
      CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dcthash_generic<KK>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));  \
    hipLaunchKernelGGL(k_dcthash_generic<KK>, grid, block, smem, stream, d_imgs, w, h,      \
                       row_stride, img_stride, tabs, d_out, d_tiles);                       \
  } while (0)
  switch (K) {
    case 0: CBH_LAUNCH_GENERIC(0); break;
    case 3: CBH_LAUNCH_GENERIC(3); break;
    case 5: CBH_LAUNCH_GENERIC(5); break;
    default: CBH_LAUNCH_GENERIC(7); break;
  }
#undef CBH_LAUNCH_GENERIC
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace cbh
