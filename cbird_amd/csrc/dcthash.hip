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
#include <mutex>

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
    const DctTables* __restrict__ tabs, uint64_t* __restrict__ out) {
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
                   size_t img_stride, uint64_t* d_out, hipStream_t stream) {
  if (n == 0) return CBH_OK;
  if (w <= 0 || h <= 0 || row_stride < (size_t)w) return CBH_E_INVAL;
  if (w % 32 || h % 32 || w > 1024 || h > 1024) return CBH_E_UNSUPPORTED;
  if (n > 0x7fffffffull) return CBH_E_INVAL;
  const DctTables* tabs = nullptr;
  int rc = get_tables(&tabs);
  if (rc) return rc;
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
                       row_stride, img_stride, tabs, d_out);                                \
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
