// mfma_f4.hip -- micro-benchmark / layout probe (development aid, not product):
// 64-bit Hamming distance as a +-1 dot product on the gfx950 block-scaled FP4 MFMA.
//   dot(a,b) over 64 signs = 64 - 2*hamm64(a,b); FP4 E2M1 +1.0 = 0x2, -1.0 = 0xA.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f4 tools/ubench/mfma_f4.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// one 32-bit word -> 32 FP4 nibbles (4 dwords); bit k -> nibble k
__host__ __device__ inline void expand32(uint32_t w, uint32_t out[4]) {
  for (int d = 0; d < 4; ++d) {
    uint32_t v = 0;
    for (int k = 0; k < 8; ++k) {
      uint32_t bit = (w >> (d * 8 + k)) & 1u;
      v |= (bit ? 0x2u : 0xAu) << (4 * k);
    }
    out[d] = v;
  }
}

// expanded layout: row r -> 8 dwords: [lo word: 4 dwords][hi word: 4 dwords]
__global__ void k_expand(const uint64_t* h, uint32_t* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t o[4];
  expand32((uint32_t)h[i], o);
  for (int d = 0; d < 4; ++d) out[i * 8 + d] = o[d];
  expand32((uint32_t)(h[i] >> 32), o);
  for (int d = 0; d < 4; ++d) out[i * 8 + 4 + d] = o[d];
}

__device__ __forceinline__ v16f mfma_f4(v8i a, v8i b) {
  v16f c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// probe: 32 haystack rows x 32 needles, dump all 1024 dots
__global__ void k_probe(const uint32_t* hx, const uint32_t* qx, float* out) {
  const int lane = threadIdx.x;
  const int r = lane & 31, half = lane >> 5;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int d = 0; d < 4; ++d) {
    a[d] = (int)hx[r * 8 + half * 4 + d];
    b[d] = (int)qx[r * 8 + half * 4 + d];
  }
  v16f c = mfma_f4(a, b);
  for (int g = 0; g < 16; ++g) {
    int row = (g & 3) + 8 * (g >> 2) + 4 * half;  // haystack index (A row)
    int col = lane & 31;                          // needle index (B column)
    out[row * 32 + col] = c[g];
  }
}


// packed probe variants: V=0 c0 regs + both MFMAs; V=1 only first MFMA on c0; V=2 zero C + second
// (scaled) MFMA only; V=3 like 0 but scale 2^15 applied through scale_a on the needles as A operand
template <int V>
__global__ void k_probe2(const uint32_t* hx, const uint32_t* qx, float* out) {
  const int lane = threadIdx.x;
  const int r = lane & 31, half = lane >> 5;
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0}, b2 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int d = 0; d < 4; ++d) {
    a[d] = (int)hx[r * 8 + half * 4 + d];
    b[d] = (int)qx[r * 8 + half * 4 + d];
    b2[d] = (int)qx[(32 + r) * 8 + half * 4 + d];
  }
  v16f c0, z;
  for (int g = 0; g < 16; ++g) { c0[g] = 8388608.0f + 16448.0f + 2097152.0f; z[g] = 0.f; }
  asm volatile("" : "+v"(c0));
  v16f c;
  if (V == 0) {
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b2, c, 4, 4, 0, 0x7f7f7f7f, 0, (int)0x8e8e8e8e);
  } else if (V == 1) {
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  } else if (V == 2) {
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b2, z, 4, 4, 0, 0x7f7f7f7f, 0, (int)0x8e8e8e8e);
  } else {
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b2, c, 4, 4, 0, (int)0x8e8e8e8e, 0, 0x7f7f7f7f);
  }
  for (int g = 0; g < 16; ++g) {
    int row = (g & 3) + 8 * (g >> 2) + 4 * half;
    out[row * 32 + (lane & 31)] = c[g];
  }
}

// throughput: each wave keeps HT haystack tiles in VGPRs and streams needle tiles
template <int HT>
__global__ __launch_bounds__(256) void k_rate(const uint4* hx, const uint4* qx, uint32_t n_qtiles,
                                              float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  v8i a[HT];
  const uint32_t tile0 = (blockIdx.x * 4 + wave) * HT;
  for (int t = 0; t < HT; ++t) {
    uint4 v = hx[((tile0 + t) * 32 + r) * 2 + half];
    a[t] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
  }
  float m0 = -100.f, m1 = -100.f;
  uint4 nb = qx[(0 * 32 + r) * 2 + half];
  for (uint32_t qt = 0; qt < n_qtiles; ++qt) {
    v8i b = v8i{(int)nb.x, (int)nb.y, (int)nb.z, (int)nb.w, 0, 0, 0, 0};
    uint32_t nx = qt + 1 < n_qtiles ? qt + 1 : qt;
    nb = qx[(nx * 32 + r) * 2 + half];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      v16f c = mfma_f4(a[t], b);
#pragma unroll
      for (int g = 0; g < 16; g += 4) {
        m0 = __builtin_fmaxf(__builtin_fmaxf(m0, c[g]), c[g + 1]);
        m1 = __builtin_fmaxf(__builtin_fmaxf(m1, c[g + 2]), c[g + 3]);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fmaxf(m0, m1);
}


// overlap probe: NV v_max3_f32 per MFMA (0..8) on persistent accumulators
template <int HT, int NV>
__global__ __launch_bounds__(256) void k_mix(const uint4* hx, const uint4* qx, uint32_t n_qtiles,
                                             float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  v8i a[HT];
  v16f c[HT];
  const uint32_t tile0 = (blockIdx.x * 4 + wave) * HT;
  for (int t = 0; t < HT; ++t) {
    uint4 v = hx[((tile0 + t) * 32 + r) * 2 + half];
    a[t] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    for (int g = 0; g < 16; ++g) c[t][g] = 0.f;
  }
  float m0 = -100.f, m1 = -100.f;
  uint4 nb = qx[(0 * 32 + r) * 2 + half];
  for (uint32_t qt = 0; qt < n_qtiles; ++qt) {
    v8i b = v8i{(int)nb.x, (int)nb.y, (int)nb.z, (int)nb.w, 0, 0, 0, 0};
    uint32_t nx = qt + 1 < n_qtiles ? qt + 1 : qt;
    nb = qx[(nx * 32 + r) * 2 + half];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], b, c[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
      for (int g = 0; g < NV; ++g) {
        if (g & 1) m1 = __builtin_fmaxf(__builtin_fmaxf(m1, c[(t + 1) % HT][2 * g]), c[(t + 1) % HT][2 * g + 1]);
        else m0 = __builtin_fmaxf(__builtin_fmaxf(m0, c[(t + 1) % HT][2 * g]), c[(t + 1) % HT][2 * g + 1]);
      }
    }
  }
  float sum = 0;
  for (int t = 0; t < HT; ++t) for (int g = 0; g < 16; ++g) sum += c[t][g];
  out[blockIdx.x * 256 + threadIdx.x] = fmaxf(m0, m1) + sum;
}


// pure MFMA issue rate: no memory traffic in the loop
template <int HT>
__global__ __launch_bounds__(256) void k_pure(const uint4* hx, uint32_t iters, float* out) {
  const int lane = threadIdx.x & 63;
  v8i a[HT];
  v16f c[HT];
  for (int t = 0; t < HT; ++t) {
    uint4 v = hx[lane + 64 * t];
    a[t] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    for (int g = 0; g < 16; ++g) c[t][g] = 0.f;
  }
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < HT; ++t)
      c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], a[(t + 1) % HT], c[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  float sum = 0;
  for (int t = 0; t < HT; ++t) for (int g = 0; g < 16; ++g) sum += c[t][g];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}


// MFMA + NV independent VALU ops per MFMA (do the pipes overlap?)
template <int HT, int NV>
__global__ __launch_bounds__(256) void k_pure_mix(const uint4* hx, uint32_t iters, float* out) {
  const int lane = threadIdx.x & 63;
  v8i a[HT];
  v16f c[HT];
  float x[8];
  for (int t = 0; t < HT; ++t) {
    uint4 v = hx[lane + 64 * t];
    a[t] = v8i{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    for (int g = 0; g < 16; ++g) c[t][g] = 0.f;
  }
  for (int g = 0; g < 8; ++g) x[g] = (float)(lane + g);
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], a[(t + 1) % HT], c[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
      for (int g = 0; g < NV; ++g)
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[g % 8]) : "v"(x[(g + 1) % 8]), "v"(x[(g + 2) % 8]));
    }
  }
  float sum = 0;
  for (int t = 0; t < HT; ++t) for (int g = 0; g < 16; ++g) sum += c[t][g];
  for (int g = 0; g < 8; ++g) sum += x[g];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  // ---- layout probe ----
  std::vector<uint64_t> h(32), q(32);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < 32; ++i) { h[i] = rnd(); q[i] = (i % 3 == 0) ? h[(i * 7) % 32] ^ (1ull << i) : rnd(); }
  uint64_t *dh, *dq; uint32_t *dhx, *dqx; float* dout;
  CK(hipMalloc(&dh, 256)); CK(hipMalloc(&dq, 256)); CK(hipMalloc(&dhx, 1024)); CK(hipMalloc(&dqx, 1024));
  CK(hipMalloc(&dout, 4096));
  CK(hipMemcpy(dh, h.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, q.data(), 256, hipMemcpyHostToDevice));
  k_expand<<<1, 32>>>(dh, dhx, 32); k_expand<<<1, 32>>>(dq, dqx, 32);
  k_probe<<<1, 64>>>(dhx, dqx, dout);
  std::vector<float> o(1024);
  CK(hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int d = __builtin_popcountll(h[i] ^ q[j]);
      if (o[i * 32 + j] != (float)(64 - 2 * d)) { if (bad < 5) printf("mismatch h%d q%d: got %g want %d\n", i, j, o[i * 32 + j], 64 - 2 * d); ++bad; }
    }
  printf("probe: %d mismatches of 1024\n", bad);
  {
    std::vector<uint64_t> q2(64);
    for (int i = 0; i < 64; ++i) q2[i] = (i % 5 == 0) ? h[(i * 3) % 32] ^ (1ull << (i % 7)) : rnd();
    uint64_t* dq2; uint32_t* dq2x; uint32_t* dout2;
    CK(hipMalloc(&dq2, 512)); CK(hipMalloc(&dq2x, 2048)); CK(hipMalloc(&dout2, 4096));
    CK(hipMemcpy(dq2, q2.data(), 512, hipMemcpyHostToDevice));
    k_expand<<<1, 64>>>(dq2, dq2x, 64);
    for (int V = 0; V < 4; ++V) {
      if (V == 0) k_probe2<0><<<1, 64>>>(dhx, dq2x, (float*)dout2);
      if (V == 1) k_probe2<1><<<1, 64>>>(dhx, dq2x, (float*)dout2);
      if (V == 2) k_probe2<2><<<1, 64>>>(dhx, dq2x, (float*)dout2);
      if (V == 3) k_probe2<3><<<1, 64>>>(dhx, dq2x, (float*)dout2);
      std::vector<float> ob(1024);
      CK(hipMemcpy(ob.data(), dout2, 4096, hipMemcpyDeviceToHost));
      int bad2 = 0;
      const double C0 = 8388608.0 + 16448.0 + 2097152.0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          int dA = __builtin_popcountll(h[i] ^ q2[j]), dB = __builtin_popcountll(h[i] ^ q2[32 + j]);
          double want = 0;
          if (V == 0 || V == 3) want = C0 + (64 - 2 * dA) + 32768.0 * (64 - 2 * dB);
          if (V == 1) want = C0 + (64 - 2 * dA);
          if (V == 2) want = 32768.0 * (64 - 2 * dB);
          if ((double)ob[i * 32 + j] != want) { if (bad2 < 4) printf("V%d mismatch h%d q%d: got %.1f want %.1f (delta %.1f)\n", V, i, j, ob[i * 32 + j], want, ob[i * 32 + j] - want); ++bad2; }
        }
      printf("packed probe V%d: %d mismatches of 1024\n", V, bad2);
    }
  }

  // ---- rate ----
  const uint32_t n = 1u << 20, nq = 1u << 20;
  uint64_t* big; uint32_t* bigx;
  CK(hipMalloc(&big, n * 8)); CK(hipMalloc(&bigx, (size_t)n * 32));
  std::vector<uint64_t> hb(n);
  for (auto& v : hb) v = rnd();
  CK(hipMemcpy(big, hb.data(), n * 8, hipMemcpyHostToDevice));
  k_expand<<<n / 256, 256>>>(big, bigx, n);
  float* dres; CK(hipMalloc(&dres, (size_t)n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(HT)                                                                              \
  {                                                                                          \
    uint32_t blocks = n / (32 * HT * 4);                                                     \
    for (int rep = 0; rep < 2; ++rep) {                                                      \
      CK(hipEventRecord(e0));                                                                \
      k_rate<HT><<<blocks, 256>>>((const uint4*)bigx, (const uint4*)bigx, nq / 32, dres);    \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                   \
    }                                                                                        \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                          \
    printf("HT=%d blocks=%u: %.2f ms for %.3g pairs -> %.3g pairs/s\n", HT, blocks, ms,      \
           (double)n * nq, (double)n * nq / (ms * 1e-3));                                    \
  }
  RUN(1) RUN(2) RUN(4) RUN(8)
#define MIX(HT, NV)                                                                          \
  {                                                                                          \
    uint32_t blocks = n / (32 * HT * 4);                                                     \
    for (int rep = 0; rep < 2; ++rep) {                                                      \
      CK(hipEventRecord(e0));                                                                \
      k_mix<HT, NV><<<blocks, 256>>>((const uint4*)bigx, (const uint4*)bigx, nq / 32, dres); \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                   \
    }                                                                                        \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                          \
    printf("mix HT=%d NV=%d: %.2f ms -> %.1f cycles/MFMA/SIMD\n", HT, NV, ms,                \
           ms * 1e-3 * 2.4e9 * 1024 / ((double)n * nq / 1024));                              \
  }
  for (int wg = 1; wg <= 2; ++wg) {
    const uint32_t iters = 200000, blocks = 256 * wg;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      k_pure<4><<<blocks, 256>>>((const uint4*)bigx, iters, dres);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double mf = (double)blocks * 4 * iters * 4;
    printf("pure: %u WGs/CU-ish, %.2f ms, %.3g MFMA/s = %.0f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", wg, ms,
           mf / (ms * 1e-3), mf / (ms * 1e-3) * 131072 / 1e12, ms * 1e-3 * 2.4e9 * 1024 / mf);
  }
#define PMIX(NV) {                                                                          \
    const uint32_t iters = 20000, blocks = 512;                                               \
    for (int rep = 0; rep < 2; ++rep) {                                                       \
      CK(hipEventRecord(e0));                                                                 \
      k_pure_mix<4, NV><<<blocks, 256>>>((const uint4*)bigx, iters, dres);                    \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                    \
    }                                                                                         \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                           \
    double mf = (double)blocks * 4 * iters * 4;                                               \
    printf("pure_mix NV=%d: %.2f ms, %.1f cycles/MFMA/SIMD @2.4GHz\n", NV, ms, ms * 1e-3 * 2.4e9 * 1024 / mf); }
  PMIX(0) PMIX(2) PMIX(4) PMIX(6) PMIX(8) PMIX(12) PMIX(16)
  MIX(4, 0) MIX(4, 2) MIX(4, 4) MIX(4, 6) MIX(4, 8) MIX(2, 0) MIX(2, 8)
  return 0;
}
