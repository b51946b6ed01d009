// Is v_mfma_scale_f32_32x32x64_f8f6f4 exact when a K block carries the scale 2^-1 (products of +-0.5) while the
// accumulator sits at 2^23 (ulp 1)?  Needed by the four-field low-word prefilter: two chained MFMAs put four 6-bit
// fields  32 + b - dist_lo  at bits 0, 6, 12, 18 of the mantissa (block scales 2^-1, 2^5, 2^11, 2^17 on +-1 needle
// signs; C0 = 2^23 + (16 + b)(1 + 2^6 + 2^12 + 2^18)); a hit in the top field carries into the exponent (bit 23 of the
// pattern).  Every block sum of 32 products +-0.5 is an integer, so the result is exact IF the hardware adds the
// products of a block before rounding against the accumulator.  This program checks that claim on random and on
// near-duplicate inputs: it prints the number of mismatching fields.
//   hipcc --offload-arch=gfx950 -O2 -o bin/mfma_half_exact mfma_half_exact.hip && bin/mfma_half_exact
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint4 fp4_expand32(uint32_t w) {
  uint32_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t x = (w >> (8 * d)) & 0xffu;
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    o[d] = 0xaaaaaaaau ^ (x << 3);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ v8i op(uint4 e) { return v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0}; }
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(64) void k_check(uint32_t seed, uint32_t b, int near, unsigned long long* bad,
                                               unsigned long long* hits, unsigned long long* flips) {
  __shared__ uint32_t s_hay[32], s_q[128];
  const uint32_t lane = threadIdx.x, r = lane & 31u, half = lane >> 5;
  const uint32_t base = mix(seed ^ (blockIdx.x * 0x9e3779b9u));
  if (lane < 32) s_hay[lane] = mix(base + lane);
  __syncthreads();
  for (uint32_t i = lane; i < 128; i += 64) {
    uint32_t v = mix(base ^ (0x1234567u + i * 977u));
    if (near) {  // a haystack row with a few flipped bits: low-word distances 0..5
      v = s_hay[(i * 7u) & 31u];
      const uint32_t nf = mix(v + i) % 6u;
      for (uint32_t k = 0; k < nf; ++k) v ^= 1u << (mix(v + k * 31u + i) & 31u);
    }
    s_q[i] = v;
  }
  __syncthreads();
  const v8i a = op(fp4_expand32(s_hay[r]));  // low word in both K blocks
  v16f c0;
  const float C0 = 8388608.0f + (float)((16u + b) * (1u + 64u + 4096u + 262144u));
#pragma unroll
  for (int g = 0; g < 16; ++g) c0[g] = C0;
  // pair 0: tiles 0 (K block 0, lanes 0-31) and 1 (K block 1, lanes 32-63); pair 1: tiles 2 and 3
  const v8i b0 = op(fp4_expand32(s_q[lane])), b1 = op(fp4_expand32(s_q[64 + lane]));
  const int s0 = half ? 0x84848484 : 0x7e7e7e7e;  // 2^5 : 2^-1
  const int s1 = half ? (int)0x90909090 : (int)0x8a8a8a8a;  // 2^17 : 2^11
  v16f c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, c0, 4, 4, 0, 0x7f7f7f7f, 0, s0);
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, c, 4, 4, 0, 0x7f7f7f7f, 0, s1);
  unsigned long long nbad = 0, nhit = 0, nflip = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const float cf = c[g];
    const uint32_t bits = __builtin_bit_cast(uint32_t, cf);
    const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * half;
    uint32_t want[4];
    for (int f = 0; f < 4; ++f) want[f] = 32u + b - __popc(s_hay[row] ^ s_q[f * 32 + r]);
    const bool flip = want[3] >= 32u;
    nflip += flip;
    const uint32_t expo = bits >> 23;
    if (expo != (flip ? 151u : 150u)) ++nbad;
    if (!flip) {
      for (int f = 0; f < 4; ++f) {
        const uint32_t got = (bits >> (6 * f)) & 63u;
        if (f < 3) {
          if (got != want[f]) ++nbad;
          nhit += want[f] >= 32u;
        } else if (((bits >> 18) & 31u) != want[3]) {
          ++nbad;
        }
      }
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(hits, nhit);
  atomicAdd(flips, nflip);
}

int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 24);
  for (int near = 0; near < 2; ++near)
    for (uint32_t b = 0; b < 4; ++b) {
      hipMemset(d, 0, 24);
      for (uint32_t it = 0; it < 8; ++it)
        hipLaunchKernelGGL(k_check, dim3(65536), dim3(64), 0, 0, 1000u * it + b * 77u + near, b, near, d, d + 1, d + 2);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("near=%d b=%u: %llu results checked, mismatches %llu, lower-field hits %llu, top-field carries %llu\n", near, b,
             8ull * 65536ull * 1024ull, h[0], h[1], h[2]);
    }
  return 0;
}
