// fp4_sign.h -- shared by the matrix-core Hamming kernels (hamm64_mfma.hip, hamm256_mfma.hip).
//
// Hamming distance as a dot product: with s(x)_k = +1 if bit k of x is set, else -1,
//   dot(s(a), s(b)) over K bits = K - 2 * popcount(a ^ b).
// +-1.0 are exact in FP4 (E2M1: +1.0 = 0x2, -1.0 = 0xA), products and partial sums are small
// integers, so v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands and unit block scales returns
// exact distances of 32 x 32 pairs per 64 bits of K.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cbh {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kScaleOne = 0x7f7f7f7f;  // E8M0 127 = 2^0 in every byte

// 32 bits -> 32 FP4 sign nibbles: bit k -> nibble k = 0x2 (+1.0) if set, 0xA (-1.0) if clear
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

// FP4 operand of the f8f6f4 MFMA: the first 4 of the 8 operand dwords are used
__device__ __forceinline__ v8i fp4_operand(uint4 e) {
  return v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
}

}  // namespace cbh
