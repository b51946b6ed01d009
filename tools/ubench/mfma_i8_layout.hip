// Operand layout of v_mfma_i32_16x16x64_i8 on gfx950, found by experiment: random i8 A (16 x 64) and B (64 x 16) are
// placed in the lanes' registers under two candidate maps and the result is compared with the plain matrix product.
//   contiguous : lane l, byte j (of its 16)  <->  row/col l & 15, k = 16 * (l >> 4) + j
//   two halves : ...                          <->  row/col l & 15, k = 32 * (j >> 3) + 8 * (l >> 4) + (j & 7)
// C/D: lane l, register r <-> D[4 * (l >> 4) + r][l & 15]  (the 16x16 map of every dtype, cdna_hip_programming.md)
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/mfma_i8_layout mfma_i8_layout.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const v4i* a, const v4i* b, v4i* d) {
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  d[threadIdx.x] = acc;
}

static int kmap(int variant, int l, int j) {
  return variant == 0 ? 16 * (l >> 4) + j : 32 * (j >> 3) + 8 * (l >> 4) + (j & 7);
}

int main() {
  std::mt19937 rng(5);
  std::vector<int8_t> A(16 * 64), B(64 * 16);
  for (auto& x : A) x = (int8_t)(rng() % 255 - 127);
  for (auto& x : B) x = (int8_t)(rng() % 5 - 2);
  std::vector<int> want(256, 0);
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n)
      for (int kk = 0; kk < 64; ++kk) want[m * 16 + n] += (int)A[m * 64 + kk] * (int)B[kk * 16 + n];
  v4i *da, *db, *dd;
  (void)hipMalloc(&da, 64 * 16), (void)hipMalloc(&db, 64 * 16), (void)hipMalloc(&dd, 64 * 16);
  for (int va = 0; va < 2; ++va)
    for (int vb = 0; vb < 2; ++vb) {
      std::vector<int8_t> ra(1024), rb(1024);
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 16; ++j) {
          ra[l * 16 + j] = A[(l & 15) * 64 + kmap(va, l, j)];
          rb[l * 16 + j] = B[kmap(vb, l, j) * 16 + (l & 15)];
        }
      (void)hipMemcpy(da, ra.data(), 1024, hipMemcpyHostToDevice);
      (void)hipMemcpy(db, rb.data(), 1024, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
      std::vector<int> got(256);
      (void)hipMemcpy(got.data(), dd, 1024, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad += got[l * 4 + r] != want[(4 * (l >> 4) + r) * 16 + (l & 15)];
      printf("{\"A_map\": \"%s\", \"B_map\": \"%s\", \"mismatches\": %d}\n", va ? "two halves" : "contiguous",
             vb ? "two halves" : "contiguous", bad);
    }
  return 0;
}
