// mfma_cadence.hip -- micro-benchmark (development aid, not product): what the matrix pipe delivers for
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands when NOTHING else is issued -- W waves per SIMD, each looping over
// independent accumulators (chains of CH dependent MFMAs, like the scan kernels' two or three per accumulator).
// Prints PFLOP/s (131072 flop per instruction) per configuration: the attainable matrix-core rate on this box, to hold
// the scan kernels' 7.3-7.4 PFLOP/s (85 % pipe-busy by the counters) against.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cadence tools/ubench/mfma_cadence.hip && /tmp/mfma_cadence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ACC, int CH, int MINB>
__global__ __launch_bounds__(256, MINB) void k_mfma(int iters, float* out) {
  v8i a = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
  v8i b = {0x2222aaaa, 0x2a2a2a2a, (int)0xaaaa2222, 0x22aa22aa, 0, 0, 0, 0};
  a[0] ^= (int)threadIdx.x << 3 & 0x88888888;
  v16f c[ACC];
#pragma unroll
  for (int t = 0; t < ACC; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) c[t][g] = (float)t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int t = 0; t < ACC; ++t)
        c[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  float s = 0;
#pragma unroll
  for (int t = 0; t < ACC; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) s += c[t][g];
  if (s == 12345.678f) out[0] = s;  // keep the results alive
}

template <int ACC, int CH, int MINB>
void run(const char* name, int wgs_per_cu) {
  float* d = nullptr;
  CK(hipMalloc(&d, 64));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int iters = 20000 / CH;
  const dim3 grid((unsigned)(cus * wgs_per_cu));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mfma<ACC, CH, MINB>), grid, dim3(256), 0, 0, 200, d);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mfma<ACC, CH, MINB>), grid, dim3(256), 0, 0, iters, d);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double mfmas = (double)cus * wgs_per_cu * 4.0 * iters * CH * ACC;  // 4 waves per workgroup
  const double pf = mfmas * 131072.0 / (best * 1e-3) / 1e15;
  // cycles per MFMA per SIMD at the nominal 2.4 GHz (the clock under load is lower: read PFLOP/s)
  const double cyc = best * 1e-3 * 2.4e9 / (mfmas / (cus * 4.0));
  printf("%-44s waves/SIMD %d  %8.3f ms  %6.2f PFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, wgs_per_cu, best,
         pf, cyc);
  CK(hipFree(d));
}

int main() {
  run<2, 3, 1>("2 accumulators x chains of 3 (the FULL3 shape)", 1);
  run<2, 3, 2>("2 accumulators x chains of 3 (the FULL3 shape)", 2);
  run<2, 3, 3>("2 accumulators x chains of 3 (the FULL3 shape)", 3);
  run<2, 2, 4>("2 accumulators x chains of 2 (the PRE shape)", 4);
  run<4, 1, 2>("4 independent accumulators, no chain", 2);
  run<8, 1, 2>("8 independent accumulators, no chain", 2);
  run<8, 1, 1>("8 independent accumulators, no chain", 1);
  run<2, 6, 2>("2 accumulators x chains of 6 (the 256-bit prefilter)", 2);
  return 0;
}
