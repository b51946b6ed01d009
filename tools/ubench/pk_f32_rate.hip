// Packed-f32 / min issue rates on gfx950 with INDEPENDENT accumulators (the round-1 probe k_pkaddf32 chained two
// registers, i.e. it measured latency).  Decides how the colour-distance kernel is written.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_f32_rate pk_f32_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// 8 independent 64-bit accumulators %0..%7, %8/%9 = 64-bit VGPR sources, %10 = SGPR pair
#define KERNEL64(name, body)                                                                        \
  __global__ __launch_bounds__(256) void name(float* out, int iters, f2 s) {                       \
    f2 a0 = {threadIdx.x * 1.0f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,  \
       a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b0 = {0.999f, 1.001f}, b1 = {1e-3f, -1e-3f};    \
    for (int i = 0; i < iters; ++i) {                                                               \
      asm volatile(REP16(body)                                                                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b0), "v"(b1), "s"(s));                                                     \
    }                                                                                               \
    f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                   \
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;                                                \
  }

// 8 independent 32-bit accumulators
#define KERNEL32(name, body)                                                                        \
  __global__ __launch_bounds__(256) void name(float* out, int iters, f2 s2) {                      \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,        \
          a6 = a0 + 6, a7 = a0 + 7, b0 = 0.999f, b1 = 1e-3f, s = s2.x;                              \
    for (int i = 0; i < iters; ++i) {                                                               \
      asm volatile(REP16(body)                                                                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                   : "v"(b0), "v"(b1), "s"(s));                                                     \
    }                                                                                               \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                    \
  }

#define L8(pre, post)                                                                                       \
  pre "%0" post "\n" pre "%1" post "\n" pre "%2" post "\n" pre "%3" post "\n" pre "%4" post "\n" pre "%5" post \
      "\n" pre "%6" post "\n" pre "%7" post "\n"

KERNEL64(k_pk_fma_vvv, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                       " v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                       " v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
KERNEL64(k_pk_mul_vv, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n"
                      " v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n"
                      " v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
KERNEL64(k_pk_add_vv, "v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n"
                      " v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n"
                      " v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n")
KERNEL64(k_pk_add_sv, "v_pk_add_f32 %0, %0, %10\n v_pk_add_f32 %1, %1, %10\n v_pk_add_f32 %2, %2, %10\n"
                      " v_pk_add_f32 %3, %3, %10\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %10\n"
                      " v_pk_add_f32 %6, %6, %10\n v_pk_add_f32 %7, %7, %10\n")
KERNEL64(k_pk_fma_svv, "v_pk_fma_f32 %0, %0, %10, %9\n v_pk_fma_f32 %1, %1, %10, %9\n v_pk_fma_f32 %2, %2, %10, %9\n"
                       " v_pk_fma_f32 %3, %3, %10, %9\n v_pk_fma_f32 %4, %4, %10, %9\n v_pk_fma_f32 %5, %5, %10, %9\n"
                       " v_pk_fma_f32 %6, %6, %10, %9\n v_pk_fma_f32 %7, %7, %10, %9\n")
KERNEL32(k_sub_vv, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                   " v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
KERNEL32(k_sub_sv, "v_sub_f32 %0, %10, %0\n v_sub_f32 %1, %10, %1\n v_sub_f32 %2, %10, %2\n v_sub_f32 %3, %10, %3\n"
                   " v_sub_f32 %4, %10, %4\n v_sub_f32 %5, %10, %5\n v_sub_f32 %6, %10, %6\n v_sub_f32 %7, %10, %7\n")
KERNEL32(k_mul_vv, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   " v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
KERNEL32(k_fma_vvv, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                    " v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                    " v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                    " v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n")
KERNEL32(k_min3_u32, "v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n"
                     " v_min3_u32 %3, %3, %8, %9\n v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n"
                     " v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9\n")
KERNEL32(k_min_f32, "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                    " v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n")
// the shape of the colour inner loop on 32-bit ops: 3 sub, 3 mul, 2 add, 1 min per pair (operands all VGPR)
KERNEL32(k_mix_color32, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_mul_f32 %0, %0, %0\n"
                        " v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_add_f32 %3, %0, %1\n v_add_f32 %3, %3, %2\n"
                        " v_min_u32 %4, %4, %3\n")
// one (needle colour, two haystack colours) cell of k_color_dist3: 6 sub, 6 mul, 4 add, 1 min3 + half of two more
KERNEL32(k_mix_cell, "v_sub_f32 %0, %4, %8\n v_sub_f32 %1, %5, %8\n v_sub_f32 %2, %6, %8\n v_sub_f32 %3, %4, %9\n"
                     " v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_sub_f32 %7, %5, %9\n"
                     " v_add_f32 %0, %0, %1\n v_mul_f32 %3, %3, %3\n v_mul_f32 %7, %7, %7\n v_add_f32 %0, %0, %2\n"
                     " v_sub_f32 %1, %6, %9\n v_add_f32 %3, %3, %7\n v_mul_f32 %1, %1, %1\n v_add_f32 %3, %3, %1\n"
                     " v_min3_u32 %2, %2, %0, %3\n v_min3_u32 %7, %7, %0, %3\n")
// ... and on packed ops: two haystack descriptors per lane
KERNEL64(k_mix_color_pk, "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n"
                         " v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n"
                         " v_pk_add_f32 %3, %0, %1\n v_pk_add_f32 %3, %3, %2\n")
KERNEL64(k_mix_color_pkfma, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                            " v_pk_fma_f32 %0, %0, %0, %9\n v_pk_fma_f32 %1, %1, %1, %9\n v_pk_fma_f32 %2, %2, %2, %9\n"
                            " v_pk_fma_f32 %3, %0, %8, %1\n v_pk_fma_f32 %3, %3, %8, %2\n")

struct K {
  const char* name;
  void (*fn)(float*, int, f2);
  int instr;        // instructions per body
  int floats;       // f32 results per lane per instruction
};

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4 * sizeof(float));
  const int iters = 2000, blocks = 256 * 8;  // 8 workgroups (32 waves) per CU: 8 waves per SIMD
  std::vector<K> ks = {{"pk_fma vvv", k_pk_fma_vvv, 8, 2}, {"pk_mul vv", k_pk_mul_vv, 8, 2}, {"pk_add vv", k_pk_add_vv, 8, 2},
                       {"pk_add sv", k_pk_add_sv, 8, 2}, {"pk_fma svv", k_pk_fma_svv, 8, 2}, {"sub vv", k_sub_vv, 8, 1},
                       {"sub sv", k_sub_sv, 8, 1}, {"mul vv", k_mul_vv, 8, 1}, {"fma vvv", k_fma_vvv, 8, 1},
                       {"min_u32", k_min_u32, 8, 1}, {"min3_u32", k_min3_u32, 8, 1}, {"min_f32", k_min_f32, 8, 1},
                       {"mix colour 32-bit", k_mix_color32, 9, 1}, {"mix cell (dist3)", k_mix_cell, 18, 1}, {"mix colour pk", k_mix_color_pk, 8, 2},
                       {"mix colour pk_fma", k_mix_color_pkfma, 8, 2}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // occupancy sweep on the colour-shaped mixes: does the ~2.2-cycle rate need many waves per SIMD?
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    for (auto& k : ks) {
      if (k.name[0] != 'm' || k.name[1] != 'i' || k.name[2] != 'x') continue;
      const int b = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.fn, dim3(b), dim3(256), 0, 0, out, iters, f2{1.f, 1.f});
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double per_simd = (double)b * 4 * iters * 16.0 * k.instr / (256.0 * 4);
      printf("waves/SIMD %d  %-20s %.2f cycles per wave-instruction per SIMD\n", wps, k.name, ms * 1e-3 * 2.4e9 / per_simd);
    }
  }
  for (auto& k : ks) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, iters, f2{1.f, 1.f});
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 /*waves*/ * iters * 16.0 * k.instr;  // wave-instructions
    const double per_simd = instr / (256.0 * 4);                                 // per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
    printf("%-20s %8.3f ms  %.2f cycles per wave-instruction per SIMD @2.4GHz  -> %.1f f32 results/clk/SIMD\n", k.name,
           ms, cyc, 64.0 * k.floats / cyc);
  }
  return 0;
}
