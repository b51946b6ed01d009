// v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8 on gfx950: (1) what they compute, against a byte-level model; (2) how fast they
// issue beside v_dot4_u32_u8, v_add_u32 and v_fma_f32 (8 independent accumulators, 1..8 waves per SIMD).
// k_dcthash_256 forms its horizontal 7-tap sums from them (two instructions per 4 pixels instead of seven v_dot4).
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/qsad_rate qsad_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

__global__ void k_sem(const uint64_t* s0, const uint32_t* s1, const uint64_t* s2, uint64_t* q, uint64_t* m, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  q[i] = __builtin_amdgcn_qsad_pk_u16_u8(s0[i], s1[i], s2[i]);
  m[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], s2[i]);
}

#define K64(name, body)                                                                                  \
  __global__ __launch_bounds__(256) void name(uint64_t* out, int iters, uint32_t ref) {                 \
    uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
             a7 = a0 + 7, b0 = 0x0102030405060708ull * (threadIdx.x + 1);                                \
    for (int i = 0; i < iters; ++i) {                                                                    \
      asm volatile(REP16(body)                                                                           \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                   : "v"(b0), "v"(ref));                                                                 \
    }                                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                         \
  }
#define K32(name, body)                                                                                  \
  __global__ __launch_bounds__(256) void name(uint64_t* out, int iters, uint32_t ref) {                 \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
             a7 = a0 + 7, b0 = 0x01020304u * (threadIdx.x + 1);                                          \
    for (int i = 0; i < iters; ++i) {                                                                    \
      asm volatile(REP16(body)                                                                           \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                   : "v"(b0), "v"(ref));                                                                 \
    }                                                                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                         \
  }
#define L8(op, tail)                                                                                              \
  op " %0, " tail ", %0\n" op " %1, " tail ", %1\n" op " %2, " tail ", %2\n" op " %3, " tail ", %3\n" op " %4, " tail \
     ", %4\n" op " %5, " tail ", %5\n" op " %6, " tail ", %6\n" op " %7, " tail ", %7\n"

K64(k_qsad, L8("v_qsad_pk_u16_u8", "%8, %9"))
K64(k_mqsad, L8("v_mqsad_pk_u16_u8", "%8, %9"))
K32(k_dot4, L8("v_dot4_u32_u8", "%8, %9"))
K32(k_sad, L8("v_sad_u8", "%8, %9"))
K32(k_add3, L8("v_add3_u32", "%8, %9"))
K32(k_perm, L8("v_perm_b32", "%8, %9"))
K32(k_alignbit, L8("v_alignbit_b32", "%8, %9"))
K32(k_fma, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
           " v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
K32(k_add, "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
           " v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n")
K64(k_pkfma, "v_pk_fma_f32 %0, %8, %8, %0\n v_pk_fma_f32 %1, %8, %8, %1\n v_pk_fma_f32 %2, %8, %8, %2\n v_pk_fma_f32 %3, %8, %8, %3\n"
             " v_pk_fma_f32 %4, %8, %8, %4\n v_pk_fma_f32 %5, %8, %8, %5\n v_pk_fma_f32 %6, %8, %8, %6\n v_pk_fma_f32 %7, %8, %8, %7\n")

static unsigned sad4(uint32_t a, uint32_t b, bool masked) {
  unsigned s = 0;
  for (int k = 0; k < 4; ++k) {
    int x = (a >> (8 * k)) & 255, y = (b >> (8 * k)) & 255;
    if (masked && y == 0) continue;
    s += (unsigned)(x > y ? x - y : y - x);
  }
  return s;
}

int main() {
  // ---- semantics
  const int n = 1 << 16;
  std::mt19937_64 rng(7);
  std::vector<uint64_t> s0(n), s2(n), q(n), m(n);
  std::vector<uint32_t> s1(n);
  for (int i = 0; i < n; ++i) {
    s0[i] = rng();
    s2[i] = rng() & 0x0fff0fff0fff0fffull;
    s1[i] = (uint32_t)rng();
    if (i % 4 == 0) s1[i] = 0;
    if (i % 4 == 1) s1[i] = 0xffffffffu;
    if (i % 4 == 2) s1[i] = 0x00ffffffu;
  }
  uint64_t *d0, *d2, *dq, *dm;
  uint32_t* d1;
  hipMalloc(&d0, n * 8), hipMalloc(&d2, n * 8), hipMalloc(&dq, n * 8), hipMalloc(&dm, n * 8), hipMalloc(&d1, n * 4);
  hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(d2, s2.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, d0, d1, d2, dq, dm, n);
  hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(m.data(), dm, n * 8, hipMemcpyDeviceToHost);
  int badq = 0, badm = 0;
  for (int i = 0; i < n; ++i) {
    uint64_t wq = 0, wm = 0;
    for (int j = 0; j < 4; ++j) {
      const uint32_t win = (uint32_t)(s0[i] >> (8 * j));
      const unsigned c = (unsigned)(s2[i] >> (16 * j)) & 0xffffu;
      wq |= (uint64_t)((sad4(win, s1[i], false) + c) & 0xffffu) << (16 * j);
      wm |= (uint64_t)((sad4(win, s1[i], true) + c) & 0xffffu) << (16 * j);
    }
    badq += wq != q[i];
    badm += wm != m[i];
  }
  printf("{\"semantics\": {\"qsad_mismatch\": %d, \"mqsad_mismatch\": %d, \"cases\": %d, \"model\": \"D.u16[j] = "
         "(M)SAD_U8(S0 >> 8j, S1) + S2.u16[j], j = 0..3; M: reference bytes equal to 0 are skipped\"}}\n", badq, badm, n);
  // ---- rates
  uint64_t* out;
  hipMalloc(&out, 4096 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int iters = 2000;
  struct K {
    const char* name;
    void (*fn)(uint64_t*, int, uint32_t);
  } ks[] = {{"v_qsad_pk_u16_u8", k_qsad}, {"v_mqsad_pk_u16_u8", k_mqsad}, {"v_dot4_u32_u8", k_dot4}, {"v_sad_u8", k_sad},
            {"v_add3_u32", k_add3},       {"v_perm_b32", k_perm},         {"v_alignbit_b32", k_alignbit},
            {"v_fma_f32", k_fma},         {"v_add_u32", k_add},           {"v_pk_fma_f32", k_pkfma}};
  for (auto& k : ks) {
    printf("{\"op\": \"%s\", \"cycles_per_wave_instr_per_simd\": {", k.name);
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: 256 CUs x 4 SIMDs x wps waves = 256 x wps workgroups of 4 waves
      const int wgs = 256 * wps;
      hipLaunchKernelGGL(k.fn, dim3(wgs), dim3(256), 0, 0, out, 10, 0x00ffffffu);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.fn, dim3(wgs), dim3(256), 0, 0, out, iters, 0x00ffffffu);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * 16 * 8 * wps;  // every SIMD runs wps waves
      printf("%s\"%d\": %.2f", wps == 1 ? "" : ", ", wps, ms * 1e-3 * clk_khz * 1e3 / instr_per_simd);
    }
    printf("}, \"clock_khz\": %d}\n", clk_khz);
  }
  return 0;
}
