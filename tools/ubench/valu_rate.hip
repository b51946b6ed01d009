// VALU issue-rate microbenchmark for gfx950: how many lanes/clk/SIMD does each opcode sustain?
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, body)                                                        \
  __global__ __launch_bounds__(256) void name(unsigned* out, int iters, unsigned s) { \
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;   \
    unsigned a4 = a0 ^ 0x55, a5 = a0 + 99, a6 = a0 * 11, a7 = a0 | 0x1000;          \
    unsigned long long l0 = a0, l1 = a1;                                            \
    for (int i = 0; i < iters; ++i) {                                               \
      asm volatile(REP16(body)                                                      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),    \
                     "+v"(a6), "+v"(a7), "+v"(l0), "+v"(l1)                         \
                   : "s"(s)                                                         \
                   : "vcc");                                                        \
    }                                                                               \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)l0 + (unsigned)l1; \
  }

// each body = 8 independent instructions (one per accumulator)
#define B8(op, tail)                                                                         \
  op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" \
  op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"

KERNEL(k_xor_vv, B8("v_xor_b32", ", %1"))
KERNEL(k_xor_sv, B8("v_xor_b32", ", %10"))
KERNEL(k_bcnt, B8("v_bcnt_u32_b32", ", 0"))
KERNEL(k_bcnt_acc, B8("v_bcnt_u32_b32", ", %2"))
KERNEL(k_min3, B8("v_min3_u32", ", %1, %2"))
KERNEL(k_add_u32, B8("v_add_u32", ", %1"))
KERNEL(k_add3, B8("v_add3_u32", ", %1, %2"))
KERNEL(k_fma, B8("v_fma_f32", ", %1, %2"))
KERNEL(k_addf, B8("v_add_f32", ", %1"))
KERNEL(k_mul24, B8("v_mul_u32_u24", ", %1"))
KERNEL(k_mulhi24, B8("v_mul_hi_u32_u24", ", %1"))
KERNEL(k_mad24, B8("v_mad_u32_u24", ", %1, %2"))
KERNEL(k_mullo, B8("v_mul_lo_u32", ", %1"))
KERNEL(k_dot4, B8("v_dot4_u32_u8", ", %1, %2"))
KERNEL(k_sad, B8("v_sad_u8", ", %1, %2"))
KERNEL(k_perm, B8("v_perm_b32", ", %1, %2"))
KERNEL(k_pkadd16, B8("v_pk_add_u16", ", %1"))
KERNEL(k_pkmad16, B8("v_pk_mad_u16", ", %1, %2"))
KERNEL(k_alignbit, B8("v_alignbit_b32", ", %1, 16"))
KERNEL(k_lshlor, B8("v_lshl_or_b32", ", 16, %1"))
KERNEL(k_bfe, B8("v_bfe_u32", ", 8, 8"))
KERNEL(k_cvt_f32_u32, B8("v_cvt_f32_u32", ""))
KERNEL(k_cvt_u32_f32, B8("v_cvt_u32_f32", ""))
KERNEL(k_rndne, B8("v_rndne_f32", ""))
KERNEL(k_add_sdwa, B8("v_add_u32_sdwa", ", %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"))
KERNEL(k_mul24_sdwa, B8("v_mul_u32_u24_sdwa", ", %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"))
KERNEL(k_mov_dpp, B8("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp_wave, B8("v_mov_b32_dpp", " wave_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_add_dpp, B8("v_add_u32_dpp", ", %1 row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_cmp_eq64, "v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n v_cmp_eq_u64 vcc, %8, %9\n")
KERNEL(k_cmp_gt32, "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %2\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_gt_u32 vcc, %3, %4\n v_cmp_gt_u32 vcc, %4, %5\n v_cmp_gt_u32 vcc, %5, %6\n v_cmp_gt_u32 vcc, %6, %7\n v_cmp_gt_u32 vcc, %7, %0\n")
KERNEL(k_pkaddf32, "v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %8\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 1, %9\n v_lshl_add_u64 %9, %9, 1, %8\n v_lshl_add_u64 %8, %8, 1, %9\n v_lshl_add_u64 %9, %9, 1, %8\n v_lshl_add_u64 %8, %8, 1, %9\n v_lshl_add_u64 %9, %9, 1, %8\n v_lshl_add_u64 %8, %8, 1, %9\n v_lshl_add_u64 %9, %9, 1, %8\n")
KERNEL(k_mix_xor_fma, "v_xor_b32 %0, %0, %1\n v_fma_f32 %4, %4, %5, %6\n v_xor_b32 %1, %1, %2\n v_fma_f32 %5, %5, %6, %7\n v_xor_b32 %2, %2, %3\n v_fma_f32 %6, %6, %7, %4\n v_xor_b32 %3, %3, %0\n v_fma_f32 %7, %7, %4, %5\n")

struct K { const char* name; void (*fn)(unsigned*, int, unsigned); };
#define E(k) {#k, k}

int main() {
  std::vector<K> ks = {E(k_xor_vv), E(k_xor_sv), E(k_bcnt), E(k_bcnt_acc), E(k_min3), E(k_add_u32), E(k_add3),
                       E(k_fma), E(k_addf), E(k_mul24), E(k_mulhi24), E(k_mad24), E(k_mullo), E(k_dot4), E(k_sad),
                       E(k_perm), E(k_pkadd16), E(k_pkmad16), E(k_alignbit), E(k_lshlor), E(k_bfe),
                       E(k_cvt_f32_u32), E(k_cvt_u32_f32), E(k_rndne), E(k_add_sdwa), E(k_mul24_sdwa),
                       E(k_mov_dpp), E(k_mov_dpp_wave), E(k_add_dpp), E(k_cmp_eq64), E(k_cmp_gt32),
                       E(k_pkaddf32), E(k_lshl_add_u64), E(k_mix_xor_fma)};
  unsigned* out;
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU -> 8 waves/SIMD
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 10, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 /*waves*/ * iters * 16 * 8;  // wave-instructions
    double lane_ops = instr * 64;
    // lanes per clk per SIMD assuming 2.4 GHz, 1024 SIMDs
    printf("%-16s %8.3f ms  %.3e lane-ops/s  %.1f lanes/clk/SIMD@2.4GHz\n", k.name, ms, lane_ops / ms * 1e3,
           lane_ops / ms * 1e3 / (1024.0 * 2.4e9));
  }
  return 0;
}
