// segread.hip -- what the row streaming of k_band_area can reach: waves that do nothing but load short row segments.
// A "group" is four images of h rows of `stride` bytes; a wave takes `seg` bytes (16 per lane, lanes = 4 images x seg / 16)
// of every row at byte offset `off`, four rows per step, `ahead` steps in flight, like the kernel's staging lanes.
//   segread <w> <h> <seg> <off> <n_strips> [xcd_aware] [lds_kb] [ahead] : GB/s of the bytes the waves asked for
// lds_kb: LDS the one-wave workgroup claims (16 -> 10 waves per CU, k_band_area's occupancy); ahead: steps in flight (2 / 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(const unsigned char* base, unsigned long long bytes, unsigned n_groups, int w, int h,
                                        int seg, int off0, int strip_step, int n_strips, int xcd, unsigned* sink, int ahead) {
  extern __shared__ unsigned char lds_[];
  if (threadIdx.x == 999) lds_[0] = 1;
  unsigned grp, sidx;
  if (xcd) {
    const unsigned wg = blockIdx.x, blk = wg / (8u * n_strips), in_blk = wg % (8u * n_strips);
    grp = blk * 8u + (in_blk & 7u), sidx = in_blk >> 3;
  } else {
    grp = blockIdx.x / n_strips, sidx = blockIdx.x % n_strips;
  }
  if (grp >= n_groups) return;
  const int lane = threadIdx.x, q = lane >> 4, n16 = lane & 15;
  const unsigned long long gbase = (unsigned long long)grp * 4ull * (unsigned long long)w * h;
  const unsigned long long left = bytes - gbase;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base + gbase), 0,
                                                                        (int)(left > 0xffffffffull ? 0xffffffffu : left), 0x27000);
  const bool act = 16 * n16 < seg;
  const unsigned voff = (unsigned)q * (unsigned)w * (unsigned)h + (unsigned)(off0 + sidx * strip_step) + 16u * n16;
  unsigned acc = 0;
  v4u buf[4][4];
  auto ld = [&](int t, v4u (&b)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned so = (unsigned)min(4 * t + r, h - 1) * (unsigned)w;
      if (ahead >= 10) {  // two 8-byte halves per lane (16-byte lane stride), as k_band_area staged its rows at first
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u lo = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, (int)so, 0);
        const v2u hi = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff + 8, (int)so, 0);
        b[r] = v4u{lo.x, lo.y, hi.x, hi.y};
      } else {
        b[r] = act ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)so, 0) : v4u{0, 0, 0, 0};
      }
    }
  };
  auto use = [&](v4u (&b)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc ^= b[r].x ^ b[r].y ^ b[r].z ^ b[r].w;
  };
  const int steps = (h + 3) / 4;
  if (ahead == 2 || ahead >= 10) {
    ld(0, buf[0]);
    ld(1, buf[1]);
    for (int t = 0; t < steps; t += 2) {
      use(buf[0]);
      if (t + 2 < steps) ld(t + 2, buf[0]);
      use(buf[1]);
      if (t + 3 < steps) ld(t + 3, buf[1]);
    }
  } else {
    ld(0, buf[0]), ld(1, buf[1]), ld(2, buf[2]), ld(3, buf[3]);
    for (int t = 0; t < steps; t += 4) {
      use(buf[0]);
      if (t + 4 < steps) ld(t + 4, buf[0]);
      use(buf[1]);
      if (t + 5 < steps) ld(t + 5, buf[1]);
      use(buf[2]);
      if (t + 6 < steps) ld(t + 6, buf[2]);
      use(buf[3]);
      if (t + 7 < steps) ld(t + 7, buf[3]);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
  const int w = atoi(argv[1]), h = atoi(argv[2]), seg = atoi(argv[3]), off = atoi(argv[4]), ns = atoi(argv[5]);
  const int xcd = argc > 6 ? atoi(argv[6]) : 0;
  const int lds_kb = argc > 7 ? atoi(argv[7]) : 0, ahead = argc > 8 ? atoi(argv[8]) : 2;
  const int strip_step = ns > 1 ? (w - seg - off) / (ns - 1) : 0;
  const unsigned long long total = 6000000000ull;
  const unsigned n_groups = (unsigned)(total / (4ull * w * h));
  const unsigned long long bytes = (unsigned long long)n_groups * 4ull * w * h;
  unsigned char* d;
  unsigned* sink;
  hipMalloc(&d, bytes + 64);
  hipMalloc(&sink, 4);
  hipMemset(d, 1, bytes);
  const unsigned wgs = (n_groups + 7) / 8 * 8 * ns;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(64), (size_t)lds_kb * 1024, 0, d, bytes, n_groups, w, h, seg, off, strip_step, ns, xcd, sink, ahead);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) best = ms < best ? ms : best;
  }
  const double asked = (double)n_groups * 4.0 * h * (double)seg * ns;
  printf("lds %d KB ahead %d | w %d h %d seg %d off %d strips %d step %d xcd %d: %.3f ms  asked %.0f GB/s  image bytes %.0f GB/s\n", lds_kb, ahead, w, h, seg, off, ns,
         strip_step, xcd, best, asked / best * 1e-6, (double)bytes / best * 1e-6);
  return 0;
}
