// color.hip -- ColorDescIndex (src/colordescindex.{h,cpp}) and ColorDescriptor::distance
// (src/cvutil.cpp:682-749) on gfx950.
//
// Reference: a descriptor is 32 x {L,u,v,w : u16} + numColors : u8 (258 bytes, src/cvutil.h:57-113);
// find() is a linear scan of distance(target, desc[i]) over the whole index (colordescindex.cpp:250-278):
//   skip when either side has 0 colours or the counts differ by more than 2 (:683-684);
//   a := the side with more colours (:697-706); score = 1 + sum_{i<numA} min_{j<numB} |a_i - b_j|_2 in
//   float, colours decompressed as l=L*100/65535, u=U*354/65535-134, v=V*262/65535-140 (cvutil.h:83-87);
//   Match(id, int(score)) for id != 0, in index order.
//
// Layout: the index is stored decompressed and planar -- l/u/v[32][capacity] f32, num[capacity] u8 -- so that a
// wave reads each colour plane with unit stride (the 258-byte AoS record would give every lane its own cache
// line).  k_color_dist: one lane per haystack descriptor; the needle's (<= 32) colours are wave-uniform SGPR
// operands.  Per colour pair 3 sub, 3 mul, 2 add in the reference's association order (no FMA contraction:
// the library is built with -ffp-contract=off), one running minimum per needle colour and one per haystack
// colour (either side can be "a"); sqrtf is monotonic, so it is applied to the 32 minima instead of the 1024
// pair distances (bit-identical result), then summed in index order starting from 1.0f.

#include <cfloat>
#include <map>

#include "cbh_index.h"

namespace {

constexpr int kNC = 32;  // ColorDescriptor::NUM_DESC_COLORS
constexpr size_t kDescBytes = 258;

struct alignas(16) NeedleF {  // 400 bytes: k_color_dist3 reads the components as float4
  float l[kNC], u[kNC], v[kNC];
  int num;
};

// Launch geometry of the distance kernels: blockIdx.x = NEEDLE (fastest), blockIdx.y/z = haystack tile.  The
// workgroups that run together then share one tile: it comes from HBM once per XCD instead of once per needle (with
// the tile fastest, 64 needles re-streamed a 384 MB index 64 times -- 1.8 TB/s of traffic that held every variant of
// the kernel at ~14 ms whatever its instruction mix).
__device__ __forceinline__ uint32_t color_tile() { return blockIdx.y + blockIdx.z * 32768u; }



// Two haystack descriptors per lane on the packed-f32 ALU (v_pk_add_f32 / v_pk_mul_f32 work on two
// independent floats per lane and issue like one VALU op): per colour pair 3 pk sub + 3 pk mul + 2 pk add
// for both descriptors, then the minima per half.  Same operations in the same order as k_color_dist (no
// contraction: products and sums are rounded separately), so results are bit-identical; missing haystack
// colours are loaded as 1e18 so their distance (~3e36, finite) can never win a minimum -- one select less.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// Missing colours on EITHER side are 1e18 in L (host: decompress_pad; device: the index planes are padded the
// same way), so a pair with a missing colour has a distance of ~1e36 (finite) that never wins a minimum and the
// loop needs no selects.  Squared distances are non-negative finite floats, whose order is the order of their
// bit patterns as unsigned integers: the minima are v_min_u32 / v_min3_u32 (no NaN canonicalisation ops).
__global__ __launch_bounds__(256) void k_color_dist2(const float* __restrict__ L, const float* __restrict__ U,
                                                     const float* __restrict__ V,
                                                     const unsigned char* __restrict__ num, size_t stride,
                                                     uint32_t n, const NeedleF* __restrict__ needles,
                                                     int* __restrict__ out /* [nq][n] */) {
  const uint32_t i0 = (color_tile() * blockDim.x + threadIdx.x) * 2u;  // descriptors i0, i0 + 1 (stride is even)
  const NeedleF& nd = needles[blockIdx.x];                             // wave-uniform, colours >= num padded
  const int nn = nd.num;
  const int hn0 = i0 < n ? (int)num[i0] : 0, hn1 = i0 + 1 < n ? (int)num[i0 + 1] : 0;
  const int hmax = max(hn0, hn1);
  constexpr uint32_t kBig = 0x7f7fffffu;  // FLT_MAX
  u2 rowmin[kNC];
#pragma unroll
  for (int p = 0; p < kNC; ++p) rowmin[p] = u2{kBig, kBig};
  f2 colacc = {1.0f, 1.0f};
  for (int h = 0; h < kNC; ++h) {
    if (__ballot(h < hmax) == 0ull) break;  // no lane of this wave has that many colours
    f2 hl = {1e18f, 1e18f}, hu = {0.f, 0.f}, hv = {0.f, 0.f};
    if (i0 < n) {  // planes are padded to an even capacity and with 1e18 beyond every descriptor's colours
      hl = *reinterpret_cast<const f2*>(L + (size_t)h * stride + i0);
      hu = *reinterpret_cast<const f2*>(U + (size_t)h * stride + i0);
      hv = *reinterpret_cast<const f2*>(V + (size_t)h * stride + i0);
    }
    u2 colmin = {kBig, kBig};
#pragma unroll
    for (int p = 0; p < kNC; p += 2) {
      const f2 dl0 = f2{nd.l[p], nd.l[p]} - hl, du0 = f2{nd.u[p], nd.u[p]} - hu, dv0 = f2{nd.v[p], nd.v[p]} - hv;
      const f2 dl1 = f2{nd.l[p + 1], nd.l[p + 1]} - hl, du1 = f2{nd.u[p + 1], nd.u[p + 1]} - hu,
               dv1 = f2{nd.v[p + 1], nd.v[p + 1]} - hv;
      const f2 e0 = dl0 * dl0 + du0 * du0 + dv0 * dv0;
      const f2 e1 = dl1 * dl1 + du1 * du1 + dv1 * dv1;
      const u2 b0 = __builtin_bit_cast(u2, e0), b1 = __builtin_bit_cast(u2, e1);
      rowmin[p] = u2{min(rowmin[p].x, b0.x), min(rowmin[p].y, b0.y)};
      rowmin[p + 1] = u2{min(rowmin[p + 1].x, b1.x), min(rowmin[p + 1].y, b1.y)};
      colmin = u2{min(min(colmin.x, b0.x), b1.x), min(min(colmin.y, b0.y), b1.y)};  // v_min3_u32
    }
    // (scalar copies first: __builtin_bit_cast on a vector-element lvalue reads element 0 with this clang)
    const uint32_t cm0 = colmin.x, cm1 = colmin.y;
    if (h < hn0) colacc.x += sqrtf(__builtin_bit_cast(float, cm0));  // haystack side is "a" (more colours)
    if (h < hn1) colacc.y += sqrtf(__builtin_bit_cast(float, cm1));
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const uint32_t i = i0 + (uint32_t)e;
    const int hn = e ? hn1 : hn0;
    int result = -1;
    if (i < n && nn != 0 && hn != 0 && abs(nn - hn) <= 2) {
      float score;
      if (nn < hn) {
        score = e ? colacc.y : colacc.x;
      } else {
        score = 1.0f;
#pragma unroll
        for (int p = 0; p < kNC; ++p)
          if (p < nn) {
            const uint32_t rm = e ? rowmin[p].y : rowmin[p].x;
            score += sqrtf(__builtin_bit_cast(float, rm));
          }
      }
      result = (int)score;
    }
    if (i < n) out[(size_t)blockIdx.x * n + i] = result;
  }
}

// k_color_dist3 (default): the same arithmetic shaped for the VALU's fast path.  tools/ubench/pk_f32_rate.hip on the
// MI355X: a wave issues one VALU instruction every ~4.3-5 cycles, and a SIMD interleaves TWO waves, so plain
// v_sub/v_mul/v_add_f32 and v_min3_u32 on VGPR operands retire every ~2.15 cycles per SIMD with 2, 4, 6 or 8 resident
// waves -- but every 2.8 with 3 (an odd wave has no partner), every 4.1 when an operand is an SGPR (k_color_dist,
// k_color_dist2 keep the needle in SGPRs), and the packed forms (v_pk_*_f32) every 4.2 for two floats per lane, i.e.
// no gain.  So: needle colours in VGPRs, 32-bit ops, and an EVEN occupancy.  All 96 needle components + 32 running
// minima do not fit 128 VGPRs (4 waves), so the needle is taken in two halves of 16 colours: a lane owns one haystack
// descriptor, walks its colours two at a time against the 16 needle colours of the half (3 sub + 3 mul + 2 add per
// pair, one v_min3_u32 per needle colour for its running minimum, one per two needle colours for each haystack
// colour's), and parks the haystack-side minima of the first half in LDS ([colour][lane], conflict-free) for the
// second half to finish.  40 KB of LDS per workgroup also pins the occupancy at 4 workgroups per CU = 4 waves per
// SIMD.  Operation order per distance and per sum is k_color_dist's (minima are order-independent; the two sums run
// over ascending colour index), so the results are bit-identical.  RAW: also the float distance (FLT_MAX where the
// reference's distance() returns FLT_MAX).
// FMA ("color_fma" 1, never the default): dl^2 + du^2 + dv^2 as one multiply and two fused multiply-adds -- 6
// floating-point instructions per colour pair instead of 8.  The squared distance then carries one rounding instead of
// three, so a distance can differ from ColorDescriptor::distance's (src/cvutil.cpp:700-735: plain float expressions,
// which GCC does not contract across statements at cbird's -O2) in its last bits: within north_star's 1e-5 on the float,
// but int(score) is not guaranteed to match at an integer boundary.  tools/bench_configs.py reports the rate both ways and
// how many scores move.
constexpr int kHalf = kNC / 2;
template <bool RAW, bool FMA>
__global__ __launch_bounds__(256) void k_color_dist3(const float* __restrict__ L, const float* __restrict__ U,
                                                     const float* __restrict__ V,
                                                     const unsigned char* __restrict__ num, size_t stride,
                                                     uint32_t n, const NeedleF* __restrict__ needles,
                                                     int* __restrict__ out /* [nq][n] */,
                                                     float* __restrict__ raw /* [nq][n] or null */) {
  __shared__ uint32_t s_cmin[kNC + 8][256];  // +8 rows: 33 KB -> 40 KB, at most 4 workgroups per CU
  const uint32_t i = color_tile() * blockDim.x + threadIdx.x;
  const NeedleF& nd = needles[blockIdx.x];  // wave-uniform, colours >= num padded with 1e18
  const int nn = nd.num;
  const int hn = i < n ? (int)num[i] : 0;
  const uint32_t ii = i < n ? i : 0;  // (lanes past the end read descriptor 0 and write nothing)
  constexpr uint32_t kBig = 0x7f7fffffu;  // FLT_MAX
  unsigned zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));  // an index the compiler cannot prove uniform: vector loads below
  const float4* np4 = reinterpret_cast<const float4*>(&nd) + zero;
  float colacc = 1.0f, rowacc = 1.0f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // this half's 48 needle components straight into VGPRs (every lane reads the same bytes: broadcast lines)
    float nl[kHalf], nu[kHalf], nv[kHalf];
#pragma unroll
    for (int p = 0; p < kHalf / 4; ++p) {
      const float4 a = np4[half * (kHalf / 4) + p], b = np4[kNC / 4 + half * (kHalf / 4) + p],
                   c = np4[2 * (kNC / 4) + half * (kHalf / 4) + p];
      nl[4 * p] = a.x, nl[4 * p + 1] = a.y, nl[4 * p + 2] = a.z, nl[4 * p + 3] = a.w;
      nu[4 * p] = b.x, nu[4 * p + 1] = b.y, nu[4 * p + 2] = b.z, nu[4 * p + 3] = b.w;
      nv[4 * p] = c.x, nv[4 * p + 1] = c.y, nv[4 * p + 2] = c.z, nv[4 * p + 3] = c.w;
    }
    uint32_t rowmin[kHalf];
#pragma unroll
    for (int p = 0; p < kHalf; ++p) rowmin[p] = kBig;
    // planes are padded with 1e18 beyond every descriptor's colours: no guards in the loop.  The next step's six
    // values are requested before this step's arithmetic.
    float al = L[ii], au = U[ii], av = V[ii];
    float bl = L[stride + ii], bu = U[stride + ii], bv = V[stride + ii];
    for (int h = 0; h < kNC; h += 2) {
      if (__ballot(h < hn) == 0ull) break;  // no lane of this wave has that many colours
      const int hnx = h + 2 < kNC ? h + 2 : h;  // (the last step re-reads its own colours: harmless)
      const float xal = L[(size_t)hnx * stride + ii], xau = U[(size_t)hnx * stride + ii],
                  xav = V[(size_t)hnx * stride + ii];
      const float xbl = L[(size_t)(hnx + 1) * stride + ii], xbu = U[(size_t)(hnx + 1) * stride + ii],
                  xbv = V[(size_t)(hnx + 1) * stride + ii];
      uint32_t cmin_a = kBig, cmin_b = kBig;
      if (half) cmin_a = s_cmin[h][threadIdx.x], cmin_b = s_cmin[h + 1][threadIdx.x];
#pragma unroll
      for (int p = 0; p < kHalf; p += 2) {
        uint32_t da[2], db[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float dl = nl[p + e] - al, du = nu[p + e] - au, dv = nv[p + e] - av;
          const float el = nl[p + e] - bl, eu = nu[p + e] - bu, ev = nv[p + e] - bv;
          const float d2a = FMA ? __builtin_fmaf(dv, dv, __builtin_fmaf(du, du, dl * dl)) : dl * dl + du * du + dv * dv;
          const float d2b = FMA ? __builtin_fmaf(ev, ev, __builtin_fmaf(eu, eu, el * el)) : el * el + eu * eu + ev * ev;
          da[e] = __builtin_bit_cast(uint32_t, d2a);
          db[e] = __builtin_bit_cast(uint32_t, d2b);
          rowmin[p + e] = min(min(rowmin[p + e], da[e]), db[e]);  // v_min3_u32
        }
        cmin_a = min(min(cmin_a, da[0]), da[1]);
        cmin_b = min(min(cmin_b, db[0]), db[1]);
      }
      if (half == 0) {
        s_cmin[h][threadIdx.x] = cmin_a, s_cmin[h + 1][threadIdx.x] = cmin_b;  // own column only: no barrier needed
      } else {
        if (h < hn) colacc += sqrtf(__builtin_bit_cast(float, cmin_a));  // haystack side is "a" (more colours)
        if (h + 1 < hn) colacc += sqrtf(__builtin_bit_cast(float, cmin_b));
      }
      al = xal, au = xau, av = xav, bl = xbl, bu = xbu, bv = xbv;
    }
#pragma unroll
    for (int p = 0; p < kHalf; ++p)  // needle side is "a": sum over needle colours in index order, across the halves
      if (half * kHalf + p < nn) rowacc += sqrtf(__builtin_bit_cast(float, rowmin[p]));
  }
  if (i >= n) return;
  int result = -1;
  float score = FLT_MAX;
  if (nn != 0 && hn != 0 && abs(nn - hn) <= 2) {
    score = nn < hn ? colacc : rowacc;
    result = (int)score;
  }
  out[(size_t)blockIdx.x * n + i] = result;
  if (RAW) raw[(size_t)blockIdx.x * n + i] = score;
}

// key = score<<32 | id for entries that match (score >= 0, id != 0), ~0 otherwise
__global__ __launch_bounds__(256) void k_color_keys(const int* __restrict__ score,
                                                    const uint32_t* __restrict__ ids, uint32_t n,
                                                    unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = score[i];
  const uint32_t id = ids[i];
  keys[i] = (s >= 0 && id != 0) ? (((unsigned long long)(uint32_t)s << 32) | id) : ~0ull;
}


// ---- top-k per needle without sorting the whole score row (find_batch) -----------------------------------
// Scores are small integers, so the k-th smallest score can be found exactly with a histogram; only entries at
// or under it are collected and ordered on the host.  The k best of ~10^6 entries lie within a few hundred of
// the minimum, so the histogram is a 2048-bin window above the per-needle minimum score (k_color_min), kept in
// LDS per block and flushed once (one pass of global atomics over ALL scores took 5 ms per 64 needles; this
// takes 0.2 ms).  A k-th score beyond the window, or more ties than kCandCap, falls back to the full sort.
constexpr int kWin = 2048;
constexpr uint32_t kCandCap = 4096;  // per needle; more ties than this -> the full-sort path

__global__ __launch_bounds__(256) void k_color_min(const int* __restrict__ score, const uint32_t* __restrict__ ids,
                                                   uint32_t n, int* __restrict__ smin /* [nq], init INT_MAX */) {
  const uint32_t q = blockIdx.y;
  const int* sc = score + (size_t)q * n;
  int m = 0x7fffffff;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const int s = sc[i];
    if (s >= 0 && ids[i] != 0) m = min(m, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m != 0x7fffffff) atomicMin(&smin[q], m);
}

__global__ __launch_bounds__(256) void k_color_hist(const int* __restrict__ score,
                                                    const uint32_t* __restrict__ ids, uint32_t n,
                                                    const int* __restrict__ smin,
                                                    uint32_t* __restrict__ hist /* [nq][kWin] */,
                                                    uint32_t* __restrict__ valid /* [nq] */) {
  __shared__ uint32_t sh[kWin];
  const uint32_t q = blockIdx.y;
  const int* sc = score + (size_t)q * n;
  const int base = smin[q];
  for (int b = threadIdx.x; b < kWin; b += 256) sh[b] = 0;
  __syncthreads();
  uint32_t local = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const int s = sc[i];
    if (s >= 0 && ids[i] != 0) {
      ++local;
      const int d = s - base;
      if (d < kWin) atomicAdd(&sh[d], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kWin; b += 256)
    if (sh[b]) atomicAdd(&hist[(size_t)q * kWin + b], sh[b]);
  if (local) atomicAdd(&valid[q], local);  // compiler aggregates per wave
}

// thr[q] = k-th smallest score when it lies inside the window, else INT_MAX (collect everything -> fallback)
__global__ __launch_bounds__(256) void k_color_thresh(const uint32_t* __restrict__ hist, uint32_t k,
                                                      const int* __restrict__ smin, int* __restrict__ thr) {
  __shared__ uint32_t part[256];
  const uint32_t q = blockIdx.x, t = threadIdx.x;
  const uint32_t* h = hist + (size_t)q * kWin + t * (kWin / 256);
  uint32_t sum = 0;
  for (int b = 0; b < kWin / 256; ++b) sum += h[b];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    int bin = -1;
    for (uint32_t j = 0; j < 256 && bin < 0; ++j) {
      if (run + part[j] >= k) {
        const uint32_t* hs = hist + (size_t)q * kWin + j * (kWin / 256);
        uint32_t c = run;
        for (int b = 0; b < kWin / 256; ++b) {
          c += hs[b];
          if (c >= k) {
            bin = (int)j * (kWin / 256) + b;
            break;
          }
        }
      }
      run += part[j];
    }
    thr[q] = bin >= 0 ? smin[q] + bin : 0x7fffffff;
  }
}

__global__ __launch_bounds__(256) void k_color_collect(const int* __restrict__ score,
                                                       const uint32_t* __restrict__ ids, uint32_t n,
                                                       const int* __restrict__ thr,
                                                       unsigned long long* __restrict__ cand /* [nq][kCandCap] */,
                                                       uint32_t* __restrict__ ncand /* [nq] */) {
  const uint32_t q = blockIdx.y;
  const int* sc = score + (size_t)q * n;
  const int T = thr[q];
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const int s = sc[i];
    const uint32_t id = ids[i];
    if (s >= 0 && id != 0 && s <= T) {
      const uint32_t slot = atomicAdd(&ncand[q], 1u);
      if (slot < kCandCap) cand[(size_t)q * kCandCap + slot] = ((unsigned long long)(uint32_t)s << 32) | id;
    }
  }
}

// distance kernels: k_color_dist2 (two descriptors per lane on packed f32; the default: 12.1 ms per 64 needles x 1M
// descriptors) and k_color_dist3 (32-bit ops on VGPR operands at 4 waves per SIMD: 12.5 ms; the one that can also hand
// out the raw floats, and the fused-square form of "color_fma").  Both sit at the VALU issue ceiling of this arithmetic
// (no FMA: the reference's rounding order) once the launch geometry lets concurrent workgroups share haystack tiles.
// (Round 1's one-descriptor-per-lane k_color_dist, 14.1 ms, is in the history: r05's "color_pk" 0.)
int g_color_fma = 0;  // "color_fma": 1 = k_color_dist3 with fused squares (faster, NOT bit-identical; see the kernel)

void decompress(const uint8_t* desc, NeedleF* out) {  // DescriptorColor::get, cvutil.h:83-87
  for (int c = 0; c < kNC; ++c) {
    uint16_t l, u, v;
    memcpy(&l, desc + c * 8 + 0, 2);
    memcpy(&u, desc + c * 8 + 2, 2);
    memcpy(&v, desc + c * 8 + 4, 2);
    out->l[c] = l * 100.0f / 65535;
    out->u[c] = u * 354.0f / 65535 - 134.0f;
    out->v[c] = v * 262.0f / 65535 - 140.0f;
  }
  out->num = desc[256];
  // colours the reference's loops never visit (index >= numColors): far away, so that a distance involving one
  // of them (~1e36, finite) can never be a minimum -- lets k_color_dist2 run without per-colour guards
  for (int c = out->num < kNC ? out->num : kNC; c < kNC; ++c) {
    out->l[c] = 1e18f;
    out->u[c] = 0.f;
    out->v[c] = 0.f;
  }
}

}  // namespace

namespace cbh {
void set_color_fma(int on) { g_color_fma = on ? 1 : 0; }
}  // namespace cbh

struct cbh_color {
  int device = 0;
  size_t n = 0, cap = 0;
  float *dL = nullptr, *dU = nullptr, *dV = nullptr;  // [32][cap]
  unsigned char* d_num = nullptr;
  uint32_t* d_ids = nullptr;
  std::vector<uint8_t> host_desc;  // AoS copy for findIndexData / slice (258 B each)
  std::vector<uint32_t> host_ids;
  std::mutex mu;
  hipStream_t stream = nullptr;
  NeedleF* d_needles = nullptr;
  size_t needles_cap = 0;
  int* d_scores = nullptr;
  size_t scores_cap = 0;
  unsigned long long *d_keys = nullptr, *d_keys_alt = nullptr;
  void* d_tmp = nullptr;
  size_t keys_cap = 0, tmp_bytes = 0;
  // find_batch top-k scratch
  uint32_t *d_hist = nullptr, *d_ncand = nullptr, *d_valid = nullptr;
  int *d_thr = nullptr, *d_smin = nullptr;
  unsigned long long* d_cand = nullptr;
  size_t topk_cap = 0;
};

namespace {

int grow_index(cbh_color* c, size_t need) {
  if (need <= c->cap) return CBH_OK;
  const size_t ncap = (std::max<size_t>(need, c->cap + c->cap / 2 + 4096) + 1) & ~(size_t)1;  // even: pair loads
  float *nL = nullptr, *nU = nullptr, *nV = nullptr;
  unsigned char* nn = nullptr;
  uint32_t* ni = nullptr;
  CBH_HIP(hipMalloc(&nL, ncap * kNC * 4));
  CBH_HIP(hipMalloc(&nU, ncap * kNC * 4));
  CBH_HIP(hipMalloc(&nV, ncap * kNC * 4));
  CBH_HIP(hipMalloc(&nn, ncap));
  CBH_HIP(hipMalloc(&ni, ncap * 4));
  if (c->n) {
    CBH_HIP(hipMemcpy2D(nL, ncap * 4, c->dL, c->cap * 4, c->n * 4, kNC, hipMemcpyDeviceToDevice));
    CBH_HIP(hipMemcpy2D(nU, ncap * 4, c->dU, c->cap * 4, c->n * 4, kNC, hipMemcpyDeviceToDevice));
    CBH_HIP(hipMemcpy2D(nV, ncap * 4, c->dV, c->cap * 4, c->n * 4, kNC, hipMemcpyDeviceToDevice));
    CBH_HIP(hipMemcpy(nn, c->d_num, c->n, hipMemcpyDeviceToDevice));
    CBH_HIP(hipMemcpy(ni, c->d_ids, c->n * 4, hipMemcpyDeviceToDevice));
  }
  for (void* p : {(void*)c->dL, (void*)c->dU, (void*)c->dV, (void*)c->d_num, (void*)c->d_ids})
    if (p) (void)hipFree(p);
  c->dL = nL;
  c->dU = nU;
  c->dV = nV;
  c->d_num = nn;
  c->d_ids = ni;
  c->cap = ncap;
  return CBH_OK;
}

// upload host descriptors [first, first+m) into the planar device arrays
int upload(cbh_color* c, size_t first, size_t m) {
  std::vector<float> pl((size_t)kNC * m), pu((size_t)kNC * m), pv((size_t)kNC * m);
  std::vector<unsigned char> pn(m);
  NeedleF f;
  for (size_t i = 0; i < m; ++i) {
    decompress(c->host_desc.data() + (first + i) * kDescBytes, &f);
    for (int k = 0; k < kNC; ++k) {
      pl[(size_t)k * m + i] = f.l[k];
      pu[(size_t)k * m + i] = f.u[k];
      pv[(size_t)k * m + i] = f.v[k];
    }
    pn[i] = (unsigned char)f.num;
  }
  CBH_HIP(hipMemcpy2D(c->dL + first, c->cap * 4, pl.data(), m * 4, m * 4, kNC, hipMemcpyHostToDevice));
  CBH_HIP(hipMemcpy2D(c->dU + first, c->cap * 4, pu.data(), m * 4, m * 4, kNC, hipMemcpyHostToDevice));
  CBH_HIP(hipMemcpy2D(c->dV + first, c->cap * 4, pv.data(), m * 4, m * 4, kNC, hipMemcpyHostToDevice));
  CBH_HIP(hipMemcpy(c->d_num + first, pn.data(), m, hipMemcpyHostToDevice));
  CBH_HIP(hipMemcpy(c->d_ids + first, c->host_ids.data() + first, m * 4, hipMemcpyHostToDevice));
  return CBH_OK;
}

int ensure_scratch(cbh_color* c, size_t nq, bool keys) {
  if (!c->stream) CBH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (nq > c->needles_cap) {
    if (c->d_needles) (void)hipFree(c->d_needles);
    c->d_needles = nullptr;
    CBH_HIP(hipMalloc(&c->d_needles, nq * sizeof(NeedleF)));
    c->needles_cap = nq;
  }
  if (nq * c->n > c->scores_cap) {
    if (c->d_scores) (void)hipFree(c->d_scores);
    c->d_scores = nullptr;
    CBH_HIP(hipMalloc(&c->d_scores, nq * c->n * 4));
    c->scores_cap = nq * c->n;
  }
  if (keys && c->n > c->keys_cap) {
    for (void* p : {(void*)c->d_keys, (void*)c->d_keys_alt, c->d_tmp})
      if (p) (void)hipFree(p);
    c->d_keys = c->d_keys_alt = nullptr;
    c->d_tmp = nullptr;
    CBH_HIP(hipMalloc(&c->d_keys, c->n * 8));
    CBH_HIP(hipMalloc(&c->d_keys_alt, c->n * 8));
    c->tmp_bytes = cbh::sort_records_scratch_bytes(c->n);
    CBH_HIP(hipMalloc(&c->d_tmp, c->tmp_bytes ? c->tmp_bytes : 16));
    c->keys_cap = c->n;
  }
  return CBH_OK;
}

// scores of nq needles against the whole index -> c->d_scores [nq][n] (enqueued on c->stream); d_raw: also the float
// distances (k_color_dist3 only)
int run_dist(cbh_color* c, const uint8_t* needle_descs, size_t nq, float* d_raw = nullptr) {
  std::vector<NeedleF> nf(nq);
  for (size_t q = 0; q < nq; ++q) decompress(needle_descs + q * kDescBytes, &nf[q]);
  CBH_HIP(hipMemcpyAsync(c->d_needles, nf.data(), nq * sizeof(NeedleF), hipMemcpyHostToDevice, c->stream));
  CBH_HIP(hipStreamSynchronize(c->stream));  // nf is a stack-lifetime buffer
  if (d_raw || g_color_fma) {
    const unsigned tiles = (unsigned)((c->n + 255) / 256);
    dim3 grid((unsigned)nq, std::min(tiles, 32768u), (tiles + 32767u) / 32768u), block(256);
#define CBH_DIST3(RAW_, FMA_)                                                                                       \
  hipLaunchKernelGGL((k_color_dist3<RAW_, FMA_>), grid, block, 0, c->stream, c->dL, c->dU, c->dV, c->d_num, c->cap, \
                     (uint32_t)c->n, c->d_needles, c->d_scores, d_raw)
    if (g_color_fma) {
      if (d_raw) CBH_DIST3(true, true); else CBH_DIST3(false, true);
    } else {
      if (d_raw) CBH_DIST3(true, false); else CBH_DIST3(false, false);
    }
#undef CBH_DIST3
  } else {  // (cap is always even: ensure_cap)
    const unsigned tiles = (unsigned)((c->n + 511) / 512);
    dim3 grid((unsigned)nq, std::min(tiles, 32768u), (tiles + 32767u) / 32768u), block(256);
    hipLaunchKernelGGL(k_color_dist2, grid, block, 0, c->stream, c->dL, c->dU, c->dV, c->d_num, c->cap,
                       (uint32_t)c->n, c->d_needles, c->d_scores);
  }
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace

extern "C" {

cbh_color* cbh_color_create(int device) {
  cbh::clear_last_error();
  if (!cbh::device_usable(device)) return (cbh_color*)cbh::fail_handle(CBH_E_NODEVICE, "cbh_color_create: no usable gfx950 device at that ordinal");
  cbh_color* c = new (std::nothrow) cbh_color;
  if (!c) return (cbh_color*)cbh::fail_handle(CBH_E_NOMEM, "cbh_color_create: host allocation failed");
  c->device = device;
  return c;
}

void cbh_color_destroy(cbh_color* c) {
  if (!c) return;
  cbh::combiner_drop(c);  // combine.hip: the queue of cbh_*_find_coalesced callers
  cbh::DeviceGuard g(c->device);
  for (void* p : {(void*)c->dL, (void*)c->dU, (void*)c->dV, (void*)c->d_num, (void*)c->d_ids, (void*)c->d_needles,
                  (void*)c->d_scores, (void*)c->d_keys, (void*)c->d_keys_alt, c->d_tmp, (void*)c->d_hist,
                  (void*)c->d_thr, (void*)c->d_smin, (void*)c->d_ncand, (void*)c->d_valid, (void*)c->d_cand})
    if (p) (void)hipFree(p);
  if (c->stream) cbh::stream_destroy(c->stream);
  delete c;
}

/* add()/load(): append n (mediaId, ColorDescriptor) entries (colordescindex.cpp:123-168, 201-213) */
int cbh_color_add(cbh_color* c, const uint32_t* ids, const void* descs, size_t n) {
  if (!c || (n && (!ids || !descs))) return CBH_E_INVAL;
  if (n == 0) return CBH_OK;
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  const size_t first = c->n;
  c->host_desc.insert(c->host_desc.end(), (const uint8_t*)descs, (const uint8_t*)descs + n * kDescBytes);
  c->host_ids.insert(c->host_ids.end(), ids, ids + n);
  int rc = grow_index(c, first + n);
  if (rc) return rc;
  c->n = first + n;
  return upload(c, first, n);
}

/* remove(): id 0 and cleared descriptor in place (:215-229) */
int cbh_color_remove(cbh_color* c, const uint32_t* ids, size_t n) {
  if (!c || (n && !ids)) return CBH_E_INVAL;
  if (c->n == 0 || n == 0) return CBH_OK;  // `if (!isLoaded()) return;`
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  std::vector<uint32_t> rm(ids, ids + n);
  std::sort(rm.begin(), rm.end());
  for (size_t i = 0; i < c->n; ++i)
    if (std::binary_search(rm.begin(), rm.end(), c->host_ids[i])) {
      c->host_ids[i] = 0;
      memset(c->host_desc.data() + i * kDescBytes, 0, kDescBytes);
      int rc = upload(c, i, 1);
      if (rc) return rc;
    }
  return CBH_OK;
}

size_t cbh_color_count(const cbh_color* c) { return c ? c->n : 0; }
int cbh_color_is_loaded(const cbh_color* c) { return c && c->n > 0; }  // `_count > 0` (:114)
size_t cbh_color_memory_usage(const cbh_color* c) { return c ? (kDescBytes + 4) * c->n : 0; }  // (:118-121)

/* findIndexData (:231-239): first entry with that id */
int cbh_color_find_index_data(const cbh_color* c, uint32_t id, void* out_desc) {
  if (!c || !out_desc) return CBH_E_INVAL;
  for (size_t i = 0; i < c->n; ++i)
    if (c->host_ids[i] == id) {
      memcpy(out_desc, c->host_desc.data() + i * kDescBytes, kDescBytes);
      return 1;
    }
  return 0;
}

int cbh_color_download(const cbh_color* c, uint32_t* ids, void* descs, size_t cap) {
  if (!c) return CBH_E_INVAL;
  const size_t m = std::min(cap, c->n);
  if (ids) memcpy(ids, c->host_ids.data(), m * 4);
  if (descs) memcpy(descs, c->host_desc.data(), m * kDescBytes);
  return CBH_OK;
}

/* find() (:250-278): every entry with a finite distance and id != 0, index order, score = int(distance) */
int cbh_color_find(cbh_color* c, const void* needle_desc, cbh_match* out, size_t cap, size_t* n_out) {
  if (!c || !needle_desc || !n_out || (cap && !out)) return CBH_E_INVAL;
  *n_out = 0;
  if (((const uint8_t*)needle_desc)[256] == 0 || c->n == 0) return CBH_OK;  // no colours (:259-266)
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  int rc = ensure_scratch(c, 1, false);
  if (rc) return rc;
  rc = run_dist(c, (const uint8_t*)needle_desc, 1);
  if (rc) return rc;
  std::vector<int> sc(c->n);
  CBH_HIP(hipMemcpyAsync(sc.data(), c->d_scores, c->n * 4, hipMemcpyDeviceToHost, c->stream));
  CBH_HIP(hipStreamSynchronize(c->stream));
  size_t m = 0;
  for (size_t i = 0; i < c->n; ++i)
    if (sc[i] >= 0 && c->host_ids[i] != 0) {
      if (m < cap) out[m] = cbh_match{c->host_ids[i], sc[i]};
      ++m;
    }
  *n_out = m;
  return CBH_OK;
}

/* find() for nq needles in one pass: needle q's matches (all of them, index order, as cbh_color_find returns them) at
 * out[out_offsets[q] .. out_offsets[q+1]); CBH_E_OVERFLOW with out_offsets complete when cap is too small */
int cbh_color_find_all_batch(cbh_color* c, const void* needle_descs, size_t nq, cbh_match* out, size_t cap,
                             uint64_t* out_offsets) {
  if (!c || !out_offsets || (nq && !needle_descs) || (cap && !out)) return CBH_E_INVAL;
  for (size_t q = 0; q <= nq; ++q) out_offsets[q] = 0;
  if (nq == 0 || c->n == 0) return CBH_OK;
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 27) / c->n));
  int rc = ensure_scratch(c, chunk, false);
  if (rc) return rc;
  std::vector<int> sc(chunk * c->n);
  uint64_t pos = 0;
  for (size_t q0 = 0; q0 < nq; q0 += chunk) {
    const size_t m = std::min(chunk, nq - q0);
    rc = run_dist(c, (const uint8_t*)needle_descs + q0 * kDescBytes, m);
    if (rc) return rc;
    CBH_HIP(hipMemcpyAsync(sc.data(), c->d_scores, m * c->n * 4, hipMemcpyDeviceToHost, c->stream));
    CBH_HIP(hipStreamSynchronize(c->stream));
    for (size_t q = 0; q < m; ++q) {
      out_offsets[q0 + q] = pos;
      if (((const uint8_t*)needle_descs)[(q0 + q) * kDescBytes + 256] == 0) continue;  // no colours (:259-266)
      const int* s = sc.data() + q * c->n;
      for (size_t i = 0; i < c->n; ++i)
        if (s[i] >= 0 && c->host_ids[i] != 0) {
          if (pos < cap) out[pos] = cbh_match{c->host_ids[i], s[i]};
          ++pos;
        }
    }
  }
  out_offsets[nq] = pos;
  return pos > cap ? CBH_E_OVERFLOW : CBH_OK;
}

/* ColorDescriptor::distance (cvutil.cpp:682-749) of nq needles against every index entry, as floats: out[q*n + i];
 * FLT_MAX where the reference returns FLT_MAX (no colours on a side, or the colour counts differ by more than 2).
 * Entry i is the i-th descriptor added, whatever its id. */
int cbh_color_distances(cbh_color* c, const void* needle_descs, size_t nq, float* out) {
  if (!c || (nq && (!needle_descs || !out))) return CBH_E_INVAL;
  if (nq == 0 || c->n == 0) return CBH_OK;
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 27) / c->n));
  int rc = ensure_scratch(c, chunk, false);
  if (rc) return rc;
  float* d_raw = nullptr;
  CBH_HIP(hipMalloc(&d_raw, chunk * c->n * sizeof(float)));
  for (size_t q0 = 0; q0 < nq && !rc; q0 += chunk) {
    const size_t m = std::min(chunk, nq - q0);
    rc = run_dist(c, (const uint8_t*)needle_descs + q0 * kDescBytes, m, d_raw);
    if (rc) break;
    hipError_t e = hipMemcpyAsync(out + q0 * c->n, d_raw, m * c->n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      cbh::set_last_error("color distances", e);
      rc = CBH_E_HIP;
    }
  }
  (void)hipFree(d_raw);
  return rc;
}

/* find() for many needles + the sort/cut of searchIndex (database.cpp:1729-1735): out[q*k..] = first
 * min(counts[q], k) matches in (score, id) order, counts[q] = all matches */
// one needle through the full sort (ties beyond kCandCap, or k > kCandCap)
static int color_full_sort_one(cbh_color* c, size_t q_in_chunk, int k, cbh_match* out_q, uint32_t valid) {
  hipLaunchKernelGGL(k_color_keys, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                     c->d_scores + q_in_chunk * c->n, c->d_ids, (uint32_t)c->n, c->d_keys);
  unsigned long long* sorted = nullptr;
  int rc_s = cbh::sort_keys64_db(c->d_keys, c->d_keys_alt, c->n, 64, c->d_tmp, c->tmp_bytes, c->stream, &sorted);
  if (rc_s) return rc_s;
  const size_t take = std::min<size_t>(std::min<size_t>((size_t)k, c->n), valid);
  std::vector<unsigned long long> head(take);
  if (take) CBH_HIP(hipMemcpyAsync(head.data(), sorted, take * 8, hipMemcpyDeviceToHost, c->stream));
  CBH_HIP(hipStreamSynchronize(c->stream));
  for (size_t j = 0; j < take; ++j) out_q[j] = cbh_match{(uint32_t)head[j], (int32_t)(head[j] >> 32)};
  return CBH_OK;
}

int cbh_color_find_batch(cbh_color* c, const void* needle_descs, size_t nq, int k, cbh_match* out,
                         uint32_t* counts) {
  if (!c || k < 0 || (nq && (!needle_descs || !counts || (k && !out)))) return CBH_E_INVAL;
  if (nq == 0) return CBH_OK;
  memset(counts, 0, nq * 4);
  if (k) memset(out, 0, nq * (size_t)k * sizeof(cbh_match));
  if (c->n == 0) return CBH_OK;
  cbh::DeviceGuard g(c->device);
  if (!g.ok) return CBH_E_NODEVICE;
  std::lock_guard<std::mutex> lk(c->mu);
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(nq, ((size_t)1 << 28) / c->n));  // <= 1 GiB of scores
  int rc = ensure_scratch(c, chunk, true);
  if (rc) return rc;
  if (chunk > c->topk_cap) {
    for (void* p : {(void*)c->d_hist, (void*)c->d_thr, (void*)c->d_smin, (void*)c->d_ncand, (void*)c->d_valid,
                    (void*)c->d_cand})
      if (p) (void)hipFree(p);
    c->d_hist = c->d_ncand = c->d_valid = nullptr;
    c->d_thr = c->d_smin = nullptr;
    c->d_cand = nullptr;
    c->topk_cap = 0;
    CBH_HIP(hipMalloc(&c->d_hist, chunk * kWin * 4));
    CBH_HIP(hipMalloc(&c->d_thr, chunk * 4));
    CBH_HIP(hipMalloc(&c->d_smin, chunk * 4));
    CBH_HIP(hipMalloc(&c->d_ncand, chunk * 4));
    CBH_HIP(hipMalloc(&c->d_valid, chunk * 4));
    CBH_HIP(hipMalloc(&c->d_cand, chunk * (size_t)kCandCap * 8));
    c->topk_cap = chunk;
  }
  const bool topk = k >= 1 && (uint32_t)k <= kCandCap;
  std::vector<uint32_t> h_valid, h_ncand;
  std::vector<unsigned long long> h_cand;
  for (size_t q0 = 0; q0 < nq; q0 += chunk) {
    const size_t m = std::min(chunk, nq - q0);
    rc = run_dist(c, (const uint8_t*)needle_descs + q0 * kDescBytes, m);
    if (rc) return rc;
    // number of matches and, exactly, the k-th smallest score of every needle
    CBH_HIP(hipMemsetAsync(c->d_hist, 0, m * kWin * 4, c->stream));
    CBH_HIP(hipMemsetAsync(c->d_valid, 0, m * 4, c->stream));
    CBH_HIP(hipMemsetAsync(c->d_ncand, 0, m * 4, c->stream));
    CBH_HIP(hipMemsetAsync(c->d_smin, 0x7f, m * 4, c->stream));  // 0x7f7f7f7f: above every score
    const unsigned gx = (unsigned)std::min<size_t>((c->n + 255) / 256, 128);
    hipLaunchKernelGGL(k_color_min, dim3(gx, (unsigned)m), dim3(256), 0, c->stream, c->d_scores, c->d_ids,
                       (uint32_t)c->n, c->d_smin);
    hipLaunchKernelGGL(k_color_hist, dim3(gx, (unsigned)m), dim3(256), 0, c->stream, c->d_scores, c->d_ids,
                       (uint32_t)c->n, c->d_smin, c->d_hist, c->d_valid);
    h_valid.resize(m);
    CBH_HIP(hipMemcpyAsync(h_valid.data(), c->d_valid, m * 4, hipMemcpyDeviceToHost, c->stream));
    if (topk) {
      hipLaunchKernelGGL(k_color_thresh, dim3((unsigned)m), dim3(256), 0, c->stream, c->d_hist, (uint32_t)k,
                         c->d_smin, c->d_thr);
      hipLaunchKernelGGL(k_color_collect, dim3(gx, (unsigned)m), dim3(256), 0, c->stream, c->d_scores, c->d_ids,
                         (uint32_t)c->n, c->d_thr, c->d_cand, c->d_ncand);
      h_ncand.resize(m);
      h_cand.resize(m * (size_t)kCandCap);
      CBH_HIP(hipMemcpyAsync(h_ncand.data(), c->d_ncand, m * 4, hipMemcpyDeviceToHost, c->stream));
      CBH_HIP(hipGetLastError());
      CBH_HIP(hipStreamSynchronize(c->stream));
      for (size_t q = 0; q < m; ++q)  // only the filled part of every candidate list
        if (h_ncand[q] && h_ncand[q] <= kCandCap)
          CBH_HIP(hipMemcpyAsync(h_cand.data() + q * kCandCap, c->d_cand + q * kCandCap, (size_t)h_ncand[q] * 8,
                                 hipMemcpyDeviceToHost, c->stream));
    }
    CBH_HIP(hipStreamSynchronize(c->stream));
    for (size_t q = 0; q < m; ++q) {
      counts[q0 + q] = h_valid[q];
      if (k == 0 || h_valid[q] == 0) continue;
      cbh_match* oq = out + (q0 + q) * (size_t)k;
      if (topk && h_ncand[q] <= kCandCap) {
        unsigned long long* cq = h_cand.data() + q * kCandCap;
        std::sort(cq, cq + h_ncand[q]);  // (score, id): Database::searchIndex order with ties by id
        const size_t take = std::min<size_t>((size_t)k, h_ncand[q]);
        for (size_t j = 0; j < take; ++j) oq[j] = cbh_match{(uint32_t)cq[j], (int32_t)(cq[j] >> 32)};
      } else {
        rc = color_full_sort_one(c, q, k, oq, h_valid[q]);
        if (rc) return rc;
      }
    }
  }
  return CBH_OK;
}

}  // extern "C"
