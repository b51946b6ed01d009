// dcthash.hip -- K1+K2: batched 64-bit DCT perceptual hash for gfx950 (CDNA4).
//
// Replaces dctHash64(const cv::Mat&, bool) (src/cvutil.cpp:435-545) for 8UC1 input:
//   stage 1  box blur, kernel 0/3/5/7 chosen by input area (:446-455), cv::blur semantics:
//            centre anchor, BORDER_REFLECT_101, u8 result = nearest(sum / k^2)        (:463)
//   stage 2  cv::resize(->32x32, INTER_AREA), integer-ratio path: block sum, 2x2 -> (s+2)>>2,
//            otherwise rint_even(float(s) * (1.f/area))                                  (:471)
//   stage 3  f32 32x32 DCT-II, only the 9x9 low-frequency block is needed                (:475-482)
//   stage 4  zig-zag order, keep positions 6..69                                         (:491-513)
//   stage 5  threshold = float(sum in double of the 64 coefficients) / 64               (:528-529)
//   stage 6  bit i (1..63) = coef[i] > threshold; hash 0 -> 1                            (:537-542)
//
// Bit-exact contract with oracle/cbird_oracle.c: stages 1-2 are integer (plus one exactly
// specified f32 multiply + round-to-nearest-even), stage 3 is the canonical separable form
//   T[r][k] = sum_j fmaf(X[r][j], C[k][j], .)  (j ascending, start 0.0f)
//   Y[u][k] = sum_r fmaf(C[u][r], T[r][k], .)  (r ascending, start 0.0f)
// with C[k][j] = f32(sqrt((k?2:1)/32) * cos(pi*(2j+1)*k/64)) supplied by the host, stage 5 is a
// sequential f64 sum in zig-zag order.  Every output element is produced by one thread in that
// fixed order, so GPU == CPU restatement bit for bit.
//
// This file: whole images -- k_dcthash_256_band / k_dcthash_256 (256 x 256), k_band_area (fractional ratios up to 1920
// columns, on the matrix cores), k_blur_area_regs / k_blur_area (+ k_tile_hash, k_tiles_hash2) for everything else, and
// launch_dcthash, which picks among them.  Rectangles, keypoint squares and the sides-below-32 corner: kphash.hip.
#include "dcthash_common.h"

namespace cbh {
namespace {
// ---------------------------------------------------------------------------------------------
// k_dcthash_256: the BASELINE configuration (256x256 tiles: 7x7 blur, 8x8 area mean).
//
// Work split: a 32-lane half-wave owns one image; lane l owns pixels [8l, 8l+8) of every row
// (one 8-pixel-wide output column), so a 256-thread workgroup hashes 8 images and every row
// load is a fully coalesced 256-B segment per half-wave (global_load_dwordx2 per lane).
// Per row and lane (44 VALU ops for 8 pixels, all integer, exact):
//   neighbours  2x ds_bpermute (LDS crossbar, no VALU) + 2x v_perm_b32 with a per-lane selector
//               that turns the two border lanes' halo into the REFLECT_101 mirror of their own
//               pixels;
//   horizontal  7-tap sums with v_dot4_u32_u8 against 0/1 byte masks (14 ops for 8 outputs);
//   vertical    7-row sliding sum on u16 pairs packed in u32 (no field ever borrows/overflows:
//               7*7*255 + 24 < 2^16): S += H(new) - H(7 rows ago), ring of 7 rows in VGPRs;
//   divide      nearest(S/49) = ((S+24) * 342393) >> 24 exactly for S <= 12495 (the +24 lives in
//               the running sum), accumulated over the 8 rows of an output cell; 8 columns of the
//               cell are this lane's 8 pixels, so the lane ends up with the 8x8 block sum and
//               rounds it half-to-even (/64) into the 32x32 tile in LDS.
// The 262 "virtual" rows -3..258 are mapped through REFLECT_101 so top/bottom borders need no
// special code; row loads run 7 rows ahead of use.
// Stages 3-6 then run per half-wave on its own tile (row pass: lane = tile row, basis values as
// wave-uniform SGPR operands; column pass and threshold through LDS) in the oracle's fma order.

// acc + (p >> 24) in one VALU op (SDWA byte select); hipcc otherwise emits shift + add
__device__ __forceinline__ unsigned add_byte3(unsigned acc, unsigned p) {
  unsigned r;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
      : "=v"(r)
      : "v"(p), "v"(acc));
  return r;
}

// nearest(S/49) is formed and summed over the 8x8 cell as an integer: ((S+24) * 342393) >> 24 with v_mul_u32_u24_sdwa +
// v_add_u32_sdwa BYTE_3 -- two "complex" VALU ops per pixel.  (Float forms -- a magic-number fma per pixel, and the one
// fma per pixel on 0x4B000000 + S that k_dcthash_256_band uses -- were built for THIS kernel too: exact, and 4 % slower
// here: simple ops cost ~2.15 cycles per wave, complex ~4.3, and they add.  r05's "hash_div" 1-3.)
template <bool DUMP>
__global__ __launch_bounds__(kThreads) void k_dcthash_256(
    const unsigned char* __restrict__ imgs, unsigned n, unsigned row_stride, unsigned img_stride,
    const DctTables* __restrict__ tabs, uint64_t* __restrict__ out,
    unsigned char* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char sTile[8][1024];
  __shared__ float sT[8][288];
  __shared__ float sY[8][84];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int l32 = lane & 31;
  const int slot = tid >> 5;  // image within the workgroup, 0..7

  const unsigned first = blockIdx.x * 8u;
  unsigned img = first + (unsigned)slot;
  const bool valid = img < n;
  if (!valid) img = n - 1;
  const unsigned char* __restrict__ base = imgs + (size_t)first * img_stride;  // wave-uniform
  const unsigned voff = (img - first) * img_stride + (unsigned)l32 * 8u;       // per lane

  // halo exchange: left neighbour's pixels 4..7, right neighbour's pixels 0..3
  const int addrL = ((l32 == 0 ? lane : lane - 1)) << 2;
  const int addrR = ((l32 == 31 ? lane : lane + 1)) << 2;
  // v_perm_b32(S0,S1,sel): selector 4..7 -> S0 byte 0..3, 0..3 -> S1 byte 0..3
  const unsigned selL = l32 == 0 ? 0x01020300u : 0x07060504u;   // lane 0: (x,px3,px2,px1)
  const unsigned selR = l32 == 31 ? 0x00000102u : 0x07060504u;  // lane 31: (px254,px253,px252,x)

  uint2 raw[7];
  unsigned ring[7][4];
  unsigned S[4];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
#pragma unroll
    for (int c = 0; c < 4; ++c) ring[j][c] = 0u;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) S[c] = 24u | (24u << 16);
  unsigned acc = 0;

  // virtual row s-3 -> REFLECT_101 source row; steps past the image (s > 261) re-read row 252,
  // their sums never reach the tile
  auto row_off = [&](int s) -> unsigned {
    int v = s - 3;
    v = v < 0 ? -v : v;
    v = v > 255 ? 510 - v : v;
    v = v < 0 ? 0 : v;
    return (unsigned)v * row_stride + voff;  // 32-bit lane offset from the uniform base
  };
  auto halo = [&](const uint2 d, unsigned& DL, unsigned& DR) {
    const unsigned bl = (unsigned)__builtin_amdgcn_ds_bpermute(addrL, (int)d.y);
    const unsigned br = (unsigned)__builtin_amdgcn_ds_bpermute(addrR, (int)d.x);
    DL = __builtin_amdgcn_perm(bl, d.x, selL);  // bytes 1..3 = px -3,-2,-1
    DR = __builtin_amdgcn_perm(br, d.y, selR);  // bytes 0..2 = px 8,9,10
  };
#pragma unroll
  for (int j = 0; j < 7; ++j) raw[j] = *reinterpret_cast<const uint2*>(base + row_off(j));
  unsigned DLn, DRn;  // halo of the row consumed by the next step (exchange runs one row ahead)
  halo(raw[0], DLn, DRn);

  for (int s0 = 0; s0 < 266; s0 += 7) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int s = s0 + j;
      const unsigned D0 = raw[j].x, D1 = raw[j].y;
      const unsigned DL = DLn, DR = DRn;
      raw[j] = *reinterpret_cast<const uint2*>(base + row_off(s + 7));
      halo(raw[(j + 1) % 7], DLn, DRn);
      const unsigned T0 = udot4(D0, 0x01010101u, 0u);
      const unsigned T1 = udot4(D1, 0x01010101u, 0u);
      const unsigned H0 = udot4(DL, 0x01010100u, T0);
      const unsigned H1 = udot4(DL, 0x01010000u, udot4(D1, 0x00000001u, T0));
      const unsigned H2 = udot4(DL, 0x01000000u, udot4(D1, 0x00000101u, T0));
      const unsigned H3 = udot4(D1, 0x00010101u, T0);
      const unsigned H4 = udot4(D0, 0x01010100u, T1);
      const unsigned H5 = udot4(D0, 0x01010000u, udot4(DR, 0x00000001u, T1));
      const unsigned H6 = udot4(D0, 0x01000000u, udot4(DR, 0x00000101u, T1));
      const unsigned H7 = udot4(DR, 0x00010101u, T1);
      const unsigned P[4] = {H0 | (H1 << 16), H2 | (H3 << 16), H4 | (H5 << 16), H6 | (H7 << 16)};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        S[c] = (S[c] - ring[j][c]) + P[c];
        ring[j][c] = P[c];
      }
      // output row y = s - 6 (garbage for s < 6: acc is reset before the first real row)
      if (s == 6) acc = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc = add_byte3(acc, (S[c] & 0xffffu) * 342393u);  // operands < 2^24 -> v_mul_u32_u24
        acc = add_byte3(acc, (S[c] >> 16) * 342393u);
      }
      const int y = s - 6;
      if (y >= 0 && (y & 7) == 7) {
        const unsigned t = (acc + 31u + ((acc >> 6) & 1u)) >> 6;  // /64, half to even
        if (y < 256) sTile[slot][(y >> 3) * 32 + l32] = (unsigned char)t;
        acc = 0;
      }
    }
  }
  __syncthreads();
  if (DUMP) {  // stage-level parity aid: the 32x32 tile after blur + area resize
    if (valid)
      for (int i = l32; i < 256; i += 32)
        reinterpret_cast<unsigned*>(tiles + (size_t)img * 1024)[i] =
            reinterpret_cast<const unsigned*>(sTile[slot])[i];
  }

  // ---- stages 3-6 per half-wave on its own tile
  const unsigned long long hv = hash_halfwave(sTile[slot], sT[slot], sY[slot], tabs, lane);
  if (l32 == 0 && valid) out[img] = hv;
}

// (Round 2's k_dcthash_256_mfma -- both box passes as f16 MFMAs, operands converted on the VALU -- was removed in round 4:
//  bit-identical but slower than k_dcthash_256 and k_dcthash_256_band; NOTES.md section 2 keeps its description.)
// ---------------------------------------------------------------------------------------------
// k_dcthash_256_band: 256x256 tiles with the 7x7 box filter on the matrix cores -- its horizontal half as the product
// with a band matrix, its vertical half as the accumulation of that product over the rows -- and half a fused
// multiply-add per pixel left for the VALU (k_dcthash_256 spends 5.9 VALU instructions per pixel and is bound by their
// issue).
//
// One wave = four images, walked top to bottom in lockstep.  v_mfma_i32_16x16x64_i8 computes D = A B + C with M = 16 =
// 4 images x 4 column STRIPS of 64 columns (one image row at a time), N = 16 output columns (tile c of every strip) and
// K = 64 = 32 source columns of the row itself (weights +1 on the 7 taps of column n) followed by the same 32 columns of
// the row SEVEN ABOVE it (weights -1): A B is the horizontal 7-tap sum of row u minus that of row u - 7, which is exactly
// what a 7-row sliding vertical sum needs -- S(u) = S(u - 1) + A B -- and with C = the accumulator of the previous row
// the matrix core adds it up itself: the same register meets the same column again on the next row.  (Through round 6's
// first half the M rows were 4 images x 4 consecutive ROWS: a register held the delta of one row and S cost one v_add per
// pixel, 128 of the loop's 339 vector instructions per eight rows in a kernel bound by its waves' instruction streams --
// NOTES 13.5e; 5.45 -> 5.8 TB/s.)  The pixels enter as p - 128 (one v_xor per four pixels when a row is staged; the bias
// cancels in the difference, the seven warm-up rows leave a constant -6272 that the initial S holds).
//   borders    a strip's first / last tile is an image edge only in strips 0 / 3, but B is shared by the sixteen M rows:
//              REFLECT_101's three columns are WRITTEN into the ring's halo bytes when a row is staged (one v_perm + one
//              ds_write_b32 per row and edge lane) and every tile takes the interior band matrix; top / bottom: the row
//              addresses.
//   division   S lives as the integer 0x4B000000 + S, i.e. the float 2^23 + S (i32 accumulation in the matrix core is
//              exact).  With c = 42799 * 2^-21 the product (2^23 + S) * c = 171196 + S * c is exact inside an fma, and
//              added to an integer-valued accumulator below 2^24 the one rounding is to the nearest integer = nearest(S / 49)
//              (S * c stays 0.0100 away from a tie for every S <= 12495; tests/test_golden_hash_stages.py).  v_pk_fma_f32
//              does two strips per instruction.
//   cell       a lane's accumulator component holds ONE column over the eight rows of a cell; two strips share a register
//              as 16-bit halves through three DPP steps over the cell's eight lanes, /64 half-to-even, one byte to the tile.
// Rows travel global -> registers (coalesced 16-byte loads, two steps of four rows ahead) -> LDS ring of 12 rows per
// image (272-byte pitch: 8 halo bytes each side, so that the K block of a tile starts 16-byte aligned; 3296 bytes between
// the images' planes, which puts the four images x four strips of a lane group's ds_read_b128 on sixteen different
// 16-byte slots of the 64 banks) -> one ds_read_b128 per tile and row in the A-operand layout.  A wave is its own
// workgroup: no barriers in the loop; 17 KB of LDS per wave, 9 waves per CU.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
// the ring is written as pairs of dwords and read back as int4 operands (and reused as floats by the tail): accesses
// that alias by design, so the compiler must not order them by type
typedef int v4i_lds __attribute__((ext_vector_type(4), may_alias));
typedef unsigned int v2u_lds __attribute__((ext_vector_type(2), may_alias));
typedef unsigned int v4u_lds __attribute__((ext_vector_type(4), may_alias));

struct BandTables {
  unsigned int w[1][64][4];  // the B operand (i8 x 16 per lane) of an interior column tile: the only one (see "borders")
};

constexpr int kBandPitch = 272;  // 8 + 256 + 8

// a one-wave workgroup orders its own LDS accesses in hardware; only the compiler must not move them across the phases
__device__ __forceinline__ void wave_order_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

template <bool DUMP>
__global__ __launch_bounds__(64) void k_dcthash_256_band(
    const unsigned char* __restrict__ imgs, unsigned n, unsigned row_stride, unsigned img_stride,
    const DctTables* __restrict__ tabs, const BandTables* __restrict__ bt, uint64_t* __restrict__ out,
    unsigned char* __restrict__ tiles) {
  constexpr int kSlots = 3, kRing = 4 * kSlots;
  constexpr int kImg = kRing * kBandPitch + 32;  // bytes per image plane (see above)
  __shared__ __attribute__((aligned(16))) unsigned char sRing[4 * kImg];  // 13 184 B; the tail reuses it
  __shared__ __attribute__((aligned(16))) unsigned char sTile[4][1024];
  const int lane = threadIdx.x & 63;
  const int n16 = lane & 15, q = lane >> 4;
  const unsigned first = blockIdx.x * 4u;
  // staging role: image q, 16-byte chunk n16 of a row
  unsigned mine = first + (unsigned)q;
  if (mine >= n) mine = n - 1;
  const unsigned char* __restrict__ base = imgs + (size_t)first * img_stride;  // workgroup-uniform
  const unsigned voff = (mine - first) * img_stride + (unsigned)n16 * 16u;
  const int wr_base = q * kImg + 8 + 16 * n16;
  // ... and the halo: chunk 0 mirrors columns 1..3 into columns -1..-3 (ring bytes 5..7 of the row), chunk 15 columns
  // 252..254 into 258..256 (bytes 264..266); the fourth byte of either dword is a column the band gives no weight
  const bool edge = n16 == 0 || n16 == 15;
  const unsigned edge_sel = n16 == 0 ? 0x01020300u : 0x00000102u;
  const int edge_base = q * kImg + (n16 == 0 ? 4 : 264);
  // A-operand role: M row n16 = image n16 >> 2, strip n16 & 3; K chunk q: 0, 1 the row, 2, 3 the row seven above it
  const int rd_base = (n16 >> 2) * kImg + (n16 & 3) * 64 + (q & 1) * 16;
  const int rd_off = q >= 2 ? kRing - 7 : 0;   // ring rows ahead of the step's first
  const int rd_wrap = q >= 2 ? kRing * kBandPitch : 0;
  const v4i_t b0 = *reinterpret_cast<const v4i_t*>(bt->w[0][lane]);

  for (int i = threadIdx.x; i < 4 * kImg / 16; i += 64)
    reinterpret_cast<v4u_lds*>(sRing)[i] = v4u_lds{0u, 0u, 0u, 0u};  // rows above the image: p - 128 = 0 contributes nothing

  // virtual row w = 0..263 is image row reflect101(w - 5); output row y = w - 8 is complete with row w
  auto src_off = [&](int w) -> unsigned {
    int v = w - 5;
    v = v < 0 ? -v : v;
    v = v > 255 ? 510 - v : v;
    return (unsigned)v * row_stride + voff;
  };
  uint4 stg[2][4];
  auto load_step = [&](int t, uint4 (&dst)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = *reinterpret_cast<const uint4*>(base + src_off(4 * t + r));
  };
  auto store_step = [&](int ts, const uint4 (&src)[4]) {  // ts = t % kSlots: the step's rows take slots 4 * ts .. + 3
    unsigned ev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned x = src[r].x ^ 0x80808080u, y = src[r].y ^ 0x80808080u, z = src[r].z ^ 0x80808080u,
                     w = src[r].w ^ 0x80808080u;
      v2u_lds* p = reinterpret_cast<v2u_lds*>(sRing + wr_base + (4 * ts + r) * kBandPitch);  // 8-byte aligned
      p[0] = v2u_lds{x, y};
      p[1] = v2u_lds{z, w};
      const unsigned e = n16 == 0 ? x : w;
      ev[r] = __builtin_amdgcn_perm(e, e, edge_sel);
    }
    if (edge) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<unsigned __attribute__((may_alias))*>(sRing + edge_base + (4 * ts + r) * kBandPitch) = ev[r];
    }
  };

  // S[c][s]: 0x4B000000 + the 7 x 7 box sum at (image q, strip s, tile c, column n16) of the newest row
  v4i_t S[4];
  v2f_t f01[4], f23[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    S[c] = v4i_t{(int)(0x4B000000u + 6272u), (int)(0x4B000000u + 6272u), (int)(0x4B000000u + 6272u),
                 (int)(0x4B000000u + 6272u)};
    f01[c] = f23[c] = v2f_t{8388608.0f, 8388608.0f};
  }
  v2f_t kC = {42799.0f / 2097152.0f, 42799.0f / 2097152.0f};
  asm volatile("" : "+v"(kC));  // VGPR operands (a literal would double the size of every fma)
  const v2f_t kInit = {8388608.0f, 8388608.0f};
  // tile byte of this lane within a cell row: even lanes carry strips 0 / 2, odd lanes strips 1 / 3 (8 cells each)
  const int st_lane = q * 1024 + (n16 >> 3) + 8 * (n16 & 1);
  const unsigned sel_shift = (unsigned)(n16 & 1) * 16u;

  // one step: rows 4t .. 4t+3 (already in the ring); FIRST = the step opens a cell row, else it closes it
  auto step = [&](auto first_tag, int t, int ts) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // ring row of the lane's K chunk for the step's first row; only the row seven above the LAST row of slot 1 wraps
    int row0 = 4 * ts + rd_off;
    row0 = row0 >= kRing ? row0 - kRing : row0;
    const unsigned char* arow = sRing + rd_base + row0 * kBandPitch;
    const unsigned char* arow3 = arow + 3 * kBandPitch - (ts == 1 ? rd_wrap : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned char* ar = r == 3 ? arow3 : arow + r * kBandPitch;
      v4i_t a[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const v4i_lds*>(ar + 16 * c);
#pragma unroll
      for (int c = 0; c < 4; ++c) S[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[c], b0, S[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // (by value: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whichever one is named)
        const int s0 = S[c][0], s1 = S[c][1], s2 = S[c][2], s3 = S[c][3];
        const v2f_t p01 = {__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1)};
        const v2f_t p23 = {__builtin_bit_cast(float, s2), __builtin_bit_cast(float, s3)};
        f01[c] = __builtin_elementwise_fma(p01, kC, (FIRST && r == 0) ? kInit : f01[c]);
        f23[c] = __builtin_elementwise_fma(p23, kC, (FIRST && r == 0) ? kInit : f23[c]);
      }
    }
    if constexpr (!FIRST) {
      const int a_row = (t - 3) >> 1;  // cell row closed by this step (negative during the warm-up)
      if (a_row >= 0) {
        unsigned char* dst = &sTile[0][0] + st_lane + a_row * 32;
        // every accumulator component: 2^23 + 8 x 171196 + the eight quotients of its column
        constexpr unsigned kBias = 0x4B000000u + 8u * 171196u;
        constexpr unsigned kBias2 = kBias + (kBias << 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            // (by value: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whichever one is named)
            const float fx = h ? f23[c].x : f01[c].x, fy = h ? f23[c].y : f01[c].y;
            unsigned wv2 = __float_as_uint(fx) + (__float_as_uint(fy) << 16) - kBias2;  // <= 2040 each; <= 16320 after the 8 lanes
            wv2 += (unsigned)__builtin_amdgcn_mov_dpp((int)wv2, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
            wv2 += (unsigned)__builtin_amdgcn_mov_dpp((int)wv2, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
            wv2 += (unsigned)__builtin_amdgcn_mov_dpp((int)wv2, 0x141, 0xf, 0xf, true);  // row_half_mirror
            const unsigned cell = (wv2 >> sel_shift) & 0xffffu;  // even lanes: strip 2h, odd lanes: strip 2h + 1
            dst[16 * h + 2 * c] = (unsigned char)((cell + 31u + ((cell >> 6) & 1u)) >> 6);  // /64, half to even
          }
        }
      }
    }
  };

  using std::integral_constant;
  load_step(0, stg[0]);
  load_step(1, stg[1]);
  __syncthreads();  // (the zero fill)
  store_step(0, stg[0]);
  load_step(2, stg[0]);
  int ts = 0;  // t % kSlots of the even step
  auto nxt = [](int v) { return v == kSlots - 1 ? 0 : v + 1; };
  for (int t = 0; t < 66; t += 2) {
    const int tsa = ts, tsb = nxt(tsa), tsc = nxt(tsb);
    wave_order_lds();
    step(integral_constant<bool, true>{}, t, tsa);
    store_step(tsb, stg[1]);  // rows of step t + 1: their slots' last readers ran just now
    load_step(min(t + 3, 65), stg[1]);  // (unconditional: see k_band_area's loop)
    wave_order_lds();
    step(integral_constant<bool, false>{}, t + 1, tsb);
    if (t + 2 < 66) store_step(tsc, stg[0]);
    load_step(min(t + 4, 65), stg[0]);
    ts = tsc;
  }
  __syncthreads();
  if (DUMP) {
    for (int g = 0; g < 4; ++g)
      if (first + (unsigned)g < n)
        for (int i = threadIdx.x; i < 256; i += 64)
          reinterpret_cast<unsigned*>(tiles + (size_t)(first + (unsigned)g) * 1024)[i] =
              reinterpret_cast<const unsigned*>(sTile[g])[i];
  }
  // ---- stages 3-6 as in k_dcthash_256: a half-wave per image, two images at a time, two passes; sT / sY live in the ring
  f32_lds* sT = reinterpret_cast<f32_lds*>(sRing);                // [2][288]
  f32_lds* sY = reinterpret_cast<f32_lds*>(sRing) + 2 * 288;      // [2][84]
  const int hw = lane >> 5;
  for (int pass = 0; pass < 2; ++pass) {
    const int slot = 2 * pass + hw;
    const unsigned img = first + (unsigned)slot;
    __syncthreads();
    const unsigned long long hv = hash_halfwave(sTile[slot], sT + hw * 288, sY + hw * 84, tabs, lane);
    if ((lane & 31) == 0 && img < n) out[img] = hv;
  }
}

// ---------------------------------------------------------------------------------------------
// Any other size (w,h >= 32, not both multiples of 32): cv::resize's general INTER_AREA path
// (resizeArea_) with fractional cell weights, in the reference's float accumulation order -- per source row
// buf += S[sx]*alpha over the x table, per output row sum = beta*buf then += beta*buf over the y table, round-half-even.

// ---------------------------------------------------------------------------------------------
// k_band_area (round 5): k_dcthash_256_band's matrix-core blur for ANY width and height, feeding cv::resize's weighted
// INTER_AREA path (fractional ratios, resizeArea_) in the reference's float chain order.
//
// One wave = four images x one COLUMN STRIP of up to 16 output cells (<= 240 source columns), walked top to bottom in
// lockstep, four rows per step, like the 256 x 256 kernel: rows travel global -> registers (two steps ahead) -> a 12-row
// LDS ring per image -> one ds_read_b128 per 16-column tile in the A-operand layout; v_mfma_i32_16x16x64_i8 against a
// band matrix (K = 32 columns of the row, +1 on the seven taps, ++ the same columns of the row seven above, -1) gives
// D = Hsum(u) - Hsum(u - 7); S += D is the 7 x 7 box sum, one fma turns 2^23 + S into 2^23 + const + nearest(S / 49)
// (k_dcthash_256_band's exact division, the constant chosen so that the quotient IS the low byte).  What differs:
//   * the blurred pixels are needed as bytes.  A lane holds four consecutive ROWS of one column: their quotients are
//     packed into one dword, sT[image][x] = rows y .. y + 3 of column x (one plane per image);
//   * horizontal INTER_AREA: lane (image, cell) walks its cell's columns in table order and runs the four rows' chains
//     `sum += float(p) * alpha` as two float pairs (v_cvt_f32_ubyte0..3, v_pk_mul_f32, v_pk_add_f32: ordinary IEEE
//     multiplies and adds, the bits of the scalar chain) -- 2 instructions per pixel, every lane busy when a strip has 16
//     cells, no per-lane trip counts (columns past a cell carry the weight +0.0f);
//   * vertical INTER_AREA in the same lane, which already holds its (image, cell)'s four horizontal sums: the y table seen
//     from the source row (YRow), k_blur_area_regs<.., FUSE>'s chain; finished tile bytes go straight to global memory;
//   * REFLECT_101 left / right is folded into per-strip band matrices made on the host (first tile of the first strip,
//     last two tiles of the last strip), top / bottom into the row addresses.
// Stages 3-6 run from the tiles (k_tiles_hash2).  Against k_blur_area_regs (blur 5.9 + area 2.75 / 0.7 VALU instructions
// per pixel): ~2.5 + ~2.2 -- see DESIGN a1.  Preconditions (launcher): whole images, or views whose vertical edges are
// the parent's or lie >= 8 (left) / >= 3 (right) columns inside it (oy, ph, ox, inner), 7 x 7 blur, fractional ratios,
// strips of at least 4 cells within 240 columns (w <= 1920), four images' strides below 2^32.
struct BaStrip {
  int xs;       // first source column of the strip's first cell
  int T;        // 16-column tiles (<= 15)
  int cell0;    // first output cell
  int ncell;    // cells (<= 16)
  int amax;     // longest cell walk
  int amin;     // shortest: entries 1 .. amin - 2 of EVERY cell weigh its wmid (checked on the host)
  int tp;       // dwords between the images' planes of sT: chosen on the host so that the cells' walks (lane = image, cell
                // reads column si0[cell] + k of its image's plane) meet in as few LDS banks as possible
  int rotm;     // integer ratios with 16 or 32 columns per cell (512, 1024 px): cell width - 1, else 0 -- see rot
  int si0[16];             // first source column of cell c, relative to xs
  // the block-sum walk of cell c starts at its column rot[c] and wraps (a sum has no order): cells a multiple of 16 columns
  // apart would otherwise read the same LDS bank in every step of the walk, 8 lanes a bank at 512 px
  int rot[16];
  // weights of cell c in table order: wfirst, then wmid for entries 1 .. amin - 2, then wtail[j] for entry amin - 1 + j
  // (+0.0f past the cell's end; amax - amin + 1 <= 4 of them)
  float wfirst[16], wmid[16], wtail[16][4];
  unsigned band[3][64][4]; // B operands: the plain band, the strip's first tile, its last tile
};

// Row bands: a small batch has too few (group, strip) pairs to fill the machine (256 frames of 1920 x 1080: 512 waves for 2560
// slots, each walking 270 steps), so the 32 output ROWS are dealt to up to 8 waves per (group, strip) as well: band b makes
// the output rows [c0, c1) and walks the source rows [ra, rb] that feed them (+ two warm-up steps for the blur's seven-row
// window; a source row that straddles two bands' cells is walked by both).
struct BaBands {
  int n;
  int ra[8], rb[8], c0[8], c1[8];
};

// T = 16-column tiles per strip (the same for every strip of a geometry).  RS = rows of a step one lane carries through the
// area walk: 4 (strips of <= 16 cells: lane = image, cell), 2 (<= 8 cells: lane = image, cell, row pair) or 1 (<= 4 cells:
// lane = image, cell, row) -- wide cells mean few cells per 240-column strip, and the rows of a step are then spread over the
// lanes that would idle; the vertical chain passes its running sum from row group to row group by DPP.
// INT = integer resize ratios on both axes (cv::resize's resizeAreaFast_: exact block sums of isx x isy blurred pixels,
// rint(float(sum) * (1.f / area)); 640 x 480, 1024 x 768, 512 x 384 ...; round 6): the same staging, ring and blur, but the
// walk adds bytes -- the four rows of a column dword as two packed 16-bit sums, 4 VALU per column instead of the float
// chains' 8 -- and the vertical pass adds row sums and rounds once per cell.
template <int T, int RS, bool INT>
__global__ __launch_bounds__(64) void k_band_area(const unsigned char* __restrict__ imgs, unsigned n, int w, int h,
                                                  unsigned row_stride, unsigned img_stride, unsigned long long buf_bytes,
                                                  const BaStrip* __restrict__ strips, int n_strips,
                                                  const YRow* __restrict__ yrow, unsigned char* __restrict__ tiles_out,
                                                  int oy, int ph /* a view: rows oy .. oy + h - 1 of images of ph rows
                                                                    (letterboxed frames after autocrop; cv::blur takes
                                                                    its border from the parent); whole images: 0, h */,
                                                  int ox, int inner /* the view's first column in the parent; bit 0 / 1:
                                                                       its left / right edge lies inside the parent (a
                                                                       pillarboxed frame): the blur reads the parent's
                                                                       pixels there instead of mirroring */,
                                                  BaBands bands, int isy /* INT: source rows per output row */) {
  // LDS, sized by T (separate arrays: the compiler must know that they do not alias): ring 4 x 12 rows x kPitch, sT =
  // blurred bytes [image][x], a dword = 4 rows (+ columns for the walk's overhang)
  // A ring row is 16 T + 16 bytes of pixels in a pitch of 16 T + 32.  The blur's ds_read_b128 serves 16 lanes = 4 images x
  // 4 rows per LDS cycle: rows a pitch apart and images 12 pitches apart must fall on different 16-byte slots of the 64
  // banks.  They do for every T but 14 (pitch 256 bytes: rows AND images on one slot, 16 ways -- 448, 608-640, 854-896 px
  // and 1696-1792 px ran at 0.65-0.75 of their neighbours' rate), 6 (rows two ways), 2 and 10 (images two ways).  One more
  // slot per row cures 14 (+35-45 %) and 6 (+1-2 %); at T = 10 the 768 bytes cost a wave of occupancy and 4 % (320 x 240,
  // 1280 x 720), so 10 and 2 keep their two-way conflict.  profiles/r06_band_area_int.txt
  constexpr int kRing = 12, kPitch = 16 * T + 32 + (T == 14 || T == 6 ? 16 : 0), kImg = kRing * kPitch;
  __shared__ __attribute__((aligned(16))) unsigned char sRing[4 * kImg];
  // sT[image][x]: the area walk has lane (image, cell) read column si0[cell] + k -- cells 28 columns apart would meet in
  // the same banks four ways in an [x][image] layout (900 px: half the speed); per-image planes of kTP = 8 (mod 32) dwords
  // keep the cells of one image and the four images apart.  The walk's weights live in registers (first, mid, tail).
  constexpr int kTP = 16 * T + 8 + 40;  // (allocation; the stride in use is st.tp <= kTP)
  __shared__ __attribute__((aligned(16))) unsigned sT[4 * kTP];
  const int lane = threadIdx.x & 63;
  const int n16 = lane & 15, q = lane >> 4;
  // Workgroup -> (group of four images, strip).  The hardware deals consecutive workgroup ids to the 8 XCDs in turn, each
  // with its own L2: the strips of ONE group overlap by the blur's halo and share the cache lines their boundaries cut
  // (224 of every 400 bytes of a row: 1.4x the bytes when each strip's XCD fetches its own copy), so they get ids that are
  // EQUAL mod 8 -- same XCD, same L2, launched within 8 n_strips ids of each other.
  const unsigned wg = blockIdx.x, ns = (unsigned)n_strips, nsb = ns * (unsigned)bands.n;
  const unsigned blk = wg / (8u * nsb), in_blk = wg % (8u * nsb);
  const unsigned grp = blk * 8u + (in_blk & 7u), sidx = (in_blk >> 3) % ns, bidx = (in_blk >> 3) / ns;
  if (grp * 4u >= n) return;  // (the last block of eight groups may be short)
  const int ra = bands.ra[bidx], rb = bands.rb[bidx], bc0 = bands.c0[bidx], bc1 = bands.c1[bidx];
  const BaStrip& st = strips[sidx];
  const int xs = st.xs, ncell = st.ncell, amax = st.amax, amin = st.amin, tp = st.tp;
  const unsigned first = grp * 4u;
  unsigned mine = first + (unsigned)q;
  if (mine >= n) mine = n - 1;
  // staging role: image q, chunk n16 of a row = columns xs - 8 + 16 n16 .. + 15, ONE 16-byte load per row and lane (raw
  // buffer load: any alignment, no divergent branch around it -- the compiler then tracks the loads' wait counts; two
  // 8-byte halves per lane cost 15-18 % of the streaming rate, tools/ubench/segread.hip).  The one chunk that would start
  // left of column 0 (first strip, n16 = 0) loads columns 0 .. 15 instead and puts them 8 bytes further into the ring row
  // -- over the first half of chunk 1, with the same bytes; the ring's bytes for columns -8 .. -1 stay zero.  A chunk past
  // the row's end reads the next row's first bytes (at the end of the buffer: zeros, by the descriptor's range) -- columns
  // to which the band matrices give no weight, like those left of column 0.
  const unsigned long long base_off = (unsigned long long)first * img_stride;
  const unsigned long long left = buf_bytes > base_off ? buf_bytes - base_off : 0ull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(imgs + base_off), 0, (int)(left > 0xffffffffull ? 0xffffffffu : (unsigned)left), 0x27000);
  // (lanes past the ring row's T + 1 chunks load chunk T's bytes again -- the same cache lines, dropped by store_step --
  // instead of up to 112 columns of the neighbouring strip: T = 8 moved 16 chunks through the texture path for every 9 used)
  const int colA = ox + xs - 8 + 16 * (n16 < T ? n16 : T);  // (an inner left edge has ox >= 8: never negative)
  const unsigned voffA = (mine - first) * img_stride + (unsigned)(colA < 0 ? 0 : colA);
  const int wr_base = q * kImg + 16 * n16 + (colA < 0 ? 8 : 0);
  // A-operand role (k_dcthash_256_band): M row n16 = image n16 >> 2, row n16 & 3 of the step; chunk q: 0, 1 the row, 2, 3
  // the row seven above
  const int rd_base = (n16 >> 2) * kImg + (q & 1) * 16;
  const int rd_row = (n16 & 3) + (q >= 2 ? kRing - 7 : 0);
  // band matrices: interior, and those of the strip's first and last tile (the image's left edge in the first strip, its
  // right edge in the last -- whose tiles end exactly at column w - 1 -- and the interior band everywhere else)
  const v4i_t b0 = *reinterpret_cast<const v4i_t*>(st.band[0][lane]);
  const v4i_t bF = *reinterpret_cast<const v4i_t*>(st.band[(inner & 1) ? 0 : 1][lane]);
  const v4i_t bL = *reinterpret_cast<const v4i_t*>(st.band[(inner & 2) ? 0 : 2][lane]);
  for (int i = lane; i < 4 * kImg / 16; i += 64) reinterpret_cast<v4u_lds*>(sRing)[i] = v4u_lds{0u, 0u, 0u, 0u};
  for (int i = lane; i < 4 * kTP; i += 64) sT[i] = 0u;
  // virtual row v = 0 .. is row reflect101(oy + ra + v - 5) of the parent; blurred row y = ra + v - 8 is complete with row v
  const int h2 = 2 * (ph - 1);
  auto row_off = [&](int v) -> unsigned {
    int ry = oy + ra + v - 5;
    ry = ry < 0 ? -ry : ry;
    ry = min(ry, h2 - ry);
    ry = max(ry, 0);
    return (unsigned)ry * row_stride;
  };
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  v4u_t stg[2][4];
  // (one code path: with a uniform branch around a faster form for the steps that touch neither the parent's top nor its
  // bottom -- one offset and three additions instead of four reflections -- the compiler waits for ALL loads where the
  // branches join: -4 .. -15 %)
  auto load_step = [&](int t, v4u_t (&a)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffA, (int)row_off(4 * t + r), 0);
  };
  auto store_step = [&](int ts, const v4u_t (&a)[4]) {
    if (n16 > T) return;  // (a ring row holds T + 1 chunks)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v2u_lds* p = reinterpret_cast<v2u_lds*>(sRing + wr_base + (4 * ts + r) * kPitch);  // (8-byte aligned)
      p[0] = v2u_lds{a[r].x ^ 0x80808080u, a[r].y ^ 0x80808080u};
      p[1] = v2u_lds{a[r].z ^ 0x80808080u, a[r].w ^ 0x80808080u};
    }
  };
  unsigned S[T];
#pragma unroll
  for (int c = 0; c < T; ++c) S[c] = 0x4B000000u + 6272u;
  v2f_t kC = {42799.0f / 2097152.0f, 42799.0f / 2097152.0f};
  asm volatile("" : "+v"(kC));
  const v2f_t kInit = {8388608.0f + 68.0f, 8388608.0f + 68.0f};  // 171196 + 68 = 0x29D00: the quotient is the low byte
  const v4i_t zero4 = {0, 0, 0, 0};
  // area / vertical role: lane = image ai, cell ac of the strip, row group rg (rows RS rg .. RS rg + RS - 1 of a step)
  constexpr int G = 4 / RS;
  const int ai = lane >> 4, ac = (lane & 15) / G, rg = (lane & 15) % G;
  const bool alive = ac < ncell && first + (unsigned)ai < n;
  const int asi = st.si0[ac < ncell ? ac : 0];
  const int rotm = INT ? st.rotm : 0, arot = INT ? st.rot[ac < ncell ? ac : 0] : 0;
  const int acc_ = ac < ncell ? ac : 0;
  const float w_first = st.wfirst[acc_], w_mid = st.wmid[acc_];
  const float w_t0 = st.wtail[acc_][0], w_t1 = st.wtail[acc_][1], w_t2 = st.wtail[acc_][2], w_t3 = st.wtail[acc_][3];
  const int ntail = amax - amin + 1;
  unsigned char* __restrict__ tdst = tiles_out + (size_t)(first + (unsigned)ai) * 1024 + (unsigned)(st.cell0 + ac);
  float vsum = 0.f;
  unsigned vacc = 0u;                      // INT: the cell's running block sum
  int ym = 0, cyc = bc0;                   // INT: row within the cell, cell row (uniform; a band starts on a cell boundary)
  const float inv_area = INT ? 1.f / (float)(amin * isy) : 0.f;
  const int steps = (rb - ra + 1 + 3) / 4 + 2;

  auto step = [&](int t, int ts) {
    // the y-table entries of the step's four output rows: scalar loads issued HERE, ahead of the blur, so that their
    // latency is over when the vertical chain wants them (behind the compiler barriers below they came one after the other,
    // each waited for: four scalar-memory round trips per step)
    const int y0 = ra + 4 * (t - 2);  // the step's blurred rows
    YRow yrs[4] = {};
    if constexpr (!INT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) yrs[r] = yrow[y0 + r];  // (the table is padded: -8 <= y0, y0 + 3 <= h + 2)
    }
    // ---- blur: rows 4t .. 4t+3 of the ring -> sT
    int slot = 4 * ts + rd_row;
    slot = slot >= kRing ? slot - kRing : slot;
    const unsigned char* arow = sRing + rd_base + slot * kPitch;
    // all T tiles as one group: the A operands are read, the MFMAs issued back to back, then the VALU work of their
    // results -- neither the LDS round trip nor the MFMA's result latency is waited for tile by tile (tile by tile the
    // compiler reuses two register sets and fills the MFMA's latency with s_nop: 120 instead of 154 VGPRs, neither of which
    // limits the 10 waves per CU the LDS allows; +3-10 % at 320 x 240, 800 x 600, 960 x 540, +-1 % elsewhere:
    // profiles/r06_band_area_blur_groups.txt)
    constexpr int GB = T;
#pragma unroll
    for (int c0 = 0; c0 < T; c0 += GB) {
      v4i_t a[GB], d[GB];
#pragma unroll
      for (int g = 0; g < GB; ++g)
        if (c0 + g < T) a[g] = *reinterpret_cast<const v4i_lds*>(arow + 16 * (c0 + g));
#pragma unroll
      for (int g = 0; g < GB; ++g)
        if (c0 + g < T) {
          const int c = c0 + g;
          d[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[g], c == 0 ? bF : c == T - 1 ? bL : b0, zero4, 0, 0, 0);
        }
#pragma unroll
      for (int g = 0; g < GB; ++g)
        if (c0 + g < T) {
          const int c = c0 + g;
          const unsigned x0 = S[c] + (unsigned)d[g][0], x1 = x0 + (unsigned)d[g][1], x2 = x1 + (unsigned)d[g][2],
                         x3 = x2 + (unsigned)d[g][3];
          S[c] = x3;
          const v2f_t p01 = {__builtin_bit_cast(float, x0), __builtin_bit_cast(float, x1)};
          const v2f_t p23 = {__builtin_bit_cast(float, x2), __builtin_bit_cast(float, x3)};
          const v2f_t f01 = __builtin_elementwise_fma(p01, kC, kInit), f23 = __builtin_elementwise_fma(p23, kC, kInit);
          const float fa = f01.x, fb = f01.y, fc = f23.x, fd = f23.y;
          // low bytes of the four quotients -> one dword (rows y .. y + 3 of column 16 c + n16).  (Four ds_write_b8 straight
          // from the fma results would save these three VALU instructions: measured 12 % SLOWER -- 52 LDS instructions per step.)
          const unsigned lo = __builtin_amdgcn_perm(__float_as_uint(fb), __float_as_uint(fa), 0x0c0c0400u);
          const unsigned hi = __builtin_amdgcn_perm(__float_as_uint(fd), __float_as_uint(fc), 0x04000c0cu);
          sT[q * tp + 16 * c + n16] = lo | hi;
        }
    }
    wave_order_lds();  // (the ring rows read above are overwritten by the caller's next store_step)
    if (t < 2) return;           // (uniform) warm-up
    if constexpr (INT) {
      // ---- integer ratios: block sums.  The lane's RS rows of its cell's isx (= amin = amax) columns, exact
      unsigned hs[4] = {0u, 0u, 0u, 0u};
      {
        const unsigned* __restrict__ src = sT + ai * tp + asi;
        const unsigned char* __restrict__ bsrc = reinterpret_cast<const unsigned char*>(src) + RS * rg;
        unsigned s02 = 0u, s13 = 0u;  // 16-bit fields: <= 60 columns x 255
        auto col = [&](int k) {
          if constexpr (RS == 4) {
            const unsigned pw_ = src[k];
            s02 += pw_ & 0x00ff00ffu;                                // rows 0 | 2
            s13 += __builtin_amdgcn_perm(0u, pw_, 0x0c030c01u);      // rows 1 | 3
          } else if constexpr (RS == 2) {
            const unsigned x_ = *reinterpret_cast<const unsigned short*>(bsrc + 4 * k);
            s02 += __builtin_amdgcn_perm(0u, x_, 0x0c010c00u);       // row a | row b << 16
          } else {
            s02 += bsrc[4 * k];
          }
        };
        if (rotm) {  // (uniform)
          int idx = arot;
#pragma unroll 4
          for (int k = 0; k < amin; ++k) {
            col(idx);
            idx = (idx + 1) & rotm;
          }
        } else {
#pragma unroll 4
          for (int k = 0; k < amin; ++k) col(k);
        }
        if constexpr (RS == 4) {
          hs[0] = s02 & 0xffffu, hs[2] = s02 >> 16, hs[1] = s13 & 0xffffu, hs[3] = s13 >> 16;
        } else if constexpr (RS == 2) {
          hs[0] = s02 & 0xffffu, hs[1] = s02 >> 16;
        } else {
          hs[0] = s02;
        }
      }
      wave_order_lds();  // (sT is rewritten by the next step's blur)
      // ---- vertical: the rows of a step in order; ym / cyc (row within the cell, cell row) are uniform and advance with
      // every row, the running sum travels between the row groups' lanes like the float chain's
#pragma unroll
      for (int ph = 0; ph < G; ++ph) {
#pragma unroll
        for (int r = 0; r < RS; ++r) {
          const int y = y0 + ph * RS + r;
          if (y <= rb) {  // (uniform)
            if (G == 1 || rg == ph) vacc += hs[r];
            if (++ym == isy) {
              if (cyc >= bc0 && cyc < bc1 && (G == 1 || rg == ph)) {
                const unsigned qv = (unsigned)__builtin_rintf((float)vacc * inv_area);
                if (alive) tdst[cyc * 32] = (unsigned char)(qv > 255u ? 255u : qv);
              }
              if (G == 1 || rg == ph) vacc = 0u;
              ym = 0;
              ++cyc;
            }
          }
        }
        if constexpr (G > 1) {
          const unsigned up = (unsigned)__builtin_amdgcn_mov_dpp((int)vacc, G == 4 ? 0x93 : 0xB1, 0xf, 0xf, true);
          if (rg == (ph + 1) % G) vacc = up;
        }
      }
      return;
    }
    // ---- horizontal INTER_AREA: the lane's RS rows of cell ac of image ai.  A column of sT is one dword = four rows; a
    // lane that carries two rows (one row) reads just its half (byte) of it -- ds_read_u16 / ds_read_u8 hand back the bytes
    // already shifted into place
    v2f_t acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    {
      const unsigned* __restrict__ src = sT + ai * tp + asi;
      const unsigned char* __restrict__ bsrc = reinterpret_cast<const unsigned char*>(src) + RS * rg;
      auto col = [&](int k, float a_) {
        if constexpr (RS == 4) {
          const unsigned pw_ = src[k];
          const v2f_t w2 = {a_, a_};
          const v2f_t p01 = {(float)(pw_ & 0xffu), (float)((pw_ >> 8) & 0xffu)};
          const v2f_t p23 = {(float)((pw_ >> 16) & 0xffu), (float)(pw_ >> 24)};
          acc01 = acc01 + p01 * w2;
          acc23 = acc23 + p23 * w2;
        } else if constexpr (RS == 2) {
          const unsigned x_ = *reinterpret_cast<const unsigned short*>(bsrc + 4 * k);
          const v2f_t w2 = {a_, a_};
          const v2f_t p01 = {(float)(x_ & 0xffu), (float)(x_ >> 8)};
          acc01 = acc01 + p01 * w2;
        } else {
          acc01.x = acc01.x + (float)bsrc[4 * k] * a_;
        }
      };
      col(0, w_first);
#pragma unroll 4  // (8 or 16 columns' reads in flight instead of 4: +-3 %, noise)
      for (int k = 1; k < amin - 1; ++k) col(k, w_mid);  // interior columns: every lane's cell has them
      {
        const int kt = amin - 1;
        col(kt, w_t0);  // (amin >= 2: entry amin - 1 exists in the shortest cell; longer cells go on)
        if (ntail > 1) col(kt + 1, w_t1);
        if (ntail > 2) col(kt + 2, w_t2);
        if (ntail > 3) col(kt + 3, w_t3);
      }
    }
    wave_order_lds();  // (sT is rewritten by the next step's blur)
    // ---- vertical INTER_AREA: the y table seen from the source row (k_blur_area_regs<.., FUSE>).  The rows of a step in
    // order: row group 0's lanes first, then the running sum moves one lane up (DPP) to row group 1's, ... and from the last
    // group back to group 0 for the next step.
    const float hv[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
    // (Most rows neither close a cell nor feed a second one: for those a multiply, a fused multiply-add and ONE scalar
    // test.  Written as `sum = opens ? t0 : sum + t0` with the store under `closes && cell in band && alive`, every row cost
    // 2 selects' worth of scalar mask arithmetic plus the whole store predicate built in SGPR pairs -- ~17 scalar
    // instructions a row, 237 a step beside 306 vector ones, in a kernel whose waves are too few to hide their own
    // instruction streams.  The empty asm keeps the rare part behind a scalar branch: merged into one predicate it is
    // evaluated on every row again.)
    auto put = [&](int di) {
      if (di >= bc0 && di < bc1) {  // (a cell of another band: its rows are walked there)
        const float rr = __builtin_rintf(vsum);
        if (alive) tdst[di * 32] = (unsigned char)(rr < 0.f ? 0.f : rr > 255.f ? 255.f : rr);
      }
    };
    auto vrow = [&](const YRow& yr, float v) {
      vsum = __builtin_fmaf(vsum, yr.k0, yr.a0 * v);
      if (yr.info & 0x600) {  // (uniform)
        asm volatile("");
        const int di = yr.info & 0xff;
        if (yr.info & 0x200) put(di);
        if (yr.info & 0x400) {
          const float t1 = yr.a1 * v;
          vsum = (yr.info & 0x800) ? t1 : vsum + t1;
          if (yr.info & 0x1000) put(di + 1);
        }
      }
    };
    const int nrow = min(4, rb - y0 + 1);  // rows of the step that belong to the band (4 but for its last step)
#pragma unroll
    for (int ph_ = 0; ph_ < G; ++ph_) {
      if (G == 1 || rg == ph_) {
#pragma unroll
        for (int r = 0; r < RS; ++r)
          if (ph_ * RS + r < nrow) vrow(yrs[ph_ * RS + r], hv[r]);  // (uniform)
      }
      if constexpr (G > 1) {
        // lane j takes the sum of lane j - 1 of its cell's G lanes (wrapping): quad_perm [3,0,1,2] / [1,0,3,2]
        const float up = __builtin_bit_cast(
            float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, vsum), G == 4 ? 0x93 : 0xB1, 0xf, 0xf, true));
        if (rg == (ph_ + 1) % G) vsum = up;
      }
    }
  };

  load_step(0, stg[0]);
  load_step(1, stg[1]);
  __syncthreads();  // (the fills)
  store_step(0, stg[0]);
  load_step(2, stg[0]);
  int ts = 0;
  auto nxt = [](int v) { return v == 2 ? 0 : v + 1; };
  for (int t = 0; t < steps; t += 2) {
    const int tsa = ts, tsb = nxt(tsa), tsc = nxt(tsb);
    wave_order_lds();
    step(t, tsa);
    store_step(tsb, stg[1]);
    // (unconditional, with the step index clamped: loads behind a uniform branch make the compiler wait for every
    // outstanding one where the paths join; the rows a clamped index fetches again are never stored)
    load_step(min(t + 3, steps - 1), stg[1]);
    wave_order_lds();
    if (t + 1 < steps) step(t + 1, tsb);
    if (t + 2 < steps) store_step(tsc, stg[0]);
    load_step(min(t + 4, steps - 1), stg[0]);
    ts = tsc;
  }
}

// ---------------------------------------------------------------------------------------------
// The VALU kernels for every geometry other than 256x256 and 32x32 (what k_band_area does not take: integer ratios,
// 3 x 3 / 5 x 5 blurs, more than 1920 columns, odd views, small batches).  Every lane owns 8 adjacent columns like in
// k_dcthash_256: horizontal K-tap sums with v_dot4_u32_u8 against compile-time byte masks, vertical sliding sums on
// packed u16 pairs in a register ring, nearest(S / K^2) by multiply-shift (exact: see BlurK); cv::resize INTER_AREA's
// horizontal part as float chains  buf += S[sx] * alpha  in table order (the order is part of the result; integer-ratio
// geometries keep exact integer block sums instead); k_tile_hash: the vertical part (sum = beta * buf, then += in row
// order; or the integer block sum and resizeAreaFast_'s rounding), the 32x32 tile, stages 3-6.
//   k_blur_area       a workgroup per 16-row band, rows staged in LDS: any view, any batch size
//   k_blur_area_regs  a workgroup walks a strip (or the whole image: FUSE) with the rows streamed through registers
// (Rounds 1-3's k_blur_u8 + k_area_hash, k_blur_rows + k_area_rows and the LDS-streaming k_blur_area_stream are in the
//  history: r05's "hash_fast_any" 0, "hash_fused" 0, "hash_regs" 0.)



// Blur and horizontal INTER_AREA in one pass: the blurred band stays in LDS and is reduced to 32 floats per source row
// straight away, so the blurred plane never reaches HBM.  A workgroup owns `cpw` of the 32 output columns (all 32
// up to 2048 image columns, 16 up to 4096, 8 up to 8192): its window starts at the first source column of its first
// cell, so every cell's float accumulation  buf += S * alpha  runs start to end inside one workgroup, in table order.
// rows[] is indexed by SOURCE row here (k_tile_hash by_src = 1).
template <int K>
__global__ __launch_bounds__(256) void k_blur_area(const unsigned char* __restrict__ imgs, int w, int h,
                                                   size_t row_stride, size_t img_stride,
                                                   const AreaTab* __restrict__ xtab, const int* __restrict__ xfirst,
                                                   int isx, int cpw, int pitch /* 8 * blockDim.x + 8 */,
                                                   float* __restrict__ rows /* n * h * 32 */,
                                                   int pw, int ph, int ox, int oy /* the image is the w x h view at
                                                   (ox, oy) of a pw x ph parent starting at imgs: the blur takes its
                                                   border pixels from the parent, like cv::blur on a cv::Mat view */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_fused[];
  constexpr int R = K / 2;
  constexpr int kMaxRows = kBlurRB + K - 1;
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;
  const int bpitch = 8 * T;
  unsigned char* __restrict__ sband = s_fused;  // kMaxRows x pitch; later the kBlurRB x bpitch blurred rows
  unsigned char* __restrict__ sblur = s_fused;
  float* __restrict__ salpha = reinterpret_cast<float*>(s_fused + (size_t)kMaxRows * pitch);
  const int c0 = (int)blockIdx.x * cpw, c1 = c0 + cpw;
  const int k_base = isx ? 0 : xfirst[c0], k_end = isx ? 0 : xfirst[c1];
  const int cx0 = isx ? c0 * isx : xtab[k_base].si;             // first image column of this workgroup's cells
  const int cxe = isx ? c1 * isx : xtab[k_end - 1].si + 1;      // one past their last column
  const int y0 = (int)blockIdx.y * kBlurRB;
  const int gx0 = cx0 + ox;  // parent column of the window's first cell column
  const unsigned char* __restrict__ img = imgs + (size_t)blockIdx.z * img_stride;
  const int out_rows = min(kBlurRB, h - y0);
  const int nrows = out_rows + 2 * R;
  const int ndw = pitch >> 2;
  for (int i = tid; i < k_end - k_base; i += T) salpha[i] = xtab[k_base + i].alpha;
  // LDS column c <-> image x = cx0 - 4 + c (as in k_blur_rows)
  for (int dwi = tid; dwi < ndw; dwi += T) {
    const int x = gx0 - 4 + 4 * dwi;  // parent column
    if (x >= 0 && x + 3 < pw) {
      unsigned v[kMaxRows];
#pragma unroll
      for (int rr = 0; rr < kMaxRows; ++rr) {
        int ry = oy + y0 - R + rr;  // parent row
        ry = ry < 0 ? -ry : (ry >= ph ? 2 * (ph - 1) - ry : ry);
        ry = ry < 0 ? 0 : (ry >= ph ? ph - 1 : ry);
        v[rr] = *reinterpret_cast<const u32_any_align*>(img + (size_t)ry * row_stride + x);
      }
#pragma unroll
      for (int rr = 0; rr < kMaxRows; ++rr)
        if (rr < nrows) *reinterpret_cast<unsigned*>(sband + (size_t)rr * pitch + 4 * dwi) = v[rr];
    }
  }
  {
    const int nl = gx0 < 4 ? 4 : 0;  // the first window dword reaches across the parent's left edge
    const int c_right = ((pw - gx0 + 4) >> 2) << 2;
    const int nr = max(0, min(pitch, pw - gx0 + 7) - c_right);
    const int per_row = nl + nr;
    for (int e = tid; e < per_row * nrows; e += T) {
      const int rr = e / per_row, k = e - rr * per_row;
      const int c = k < nl ? k : c_right + (k - nl);
      int xx = gx0 - 4 + c;
      xx = xx < 0 ? -xx : xx;
      xx = xx >= pw ? 2 * (pw - 1) - xx : xx;
      xx = xx < 0 ? 0 : (xx >= pw ? pw - 1 : xx);
      sband[(size_t)rr * pitch + c] = img[(size_t)reflect101(oy + y0 - R + rr, ph) * row_stride + xx];
    }
  }
  __syncthreads();
  // blur: this lane's 8 columns, all rows of the band.  The results wait in registers until every lane is done with
  // the band, then overwrite it: the blurred rows alias the band's LDS (one buffer instead of two -> more workgroups
  // per CU).  The row loop is fully unrolled so that the result registers are indexed statically.
  uint2 qo[kBlurRB];
  const bool lane_live = cx0 + 8 * tid < cxe;
  if (lane_live) {
    unsigned ring[K][4];
    unsigned S[4];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) ring[j][c] = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) S[c] = BlurK<K>::add | (BlurK<K>::add << 16);
    const unsigned char* __restrict__ win = sband + 8 * tid;
#pragma unroll
    for (int rr = 0; rr < kMaxRows; ++rr) {
      if (rr < nrows) {
        const int j = rr % K;
        const uint2 a = *reinterpret_cast<const uint2*>(win + (size_t)rr * pitch);
        const uint2 b = *reinterpret_cast<const uint2*>(win + (size_t)rr * pitch + 8);
        const unsigned W[4] = {a.x, a.y, b.x, b.y};
        unsigned P[4];
        hsum_pairs<R>(W, P);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          S[c] = (S[c] - ring[j][c]) + P[c];
          ring[j][c] = P[c];
        }
        if (rr >= 2 * R) {
          qo[rr - 2 * R] = blur_quotients<K>(S);
        }
      }
    }
  }
  __syncthreads();  // every lane has read what it needs of the band
  if (lane_live) {
#pragma unroll
    for (int i = 0; i < kBlurRB; ++i)
      if (i < out_rows) *reinterpret_cast<uint2*>(sblur + (size_t)i * bpitch + 8 * tid) = qo[i];
  }
  __syncthreads();
  // horizontal INTER_AREA pass: one (row, output column) chain per task, in table order.  A lane takes one output
  // column and two rows at a time (they share the weights), operands fetched eight ahead of the dependent adds.
  {
    const int groups = max(1, T / cpw);
    const int slot = tid / cpw, c = c0 + (tid - slot * cpw);
    if (slot < groups) {
      const int k0 = isx ? 0 : xfirst[c], nk = isx ? isx : xfirst[c + 1] - k0;
      const int col = (isx ? c * isx : xtab[k0].si) - cx0;  // si is consecutive within an output column
      const float* __restrict__ al = salpha + (k0 - k_base);
      for (int r = slot; r < out_rows; r += 2 * groups) {
        const int r2 = r + groups;
        const bool two = r2 < out_rows;
        const unsigned char* __restrict__ Sa = sblur + (size_t)r * bpitch + col;
        const unsigned char* __restrict__ Sb = sblur + (size_t)(two ? r2 : r) * bpitch + col;
        float* __restrict__ oa = rows + ((size_t)blockIdx.z * (size_t)h + (size_t)(y0 + r)) * 32 + c;
        float* __restrict__ ob = rows + ((size_t)blockIdx.z * (size_t)h + (size_t)(y0 + (two ? r2 : r))) * 32 + c;
        if (isx) {
          unsigned sa = 0, sb = 0;
          for (int u = 0; u < nk; ++u) {
            sa += Sa[u];
            sb += Sb[u];
          }
          *oa = __uint_as_float(sa);
          if (two) *ob = __uint_as_float(sb);
        } else {
          float ba = 0.f, bb = 0.f;
          int k = 0;
          for (; k + 8 <= nk; k += 8) {
            float a[8];
            unsigned pa[8], pb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              a[u] = al[k + u];
              pa[u] = Sa[k + u];
              pb[u] = Sb[k + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              ba += (float)pa[u] * a[u];
              bb += (float)pb[u] * a[u];
            }
          }
          for (; k < nk; ++k) {
            const float av = al[k];
            ba += (float)Sa[k] * av;
            bb += (float)Sb[k] * av;
          }
          *oa = ba;
          if (two) *ob = bb;
        }
      }
    }
  }
}

// Streaming form for batches with enough workgroups to spare: a workgroup walks down a strip of the image in steps of
// kStep source rows (a multiple of K, so the register ring keeps its phase) and carries the sliding column sums from
// step to step -- the K-1 halo rows are read and summed once per strip instead of once per 16-row band.
template <int K>
struct StreamK {
  static constexpr int step = K == 7 ? 14 : 15;  // rows per step: a multiple of K
};

// k_blur_area_regs: the streaming kernel with the blur input taken STRAIGHT FROM GLOBAL MEMORY into registers, the way
// k_dcthash_256 does it: per source row a lane loads its own 8 pixels (one
// aligned 8-byte load) and the dword on either side of them (the neighbours' pixels come from L1/L2; the two border
// lanes mirror their own pixels with a per-lane v_perm selector = REFLECT_101), a ring of PF rows in flight across the
// step boundaries.  LDS only holds the blurred rows of a step for the horizontal INTER_AREA chains.
// Preconditions (checked by the launcher; everything else takes k_blur_area): whole images, or views whose left
// and right edges are the parent's or lie at least 4 (+ the last lane's overhang) pixels inside it (autocrop of
// letterboxed / pillarboxed frames: the row mapping and the edge lanes' roles change, nothing else),
// 32 <= w <= 2048 (one workgroup spans the row).  GEN = false: w a multiple of 8 and image base / row stride / image
// stride multiples of 8 (aligned 8-byte loads, no extra shuffles); GEN = true: any width and alignment (+3 v_perm per
// row and lane, unaligned dword loads).
// FUSE (round 3): the workgroup walks the WHOLE image (one strip) and keeps going where the split version hands over to
// k_tile_hash: the horizontal sums of a step stay in LDS (ipb x kStep x 32 floats), 32 lanes per image carry the vertical
// INTER_AREA accumulation from step to step in registers -- entries of the y table in table order, `sum = alpha * v`
// for a cell's first entry, `sum += alpha * v` after it, exactly k_tile_hash's chain; integer ratios: exact block sums
// -- and round each finished output row into the image's 32 x 32 tile in LDS.  What leaves the kernel is 1024 bytes per
// image instead of h x 32 floats (0.64 GB per 2 GB of 400x300 input, written and read back); stages 3-6 run from the
// tiles (k_tiles_hash).  The vertical lanes do their rows right behind the barrier that ends a step, before their
// next blur rows: no extra barrier.
// KS = source rows per step (a multiple of K and of the rows in flight).  The area phase hands the step's rows to the
// workgroup's T / 32 row groups two at a time (x 32 / ncell on a column strip), so a step of 14 rows fills a turn of
// 12 (T = 192), 24, 32, 48 or 64 row slots badly: the launcher picks 21 or 28 rows where that fills the turns better
// (pick_rows_per_step below; K = 7 only).
template <int K, bool GEN, bool FUSE, int KS = StreamK<K>::step>
__global__ __launch_bounds__(256) void k_blur_area_regs(const unsigned char* __restrict__ imgs, int w, int h,
                                                        unsigned row_stride, size_t img_stride,
                                                        const AreaTab* __restrict__ xtab,
                                                        const int* __restrict__ xfirst, int isx,
                                                        int steps /* per strip */,
                                                        float* __restrict__ rows /* n * h * 32 */,
                                                        int ipb /* images side by side in the workgroup */,
                                                        unsigned n_imgs, const YRow* __restrict__ yrow = nullptr,
                                                        int isy = 0, unsigned char* __restrict__ tiles_out = nullptr,
                                                        int oy = 0, int ph = 0, int ox = 0, int pw = 0,
                                                        /* a view: the w x h rectangle at (ox, oy) of a pw x ph parent
                                                           that starts at imgs; ph = 0: whole images */
                                                        int cell0 = 0, int ncell = 32
                                                        /* a column strip of an image wider than 2048: this launch makes
                                                           the output cells cell0 .. cell0 + ncell - 1 of every row (xtab /
                                                           xfirst describe those cells, columns relative to the strip) */,
                                                        int cpad = 0
                                                        /* integer ratios with cells of an even number of dwords: one pad
                                                           dword behind every cell of a blurred row in LDS (see `bp`) */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_fused[];
  constexpr int R = K / 2;
  constexpr int kStep = KS;
  constexpr int PF = K == 7 ? 7 : 5;  // rows in flight; divides kStep so that the ring slot of a row is static
  static_assert(KS % K == 0 && KS % PF == 0, "the register ring and the prefetch ring keep their phase across steps");
  const int T = (int)blockDim.x, tid = (int)threadIdx.x;
  const int L = (w + 7) >> 3;  // lanes per image: 8 columns each.  A 640-pixel row needs 80 lanes: three images share 256
  // LDS pitch of a blurred row.  The area phase has the 32 lanes of a row group read dword u of 32 different cells at
  // once (ds_read_b32: bank = dword mod 32): cells of nd = isx / 4 dwords put them gcd(nd, 32) to a bank -- 8-way at 1024
  // px, 16-way at 2048 (SQ_LDS_BANK_CONFLICT: 31 % of the kernel's cycles at 1024 x 768).  cpad: every cell is followed
  // by one pad dword, cell stride nd + 1 (odd): no conflicts; the blur lanes' two dwords never straddle a cell (nd even).
  const int bp = 8 * L + (cpad ? 128 : 0);
  const int islot = tid / L;
  const unsigned img_i = blockIdx.z * (unsigned)ipb + (unsigned)islot;
  const bool lane_live = islot < ipb && img_i < n_imgs;
  const int tl = lane_live ? tid - islot * L : 0;
  unsigned char* __restrict__ sblur_all = s_fused;  // ipb x kStep x bp blurred rows
  unsigned char* __restrict__ sblur = sblur_all + (size_t)(lane_live ? islot : 0) * (size_t)kStep * (size_t)bp;
  float* __restrict__ salpha = reinterpret_cast<float*>(s_fused + (size_t)ipb * (size_t)kStep * (size_t)bp);
  const int k_end = isx ? 0 : xfirst[32];
  // per-cell edge weights of the area walk (16 floats per cell; fractional ratios only), 16-byte aligned
  float* __restrict__ swt = salpha + ((k_end + 3) & ~3);
  // FUSE: horizontal sums of the current step and the tiles, behind the weights
  float* __restrict__ shrow = swt + (isx ? 0 : 512);
  unsigned char* __restrict__ stile = reinterpret_cast<unsigned char*>(shrow + (FUSE ? (size_t)ipb * kStep * 32 : 0));
  const int strip_out = steps * kStep - 2 * R;
  const int o0 = (int)blockIdx.y * strip_out;
  const int o1 = min(h, o0 + strip_out);
  // Addresses = a workgroup-uniform 64-bit base (first image of the workgroup + the row: scalar registers) + a 32-bit
  // per-lane offset that never changes (image slot + column): global loads in their `saddr` form, no vector address
  // arithmetic per row (it took 4 of the 47 VALU instructions of a blur row).  The launcher keeps ipb * img_stride < 2^32.
  const unsigned char* __restrict__ img = imgs + (size_t)(blockIdx.z * (unsigned)ipb) * img_stride;
  const unsigned ioff = lane_live ? (unsigned)islot * (unsigned)img_stride : 0u;
  for (int i = tid; i < k_end; i += T) salpha[i] = xtab[i].alpha;
  // The 16-byte window of a lane (image bytes 8*tl - 4 .. 8*tl + 11) is assembled from three loads and v_perm_b32
  // with per-lane selectors (v_perm_b32(S0, S1, sel): selector 4..7 -> S0 byte 0..3, 0..3 -> S1 byte 0..3):
  //   W[0] = perm(left dword, own.x, selL)   W[1] = perm(own.y, own.x, sel1)   W[2] = perm(own.y, own.x, sel2)
  //   W[3] = perm(right dword, perm(own.y, own.x, selT), selR)
  // Interior lanes use identity selectors.  REFLECT_101 at the image edges is a matter of selectors and load
  // offsets of the edge lanes only: lane 0 mirrors its own pixels into the left halo; with w a multiple of 8 the
  // last lane mirrors its own pixels into the right halo.  GEN (any width): the last lane owns only m = w mod 8 real
  // pixels -- it loads the LAST eight bytes of the row and shuffles real and mirrored pixels into place, its right
  // halo comes from those and the dword before them; for m <= 3 the lane before it reads its right halo from the
  // last dword of the row.  (sel1 / sel2 / selT are the identity and cost nothing unless GEN.)
  // A view (cv::blur on a cv::Mat view takes its border from the parent): where the parent has at least 4 more pixels
  // on the left (li) the first lane is an interior lane; where it has room for the last lane's 8 pixels + 4 on the right
  // (ri) so is the last one, whatever w mod 8 is -- its columns past the view are computed from real pixels and never
  // read.  The launcher admits ox = 0 or >= 4 and ox + w = pw or ri; all offsets below are shifted by ox at the end.
  const bool li = ox >= 4, ri = pw != 0 && ox + 8 * L + 4 <= pw;
  unsigned selL = (tl == 0 && !li) ? 0x01020300u : 0x07060504u;  // lane 0: (x, px3, px2, px1) from its own pixels
  unsigned selR = 0x07060504u, sel1 = 0x03020100u, sel2 = 0x07060504u, selT = 0x07060504u;
  // where the lane's blurred pixels go in the LDS row
  const unsigned offS = 8u * (unsigned)tl + (cpad ? 4u * ((2u * (unsigned)tl) / (unsigned)(isx >> 2)) : 0u);
  unsigned offC = 8u * (unsigned)tl;  // the lane's own pixels in the source row
  unsigned offL = (tl == 0 && !li) ? offC : offC - 4u;  // (lane 0 with li: ox - 4 after the shift)
  unsigned offR = offC + 8u;
  const int m = w & 7;
  if (ri) {
    // every lane reads real pixels on its right
  } else if (!GEN || m == 0) {
    if (tl == L - 1) selR = 0x00000102u, offR = offC + 4u;  // (px w-2, w-3, w-4, x) from its own pixels
  } else {
    auto src_of = [&](int x) { return x <= w - 1 ? x : 2 * (w - 1) - x; };
    if (tl == L - 1) {
      offC = (unsigned)(w - 8);
      offR = (unsigned)(w - 12);
      sel1 = sel2 = selT = selR = 0u;
      for (int i = 0; i < 4; ++i) {
        sel1 |= (unsigned)(src_of(w - m + i) - (w - 8)) << (8 * i);
        sel2 |= (unsigned)(src_of(w - m + 4 + i) - (w - 8)) << (8 * i);
      }
      for (int j = 0; j < 3; ++j) {
        const int sp = src_of(w - m + 8 + j);
        if (sp >= w - 8) {
          selT |= (unsigned)(sp - (w - 8)) << (8 * j);
          selR |= (unsigned)j << (8 * j);
        } else {
          selR |= (unsigned)(4 + sp - (w - 12)) << (8 * j);
        }
      }
    } else if (tl == L - 2 && m <= 3) {
      offR = (unsigned)(w - 4);
      selR = 0u;
      for (int j = 0; j < 3; ++j) selR |= (unsigned)(4 + src_of(w - m + j) - (w - 4)) << (8 * j);
    }
  }
  offC += (unsigned)ox + ioff, offL += (unsigned)ox + ioff, offR += (unsigned)ox + ioff;
  const int hp = ph ? ph : h;  // the rows above and below a view are the parent's; REFLECT_101 at the parent's edges
  // byte offset of (reflected, clamped) source row s.  REFLECT_101 as min(|y|, 2 (hp - 1) - |y|), clamped at 0 for the rows
  // prefetched far below a small image: five scalar instructions per row instead of the twelve of the compare-and-select
  // form (equal for hp >= 4: rows above the image reach -3 at most)
  const int hp2 = 2 * (hp - 1);
  auto row_base = [&](int s) -> unsigned {
    int ry = oy + s;
    ry = ry < 0 ? -ry : ry;
    ry = min(ry, hp2 - ry);
    ry = max(ry, 0);
    return (unsigned)ry * row_stride;
  };
  unsigned ring[K][4];
  unsigned S[4];
#pragma unroll
  for (int j = 0; j < K; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) ring[j][c] = 0u;
#pragma unroll
  for (int c = 0; c < 4; ++c) S[c] = BlurK<K>::add | (BlurK<K>::add << 16);
  // area-pass roles: lane = output column cc of row group rg; the T / 32 row groups walk the (image, row) list of the
  // whole workgroup, two entries per turn (they share the weights)
  const int G = T >> 5, rg = tid >> 5, cc = tid & 31;
  // a strip owns ncell = 16 / 8 / 4 of the 32 cells: the 32 lanes of a row group then take 2 / 4 / 8 rows of those cells at
  // once instead of idling (lane cc -> cell cc mod ncell, row sub-group cc / ncell); ncell = 32: one row, as ever
  const int ccl = cc & (ncell - 1);
  const int Gm = 32 / ncell, Ge = G * Gm, rge = rg * Gm + cc / ncell;
  const int ak0 = isx ? 0 : xfirst[ccl];
  const int ank = isx ? isx : xfirst[ccl + 1] - ak0;
  const int acol = isx ? ccl * isx + (cpad ? 4 * ccl : 0) : xtab[ak0].si;
  const float* __restrict__ al = salpha + ak0;

  // weights of the lane's cell by pixel position: (partial first) mid ... mid (partial last), +0 past the cell --
  // make_area_tab gives every interior pixel the same float(1 / cellWidth).  Word 0 and the last two words of the
  // uniform walk of nw_u words are kept per CELL in LDS (swt: 16 floats per cell, read at the head of a turn so that
  // they occupy registers in the area phase only); the words between are a_mid for every lane (mid_ok, checked).
  const unsigned ama = (unsigned)acol & 3u;  // (rows start on 8-byte boundaries of the LDS buffer)
  int nw_u = 0;
  bool mid_ok = true;
  if (!isx) {
    __syncthreads();  // salpha
    const float a_first = al[0], a_mid_ = al[ank > 1 ? 1 : 0], a_last = al[ank - 1];
    auto area_wt = [&](int k) -> float {
      return k == 0 ? a_first : (k < ank - 1 ? a_mid_ : (k == ank - 1 ? a_last : 0.f));
    };
    int amax = ank, amin = ank;  // over the 32 cells (every wave holds all of them)
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      amax = max(amax, __shfl_xor(amax, d));
      amin = min(amin, __shfl_xor(amin, d));
    }
    nw_u = __builtin_amdgcn_readfirstlane((amax + 3) >> 2);
    mid_ok = __builtin_amdgcn_readfirstlane(4 * nw_u - 7 <= amin ? 1 : 0) != 0;  // pixels 4 .. 4 nw_u - 9 are interior
    if (tid < 32) {
      float* __restrict__ o = swt + 16 * cc;
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = area_wt(u);
#pragma unroll
      for (int u = 0; u < 8; ++u) o[4 + u] = area_wt(4 * (nw_u - 2) + u);
      o[12] = a_mid_, o[13] = a_first, o[14] = a_last, o[15] = 0.f;
    }
  }
  const int sfirst = o0 - R;  // first source row of the strip
  uint2 rawC[PF];
  unsigned rawL[PF], rawR[PF];
  // (word 3: 32-bit data format, as the runtime's own descriptors on gfx9; untyped loads ignore it.  No stride, 4 GB of range)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(img), 0, (int)0xffffffffu, 0x27000);
  auto load_row = [&](int s, uint2& c, unsigned& l, unsigned& r) {
    const unsigned char* __restrict__ p = GEN ? img + row_base(s) : img;
    if constexpr (GEN) {  // any alignment
      // (raw buffer loads work at odd addresses too but buy nothing here: 900x600 -5 %, 533x400 / 1366x768 +-1 %)
      c.x = *reinterpret_cast<const u32_any_align*>(p + offC);
      c.y = *reinterpret_cast<const u32_any_align*>(p + offC + 4);
      l = *reinterpret_cast<const u32_any_align*>(p + offL);
      r = *reinterpret_cast<const u32_any_align*>(p + offR);
    } else {
      // raw buffer loads: address = descriptor base (the workgroup's first image) + per-lane offset (VGPR, constant) + row
      // offset (SGPR): no vector address arithmetic per row at all (the flat form spent 3-4 v_lshl_add_u64 per row)
      (void)p;
      const unsigned so = row_base(s);
      const v2u_t c2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)offC, (int)so, 0);
      c.x = c2.x, c.y = c2.y;
      l = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)offL, (int)so, 0);
      r = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)offR, (int)so, 0);
    }
  };
  // narrow images (L < 32) leave whole waves without a blur lane (the workgroup keeps 32 lanes per image for the area and
  // vertical passes): those waves skip the row loop and only meet the barriers
  const bool wave_live = (__builtin_amdgcn_readfirstlane(tid) & ~63) / L < ipb;
  if (wave_live) {
#pragma unroll
    for (int j = 0; j < PF; ++j) load_row(sfirst + j, rawC[j], rawL[j], rawR[j]);
  }
  // FUSE: vertical INTER_AREA state of lane (image vi, output column vc), carried across the steps
  const int vi = tid >> 5, vc = tid & 31;
  const bool vlane = FUSE && vi < ipb;
  float vsum = 0.f;
  unsigned vacc = 0u;
  for (int st = 0; st < steps; ++st) {
    const int s0 = sfirst + st * kStep;  // first source row consumed in this step
    if (s0 - R >= o1) break;             // nothing left to output (uniform)
    if (wave_live) {
#pragma unroll
    for (int rr = 0; rr < kStep; ++rr) {
      const int j = rr % K, pj = rr % PF;
      const uint2 dC = rawC[pj];
      const unsigned dl = rawL[pj], dr = rawR[pj];
      load_row(s0 + rr + PF, rawC[pj], rawL[pj], rawR[pj]);
      unsigned W[4];
      W[0] = __builtin_amdgcn_perm(dl, dC.x, selL);
      if constexpr (GEN) {
        W[1] = __builtin_amdgcn_perm(dC.y, dC.x, sel1);
        W[2] = __builtin_amdgcn_perm(dC.y, dC.x, sel2);
        W[3] = __builtin_amdgcn_perm(dr, __builtin_amdgcn_perm(dC.y, dC.x, selT), selR);
      } else {
        W[1] = dC.x, W[2] = dC.y;
        W[3] = __builtin_amdgcn_perm(dr, dC.y, selR);
      }
      unsigned P[4];
      hsum_pairs<R>(W, P);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        S[c] = (S[c] - ring[j][c]) + P[c];
        ring[j][c] = P[c];
      }
      const uint2 qo = blur_quotients<K>(S);
      if (lane_live) {
        if (cpad) {  // (uniform) dword-aligned only: ds_write2_b32
          unsigned* __restrict__ po = reinterpret_cast<unsigned*>(sblur + (size_t)rr * (size_t)bp + offS);
          po[0] = qo.x, po[1] = qo.y;
        } else {
          *reinterpret_cast<uint2*>(sblur + (size_t)rr * (size_t)bp + offS) = qo;
        }
      }
    }
    }
    __syncthreads();
    // ---- horizontal INTER_AREA chains for the valid blurred rows of this step: local row rr <-> image row s0 + rr - R.
    // Row group rg takes the local rows lo + rg, lo + rg + G, ... of every image, two per turn (they share the lane's
    // weights and give the chain a second, independent accumulator).  No division, no per-lane trip counts: every
    // lane walks nw_u words; pixels past its own cell carry the weight +0.0f (x + 0.0f == x: the sums are the chains
    // `ba += float(p[k]) * alpha[k]`, k ascending, of make_area_tab to the bit).
    {
      const int ob = s0 - R;
      const int lo = max(0, o0 - ob), hi = min(kStep, o1 - ob);  // valid local rows [lo, hi), the same for every image
      for (int ia = 0; ia < ipb; ++ia) {
        const unsigned ga = blockIdx.z * (unsigned)ipb + (unsigned)ia;
        if (ga >= n_imgs) break;  // uniform
        const unsigned char* __restrict__ Si = sblur_all + (size_t)ia * (size_t)kStep * (size_t)bp + acol - ama;
        float* __restrict__ Oi = FUSE ? shrow + (size_t)ia * kStep * 32 + cc
                                      : rows + ((ptrdiff_t)ga * (ptrdiff_t)h + (ptrdiff_t)ob) * 32 + cell0 + ccl;
        for (int ra = lo + rge; ra < hi; ra += 2 * Ge) {
          const int rb_ = ra + Ge;
          const bool two = rb_ < hi;
          const unsigned* __restrict__ A4 = reinterpret_cast<const unsigned*>(Si + (size_t)ra * (size_t)bp);
          const unsigned* __restrict__ B4 = reinterpret_cast<const unsigned*>(Si + (size_t)(two ? rb_ : ra) * (size_t)bp);
          if (isx) {
            unsigned sa = 0, sb = 0;
            if (((isx | bp) & 3) == 0) {  // cells start on dword boundaries: four pixels per v_dot4 (uniform branch)
              for (int u = 0; u < (isx >> 2); ++u) {
                sa = udot4(A4[u], 0x01010101u, sa);
                sb = udot4(B4[u], 0x01010101u, sb);
              }
            } else {
              unsigned lo_a = A4[0], lo_b = B4[0];
              for (int c = 0; c < ((isx + 3) >> 2); ++c) {
                const unsigned hi_a = A4[c + 1], hi_b = B4[c + 1];
                unsigned wa = __builtin_amdgcn_alignbyte(hi_a, lo_a, ama), wb = __builtin_amdgcn_alignbyte(hi_b, lo_b, ama);
                lo_a = hi_a, lo_b = hi_b;
                const int left = isx - 4 * c;  // uniform
                if (left < 4) wa &= (1u << (8 * left)) - 1u, wb &= (1u << (8 * left)) - 1u;
                sa = udot4(wa, 0x01010101u, sa);
                sb = udot4(wb, 0x01010101u, sb);
              }
            }
            Oi[ra * 32] = __uint_as_float(sa);
            if (two) Oi[rb_ * 32] = __uint_as_float(sb);
          } else {
            v2f_t bab = {0.f, 0.f};  // (row ra, row rb_)
            unsigned lo_a = A4[0], lo_b = B4[0], wa, wb;
            const float4 wF4 = *reinterpret_cast<const float4*>(swt + 16 * cc);
            const float4 wTa = *reinterpret_cast<const float4*>(swt + 16 * cc + 4);
            const float4 wTb = *reinterpret_cast<const float4*>(swt + 16 * cc + 8);
            const float4 wM = *reinterpret_cast<const float4*>(swt + 16 * cc + 12);  // a_mid, a_first, a_last
            const float a_mid = wM.x;
#define CBH_AREA_WORD(c)                                     \
  {                                                          \
    const unsigned hi_a = A4[(c) + 1], hi_b = B4[(c) + 1];   \
    wa = __builtin_amdgcn_alignbyte(hi_a, lo_a, ama);        \
    wb = __builtin_amdgcn_alignbyte(hi_b, lo_b, ama);        \
    lo_a = hi_a, lo_b = hi_b;                                \
  }
// The two rows of a turn share the weights and have independent accumulators: as a float pair their `sum += p * alpha` is one
// v_pk_mul_f32 + one v_pk_add_f32 (each component an ordinary IEEE multiply / add: the same bits as the scalar chain), i.e. two
// instructions for two pixels instead of four -- on this VALU every instruction of a mixed stream costs ~4 cycles (NOTES 9).
#define CBH_AREA_PIX1(SH, W)                                                                      \
  {                                                                                              \
    const v2f_t px_ = {(float)((wa >> (SH)) & 0xffu), (float)((wb >> (SH)) & 0xffu)};             \
    const v2f_t w_ = {(W), (W)};                                                                 \
    bab = bab + px_ * w_;                                                                         \
  }
#define CBH_AREA_PIX4(W0, W1, W2, W3) \
  CBH_AREA_PIX1(0, W0) CBH_AREA_PIX1(8, W1) CBH_AREA_PIX1(16, W2) CBH_AREA_PIX1(24, W3)
            CBH_AREA_WORD(0);
            CBH_AREA_PIX4(wF4.x, wF4.y, wF4.z, wF4.w);
            if (mid_ok) {
#pragma unroll 2  // (the carried dwords lo_a / lo_b ping-pong between registers instead of being moved every word)
              for (int c = 1; c < nw_u - 2; ++c) {  // interior words: every pixel of every lane weighs a_mid
                CBH_AREA_WORD(c);
                CBH_AREA_PIX4(a_mid, a_mid, a_mid, a_mid);
              }
              if (nw_u >= 3) {
                CBH_AREA_WORD(nw_u - 2);
                CBH_AREA_PIX4(wTa.x, wTa.y, wTa.z, wTa.w);
              }
              if (nw_u >= 2) {
                CBH_AREA_WORD(nw_u - 1);
                CBH_AREA_PIX4(wTb.x, wTb.y, wTb.z, wTb.w);
              }
            } else {  // cells of very different widths (no area table does this): weights word by word
              auto area_wt = [&](int k) -> float {
                return k == 0 ? wM.y : (k < ank - 1 ? a_mid : (k == ank - 1 ? wM.z : 0.f));
              };
              for (int c = 1; c < nw_u; ++c) {
                CBH_AREA_WORD(c);
                CBH_AREA_PIX4(area_wt(4 * c), area_wt(4 * c + 1), area_wt(4 * c + 2), area_wt(4 * c + 3));
              }
            }
#undef CBH_AREA_WORD
#undef CBH_AREA_PIX4
#undef CBH_AREA_PIX1
            Oi[ra * 32] = bab.x;
            if (two) Oi[rb_ * 32] = bab.y;
          }
        }
      }
    }
    __syncthreads();  // the blurred rows are consumed before the next step overwrites them
    if constexpr (FUSE) {
      // ---- vertical INTER_AREA for the rows this step produced (the barrier above completed them in LDS)
      const int ob = s0 - R;
      const int lo = max(0, o0 - ob), hi = min(kStep, o1 - ob);
      if (vlane) {
        const float* __restrict__ hr = shrow + (size_t)vi * kStep * 32 + vc;
        unsigned char* __restrict__ tl_ = stile + (size_t)vi * 1024 + vc;
        for (int ra = lo; ra < hi; ++ra) {
          const int y = ob + ra;
          const float v = hr[ra * 32];
          if (isx) {  // resizeAreaFast_: exact block sum; 2x2 -> (s+2)>>2, else rint(s * (1.f/area))
            vacc += __float_as_uint(v);
            if ((y + 1) % isy == 0) {
              const unsigned q = (isx == 2 && isy == 2) ? (vacc + 2u) >> 2
                                                        : (unsigned)__builtin_rintf((float)vacc * (1.f / (float)(isx * isy)));
              tl_[(y / isy) * 32] = (unsigned char)(q > 255u ? 255u : q);
              vacc = 0u;
            }
          } else {
            const YRow yr = yrow[y];  // uniform address: a scalar load, issued ahead of the dependent adds
            const float t0 = yr.a0 * v;
            vsum = (yr.info & 0x100) ? t0 : vsum + t0;
            if (yr.info & 0x200) {
              const float r = __builtin_rintf(vsum);
              tl_[(yr.info & 0xff) * 32] = (unsigned char)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
            }
            if (yr.info & 0x400) {  // the row straddles two cells
              const float t1 = yr.a1 * v;
              vsum = (yr.info & 0x800) ? t1 : vsum + t1;
              if (yr.info & 0x1000) {
                const float r = __builtin_rintf(vsum);
                tl_[((yr.info & 0xff) + 1) * 32] = (unsigned char)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
              }
            }
          }
        }
      }
    }
  }
  if constexpr (FUSE) {
    __syncthreads();
    // the tiles of this workgroup's images, coalesced (1024 bytes each)
    for (int i = tid; i < ipb * 256; i += T) {
      const unsigned g = blockIdx.z * (unsigned)ipb + (unsigned)(i >> 8);
      if (g < n_imgs)
        reinterpret_cast<unsigned*>(tiles_out + (size_t)g * 1024)[i & 255] = reinterpret_cast<const unsigned*>(stile)[i];
    }
  }
}


// stages 3-6 from finished 32 x 32 tiles (k_band_area, k_blur_area_regs<.., FUSE>): two tiles per 64-thread workgroup,
// a half-wave per image (the tail of k_dcthash_256_band as a kernel of its own)
__global__ __launch_bounds__(64) void k_tiles_hash2(const unsigned char* __restrict__ tiles_in, unsigned n,
                                                    const DctTables* __restrict__ tabs, uint64_t* __restrict__ out,
                                                    unsigned char* __restrict__ tiles_copy) {
  __shared__ __attribute__((aligned(16))) unsigned char sTile[2][1024];
  __shared__ __attribute__((aligned(16))) float sT[2 * 288], sY[2 * 84];
  const int lane = threadIdx.x, l32 = lane & 31, hw = lane >> 5;
  const unsigned img = blockIdx.x * 2u + (unsigned)hw;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const unsigned gi = min(blockIdx.x * 2u + (unsigned)g, n - 1u);
    const uint4 v = reinterpret_cast<const uint4*>(tiles_in + (size_t)gi * 1024)[lane];
    reinterpret_cast<uint4*>(sTile[g])[lane] = v;
    if (tiles_copy && blockIdx.x * 2u + (unsigned)g < n) reinterpret_cast<uint4*>(tiles_copy + (size_t)gi * 1024)[lane] = v;
  }
  __syncthreads();
  const unsigned long long hv = hash_halfwave(sTile[hw], sT + hw * 288, sY + hw * 84, tabs, lane);
  if (l32 == 0 && img < n) out[img] = hv;
}

__global__ __launch_bounds__(kThreads) void k_tile_hash(const float* __restrict__ rows, int yn,
                                                        const AreaTab* __restrict__ ytab,
                                                        const int* __restrict__ yfirst, int isx, int isy,
                                                        int by_src /* rows[] indexed by source row, not table row */,
                                                        const DctTables* __restrict__ tabs,
                                                        uint64_t* __restrict__ out,
                                                        unsigned char* __restrict__ tiles) {
  __shared__ __attribute__((aligned(16))) float sT[288], sY[84];
  __shared__ __attribute__((aligned(16))) unsigned char tile[1024], sZ[64];
  const int tid = threadIdx.x;
  const float* __restrict__ R = rows + (size_t)blockIdx.x * (size_t)yn * 32;
  if (tid < 64) sZ[tid] = tabs->zz[tid];
  {
    // a lane owns output column dx and rows dy0, dy0+8, dy0+16, dy0+24: four independent chains, advanced together
    // with the operands of four steps fetched ahead of the dependent adds (the chains themselves stay in row order)
    const int dx = tid & 31, dy0 = tid >> 5;
    if (isx) {  // resizeAreaFast_: exact block sum; 2x2 -> (s+2)>>2, else rint(s * (1.f/area))
      unsigned acc[4] = {0u, 0u, 0u, 0u};
      int yy = 0;
      for (; yy + 4 <= isy; yy += 4) {
        unsigned v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u) v[i][u] = __float_as_uint(R[(size_t)((dy0 + 8 * i) * isy + yy + u) * 32 + dx]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[i] += v[i][u];
      }
      for (; yy < isy; ++yy)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += __float_as_uint(R[(size_t)((dy0 + 8 * i) * isy + yy) * 32 + dx]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned v = (isx == 2 && isy == 2) ? (acc[i] + 2u) >> 2
                                                   : (unsigned)__builtin_rintf((float)acc[i] * (1.f / (float)(isx * isy)));
        tile[(dy0 + 8 * i) * 32 + dx] = (unsigned char)(v > 255u ? 255u : v);
      }
    } else {
      int j0[4], len[4];
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
      int nmax = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        j0[i] = yfirst[dy0 + 8 * i];
        len[i] = yfirst[dy0 + 8 * i + 1] - j0[i];
        nmax = max(nmax, len[i]);
      }
      for (int k = 0; k < nmax; k += 4) {
        float al[4][4], v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0[i] + min(k + u, len[i] - 1);  // past the end: a valid entry, result unused
            al[i][u] = ytab[j].alpha;
            v[i][u] = R[(size_t)(by_src ? ytab[j].si : j) * 32 + dx];
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k + u < len[i]) {
              const float t = al[i][u] * v[i][u];
              sum[i] = (k + u == 0) ? t : sum[i] + t;
            }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = __builtin_rintf(sum[i]);
        tile[(dy0 + 8 * i) * 32 + dx] = (unsigned char)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
      }
    }
  }
  __syncthreads();
  if (tiles)
    for (int i = tid; i < 1024; i += kThreads) tiles[(size_t)blockIdx.x * 1024 + i] = tile[i];
  hash_from_tile(tile, sZ, sT, sY, out + blockIdx.x, tabs);
}


struct TableCache {
  std::mutex mu;
  DctTables* d[16] = {};  // [device]
} g_tabs;

}  // namespace

namespace {

// computeResizeAreaTab (OpenCV 2.4 imgwarp.cpp, as recalled): see oracle/cbird_oracle.c for the prose
// cv::resize: inv_scale = (double)dsize/ssize; scale = 1./inv_scale (two roundings -- it is not ssize/dsize), and the
// integer "area fast" path only when |scale - round(scale)| < DBL_EPSILON on both axes: for ssize = 32*m with
// m = 49, 93, 98, 99, 103 ... the double rounding misses m by an ulp and the weighted tables are used instead.
double cv_resize_scale(int ssize, int dsize) {
  const double inv_scale = (double)dsize / ssize;
  return 1. / inv_scale;
}
}  // namespace
bool area_fast(int w, int h) {
  const double sx = cv_resize_scale(w, 32), sy = cv_resize_scale(h, 32);
  return std::fabs(sx - std::nearbyint(sx)) < DBL_EPSILON && std::fabs(sy - std::nearbyint(sy)) < DBL_EPSILON;
}

std::vector<AreaTab> make_area_tab(int ssize, int dsize, std::vector<int>* first) {
  std::vector<AreaTab> tab;
  const double scale = cv_resize_scale(ssize, dsize);
  first->assign((size_t)dsize + 1, 0);
  for (int dx = 0; dx < dsize; dx++) {
    (*first)[(size_t)dx] = (int)tab.size();
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cellWidth = std::min(scale, ssize - fsx1);
    int sx1 = (int)__builtin_ceil(fsx1), sx2 = (int)__builtin_floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back(AreaTab{sx1 - 1, dx, (float)((sx1 - fsx1) / cellWidth)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back(AreaTab{sx, dx, (float)(1.0 / cellWidth)});
    if (fsx2 - sx2 > 1e-3)
      tab.push_back(AreaTab{sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
  }
  (*first)[(size_t)dsize] = (int)tab.size();
  return tab;
}

namespace {

struct AreaTabsDev {
  AreaTab *x = nullptr, *y = nullptr;
  int *xfirst = nullptr, *yfirst = nullptr;
  int xn = 0, yn = 0;
  YRow* yrow = nullptr;  // h entries, nullptr when some source row has more than two entries (never for h >= 32);
                         // kYRowPad neutral entries before and behind them (k_band_area reads the four rows of a step
                         // without clamping their indices)
  YRow* yrow_base = nullptr;
};
constexpr int kYRowPad = 8;
std::mutex g_area_mu;
std::map<std::tuple<int, int, int>, AreaTabsDev> g_area;  // (device, w, h)

int get_area_tabs(int w, int h, AreaTabsDev* out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_area_mu);
  auto key = std::make_tuple(dev, w, h);
  auto it = g_area.find(key);
  if (it == g_area.end()) {
    std::vector<int> xf, yf;
    std::vector<AreaTab> xt = make_area_tab(w, 32, &xf), yt = make_area_tab(h, 32, &yf);
    AreaTabsDev d;
    d.xn = (int)xt.size();
    d.yn = (int)yt.size();
    // all or nothing: a table that could not be made leaves nothing behind (the next call starts over)
    hipError_t e = hipMalloc(&d.x, xt.size() * sizeof(AreaTab));
    if (e == hipSuccess) e = hipMalloc(&d.y, yt.size() * sizeof(AreaTab));
    if (e == hipSuccess) e = hipMalloc(&d.xfirst, 33 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&d.yfirst, 33 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d.x, xt.data(), xt.size() * sizeof(AreaTab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.y, yt.data(), yt.size() * sizeof(AreaTab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.xfirst, xf.data(), 33 * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d.yfirst, yf.data(), 33 * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      std::vector<YRow> yr_all((size_t)h + 2 * kYRowPad, YRow{0.f, 0.f, 0, 1.f});
      YRow* const yr = yr_all.data() + kYRowPad;
      std::vector<int> cnt((size_t)h, 0);
      bool ok = true;
      for (size_t j = 0; j < yt.size() && ok; ++j) {
        const AreaTab& en = yt[j];
        if (en.si < 0 || en.si >= h || en.di < 0 || en.di > 31) {
          ok = false;
          break;
        }
        const bool opens = (int)j == yf[(size_t)en.di], closes = (int)j + 1 == yf[(size_t)en.di + 1];
        YRow& r = yr[(size_t)en.si];
        if (cnt[(size_t)en.si] == 0) {
          r.a0 = en.alpha;
          r.info = en.di | (opens ? 0x100 : 0) | (closes ? 0x200 : 0);
          r.k0 = opens ? 0.f : 1.f;
        } else if (cnt[(size_t)en.si] == 1 && en.di == (r.info & 0xff) + 1) {
          r.a1 = en.alpha;
          r.info |= 0x400 | (opens ? 0x800 : 0) | (closes ? 0x1000 : 0);
        } else {
          ok = false;
        }
        cnt[(size_t)en.si]++;
      }
      for (int y = 0; y < h && ok; ++y) ok = cnt[(size_t)y] >= 1;
      if (ok) {
        e = hipMalloc(&d.yrow_base, yr_all.size() * sizeof(YRow));
        if (e == hipSuccess) e = hipMemcpy(d.yrow_base, yr_all.data(), yr_all.size() * sizeof(YRow), hipMemcpyHostToDevice);
        if (e == hipSuccess) d.yrow = d.yrow_base + kYRowPad;
      }
    }
    if (e != hipSuccess) {
      for (void* q : {(void*)d.x, (void*)d.y, (void*)d.xfirst, (void*)d.yfirst, (void*)d.yrow_base})
        if (q) (void)hipFree(q);
      CBH_HIP(e);
    }
    it = g_area.emplace(key, d).first;
  }
  *out = it->second;
  return CBH_OK;
}

// Column strips of an image wider than 2048 pixels (k_blur_area_regs spans a row with one workgroup of <= 256 lanes x 8
// columns): strip s of ncol makes the output cells [s * 32 / ncol, (s + 1) * 32 / ncol); it covers the source columns of
// exactly those cells (the blur's 3-pixel border comes from the parent through the view mechanism).
struct StripTabs {
  AreaTab* x = nullptr;  // fractional ratios: the cells' table entries, si relative to x0
  int* xfirst = nullptr;
  int xn = 0, x0 = 0, ws = 0;
};
std::map<std::tuple<int, int, int, int>, StripTabs> g_strips;  // (device, w or -w, ncol, s), under g_area_mu

// integer: the image takes resizeAreaFast_ (both ratios integral, area_fast()): cells are whole pixel runs, no tables
int get_strip_tabs(int w, bool integer, int ncol, int s, StripTabs* out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_area_mu);
  auto key = std::make_tuple(dev, integer ? -w : w, ncol, s);
  auto it = g_strips.find(key);
  if (it == g_strips.end()) {
    const int cpw = 32 / ncol, c_lo = s * cpw, c_hi = c_lo + cpw;
    StripTabs d;
    if (integer) {
      d.x0 = c_lo * (w / 32);
      d.ws = cpw * (w / 32);
    } else {
      std::vector<int> xf;
      const std::vector<AreaTab> xt = make_area_tab(w, 32, &xf);
      const int e0 = xf[(size_t)c_lo], e1 = xf[(size_t)c_hi];
      d.x0 = xt[(size_t)e0].si;
      d.ws = xt[(size_t)e1 - 1].si + 1 - d.x0;
      std::vector<AreaTab> st(xt.begin() + e0, xt.begin() + e1);
      for (AreaTab& e : st) e.si -= d.x0, e.di -= c_lo;
      std::vector<int> sf(33);
      for (int c = 0; c <= 32; ++c) sf[(size_t)c] = xf[(size_t)std::min(c_lo + c, c_hi)] - e0;
      d.xn = (int)st.size();
      hipError_t e = hipMalloc(&d.x, st.size() * sizeof(AreaTab));
      if (e == hipSuccess) e = hipMalloc(&d.xfirst, 33 * sizeof(int));
      if (e == hipSuccess) e = hipMemcpy(d.x, st.data(), st.size() * sizeof(AreaTab), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(d.xfirst, sf.data(), 33 * sizeof(int), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        if (d.x) (void)hipFree(d.x);
        if (d.xfirst) (void)hipFree(d.xfirst);
        CBH_HIP(e);
      }
    }
    it = g_strips.emplace(key, d).first;
  }
  *out = it->second;
  return CBH_OK;
}

}  // namespace

namespace {

int reflect101_host(int p, int len);

// Strips and band matrices of k_band_area for images of width w: n_strips >= 2 strips of cps = ceil(32 / n_strips) <= 16
// cells, each at most 240 source columns wide with cells of at most 32 table entries.  *out = nullptr (and CBH_OK) when
// the geometry does not fit (the caller takes k_blur_area_regs).
// "hash_band_area": 1 (default) = fractional-ratio geometries whose strips hold >= 4 cells (w <= 1920) take k_band_area (blur
// on the matrix cores, up to four rows per area walk), 0 = k_blur_area_regs as through round 4
int g_hash_band_area = 1;

struct BaTabsDev {
  BaStrip* strips = nullptr;
  int n_strips = 0, T = 0, amax = 0, RS = 4;
};
std::map<std::pair<int, int>, BaTabsDev> g_ba_tabs;  // (device, w), under g_area_mu

int get_ba_tabs(int w, BaTabsDev* out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_area_mu);
  auto key = std::make_pair(dev, w);
  auto it = g_ba_tabs.find(key);
  if (it == g_ba_tabs.end()) {
    BaTabsDev d;
    std::vector<int> xf;
    const std::vector<AreaTab> xt = make_area_tab(w, 32, &xf);
    std::vector<BaStrip> host;
    auto band_of = [&](int xs0, int c, bool plain, unsigned (*dst)[4]) {
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 16; ++j) {
          const int k = 16 * (l >> 4) + j, kk = k & 31;
          const int col = xs0 + 16 * c - 8 + kk, xo = xs0 + 16 * c + (l & 15);
          int wgt = 0;
          if (plain) {
            wgt = std::abs(col - xo) <= 3 ? 1 : 0;
          } else if (col >= 0 && col < w && xo < w) {
            for (int dd = -3; dd <= 3; ++dd) wgt += reflect101_host(xo + dd, w) == col;
          }
          if (k >= 32) wgt = -wgt;
          dst[l][j >> 2] |= (unsigned)(wgt & 0xff) << (8 * (j & 3));
        }
    };
    // cells per strip, in order of preference: 16 or 11 (a lane carries the four rows of a step: RS 4), 8 (two rows: RS 2),
    // 4 (one row: RS 1; w <= 1920).  Two cells per strip (w <= 3840) leave half the walk's lanes idle: measured 1.45 TB/s at
    // 3840 x 2160 against k_blur_area_regs' 2.6 -- not offered.
    // (fewer cells per strip than the geometry allows -- shorter rings, 15 instead of 10 waves per CU -- lose more to the
    // strips' halos than the occupancy returns: 400 x 300 2.92 -> 2.72 TB/s with 8 cells, 160 x 120 2.03 -> 1.42;
    // profiles/r06_band_area_cells_per_strip.txt)
    const int cps_try[4] = {16, 11, 8, 4};
    for (int ci = 0; ci < 4 && host.empty(); ++ci) {
      const int cps = cps_try[ci];
      bool ok = true;
      // one tile count for every strip: the widest strip's
      int Tc = 0;
      for (int c0 = 0; c0 < 32 && ok; c0 += cps) {
        const int c1 = std::min(32, c0 + cps);
        Tc = std::max(Tc, (xt[(size_t)xf[(size_t)c1] - 1].si + 1 - xt[(size_t)xf[(size_t)c0]].si + 15) / 16);
      }
      ok = ok && Tc >= 2 && Tc <= 15 && 16 * Tc + 8 <= w;
      if (!ok) continue;
      d.RS = cps > 8 ? 4 : cps == 8 ? 2 : 1;
      std::vector<BaStrip> cand;
      for (int c0 = 0; c0 < 32 && ok; c0 += cps) {
        const int c1 = std::min(32, c0 + cps);
        BaStrip b;
        memset(&b, 0, sizeof b);
        const int xs_cells = xt[(size_t)xf[(size_t)c0]].si;
        // the strip's tiles start at its first cell -- or further left, so that the LAST strip's tiles end exactly at
        // column w - 1 (then only its last tile sees the right edge) and no strip reaches past it
        b.xs = c0 == 0 ? 0 : std::min(xs_cells, w - 16 * Tc);
        b.T = Tc;
        b.cell0 = c0;
        b.ncell = c1 - c0;
        ok = b.xs == 0 || b.xs >= 8;
        if (c1 == 32) ok = ok && b.xs == w - 16 * Tc;
        for (int c = c0; c < c1 && ok; ++c) {
          const int e0 = xf[(size_t)c], e1 = xf[(size_t)c + 1];
          ok = e1 - e0 >= 2;
          b.amax = std::max(b.amax, e1 - e0);
          b.amin = c == c0 ? e1 - e0 : std::min(b.amin, e1 - e0);
          b.si0[c - c0] = xt[(size_t)e0].si - b.xs;
          ok = ok && b.si0[c - c0] >= 0 && b.si0[c - c0] + (e1 - e0) <= 16 * Tc;
          for (int e = e0; e < e1 && ok; ++e) {
            ok = xt[(size_t)e].si == xt[(size_t)e0].si + (e - e0) && xt[(size_t)e].di == c;  // consecutive columns, in order
          }
        }
        ok = ok && b.amax - b.amin + 1 <= 4;
        for (int c = c0; c < c1 && ok; ++c) {  // the walk's weights: first, mid (entries 1 .. amin - 2), tail
          const int e0 = xf[(size_t)c], nn = xf[(size_t)c + 1] - e0;
          b.wfirst[c - c0] = xt[(size_t)e0].alpha;
          b.wmid[c - c0] = xt[(size_t)e0 + 1].alpha;
          for (int k = 1; k <= b.amin - 2 && ok; ++k) ok = xt[(size_t)(e0 + k)].alpha == b.wmid[c - c0];
          for (int j = 0; j < 4; ++j) b.wtail[c - c0][j] = b.amin - 1 + j < nn ? xt[(size_t)(e0 + b.amin - 1 + j)].alpha : 0.f;
        }
        if (b.amin == b.amax && (b.amin == 16 || b.amin == 32)) {
          b.rotm = b.amin - 1;
          for (int c = 0; c < b.ncell; ++c) b.rot[c] = c * std::max(1, b.amin / b.ncell) & b.rotm;
        }
        {  // plane stride: the fewest lanes per LDS bank over the walk (every lane advances by one column per read)
          int best = 1 << 30;
          for (int cand_tp = 16 * Tc + 8; cand_tp <= 16 * Tc + 8 + 39; ++cand_tp) {
            int cnt[32] = {0};
            for (int im = 0; im < 4; ++im)
              for (int c = 0; c < b.ncell; ++c) cnt[(b.si0[c] + b.rot[c] + im * cand_tp) & 31]++;
            int worst = 0;
            for (int k = 0; k < 32; ++k) worst = std::max(worst, cnt[k]);
            // (ties: prefer strides that also keep the blur's stores -- 16 consecutive columns per image -- apart)
            const int score = worst * 64 + ((cand_tp & 31) == 16 ? 0 : (cand_tp & 15) == 8 ? 1 : 2);
            if (score < best) best = score, b.tp = cand_tp;
          }
        }
        band_of(b.xs, 0, true, b.band[0]);
        band_of(b.xs, 0, false, b.band[1]);
        band_of(b.xs, Tc - 1, false, b.band[2]);
        // every tile between must be the plain band: no image edge within three columns of it
        for (int c = 1; c + 1 < Tc && ok; ++c) {
          unsigned m[64][4];
          memset(m, 0, sizeof m);
          band_of(b.xs, c, false, m);
          ok = memcmp(m, b.band[0], sizeof m) == 0;
        }
        cand.push_back(b);
      }
      if (ok) host.swap(cand);
    }
    if (!host.empty()) {
      for (BaStrip& b : host) d.amax = std::max(d.amax, b.amax);
      hipError_t e = hipMalloc(&d.strips, host.size() * sizeof(BaStrip));
      if (e == hipSuccess) e = hipMemcpy(d.strips, host.data(), host.size() * sizeof(BaStrip), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        if (d.strips) (void)hipFree(d.strips);
        CBH_HIP(e);
      }
      d.n_strips = (int)host.size();
      d.T = host[0].T;
    }
    it = g_ba_tabs.emplace(key, d).first;
  }
  *out = it->second;
  return CBH_OK;
}

}  // namespace

namespace {

int reflect101_host(int p, int len) {
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// The B operand of k_dcthash_256_band: lane l = (column n = l & 15 of the tile, chunk q = l >> 4), byte j <-> k = 16 q + j.
// k < 32: source column 16 c - 8 + k of the row, weight 1 on the 7 taps of output column 16 c + n (an interior tile: the
// kernel mirrors the image's edge columns into the ring); k >= 32: the same columns of the row seven above, weights negated.
void make_band_tables(BandTables* bt) {
  memset(bt, 0, sizeof(*bt));
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 16; ++j) {
      const int k = 16 * (l >> 4) + j, kk = k & 31;
      int wgt = std::abs((kk - 8) - (l & 15)) <= 3 ? 1 : 0;
      if (k >= 32) wgt = -wgt;
      bt->w[0][l][j >> 2] |= (unsigned)(wgt & 0xff) << (8 * (j & 3));
    }
}

struct BandTabCache {
  std::mutex mu;
  std::map<int, BandTables*> d;  // by device ordinal (usable-device masks go up to ordinal 31)
} g_band_tabs;

int get_band_tables(const BandTables** out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_band_tabs.mu);
  if (!g_band_tabs.d.count(dev)) {
    std::vector<BandTables> host(1);
    make_band_tables(host.data());
    BandTables* d = nullptr;
    CBH_HIP(hipMalloc(&d, sizeof(BandTables)));
    hipError_t e = hipMemcpy(d, host.data(), sizeof(BandTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      CBH_HIP(e);
    }
    g_band_tabs.d[dev] = d;
  }
  *out = g_band_tabs.d[dev];
  return CBH_OK;
}

}  // namespace

// "hash_mfma": which kernel hashes 256 x 256 tiles -- non-zero (default 2) k_dcthash_256_band (horizontal box sums on the
// matrix cores; needs 16-byte aligned rows, else 0 is taken), 0 k_dcthash_256 (all VALU: what runs when the band tables
// cannot be made; the parity suite runs both).
int g_hash_mfma = 2;
int g_hash_mfma_set(int v) { return g_hash_mfma = v; }
void set_hash_band_area(int v) { g_hash_band_area = v < 0 ? 0 : v > 2 ? 2 : v; }
int g_hash_fuse = 1;  // "hash_fuse": 1 = k_blur_area_regs<.., FUSE> (vertical pass + tile in the strip kernel) when the batch
                      // gives >= 512 workgroups, 2 = always, 0 = never (k_tile_hash reads the rows back)
void set_hash_fuse(int v) {
  if (v >= 0 && v <= 2) g_hash_fuse = v;
}
// "hash_stream": k_blur_area_regs on strips: 0 never (k_blur_area's 16-row bands for every batch), 1 = when
                        // the batch is large enough (strips of 3..8 steps), v >= 2 = always, strips of v steps (the parity
                        // suite's way to put a handful of images through the strip kernel)
int g_hash_stream = 1;
void set_hash_stream(int v) {
  if (v >= 0) g_hash_stream = v;
}
// rows per step for a workgroup of T threads making ncell cells per row: a turn of the area phase has 2 T / ncell row
// slots, and 14 rows fill a turn of 12 (T = 192), 24, 32, 48 or 64 slots badly -- 21 where that fills the turns better
static int pick_rows_per_step(int K, unsigned T, int ncell, size_t lds_per_row, size_t lds_fixed) {
  if (K != 7) return 15;
  auto fits = [&](int ks) { return lds_fixed + (size_t)ks * lds_per_row <= (size_t)64 * 1024; };
  const int cap = (int)(2 * T) / ncell;
  int best = 14;
  double best_u = 0.0;
  for (int ks : {14, 21}) {  // (28 measured too: slower than 21 everywhere, 15-30 % slower than 14 on one-wave workgroups)
    if (!fits(ks)) break;
    const int turns = (ks + cap - 1) / cap;
    const double u = (double)ks / ((double)turns * cap);
    if (u > best_u + 0.05) best = ks, best_u = u;  // (more rows per step only for a real gain: they cost LDS)
  }
  return best;
}
// steps per strip of k_blur_area_regs: every strip re-reads the 6 rows of blur halo and its last step may be partly
// empty, so the rows PROCESSED depend on how h divides: 960 rows in strips of 8 x 14 = 106 output rows are 10 strips =
// 1120 rows, in strips of 9 x 14 = 120 they are 8 = 1008.  Around the target (which the batch size sets: enough
// workgroups), the count that processes the fewest rows; ties go to the longer strip.  "hash_stream" >= 2 forces its value.
static int pick_steps_per_strip(int h, int ks, int target, int halo) {
  const int all = (h + halo + ks - 1) / ks;  // one strip spanning the image
  if (g_hash_stream >= 2 || target >= all) return std::max(1, std::min(target, all));
  int best = target;
  long long best_rows = -1;
  for (int st = std::max(2, target - 1); st <= std::min(all, target + 3); ++st) {
    const int out = st * ks - halo;
    if (out <= 0) continue;
    const long long rows = (long long)((h + out - 1) / out) * st * ks;
    if (best_rows < 0 || rows <= best_rows) best = st, best_rows = rows;
  }
  return best;
}
// integer ratios, cells of nd = isx / 4 dwords: the 32 lanes of a row group read dword u of their cells together, gcd(nd, 32)
// to a bank -- one pad dword behind every cell of a blurred LDS row where that would be 4 ways or more (512, 1024, 1536 px ...)
static int cell_pad_for(bool integer, int isx) {
  if (!integer || (isx & 7)) return 0;  // (nd even: a blur lane's two dwords stay in one cell)
  int g = 1;
  while (g < 32 && ((isx >> 2) % (2 * g)) == 0) g *= 2;
  return g >= 4 ? 1 : 0;
}

// Host-side table construction (same closed forms as the oracle, computed independently here).
static void make_tables(DctTables* t) {
  // 9x9 zig-zag, first step downwards (equals the table at cvutil.cpp:491-495); keep 6..69
  int zz[81], n = 0;
  for (int s = 0; s <= 16; ++s) {
    if (s & 1) {
      for (int r = (s < 8 ? s : 8); r >= 0 && s - r <= 8; --r) zz[n++] = r * 9 + (s - r);
    } else {
      for (int r = (s > 8 ? s - 8 : 0); r <= 8 && r <= s; ++r) zz[n++] = r * 9 + (s - r);
    }
  }
  for (int i = 0; i < 64; ++i) t->zz[i] = (unsigned char)zz[6 + i];
  cv_dct32_make_tabs(&t->cv);
}

int get_tables(const DctTables** out) {
  int dev = 0;
  CBH_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return CBH_E_INVAL;
  std::lock_guard<std::mutex> lk(g_tabs.mu);
  if (!g_tabs.d[dev]) {
    DctTables host;
    make_tables(&host);
    DctTables* d = nullptr;
    CBH_HIP(hipMalloc(&d, sizeof(DctTables)));
    hipError_t e = hipMemcpy(d, &host, sizeof(DctTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      CBH_HIP(e);
    }
    g_tabs.d[dev] = d;
  }
  *out = g_tabs.d[dev];
  return CBH_OK;
}

int launch_dcthash(const uint8_t* d_imgs, size_t n, int w, int h, size_t row_stride,
                   size_t img_stride, uint64_t* d_out, hipStream_t stream, uint8_t* d_tiles, const HashView* view) {
  if (n == 0) return CBH_OK;
  if (w <= 0 || h <= 0 || row_stride < (size_t)w) return CBH_E_INVAL;
  if (w > 8192 || h > 8192) return CBH_E_UNSUPPORTED;
  if (n > 0x7fffffffull) return CBH_E_INVAL;
  // a view: the w x h images are sub-rectangles at (ox, oy) of pw x ph parents that start at d_imgs (+ i*img_stride);
  // the blur takes its border from the parent.  Whole images: the view is the image.
  HashView vw{w, h, 0, 0};
  if (view) {
    vw = *view;
    if (vw.ox < 0 || vw.oy < 0 || vw.ox + w > vw.pw || vw.oy + h > vw.ph || row_stride < (size_t)vw.pw) return CBH_E_INVAL;
    if (vw.ox == 0 && vw.oy == 0 && vw.pw == w && vw.ph == h) view = nullptr;
  }
  if (view && w == 32 && h == 32)  // no blur: nothing is read outside the view
    return launch_dcthash(d_imgs + (size_t)vw.oy * row_stride + vw.ox, n, w, h, row_stride, img_stride, d_out, stream,
                          d_tiles, nullptr);
  if (w < 32 || h < 32 || (w == 32 && h == 32)) {
    // a side enlarges (cv::resize's bilinear emulation), or nothing is resized at all: the rectangle kernel, one
    // rectangle = the whole image
    const size_t per_chunk = (size_t)1 << 20;
    for (size_t i0 = 0; i0 < n; i0 += per_chunk) {
      const size_t m = std::min(per_chunk, n - i0);
      std::vector<RectImageDesc> images(m);
      std::vector<int> rects(4 * m);
      for (size_t i = 0; i < m; ++i) {
        images[i] = RectImageDesc{(unsigned long long)(i * img_stride), vw.pw, vw.ph, (unsigned)row_stride, (unsigned)i,
                                  1u};
        rects[4 * i] = vw.ox, rects[4 * i + 1] = vw.oy, rects[4 * i + 2] = w, rects[4 * i + 3] = h;
      }
      int rc2 = launch_rect_hashes(const_cast<uint8_t*>(d_imgs) + i0 * img_stride, images, rects, 0, d_out + i0, stream,
                                   d_tiles ? d_tiles + i0 * 1024 : nullptr);
      if (rc2) return rc2;
    }
    return CBH_OK;
  }
  const DctTables* tabs = nullptr;
  int rc = get_tables(&tabs);
  if (rc) return rc;
  // ---- 256 x 256 (8-byte aligned rows; anything else of that size takes the general kernels)
  if (!view && w == 256 && h == 256 && ((uintptr_t)d_imgs % 8) == 0 && row_stride % 8 == 0 && img_stride % 8 == 0 &&
      row_stride * 256 < (1u << 24) && img_stride < (1u << 28)) {
    const BandTables* btab = nullptr;
    if (g_hash_mfma != 0 && ((uintptr_t)d_imgs % 16) == 0 && row_stride % 16 == 0 && img_stride % 16 == 0 &&
        get_band_tables(&btab) == CBH_OK) {  // (no table on this device: the all-VALU kernel needs none)
      dim3 gridb((unsigned)((n + 3) / 4));
      if (d_tiles)
        hipLaunchKernelGGL(k_dcthash_256_band<true>, gridb, dim3(64), 0, stream, d_imgs, (unsigned)n, (unsigned)row_stride,
                           (unsigned)img_stride, tabs, btab, d_out, d_tiles);
      else
        hipLaunchKernelGGL(k_dcthash_256_band<false>, gridb, dim3(64), 0, stream, d_imgs, (unsigned)n, (unsigned)row_stride,
                           (unsigned)img_stride, tabs, btab, d_out, d_tiles);
    } else {
      cbh_clear_error();
      dim3 grid((unsigned)((n + 7) / 8)), block(kThreads);
      if (d_tiles)
        hipLaunchKernelGGL(k_dcthash_256<true>, grid, block, 0, stream, d_imgs, (unsigned)n, (unsigned)row_stride,
                           (unsigned)img_stride, tabs, d_out, d_tiles);
      else
        hipLaunchKernelGGL(k_dcthash_256<false>, grid, block, 0, stream, d_imgs, (unsigned)n, (unsigned)row_stride,
                           (unsigned)img_stride, tabs, d_out, d_tiles);
    }
    CBH_HIP(hipGetLastError());
    return CBH_OK;
  }
  // ---- every other geometry
  const long long area_ = (long long)w * h;
  const int K_ = area_ <= 64 * 64 ? 3 : area_ <= 128 * 128 ? 5 : 7;  // (area <= 32*32 is only 32x32 itself)
  AreaTabsDev at;
  if ((rc = get_area_tabs(w, h, &at))) return rc;
  const bool integer = area_fast(w, h);
  const int isx = integer ? w / 32 : 0, isy = integer ? h / 32 : 0;
  // (views: a vertical edge is the parent's, or lies far enough inside it that the blur's three columns -- and on the left
  // the staging chunk's eight -- are the parent's pixels: letterboxed and pillarboxed frames after autocrop)
  // (integer ratios: every width but 1024 -- strips of 128 columns on rows a multiple of the cache line apart put every row's
  // 144 staged bytes across three lines, 2.84 TB/s against k_blur_area_regs' 3.1; profiles/r06_band_area_int.txt)
  const bool ba_view = !view || ((vw.ox == 0 || vw.ox >= 8) && (vw.ox + w == vw.pw || vw.ox + w + 3 <= vw.pw));
  if (g_hash_band_area && ba_view && K_ == 7 && (integer ? isx >= 2 && w != 1024 : at.yrow != nullptr) && w >= 64 &&
      4ull * img_stride < (1ull << 32) && (size_t)vw.ph * row_stride < ((size_t)1 << 32)) {
    BaTabsDev bat;
    if (get_ba_tabs(w, &bat) != CBH_OK) {  // (its table could not be made: k_blur_area_regs needs none)
      cbh_clear_error();
      bat.strips = nullptr;
    }
    if (bat.strips) {
      const size_t per_chunk_b = 200000;  // (grid y)
      unsigned char* d_btiles = nullptr;
      cbh::Scratch scratch(stream);
      CBH_HIP(scratch.get(&d_btiles, std::min(per_chunk_b, n) * 1024));
      for (size_t i0 = 0; i0 < n; i0 += per_chunk_b) {
        const size_t m = std::min(per_chunk_b, n - i0);
        const unsigned char* src = d_imgs + i0 * img_stride;
        const unsigned long long bytes = (unsigned long long)(m - 1) * img_stride + (unsigned long long)(vw.ph - 1) * row_stride + (unsigned)vw.pw;
        // row bands: enough waves for ~two rounds of the machine's 2560 slots, at least 4 output rows per band
        BaBands bands;
        bands.n = 1;
        while (bands.n < 8 && (m + 3) / 4 * (size_t)bat.n_strips * (size_t)bands.n < 5120) bands.n *= 2;
        if (integer) {
          for (int b_ = 0; b_ < bands.n; ++b_) {
            const int c0_ = 32 * b_ / bands.n, c1_ = 32 * (b_ + 1) / bands.n;
            bands.c0[b_] = c0_, bands.c1[b_] = c1_;
            bands.ra[b_] = c0_ * isy, bands.rb[b_] = c1_ * isy - 1;
          }
        } else {
          std::vector<int> yf;
          const std::vector<AreaTab> yt = make_area_tab(h, 32, &yf);
          for (int b_ = 0; b_ < bands.n; ++b_) {
            const int c0_ = 32 * b_ / bands.n, c1_ = 32 * (b_ + 1) / bands.n;
            bands.c0[b_] = c0_, bands.c1[b_] = c1_;
            bands.ra[b_] = yt[(size_t)yf[(size_t)c0_]].si;
            bands.rb[b_] = yt[(size_t)yf[(size_t)c1_] - 1].si;
          }
        }
#define CBH_BA_(TT, RR)                                                                                                  \
  if (integer) CBH_BA__(TT, RR, true); else CBH_BA__(TT, RR, false)
#define CBH_BA__(TT, RR, II)                                                                                             \
  hipLaunchKernelGGL((k_band_area<TT, RR, II>), dim3((unsigned)(((m + 3) / 4 + 7) / 8 * 8 * (size_t)bat.n_strips * (size_t)bands.n)), dim3(64), 0, \
                     stream, src, (unsigned)m, w, h, (unsigned)row_stride, (unsigned)img_stride, bytes, bat.strips,      \
                     bat.n_strips, at.yrow, d_btiles, view ? vw.oy : 0, view ? vw.ph : h, view ? vw.ox : 0,                \
                     view ? (vw.ox > 0 ? 1 : 0) | (vw.ox + w < vw.pw ? 2 : 0) : 0, bands, isy)
#define CBH_BA(TT)                         \
  case TT:                                 \
    if (bat.RS == 4) { CBH_BA_(TT, 4); }   \
    else if (bat.RS == 2) { CBH_BA_(TT, 2); } \
    else { CBH_BA_(TT, 1); }               \
    break
        switch (bat.T) {
          CBH_BA(2); CBH_BA(3); CBH_BA(4); CBH_BA(5); CBH_BA(6); CBH_BA(7); CBH_BA(8); CBH_BA(9); CBH_BA(10); CBH_BA(11);
          CBH_BA(12); CBH_BA(13); CBH_BA(14); CBH_BA(15);
          default: return CBH_E_UNSUPPORTED;
        }
#undef CBH_BA__
#undef CBH_BA_
#undef CBH_BA
        hipLaunchKernelGGL(k_tiles_hash2, dim3((unsigned)((m + 1) / 2)), dim3(64), 0, stream, d_btiles, (unsigned)m, tabs,
                           d_out + i0, d_tiles ? d_tiles + i0 * 1024 : nullptr);
        CBH_HIP(hipGetLastError());
      }
      return CBH_OK;
    }
  }
  // ---- the VALU kernels: k_blur_area_regs on strips (batches that fill the machine that way), k_blur_area on 16-row bands
  // column strips: the fewest (1, 2, 4, 8) whose widest window -- first source column of a strip's first cell .. last of
  // its last -- fits 256 lanes x 8 columns.  (Fractional cells overlap by a pixel: 8191 columns need 8 strips.)
  int ncol = w <= 2048 ? 1 : w <= 4096 ? 2 : 4, cpw = 32 / ncol, win = 0;
  std::vector<int> xf;
  const std::vector<AreaTab> xt_full = integer ? std::vector<AreaTab>() : make_area_tab(w, 32, &xf);
  for (;; ncol *= 2) {
    cpw = 32 / ncol;
    win = 0;
    if (integer) {
      win = cpw * isx;
    } else {
      for (int c = 0; c < 32; c += cpw)
        win = std::max(win, xt_full[(size_t)xf[(size_t)(c + cpw)] - 1].si + 1 - xt_full[(size_t)xf[(size_t)c]].si);
    }
    if ((win + 7) / 8 <= 256 || ncol == 8) break;
  }
  const int Tf = std::min(256, ((win + 7) / 8 + 63) / 64 * 64);
  if ((win + 7) / 8 > 256) return CBH_E_UNSUPPORTED;  // cannot happen for w <= 8192
  const int fpitch = Tf * 8 + 8;
  const size_t fsmem = (size_t)(kBlurRB + K_ - 1) * fpitch + (size_t)(integer ? 0 : at.xn) * sizeof(float);
  const size_t per_chunk_f = std::max<size_t>(1, ((size_t)1 << 30) / ((size_t)h * 128));
  float* d_rowsf = nullptr;
  unsigned char* d_ftiles = nullptr;  // the fused kernel's 32 x 32 tiles (1 KB per image)
  cbh::Scratch scratch(stream);
  CBH_HIP(scratch.get(&d_rowsf, std::min(per_chunk_f, n) * (size_t)h * 32 * sizeof(float)));
  CBH_HIP(scratch.get(&d_ftiles, std::min(per_chunk_f, n) * 1024));
  for (size_t i0 = 0; i0 < n; i0 += per_chunk_f) {
    const size_t m = std::min(per_chunk_f, n - i0);
    const unsigned char* src = d_imgs + i0 * img_stride;
    unsigned char* tcopy = d_tiles ? d_tiles + i0 * 1024 : nullptr;
    // strips of `steps` steps when the batch leaves enough workgroups ("hash_stream": 0 never, >= 2 always, of that many)
    const int kstep = K_ == 7 ? 14 : 15;
    const long long band_wgs = (long long)ncol * ((h + kBlurRB - 1) / kBlurRB) * (long long)m;
    int steps = (int)std::min<long long>(8, band_wgs / 3072);
    steps = std::min(steps, (h + 2 * (K_ / 2) + kstep - 1) / kstep);
    if (g_hash_stream >= 2) steps = g_hash_stream;
    if (!g_hash_stream) steps = 0;
    // a view whose vertical edges are the parent's or lie >= 4 pixels inside it (letterboxed / pillarboxed frames
    // after autocrop) differs from a whole image in the row mapping and in which lanes mirror: the
    // register-streaming kernel takes it; other views (a margin of 1..3 pixels) stay on the band kernel
    const int Lv = (w + 7) / 8;
    const bool lbox = view && (vw.ox == 0 || vw.ox >= 4) && (vw.ox + w == vw.pw || vw.ox + 8 * Lv + 4 <= vw.pw);
    if (steps >= 3 && ncol == 1 && (!view || lbox) && (size_t)vw.ph * row_stride < ((size_t)1 << 31)) {
      const int v_oy = view ? vw.oy : 0, v_ph = view ? vw.ph : 0, v_ox = view ? vw.ox : 0, v_pw = view ? vw.pw : 0;
      const bool v_ri = view && vw.ox + 8 * Lv + 4 <= vw.pw;  // the last lane reads real pixels: any w mod 8
      const bool gen = !((w % 8 == 0 || v_ri) && ((uintptr_t)(src + v_ox) % 8) == 0 && row_stride % 8 == 0 &&
                         img_stride % 8 == 0);
      // lanes per image = w / 8.  Images whose last wave would be mostly empty share a 256-lane workgroup side by
      // side (640 px: 80 of 128 lanes busy alone, 240 of 256 as three: +12 %) where that puts 15 % more of the lanes to
      // work (400 px: 50 of 64 alone, 250 of 256 as five: +3..4 %; 720 / 800 px: two images fill 256 lanes no better than one
      // fills 128: -2 % packed); where the lanes are already well used the larger workgroup only costs (more waves per
      // barrier: -5..-9 % measured at 400, 512, 1024 px).  Integer ratios (cheap area phase: the blur lanes decide) count
      // the packed workgroup's own size -- 704 x 576: 176 of 192 lanes as two against 88 of 128 alone, +10 % --
      // fractional ones count it as 256 lanes: 720 x 540 and 688 x 516 run 4-5 % faster alone in 128 lanes than as two in 192.
      const int Lr = (w + 7) / 8, Lw = (Lr + 63) / 64 * 64;
      const int ipb_try = Lr <= 128 ? 256 / Lr : 1;
      const int Tp = integer ? (ipb_try * Lr + 63) / 64 * 64 : 256;
      const bool pack = ipb_try > 1 && (long long)ipb_try * Lr * Lw * 100 >= 115LL * Lr * Tp;
      const int ipb = (pack && (unsigned long long)ipb_try * img_stride < (1ull << 32)) ? ipb_try : 1;
      const size_t k_end_r = integer ? 0 : (((size_t)at.xn + 3) & ~(size_t)3) + 512;  // weights + per-cell edge weights
      const int cpad = cell_pad_for(integer, isx);
      const size_t rowb = (size_t)(8 * Lr) + (cpad ? 128 : 0);  // LDS bytes of one blurred row
      const unsigned Tr = (unsigned)std::max(64, (ipb * Lr + 63) / 64 * 64);
      // rows per step (14 / 21) by how they fill the area phase's turns; the strips keep their length in rows
      const int ks_r = pick_rows_per_step(K_, Tr, 32, (size_t)ipb * rowb, k_end_r * sizeof(float) + 16);
      const int steps_r = pick_steps_per_strip(h, ks_r, (steps * kstep + ks_r - 1) / ks_r, 2 * (K_ / 2));
      const int strip_out_r = steps_r * ks_r - 2 * (K_ / 2);
      const unsigned gsy_r = (unsigned)((h + strip_out_r - 1) / strip_out_r);
      const size_t rsmem = (size_t)ipb * ks_r * rowb + k_end_r * sizeof(float) +
                           16;  // the area walk reads whole words: up to 7 bytes past the last blurred row
      // whole image per workgroup, vertical pass and tile inside the kernel (FUSE) when the batch still fills
      // the machine that way: at least two workgroups per CU
      const int ipb_f = std::min(ipb, 8);
      const size_t k_end_f = k_end_r;
      const unsigned Tf_ = (unsigned)std::max(64, (ipb_f * std::max(Lr, 32) + 63) / 64 * 64);
      const int ks_f = pick_rows_per_step(K_, std::min(256u, Tf_), 32, (size_t)ipb_f * (rowb + 32 * sizeof(float)),
                                          k_end_f * sizeof(float) + (size_t)ipb_f * 1024);
      const int steps_f = (h + 2 * (K_ / 2) + ks_f - 1) / ks_f;
      const size_t fsm = (size_t)ipb_f * ks_f * rowb + k_end_f * sizeof(float) +
                         (size_t)ipb_f * ks_f * 32 * sizeof(float) + (size_t)ipb_f * 1024;
      // (measured, hash_fuse 0 -> 2: 320x240 +30 %, 400x300 +22 %, 533x400 +19 %, 640x480 +8 %, 800x600 +6 %,
      // 1366x768 +6 %, 1024x768 -2 %, 1280x960 -8 %, 1080p -7 %: large images spend little in k_tile_hash and
      // lose occupancy to the extra LDS; fractional ratios gain up to ~1 MP)
      // a fused workgroup walks its whole image alone: with one or two waves per workgroup the machine needs
      // thousands of them before that beats strips of 8 steps (tools/ab/hash_small_batches.py: 400x300, one wave per
      // image, 1024 images 137 us fused / 86 split, 2048: 160 / 148, 4096: 289 / 298; 800x600, two waves:
      // 2048 images 553 / 483, 4096: 945 / 951; four-wave workgroups of 3-6 images pay from ~600 up)
      const size_t fuse_min_wgs = Tf_ <= 64 ? 4096 : Tf_ <= 128 ? 3072 : 512;
      const bool fuse = g_hash_fuse && fsm <= 160 * 1024 - 1024 && (integer || at.yrow) &&
                        (g_hash_fuse >= 2 ||
                         ((m + (size_t)ipb_f - 1) / (size_t)ipb_f >= fuse_min_wgs &&
                          (size_t)w * (size_t)h <= (integer ? 1000000u : 1100000u)));
#define CBH_REGS_L(KK, GG, KSV)                                                                              \
  do {                                                                                                       \
    if (fuse) {                                                                                              \
      if (fsm > 64 * 1024)                                                                                   \
        CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_area_regs<KK, GG, true, KSV>),      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsm));                  \
      hipLaunchKernelGGL((k_blur_area_regs<KK, GG, true, KSV>), dim3(1, 1, (unsigned)((m + ipb_f - 1) / ipb_f)), \
                         dim3(std::min(256u, Tf_)), fsm, stream, src, w, h, (unsigned)row_stride, img_stride, at.x,       \
                         at.xfirst, isx, steps_f, (float*)nullptr, ipb_f, (unsigned)m, at.yrow, isy, d_ftiles,       \
                         v_oy, v_ph, v_ox, v_pw, 0, 32, cpad);                                               \
      break;                                                                                                 \
    }                                                                                                        \
    if (rsmem > 64 * 1024)                                                                                   \
      CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_area_regs<KK, GG, false, KSV>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)rsmem));                  \
    hipLaunchKernelGGL((k_blur_area_regs<KK, GG, false, KSV>), dim3(1, gsy_r, (unsigned)((m + ipb - 1) / ipb)), dim3(Tr), rsmem, \
                       stream, src, w, h, (unsigned)row_stride, img_stride, at.x, at.xfirst, isx, steps_r, d_rowsf, \
                       ipb, (unsigned)m, (const YRow*)nullptr, 0, (unsigned char*)nullptr, v_oy, v_ph, v_ox, \
                       v_pw, 0, 32, cpad);                                                                   \
  } while (0)
#define CBH_REGS_(KK, GG)                                    \
  do {                                                       \
    const int ks_ = fuse ? ks_f : ks_r;                      \
    if (KK == 7 && ks_ == 21) CBH_REGS_L(7, GG, 21);         \
    else CBH_REGS_L(KK, GG, StreamK<KK>::step);              \
  } while (0)
#define CBH_REGS(KK)              \
  do {                            \
    if (gen) CBH_REGS_(KK, true); \
    else CBH_REGS_(KK, false);    \
  } while (0)
      switch (K_) {
        case 3: CBH_REGS(3); break;
        case 5: CBH_REGS(5); break;
        default: CBH_REGS(7); break;
      }
#undef CBH_REGS_L
#undef CBH_REGS_
#undef CBH_REGS
      if (fuse)
        hipLaunchKernelGGL(k_tiles_hash2, dim3((unsigned)((m + 1) / 2)), dim3(64), 0, stream, d_ftiles, (unsigned)m, tabs,
                           d_out + i0, tcopy);
      else
        hipLaunchKernelGGL(k_tile_hash, dim3((unsigned)m), dim3(kThreads), 0, stream, d_rowsf, h, at.y, at.yfirst,
                           isx, isy, 1, tabs, d_out + i0, tcopy);
      continue;
    }
    // images wider than 2048 pixels: the register-streaming kernel on ncol column strips, each a view of the
    // parent that makes its share of the 32 output cells
    if (steps >= 3 && ncol > 1 && !view && K_ == 7 && (size_t)h * row_stride < ((size_t)1 << 31)) {
      StripTabs stt[8];
      bool ok = true;
      for (int sidx = 0; sidx < ncol && ok; ++sidx) {
        if ((rc = get_strip_tabs(w, integer, ncol, sidx, &stt[sidx]))) return rc;
        const StripTabs& S_ = stt[sidx];
        const int Ls = (S_.ws + 7) / 8;
        ok = S_.ws >= 64 && Ls <= 256 && (S_.x0 == 0 || S_.x0 >= 4) &&
             (S_.x0 + S_.ws == w || S_.x0 + 8 * Ls + 4 <= w);
      }
      if (ok) {
        for (int sidx = 0; sidx < ncol; ++sidx) {
          const StripTabs& S_ = stt[sidx];
          const int ws = S_.ws, Ls = (ws + 7) / 8;
          const bool s_ri = S_.x0 + 8 * Ls + 4 <= w;
          const bool gen = !((ws % 8 == 0 || s_ri) && ((uintptr_t)(src + S_.x0) % 8) == 0 && row_stride % 8 == 0 &&
                             img_stride % 8 == 0);
          const size_t k_end_s = integer ? 0 : (((size_t)S_.xn + 3) & ~(size_t)3) + 512;
          const int cpad_s = cell_pad_for(integer, isx);
          const unsigned Ts = (unsigned)std::max(64, (Ls + 63) / 64 * 64);
          // a strip makes cpw of the 32 cells: 2 Ts / cpw row slots per turn of the area phase, 14 rows fill them badly
          const int ks_s = pick_rows_per_step(7, Ts, cpw, (size_t)(8 * Ls + (cpad_s ? 128 : 0)), k_end_s * sizeof(float) + 16);
          const int steps_s = pick_steps_per_strip(h, ks_s, (steps * kstep + ks_s - 1) / ks_s, 6);
          const int strip_out_s = steps_s * ks_s - 6;
          const unsigned gsy_s = (unsigned)((h + strip_out_s - 1) / strip_out_s);
          const size_t smem_s = (size_t)ks_s * (size_t)(8 * Ls + (cpad_s ? 128 : 0)) + k_end_s * sizeof(float) + 16;
#define CBH_STRIP_L(GG, KSV)                                                                                      \
  do {                                                                                                            \
    if (smem_s > 64 * 1024)                                                                                       \
      CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_area_regs<7, GG, false, KSV>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s));                      \
    hipLaunchKernelGGL((k_blur_area_regs<7, GG, false, KSV>), dim3(1, gsy_s, (unsigned)m), dim3(Ts), smem_s, stream, src, \
                       ws, h, (unsigned)row_stride, img_stride, integer ? at.x : S_.x, integer ? at.xfirst : S_.xfirst, \
                       isx, steps_s, d_rowsf, 1, (unsigned)m, (const YRow*)nullptr, 0, (unsigned char*)nullptr, 0, h,  \
                       S_.x0, w, sidx * cpw, cpw, cpad_s);                                                        \
  } while (0)
#define CBH_STRIP(GG)                               \
  do {                                              \
    if (ks_s == 21) CBH_STRIP_L(GG, 21);            \
    else CBH_STRIP_L(GG, 14);                       \
  } while (0)
          if (gen) CBH_STRIP(true);
          else CBH_STRIP(false);
#undef CBH_STRIP_L
#undef CBH_STRIP
        }
        hipLaunchKernelGGL(k_tile_hash, dim3((unsigned)m), dim3(kThreads), 0, stream, d_rowsf, h, at.y, at.yfirst,
                           isx, isy, 1, tabs, d_out + i0, tcopy);
        continue;
      }
    }
    // everything else -- small batches, views with a margin of 1..3 pixels, views of more than 2048 columns: a workgroup
    // per 16-row band, rows staged in LDS
    dim3 gf((unsigned)ncol, (unsigned)((h + kBlurRB - 1) / kBlurRB), (unsigned)m);
#define CBH_FUSED(KK)                                                                                    \
  do {                                                                                                   \
    if (fsmem > 64 * 1024)                                                                               \
      CBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_blur_area<KK>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)fsmem));              \
    hipLaunchKernelGGL(k_blur_area<KK>, gf, dim3((unsigned)Tf), fsmem, stream, src, w, h, row_stride,    \
                       img_stride, at.x, at.xfirst, isx, cpw, fpitch, d_rowsf, vw.pw, vw.ph, vw.ox, vw.oy);  \
  } while (0)
    switch (K_) {
      case 3: CBH_FUSED(3); break;
      case 5: CBH_FUSED(5); break;
      default: CBH_FUSED(7); break;
    }
#undef CBH_FUSED
    hipLaunchKernelGGL(k_tile_hash, dim3((unsigned)m), dim3(kThreads), 0, stream, d_rowsf, h, at.y, at.yfirst, isx,
                       isy, 1, tabs, d_out + i0, tcopy);
  }
  CBH_HIP(hipGetLastError());
  return CBH_OK;
}

}  // namespace cbh
