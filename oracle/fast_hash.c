/* fast_hash.c -- TEST INFRASTRUCTURE / CPU BASELINE, not product: dctHash64 of 256 x 256 8-bit images (the BASELINE
 * geometry) written the way a tuned CPU library does stages 1-2, so that bench.py's `cpu_baseline.hash_images_per_s`
 * is not the figure of a scalar per-pixel port.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may call this.
 *
 * Same arithmetic as oracle/cbird_oracle.c (and so the same hashes, which tests/test_oracle.py checks image by image):
 *   cv::blur 7x7, BORDER_REFLECT_101 (src/cvutil.cpp:463): box sums are exact integers whatever the order, so they are
 *     formed as running COLUMN sums (add the entering row, subtract the leaving one: two vector ops per row of 256
 *     columns) followed by seven shifted adds along the row; the rounded mean nearest(S / 49) is the exact
 *     multiply-shift ((S + 24) * 342393) >> 24 (49 is odd: no ties);
 *   cv::resize(-> 32 x 32, INTER_AREA), integer ratio 8 (src/cvutil.cpp:471): exact 8 x 8 block sums,
 *     rint-half-even(float(sum) * (1.f / 64));
 *   stages 3-6 from the tile: orc_hash_from_tile32 (oracle/cbird_oracle.c), unchanged.
 * The loops are plain C over uint16_t / uint32_t arrays; gcc -O3 vectorises them, and target_clones builds an AVX2
 * version next to the portable one (the host picks at load time).  No reference source is involved. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint64_t orc_hash_from_tile32(const uint8_t* tile, float* coefs, float* thr_out);

#define W 256
#define H 256
#define R 3

static inline int reflect101(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); }

__attribute__((target_clones("avx2", "default"))) static void tile_256(const uint8_t* img, size_t stride,
                                                                       uint8_t* tile /* 32 x 32 */) {
  uint16_t cs[W + 2 * R]; /* column sums of the current 7-row window, with the reflected columns on both sides */
  uint16_t colacc[W];     /* blurred pixels of the current cell row, summed over its rows so far */
  uint16_t* c = cs + R;
  memset(c, 0, W * sizeof(uint16_t));
  for (int dy = -R; dy <= R; ++dy) { /* window of output row 0 */
    const uint8_t* row = img + (size_t)reflect101(dy, H) * stride;
    for (int x = 0; x < W; ++x) c[x] = (uint16_t)(c[x] + row[x]);
  }
  memset(colacc, 0, sizeof colacc);
  for (int y = 0; y < H; ++y) {
    if (y > 0) { /* slide the window down by one row */
      const uint8_t* add = img + (size_t)reflect101(y + R, H) * stride;
      const uint8_t* sub = img + (size_t)reflect101(y - R - 1, H) * stride;
      for (int x = 0; x < W; ++x) c[x] = (uint16_t)(c[x] + add[x] - sub[x]);
    }
    for (int i = 1; i <= R; ++i) { /* REFLECT_101 along the row: column sums mirror like the pixels do */
      cs[R - i] = c[i];
      cs[R + W - 1 + i] = c[W - 1 - i];
    }
    for (int x = 0; x < W; ++x) {
      const uint32_t s = (uint32_t)cs[x] + cs[x + 1] + cs[x + 2] + cs[x + 3] + cs[x + 4] + cs[x + 5] + cs[x + 6];
      colacc[x] = (uint16_t)(colacc[x] + (((s + 24u) * 342393u) >> 24)); /* nearest(s / 49) */
    }
    if ((y & 7) == 7) { /* a row of 8 x 8 cells is complete */
      uint8_t* t = tile + (y >> 3) * 32;
      for (int cx = 0; cx < 32; ++cx) {
        uint32_t s = 0;
        for (int i = 0; i < 8; ++i) s += colacc[cx * 8 + i];
        const float v = rintf((float)s * (1.f / 64.f)); /* resizeAreaFast_: rint(sum * (1.f / area)), half to even */
        t[cx] = (uint8_t)(v > 255.f ? 255.f : v);
      }
      memset(colacc, 0, sizeof colacc);
    }
  }
}

/* n images of 256 x 256; hashes out.  Returns 0. */
int orc_dcthash64_fast256_batch(const uint8_t* imgs, size_t n, size_t row_stride, size_t img_stride, uint64_t* out) {
  for (size_t i = 0; i < n; ++i) {
    uint8_t tile[1024];
    tile_256(imgs + i * img_stride, row_stride, tile);
    out[i] = orc_hash_from_tile32(tile, NULL, NULL);
  }
  return 0;
}

int orc_tile32_fast256(const uint8_t* img, size_t row_stride, uint8_t* tile) {
  tile_256(img, row_stride, tile);
  return 0;
}
