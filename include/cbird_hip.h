/* cbird_hip.h -- C-ABI of libcbird_hip.so: cbird's perceptual-hash build + Hamming find hot
 * path on AMD MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary: plain pointers and sizes only, no C++/torch/Qt types.  Each
 * entry point names the reference interface it replaces (paths relative to the cbird source
 * tree, v0.8.1).  A cbird maintainer binds these from a thin `Index` subclass -- see
 * INTEGRATION.md and cbird_amd/cpp/gpu_dcthashindex.h.
 *
 * Conventions
 *  - return 0 (CBH_OK) or a negative CBH_E_* code; nothing aborts, nothing throws.
 *  - the caller owns every buffer it passes; the library copies what it keeps.
 *  - "host" entry points take host pointers and perform the transfers themselves;
 *    "*_dev" entry points take device pointers (hipMalloc'ed memory of the index's device)
 *    and enqueue on `stream` (a hipStream_t passed as void*; NULL = the library's stream for
 *    that call, synchronised before return).
 *  - find/find_batch/scan may be called concurrently from many host threads on one index
 *    (cbird calls Index::find from QThreadPool workers under a read lock,
 *    src/database.cpp:1400-1432,1698); load/add/remove/destroy need exclusive access
 *    (cbird holds the write lock there, src/database.cpp:371,544,1673).
 *  - there is NO CPU fallback: without a usable gfx950 device every compute entry point
 *    returns CBH_E_NODEVICE.
 */
#ifndef CBIRD_HIP_H
#define CBIRD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBH_VERSION 100 /* 0.1.0 */

enum {
  CBH_OK = 0,
  CBH_E_INVAL = -1,       /* bad argument */
  CBH_E_UNSUPPORTED = -2, /* valid in cbird, not implemented by this build (e.g. image size) */
  CBH_E_NODEVICE = -3,    /* no HIP device / kernel image not loadable */
  CBH_E_NOMEM = -4,       /* host or device allocation failed */
  CBH_E_HIP = -5,         /* HIP runtime error, see cbh_last_error() */
  CBH_E_OVERFLOW = -6,    /* result does not fit the record buffer; see cbh_idx64_set_record_capacity */
  CBH_E_NOTLOADED = -7    /* index used before load (src/index.h:233-237) */
};

/* Index::Match without the MatchRange (src/index.h:157-166); score = Hamming distance. */
typedef struct cbh_match {
  uint32_t id;   /* mediaId */
  int32_t score; /* lower is better */
} cbh_match;

/* Raw scan record, one per (query, haystack entry) pair under threshold:
 *   bits 63..39  query index within the call (< 2^25)
 *   bits 38..32  Hamming distance (0..64)
 *   bits 31..0   mediaId
 * so that an ascending u64 sort yields cbird's order: query, then score, then mediaId. */
typedef uint64_t cbh_record;
#define CBH_REC_QUERY(r) ((uint32_t)((r) >> 39))
#define CBH_REC_DIST(r) ((int)(((r) >> 32) & 0x7f))
#define CBH_REC_ID(r) ((uint32_t)((r)&0xffffffffu))
#define CBH_MAX_QUERIES_PER_CALL (1u << 25)

typedef struct cbh_idx64 cbh_idx64; /* opaque: DctHashIndex state on one device */
typedef struct cbh_idx256 cbh_idx256; /* opaque: CvFeaturesIndex state */

/* ---- library ------------------------------------------------------------------------- */
int cbh_version(void);
int cbh_device_count(void);             /* number of usable gfx950 devices, 0 if none */
/* bit d set for every HIP ordinal d < 32 that is a usable gfx950 device (the ordinals need not be contiguous: a node
 * may list another device first) -- what GpuDeviceSet::all() / cbh_idx64_create_sharded take */
uint32_t cbh_usable_device_mask(void);
const char* cbh_strerror(int code);     /* static string */
const char* cbh_last_error(void);       /* thread-local detail of the last failure (HIP call and reason) */
/* the CBH_E_* that goes with cbh_last_error() -- for the entry points that return a handle (create / slice): NULL
 * says "failed", this says whether it was CBH_E_NOMEM (transient: cbh_trim and try again) or something else */
int cbh_last_error_code(void);
/* Forget the thread's last error (text and code).  The entry points that return a handle do this themselves on entry;
 * a caller that decides on cbh_last_error_code() after an int-returning call (the adapters' retry-after-trim rule,
 * cbird_amd/cpp/gpu_errors.h) clears first so that the code it reads belongs to that call. */
void cbh_clear_error(void);
/* Scratch memory.  Kernel scratch comes from the library's own stream-ordered arena: hipMalloc'ed blocks cached per
 * (device, stream) -- a freed block is reused only by the stream that freed it, so the next call on that stream finds
 * its buffers mapped.  A stream's cache lives as long as the stream: when the library destroys one of its own streams,
 * or finds a caller's stream gone or idle while more than 32 streams have caches, the blocks move to a per-device
 * list any stream may take from, of which at most "pool_keep_mb" (default 16384) stay cached; a live stream's own
 * cache holds at most "pool_live_keep_mb" (default: a quarter of the device's memory).  cbh_trim synchronises
 * `device` and returns EVERY cached block to the driver; *released_bytes (optional) = what that gave back.  Safe to
 * call at any time between calls. */
int cbh_trim(int device, unsigned long long* released_bytes);

/* ---- hash build: replaces dctHash64(const cv::Mat&, bool) -- src/cvutil.cpp:435-545,
 * called once per image from Scanner::processImage (src/scanner.cpp:862).
 * imgs: n 8-bit single-channel images (cv::Mat CV_8UC1 after grayscale()), image i at
 * imgs + i*img_stride, row y at + y*row_stride.  Supported geometry: 1 <= w,h <= 8192 (cv::resize INTER_AREA's
 * integer-ratio and weighted paths; with a side < 32 its bilinear emulation for enlarging axes); larger:
 * CBH_E_UNSUPPORTED.  out[i] = 64-bit hash (bit 0 clear unless the hash would be 0 -> 1). */
int cbh_dcthash_batch(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride,
                      size_t img_stride, uint64_t* out, int device);
int cbh_dcthash_batch_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride, void* d_out, int device, void* stream);

/* ---- keypoint hashes: replaces Media::makeKeyPointHashes(cvImg, keyPoints, outHashes) -- src/media.cpp:874-923,
 * called per image from Scanner::processImage (src/scanner.cpp:888) to produce the DctFeaturesIndex rows.
 * kp: (x, y, size) float triples = cv::KeyPoint::pt.x, pt.y, size; image i owns kp[3*kp_first[i] .. 3*kp_first[i+1]).
 * A keypoint becomes the square Rect(floor x, floor y, ceil size) when size >= 31 and it lies inside the image as
 * :887-894 tests; cbh_keypoint_rects is that rule alone (rects = x, y, s triples; returns their number).
 * Each square is hashed with dctHash64(sub, inPlace = true), in keypoint order: a blurred square is written back into
 * the image before the next (possibly overlapping) one is read, exactly like the reference's shared cv::Mat.
 * Images: 8-bit grey, image i at imgs + img_off[i], img_w[i] x img_h[i], rows img_row_stride[i] bytes apart (sizes may
 * differ per image; cbird scales to <= 400 px first, scanner.cpp:876).  out_first[i] = index of image i's first hash
 * (n + 1 entries; out_first[n] = total <= kp_first[n], the capacity out_hashes needs).  imgs_after (NULL or imgs_bytes
 * bytes) receives the images as the in-place blurs left them.  _dev: d_imgs is modified in place, d_out on the
 * device, the descriptor arrays on the host; the call returns when the hashes are complete. */
long long cbh_keypoint_rects(int cols, int rows, const float* kp, size_t nkp, int32_t* rects);
/* The building block of the above, also the drop-in for dctHash64(view, inPlace) on arbitrary sub-rectangle views
 * (cvImg.colRange(..).rowRange(..), src/media.cpp:908-910): rects = (x, y, w, h) int32 quadruples, image i owns
 * rects[4*rect_first[i] .. 4*rect_first[i+1]) (rect_first[0] = 0), hashed in that order; out_hashes[j] belongs to
 * rectangle j.  The blur takes the pixels around a rectangle from the parent image, as cv::blur does on a view.
 * in_place != 0: each blurred rectangle is written back before the next is read (inPlace = true); 0: the image is
 * left alone (inPlace = false on an 8UC1 view copies). */
int cbh_dcthash_rects(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off,
                      const uint32_t* img_w, const uint32_t* img_h, const uint32_t* img_row_stride,
                      const int32_t* rects, const uint32_t* rect_first, int in_place, uint64_t* out_hashes,
                      uint8_t* imgs_after, int device);
int cbh_keypoint_hashes(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off,
                        const uint32_t* img_w, const uint32_t* img_h, const uint32_t* img_row_stride,
                        const float* kp, const uint32_t* kp_first, uint64_t* out_hashes, uint32_t* out_first,
                        uint8_t* imgs_after, int device);
int cbh_keypoint_hashes_dev(void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                            const uint32_t* img_h, const uint32_t* img_row_stride, const float* kp,
                            const uint32_t* kp_first, void* d_out, uint32_t* out_first, int device, void* stream);

/* ---- ORB keypoints and descriptors -- Media::makeKeyPoints / makeKeyPointDescriptors (src/media.cpp:859-872) ----
 * cbird configures OpenCV 2.4's ORB as OrbFeatureDetector(numKeyPoints, 1.2f, 12, 31, 0, 2, HARRIS_SCORE, 31) and a
 * default OrbDescriptorExtractor (256-bit rBRIEF): 12-level pyramid at 1.2 (cv::resize INTER_LINEAR), FAST-9/16
 * (threshold 20, 3x3 non-maximum suppression), the 31-pixel border, retainBest(2N) on the FAST score, Harris response
 * (7x7, k 0.04), retainBest(N), intensity-centroid orientation (cv::fastAtan2), 7x7 sigma-2 Gaussian, rotated test
 * pairs.  n grey images of any sizes in one buffer (image i: img_w[i] x img_h[i] at byte img_off[i], pitch
 * img_row_stride[i]; cbird feeds <= 400 px on the longest side, src/scanner.cpp:876).
 *   kp[i*kp_cap + j]       j-th keypoint of image i as detect() returns it: pyramid level by level, inside a level in
 *                          the retainBest order selected below; x, y in image coordinates, size = 31 * 1.2^octave,
 *                          angle in degrees
 *   kp_after[2*(i*kp_cap+j)]  (optional) the keypoint's x, y as compute() leaves them in cbird's non-const list
 *                          (pt * (1/scale) * scale): what Media::makeKeyPointHashes sees when both algorithms run
 *   desc[(i*kp_cap+j)*32]  (optional) its descriptor row, the layout CvFeaturesIndex stores
 *   counts[i]              keypoints found for image i; only the first kp_cap are written -- a count above kp_cap
 *                          means the call must be repeated with more room (ties at a cut can be kept: see below)
 * Two things depend on what the cbird binary was built with and are stated in oracle/orb_oracle.c.
 * (1) KeyPointsFilter::retainBest is std::nth_element + std::partition: which of several keypoints with EQUAL response
 * survive a cut, and the ORDER of the survivors, follow from the C++ library.  cbh_set_tuning("orb_retain_order", v):
 *   1 (default)  what libstdc++ (the Linux builds' library) leaves: its introselect and partition restated on the
 *                device and checked against the real std::nth_element (oracle/retain_stl.cpp);
 *   0            canonical: every keypoint whose response is >= the n-th best is kept, raster order inside a level
 *                (a superset of what any library leaves).
 * cbh_orb_retain_best_dev is that one step on its own: count float responses in device memory -> d_order (uint32, room
 * for count) = the original positions of the survivors of retainBest(n_points) in the order they are left, *d_count =
 * how many.  depth_limit < 0 is nth_element's own 2 * lg(count); >= 0 enters the selection with that limit (its
 * heap-select branch, which only adversarial orders reach).
 * (2) The 256 test pairs are a learned table inside OpenCV (bit_pattern_31_), an input here: cbh_orb_set_pattern(xy)
 * takes its 1024 integers (x0, y0, x1, y1 per bit, each within [-15, 15]).  Without a pattern a call that asks for
 * descriptors fails with CBH_E_INVAL; detection alone (desc == NULL) needs none. */
typedef struct cbh_keypoint {
  float x, y, size, angle, response;
  int32_t octave;
} cbh_keypoint;
int cbh_orb_set_pattern(const int8_t* xy);
int cbh_orb_retain_best_dev(const void* d_responses, uint32_t count, int n_points, int depth_limit, void* d_order,
                            void* d_count, int device, void* stream);
int cbh_orb(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
            const uint32_t* img_h, const uint32_t* img_row_stride, int nfeatures, int kp_cap, cbh_keypoint* kp,
            float* kp_after, uint8_t* desc, uint32_t* counts, int device);
/* Media::makeKeyPointDescriptors alone (src/media.cpp:868-872), on keypoints the caller provides -- ORB::operator()
 * with useProvidedKeypoints: keypoints whose rounded position lies within 31 pixels of the image border are dropped,
 * the rest grouped by octave (order kept inside an octave), pt scaled to the level, described, scaled back.  Image i
 * provides kp[kp_first[i] .. kp_first[i+1]); out_kp / out_desc (room for kp_first[n] entries) receive what compute()
 * leaves in cbird's non-const keypoint list and the descriptor rows, image i owning [out_first[i], out_first[i+1]).
 * A keypoint whose octave has no pyramid level in its image, or whose patch leaves that level, is CBH_E_INVAL
 * (OpenCV would read outside the image). */
int cbh_orb_describe(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                     const uint32_t* img_h, const uint32_t* img_row_stride, const cbh_keypoint* kp,
                     const uint32_t* kp_first, cbh_keypoint* out_kp, uint8_t* out_desc, uint32_t* out_first, int device);
/* images and outputs in device memory (the size arrays stay on the host); stream NULL = synchronous */
int cbh_orb_dev(const void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
                const uint32_t* img_row_stride, int nfeatures, int kp_cap, void* d_kp, void* d_kp_after, void* d_desc,
                void* d_counts, int device, void* stream);

/* ---- ColorDescriptor::create (src/cvutil.cpp:790-1099) for a batch of 8-bit BGR (channels 3) or BGRA (4) images --
 * sizeLongestSide(rgb, 256, INTER_NEAREST) when a side exceeds 256; the elliptic centre mask (cv::ellipse of 0.9 x the
 * image, pix * alpha >> 8); float BGR -> Luv (cv::cvtColor's spline tables); samples with l > 4; cv::kmeans(K = 32,
 * TermCriteria(ITER|EPS, 100, 10), 1 attempt, KMEANS_PP_CENTERS); per quantised centre colour the sum of
 * (maxDist - dist) / maxDist over its pixels; colours by descending frequency; w = int(freq * 65535 / maxFreq);
 * numColors = index of the last colour (as the reference writes it).  descs: n x 258 bytes (32 x {l, u, v, w : u16},
 * numColors : u8, pad) -- the record ColorDescIndex stores; ok[i] = 0 where the reference returns without touching the
 * descriptor ("not enough colors": fewer than 32 samples), the record is then all zero.
 * Three things cannot match the cbird binary by its own construction (oracle/colordesc_oracle.c): kmeans draws from the
 * worker thread's never-reseeded cv::theRNG() -- here every image starts from a fresh thread's state; equal frequencies
 * are ordered by key; the rim of the ellipse is OpenCV's polygon fill as recalled.
 * The _dev call returns when the descriptors are complete (it synchronises the stream). */
void cbh_color_descriptor_dims(int w, int h, int* cols, int* rows);      /* the working size after the resize */
int cbh_color_ellipse_mask(int cols, int rows, uint8_t* mask);           /* the mask itself (cols x rows bytes) */
int cbh_color_descriptors(const uint8_t* imgs, size_t imgs_bytes, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                          const uint32_t* img_h, const uint32_t* img_row_stride, int channels, uint8_t* descs, uint8_t* ok,
                          int device);
int cbh_color_descriptors_dev(const void* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                              const uint32_t* img_h, const uint32_t* img_row_stride, int channels, void* d_descs,
                              void* d_ok, int device, void* stream);

/* ---- the steps in front of dctHash64 in Scanner::processImage (src/scanner.cpp:852-862) -------------------
 * grayscale(): cv::cvtColor(BGR2GRAY/BGRA2GRAY) on 8-bit data (src/cvutil.cpp:1265-1283); d_gray is packed
 * n*w*h bytes. */
int cbh_bgr2gray_dev(const void* d_src, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                     int channels, void* d_gray, int device, void* stream);
/* autocrop(gray, range) (src/cvutil.cpp:1285-1402): d_rects[i] = {left, top, right, bottom} (int32 x4) of the
 * region kept (the whole image when nothing is cropped). */
int cbh_autocrop_dev(const void* d_gray, size_t n, int w, int h, size_t row_stride, size_t img_stride, int range,
                     void* d_rects, int device, void* stream);
/* processImage's hash for n decoded images of one size in host memory: grayscale (channels 1/3/4) ->
 * autocrop when autocrop_range >= 0 (cbird uses 20) -> dctHash64 of the kept region.  autocrop() leaves cvGray as a
 * colRange/rowRange VIEW of the full image (src/cvutil.cpp:1397-1401), so -- as with any cv::Mat view -- the blur inside
 * dctHash64 still reads the cropped-away margins at the view's edges; the same is done here.  rects may be NULL. */
int cbh_process_images(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                       int channels, int autocrop_range, uint64_t* out, int32_t* rects, int device);
/* The same, and from the same upload the image processImage hands to ORB next: sizeLongestSide(cvGray, resize_size) of
 * the (autocropped) grey image (src/scanner.cpp:876; cbird: 400).  resized = n slots of resize_size^2 bytes, image i
 * packed (pitch = its width) at the start of slot i with the size resized_dims[2i] x resized_dims[2i+1] (0 x 0 where
 * the reference's sizeLongestSide throws).  resize_size 0: exactly cbh_process_images. */
int cbh_process_images_ex(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                          int channels, int autocrop_range, uint64_t* out, int32_t* rects, int resize_size,
                          uint8_t* resized, int32_t* resized_dims, int device);

/* ---- Scanner::processImage for a batch (src/scanner.cpp:828-895): every feature stage of one image, chained on the
 * device from ONE upload: grayscale -> autocrop -> dctHash64 of the kept region; ColorDescriptor::create of the colour
 * image; sizeLongestSide(kept region, resize_longest_side); makeKeyPoints; makeKeyPointDescriptors; makeKeyPointHashes
 * on the keypoints as compute() left them.  n decoded images of one geometry in host memory (channels 1 / 3 / 4).
 * algos = IndexParams::algos, bit (1 << SearchParams::Algo*): 1 dct, 2 dct features (keypoint hashes), 4 cv features
 * (ORB descriptors; needs cbh_orb_set_pattern), 8 colour.  Outputs of stages that are switched off may be NULL.
 *   dct_hashes[n], rects[4n] (kept region: left, top, right, bottom), resized_dims[2n] (0 x 0: the reference throws,
 *   no features); kp_counts[n] (found; at most kp_cap are written), kp[n * kp_cap] (the keypoint list as processImage
 *   holds it at the end), desc[n * kp_cap * 32]; kph_counts[n], kp_hashes[n * kp_cap]; color_descs[n * 258],
 *   color_ok[n]. */
typedef struct cbh_index_params {
  int autocrop_range;       /* cbird: 20; < 0: no autocrop (IndexParams::autocrop = false) */
  int algos;                /* IndexParams::algos */
  int resize_longest_side;  /* IndexParams::resizeLongestSide = 400 */
  int num_features;         /* IndexParams::numFeatures = 400 */
  int kp_cap;               /* room per image in kp / desc / kp_hashes */
} cbh_index_params;
int cbh_index_images(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride, int channels,
                     const cbh_index_params* p, uint64_t* dct_hashes, int32_t* rects, int32_t* resized_dims,
                     uint32_t* kp_counts, cbh_keypoint* kp, uint8_t* desc, uint32_t* kph_counts, uint64_t* kp_hashes,
                     uint8_t* color_descs, uint8_t* color_ok, int device);

/* sizeLongestSide(cv::Mat& img, int size, int filter = INTER_LANCZOS4) -- src/cvutil.cpp:1932-1950, the resize in
 * front of ORB detection (src/scanner.cpp:876, size = IndexParams::resizeLongestSide = 400): target size from the
 * float aspect ratio (cbh_longest_side_dims; a zero side is the reference's std::invalid_argument -> CBH_E_INVAL),
 * then cv::resize's 8-bit Lanczos-4 path.  n grey images of one geometry; out / d_dst: n packed out_w x out_h
 * images.  cbh_resize_lanczos4_dev resizes to any dw x dh. */
void cbh_longest_side_dims(int w, int h, int size, int* out_w, int* out_h);
int cbh_size_longest_side(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride, int size,
                          uint8_t* out, int* out_w, int* out_h, int device);
int cbh_resize_lanczos4_dev(const void* d_src, size_t n, int w, int h, size_t row_stride, size_t img_stride, int dw,
                            int dh, void* d_dst, int device, void* stream);

/* Stage-level diagnostics: as cbh_dcthash_batch_dev, and additionally writes the 32x32 u8 tile
 * each image is reduced to after stages 1-2 (blur + INTER_AREA) to d_tiles[i*1024 ..]. */
int cbh_dcthash_tiles_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride, void* d_out, void* d_tiles, int device, void* stream);

/* ---- DctHashIndex: src/dcthashindex.{h,cpp} -------------------------------------------- */
cbh_idx64* cbh_idx64_create(int device);                         /* DctHashIndex() :30-41 */
void cbh_idx64_destroy(cbh_idx64*);                              /* ~DctHashIndex/unload :43-54 */
/* ONE process, several GPUs.  cbird registers each Index once (src/engine.cpp:38-45) and fans find() out from its own
 * thread pool (src/database.cpp:1400-1432), so a drop-in that is to use every GPU of the node shards INSIDE the handle:
 * device_mask = bit d set for every HIP device d that takes a share (all must be usable gfx950 devices, else NULL);
 * shards_per_device >= 2 additionally cuts each device's share into that many logical shards with their own streams
 * (0 and 1 mean one).  The handle is a cbh_idx64 like any other: load/add/remove/find/find_batch/find_coalesced/slice/
 * search_index_batch, the fdct_* calls and cbh_vidx / cbh_color built on it all accept it and return what the
 * one-device index returns, bit for bit.  Rows are sharded in load order, shard s of R owning [s*n/R, (s+1)*n/R); add()
 * appends to the emptiest shard; needles are replicated; per search every shard scans its rows on its own device into
 * a { count, records } block, and the records meet on the first device of the mask, which alone consumes them: copies
 * of exactly count_s records per shard (device-to-device inside a device, hipMemcpyPeerAsync over xGMI between
 * devices), or -- cbh_set_tuning("shard_exchange", 0) -- ONE grouped ncclAllGather of the per-device blocks (librccl,
 * dlopen'ed on first use; an index that cannot get a communicator keeps using the copies and says so once in
 * cbh_last_error / stderr and in cbh_shard_stats.collective_fallbacks) (cbird_amd/csrc/sharded.hip).
 * The first device of the mask is the handle's device: *_dev calls take pointers on it. */
cbh_idx64* cbh_idx64_create_sharded(uint32_t device_mask, int shards_per_device);
uint32_t cbh_idx64_device_mask(const cbh_idx64*);   /* a plain index: 1 << device */
int cbh_idx64_shards_per_device(const cbh_idx64*);  /* a plain index: 1 */
int cbh_idx64_shard_count(const cbh_idx64*);        /* a plain index: 1 */
/* shard i as a plain one-device index, BORROWED (never destroy it): its count / download / scan_dev / time_scan_dev
 * show what one shard holds and does; a plain index is its own shard 0 */
cbh_idx64* cbh_idx64_shard(cbh_idx64*, int i);
typedef struct cbh_shard_stats {
  uint32_t shards, devices, device_mask;
  uint64_t segments;     /* runs of the global slot order (1 per shard after load; add() appends runs) */
  uint64_t scans;        /* shard-local scan launches */
  uint64_t rescans;      /* ... of which repeated because a shard's block overflowed */
  uint64_t collectives;  /* grouped ncclAllGather calls */
  uint64_t peer_copies;  /* hipMemcpyPeerAsync calls (needles out; records back unless "shard_exchange" is 0) */
  uint64_t local_copies; /* device-to-device copies of shard blocks inside one device */
  uint64_t collective_fallbacks; /* exchanges that asked for ncclAllGather and went by copies (no communicator) */
} cbh_shard_stats;
int cbh_idx64_shard_stats(const cbh_idx64*, cbh_shard_stats* out);
/* load(): replaces the SoA fill of DctHashIndex::load (:70-114); the caller runs the SQL
 * (`select id,phash_dct from media where type=1`, :89) and hands over the two columns.
 * n == 0 is valid (loaded, empty). */
int cbh_idx64_load(cbh_idx64*, const uint64_t* hashes, const uint32_t* ids, size_t n);
int cbh_idx64_load_dev(cbh_idx64*, const void* d_hashes, const void* d_ids, size_t n, void* stream);
int cbh_idx64_is_loaded(const cbh_idx64*);                       /* isLoaded() */
int cbh_idx64_add(cbh_idx64*, const uint64_t* hashes, const uint32_t* ids, size_t n); /* add() :158-173 */
/* remove(): zero id AND hash of every slot whose id is listed, in place (:175-191);
 * count() does not shrink. */
int cbh_idx64_remove(cbh_idx64*, const uint32_t* ids, size_t n);
size_t cbh_idx64_count(const cbh_idx64*);                        /* count() = _numHashes */
size_t cbh_idx64_memory_usage(const cbh_idx64*);                 /* memoryUsage() = 12 B * count (:56-59) */
/* mediaIds(), loaded branch (:129-133): ids of slots with hash != 0.  Writes up to cap,
 * *n_out = full number. */
int cbh_idx64_media_ids(const cbh_idx64*, uint32_t* out, size_t cap, size_t* n_out);
/* slice(): new index holding the slots whose id is in ids[], original order (:222-250);
 * caller destroys it (src/database.cpp:1435,1491). */
cbh_idx64* cbh_idx64_slice(const cbh_idx64*, const uint32_t* ids, size_t n);
/* copy the resident SoA back (tests, save()) */
int cbh_idx64_download(const cbh_idx64*, uint64_t* hashes, uint32_t* ids, size_t cap);

/* find(): every slot with hamm64(q, hash) < thresh and id != 0 (:193-220 with the brute-force
 * predicate of :210-217), as Match(id, distance), ascending (score, id).  q == 0 -> CBH_OK with
 * *n_out = 0 (:196-200).  Writes the first min(*n_out, cap) matches; *n_out is the full count. */
int cbh_idx64_find(cbh_idx64*, uint64_t q, int thresh, cbh_match* out, size_t cap, size_t* n_out);
/* Batched find + the sort/truncate of Database::searchIndex (src/database.cpp:1729-1735):
 * for query j, out[j*max_per_query ..] receives its first min(counts[j], max_per_query)
 * matches in (score, id) order (unused slots zeroed), counts[j] the full match count.
 * nq <= CBH_MAX_QUERIES_PER_CALL. */
int cbh_idx64_find_batch(cbh_idx64*, const uint64_t* q, size_t nq, int thresh, int max_per_query,
                         cbh_match* out, uint32_t* counts);
/* Same with device-resident queries and outputs (d_out: nq*max_per_query cbh_match,
 * d_counts: nq u32); *total_out (host) = total number of matching pairs. */
/* find_batch with one equal-bits mask per needle: additionally requires ((q ^ hash) & qmask) == 0.  This is
 * how the reference's APPROXIMATE structures restrict a search -- a HammingTree needle only sees the leaf
 * that shares its low bits (src/tree/hammingtree.h:244-252), a RadixMap needle only its bucket
 * (src/tree/radix.h:135-141) -- so with the right masks the results equal the reference's instead of being
 * a superset.  qmask == NULL is cbh_idx64_find_batch. */
int cbh_idx64_find_batch_masked(cbh_idx64*, const uint64_t* q, const uint64_t* qmask, size_t nq, int thresh,
                                int max_per_query, cbh_match* out, uint32_t* counts);
/* masks that make a search see exactly what HammingTree::search (hammingtree.h:103-108,244-293) would see
 * for the index's current contents: out_masks[i] = (1 << depth of q[i]'s leaf) - 1.  The tree shape (a node
 * splits on bit = depth once more than 8192 values were routed to it, :384-414) is rebuilt lazily after
 * load/add. */
int cbh_idx64_tree_masks(cbh_idx64*, const uint64_t* q, size_t nq, uint64_t* out_masks);
/* find() for an UNMODIFIED caller that issues one synchronous call per needle from many threads
 * (Database::similar: QtConcurrent::map -> searchIndex -> index->find, src/database.cpp:1400-1432,1698-1700).
 * Same results as cbh_idx64_find, bit for bit.  Concurrent callers are combined into one scan per round trip, and
 * once the time spent that way for a threshold exceeds what scanning the whole index against itself would cost
 * (the needles of an all-pairs search ARE the index entries), that self-join is run once, kept on the host, and
 * later calls whose needle hash is an index entry become table lookups (cbird_amd/csrc/coalesce.hip).  load / add /
 * remove drop the cache.  Thread-safe for concurrent readers; writers exclusive, as the reference's lock provides. */
int cbh_idx64_find_coalesced(cbh_idx64*, uint64_t q, int thresh, cbh_match* out, size_t cap, size_t* n_out);
typedef struct cbh_coalesce_stats {
  uint64_t finds;           /* calls */
  uint64_t cache_hits;      /* answered from a cached self-join */
  uint64_t rounds;          /* combined scans */
  uint64_t scanned_needles; /* needles served by those scans (finds - cache_hits, once everything has returned) */
  uint64_t self_joins;      /* whole-index self-joins built */
} cbh_coalesce_stats;
int cbh_idx64_coalesce_stats(cbh_idx64*, cbh_coalesce_stats* out);
int cbh_idx64_coalesce_set_self_join(cbh_idx64*, int enabled); /* default 1; 0 = combining only */
int cbh_idx64_find_batch_dev(cbh_idx64*, const void* d_q, size_t nq, int thresh,
                             int max_per_query, void* d_out, void* d_counts,
                             uint64_t* total_out, void* stream);
/* Raw scan stage only (the Hamming kernel): appends one cbh_record per matching pair to
 * d_records (capacity cap records, unordered) and adds the number of matching pairs to
 * *d_total (u64 on device, NOT reset by the call).  Records beyond cap are dropped but still
 * counted.  Asynchronous on `stream`.  This is the shard-local step of the multi-GPU path. */
int cbh_idx64_scan_dev(cbh_idx64*, const void* d_q, size_t nq, int thresh, void* d_records,
                       size_t cap, void* d_total, void* stream);
/* K4, the per-needle cut of Database::searchIndex (std::sort by score + stop at maxMatches,
 * src/database.cpp:1729-1737) as a counting select over UNORDERED scan records -- no global sort, no record count on
 * the host, nothing synchronises (cbird_amd/csrc/topk.hip).  Input: n_blocks blocks of u64 words, block b at
 * d_blocks + b*block_stride words = { count, records[cap] } -- what cbh_idx64_scan_dev produces when d_total points
 * at word 0 and d_records at word 1, and what ONE all_gather_into_tensor of such blocks delivers in the multi-GPU
 * path.  Output as cbh_idx64_find_batch_dev: the first max_per_query matches of every needle in ascending
 * (score, mediaId) order + the full match count per needle.  *d_status (u32 on device, NOT reset by the call) gets
 * bit 0 set when some block's count exceeds cap (records were dropped: rescan with a larger cap).  Asynchronous
 * on `stream` (NULL: synchronous).  nq <= 2^25, n_blocks*cap < 2^32. */
int cbh_records_topk_dev(const void* d_blocks, size_t n_blocks, size_t block_stride, size_t cap, size_t nq,
                         int max_per_query, void* d_out, void* d_counts, void* d_status, int device, void* stream);
/* Sort records ascending in place (device), n records, only bits [0, 39+ceil(log2(nq))). */
int cbh_sort_records_dev(void* d_records, size_t n, size_t nq, int device, void* stream);
/* Records (sorted) -> per-query top max_per_query + counts, as find_batch_dev does. */
int cbh_select_records_dev(const void* d_sorted_records, size_t n, size_t nq, int max_per_query,
                           void* d_out, void* d_counts, int device, void* stream);
/* capacity (records) of the internal match buffer used by find/find_batch; default 1<<24 */
int cbh_idx64_set_record_capacity(cbh_idx64*, size_t records);

/* Per-index counters of the Hamming scan kernel as launched by find/find_batch(_dev): number of
 * launches, GPU time between HIP events recorded around each launch on its own stream, and the
 * (needle x slot) pairs those launches evaluated. */
typedef struct cbh_stats {
  uint64_t scan_launches;
  uint64_t scan_pairs;
  double scan_ms;
} cbh_stats;
int cbh_idx64_get_stats(const cbh_idx64*, cbh_stats* out);
int cbh_idx64_reset_stats(cbh_idx64*);
int cbh_idx256_get_stats(const cbh_idx256*, cbh_stats* out);

/* ---- Database::searchIndex / Database::similar for a whole needle batch (cbird_amd/csrc/search.hip) ------------
 * searchIndex (src/database.cpp:1691-1757) over DctHashIndex for nq needles: find at `thresh`; when max_thresh > 0 the
 * needles whose match count is <= min_matches are searched again at thresh + 1, + 2, ... <= max_thresh (:1703-1725);
 * matches in ascending (score, mediaId) order (:1729, ties fixed); the needle's own id dropped when filter_self
 * (:1735); at most max_matches kept (:1736); ids not in valid_ids_sorted (the caller's idMap; NULL = every id is
 * known) skipped without consuming a place (:1755).  out[j*max_matches ..], out_counts[j] = kept matches of needle j.
 * max_matches <= 55. */
int cbh_search_index_batch(cbh_idx64*, const uint64_t* q, const uint32_t* needle_ids, size_t nq, int thresh,
                           int max_thresh, int min_matches, int max_matches, int filter_self,
                           const uint32_t* valid_ids_sorted, size_t n_valid, cbh_match* out, uint32_t* out_counts);
/* The group filtering of Database::similar on those results (host code): a needle with no match is no group (:1409);
 * a group needs more than min_matches members, needle included (filterMatch, :1245); with filter_groups a group whose
 * set of paths equals an earlier group's is dropped (filterMatches, :1252-1272), "earlier" in needle-path order; the
 * result is in needle-path order (:1463).  Paths enter as ranks: path_rank[i] = position of the path of media
 * ids_sorted[i] in the sorted order of all paths.  out_group[0..*n_out) = needle indices of the surviving groups. */
int cbh_filter_groups(const uint32_t* needle_ids, const cbh_match* matches, const uint32_t* counts, size_t nq,
                      int max_matches, int min_matches, int filter_groups, const uint32_t* ids_sorted,
                      const uint32_t* path_rank, size_t n_ids, uint32_t* out_group, size_t* n_out);

/* searchIndex for a needle batch over the other four indexes (cbird_amd/csrc/searchbatch.hip): per threshold level ONE
 * batched find of every needle still pending (the *_find_batch entry points below), needles with <= min_matches results
 * go on to the next level -- dctThresh + 1 for dct features and video, cvThresh + 5 for ORB, no levels for colour
 * (:1707-1718) -- while it does not pass max_thresh (max_thresh 0: no escalation); then (score, mediaId) order,
 * filter_self, the max_matches cut and the idMap rule as above.  out[j*max_matches ..], out_counts[j].  Needle data as
 * in the corresponding *_find_batch call. */
int cbh_fdct_search_index_batch(cbh_idx64*, const uint64_t* hashes, const uint64_t* offsets, const uint32_t* needle_ids,
                                size_t n_needles, int thresh, int max_thresh, int tree_compat, int min_matches,
                                int max_matches, int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid,
                                cbh_match* out, uint32_t* out_counts);
struct cbh_vidx;
struct cbh_vmatch;
int cbh_vidx_search_index_batch(struct cbh_vidx*, const int32_t* frames, const uint64_t* hashes, const uint64_t* offsets,
                                const uint32_t* needle_ids, size_t n_needles, int thresh, int max_thresh, int skip_frames,
                                int min_frames_matched, int min_frames_near, int min_matches, int max_matches,
                                int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid, struct cbh_vmatch* out,
                                uint32_t* out_counts);
int cbh_idx256_search_index_batch(cbh_idx256*, const uint8_t* rows, const uint64_t* offsets, const uint32_t* needle_ids,
                                  size_t n_needles, int thresh, int max_thresh, int k, int min_matches, int max_matches,
                                  int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid, cbh_match* out,
                                  uint32_t* out_counts);
struct cbh_color;
int cbh_color_search_index_batch(struct cbh_color*, const void* needle_descs, const uint32_t* needle_ids, size_t n_needles,
                                 int max_matches, int filter_self, const uint32_t* valid_ids_sorted, size_t n_valid,
                                 cbh_match* out, uint32_t* out_counts);

/* find() for an UNMODIFIED caller on the other four indexes (cbird_amd/csrc/combine.hip): same arguments and results as
 * cbh_fdct_find_ex / cbh_vidx_find_video / cbh_idx256_find / cbh_color_find; callers that arrive from other threads
 * while a search is in flight are served together by ONE call of the index's batch entry point (leader / follower, up
 * to 256 needles per round trip, requests grouped by equal search parameters).  This is what the Gpu*Index::find
 * adapters call.  cbh_combine_stats(handle): calls and combined searches so far. */
int cbh_fdct_find_coalesced(cbh_idx64*, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh, int tree_compat,
                            cbh_match* out, size_t cap, size_t* n_out);
int cbh_vidx_find_video_coalesced(struct cbh_vidx*, const int32_t* frames, const uint64_t* hashes, size_t n,
                                  uint32_t needle_id, int thresh, int skip_frames, int min_frames_matched,
                                  int min_frames_near, int filter_self, struct cbh_vmatch* out, size_t cap, size_t* n_out);
int cbh_idx256_find_coalesced(cbh_idx256*, const uint8_t* needle_rows, size_t n_desc, int thresh, int k, cbh_match* out,
                              size_t cap, size_t* n_out);
int cbh_color_find_coalesced(struct cbh_color*, const void* needle_desc, cbh_match* out, size_t cap, size_t* n_out);
int cbh_combine_stats(const void* handle, uint64_t* finds, uint64_t* rounds);

/* All of filterMatch (src/database.cpp:1209-1248) and filterMatches (:1250-1278) on those results, host code:
 *   path_mode      params.path / inPath (:1217-1229): 0 = no path filter, 1 = keep only matches under the prefix
 *                  (inPath), 2 = keep only matches NOT under it; the needle always stays
 *   filter_parent  a match in the needle's directory -- or zip archive, Media::dirPath (src/media.cpp:198-208) -- is
 *                  removed (:1231-1242)
 *   min_matches    a group needs more than min_matches members, needle included, AFTER those removals (:1245)
 *   filter_groups  groups in needle-path order, a group whose set of paths was seen before is dropped (:1253-1272)
 *   merge_groups   Media::mergeGroupList (src/media.cpp:300-324): if group a contains the first member of group b, b's
 *                  other members join a, a is re-ordered by score (ties: by path) and b disappears
 *   expand_groups  (when not merging) Media::expandGroupList (:326-331): a,b,c,d becomes (a,b), (a,c), (a,d)
 * and the final stable order by the first member's path (:1463).  negativeMatch and the weed marks need other tables of
 * the database and stay with the caller.  Media enter as ids with three attributes the caller derives from their paths,
 * parallel to ids_sorted: path_rank (position of the path among all sorted paths), dir_id (equal numbers <=> equal
 * dirPath(); may be NULL without filter_parent), under_prefix (path.startsWith(prefix) as :1223-1227 builds the prefix;
 * may be NULL with path_mode 0).
 * Output: group g = out_members[out_first[g] .. out_first[g+1]) as (mediaId, score), the needle first with score -1 (a
 * haystack Media's default, src/media.cpp:112) unless a merge re-ordered the group.  CBH_E_OVERFLOW with *n_groups /
 * *n_members = the room needed when cap_groups (+1 entries in out_first) or cap_members is too small. */
typedef struct cbh_filter_params {
  int min_matches, filter_groups, filter_parent, path_mode, merge_groups, expand_groups;
} cbh_filter_params;
int cbh_filter_groups_ex(const uint32_t* needle_ids, const cbh_match* matches, const uint32_t* counts, size_t nq,
                         int max_matches, const cbh_filter_params* p, const uint32_t* ids_sorted,
                         const uint32_t* path_rank, const uint32_t* dir_id, const uint8_t* under_prefix, size_t n_ids,
                         uint64_t* out_first, size_t cap_groups, cbh_match* out_members, size_t cap_members,
                         size_t* n_groups, size_t* n_members);

/* ---- DctFeaturesIndex: src/dctfeaturesindex.{h,cpp} over src/tree/hammingtree.h -----------------
 * The index is a cbh_idx64 whose entries are (mediaId, keypoint hash) pairs, several per media:
 *   load/add      -> cbh_idx64_load / cbh_idx64_add with one entry per hash (:229-238, :143-156)
 *   remove        -> cbh_idx64_remove_ids_only: HammingTree::remove zeroes the index and KEEPS the
 *                    hash (hammingtree.h:349-361, dctfeaturesindex.cpp:240-249)
 *   count()       -> cbh_idx64_count (= _tree->size(), removed entries included) */
int cbh_idx64_remove_ids_only(cbh_idx64*, const uint32_t* ids, size_t n);
/* HammingTree::findIndex (hammingtree.h:110-112): hashes stored for one mediaId */
int cbh_idx64_hashes_for_id(const cbh_idx64*, uint32_t id, uint64_t* out, size_t cap, size_t* n_out);
/* DctFeaturesIndex::find (:260-358) for one needle with keypoint hashes[n] and id needle_id:
 * per needle hash the 10 nearest entries under thresh (removed entries take part in the cut and are
 * skipped afterwards, :301-308), votes per mediaId, score -1 for the needle itself, 10*avg distance
 * when no other media got more than one vote, else maxMatches - votes.  Results ascending mediaId
 * (QMap order).  Exact candidates (the reference tree is approximate). */
int cbh_fdct_find(cbh_idx64*, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                  cbh_match* out, size_t cap, size_t* n_out);
/* Many needles in one scan: needle i owns hashes[offsets[i] .. offsets[i+1]); results of needle i
 * go to out[out_offsets[i] .. out_offsets[i+1]).  CBH_E_OVERFLOW when the total exceeds cap
 * (out_offsets[n_needles] then holds the required capacity). */
/* as cbh_fdct_find / _batch; tree_compat != 0 restricts every needle hash to its HammingTree leaf
 * (cbh_idx64_tree_masks), reproducing the reference's approximate candidate sets on multi-leaf trees */
int cbh_fdct_find_ex(cbh_idx64*, const uint64_t* hashes, size_t n, uint32_t needle_id, int thresh,
                     int tree_compat, cbh_match* out, size_t cap, size_t* n_out);
int cbh_fdct_find_batch_ex(cbh_idx64*, const uint64_t* hashes, const uint64_t* offsets,
                           const uint32_t* needle_ids, size_t n_needles, int thresh, int tree_compat,
                           cbh_match* out, size_t cap, uint64_t* out_offsets);
int cbh_fdct_find_batch(cbh_idx64*, const uint64_t* hashes, const uint64_t* offsets,
                        const uint32_t* needle_ids, size_t n_needles, int thresh, cbh_match* out,
                        size_t cap, uint64_t* out_offsets);

/* ---- DctVideoIndex: src/dctvideoindex.{h,cpp}, VideoIndex: src/videoindex.{h,cpp} -----------------
 * Index::Match with its MatchRange (src/index.h:157-166, src/media.h:62-78). */
typedef struct cbh_vmatch {
  uint32_t id;   /* mediaId of the matched video */
  int32_t score; /* findVideo: 100 - percentNear; findFrame: Hamming distance */
  int32_t src_in, dst_in, len; /* MatchRange: needle frame, matched frame, length */
} cbh_vmatch;
typedef struct cbh_vidx cbh_vidx; /* opaque: DctVideoIndex state */

cbh_vidx* cbh_vidx_create(int device);
/* the search structure built over all videos' frames becomes a sharded cbh_idx64 (above); entries are in video order,
 * so the contiguous row shares are shares BY VIDEO up to one video at each boundary */
cbh_vidx* cbh_vidx_create_sharded(uint32_t device_mask, int shards_per_device);
void cbh_vidx_destroy(cbh_vidx*);
/* 0 (default) = exact search; N > 0 = the reference's RadixMap(videoRadix = N) behaviour: a needle frame only
 * sees index entries of its bucket (hash >> 1) & (2^N - 1) (src/tree/radix.h:135-141, `-p.vradix`, default 10
 * in cbird); N is limited to 24 like the reference's constructor does (:105-112) */
int cbh_vidx_set_radix(cbh_vidx*, int radix);
/* load()/add() (:172-211, :250-254) only register media ids; the per-video (frame, hash) lists come
 * from <dataPath>/<id>.vdx when the tree is built (insertHashes :61-111).  Here the caller hands the
 * decoded lists over (cbh_vdx_decode below reads the files).  Order of calls = _mediaId order. */
int cbh_vidx_add_video(cbh_vidx*, uint32_t media_id, const int32_t* frames, const uint64_t* hashes, size_t n);
int cbh_vidx_remove(cbh_vidx*, const uint32_t* media_ids, size_t n);           /* remove() :256-275 */
size_t cbh_vidx_count(const cbh_vidx*);                                         /* count() = #videos */
/* memoryUsage() (src/dctvideoindex.cpp:57-59: `_tree ? _tree->stats().memory : 0`): 0 until the search structure is
 * built; then the payload bytes the reference's RadixMap holds for the same entries -- 8 (hash_t) + 6 (the packed 24+24-bit
 * VideoTreeIndex, src/dctvideoindex.h:37-43) per entry -- without std::vector's capacity slack and the bucket table
 * (src/tree/radix.h:80-83,167), which depend on insertion history */
size_t cbh_vidx_memory_usage(const cbh_vidx*);
/* number of entries in the search structure after the insertHashes filters; builds it (buildTree
 * :113-170) with vtrim = skip_frames if it is not built yet (a built tree is NOT rebuilt for another
 * skip_frames, like `if (_tree) return;`). */
size_t cbh_vidx_entries(cbh_vidx*, int skip_frames);
/* findFrame (:291-387), image needle: nearest frame per video under thresh; results ascending video
 * index; range = (src_in<0 ? 0 : src_in, matched frame, 1). */
int cbh_vidx_find_frame(cbh_vidx*, uint64_t hash, int thresh, int skip_frames, int src_in,
                        cbh_vmatch* out, size_t cap, size_t* n_out);
/* findVideo (:399-657), video needle given as its (frames, hashes) list (needle_id 0 = not in the db).
 * thresh = dctThresh, skip_frames = vtrim (skipFrames), min_frames_matched = vfm, min_frames_near = vfn. */
int cbh_vidx_find_video(cbh_vidx*, const int32_t* frames, const uint64_t* hashes, size_t n,
                        uint32_t needle_id, int thresh, int skip_frames, int min_frames_matched,
                        int min_frames_near, int filter_self, cbh_vmatch* out, size_t cap, size_t* n_out);
/* many video needles in one scan (needle i = [offsets[i], offsets[i+1]) of frames/hashes) */
int cbh_vidx_find_videos_batch(cbh_vidx*, const int32_t* frames, const uint64_t* hashes,
                               const uint64_t* offsets, const uint32_t* needle_ids, size_t n_needles,
                               int thresh, int skip_frames, int min_frames_matched, int min_frames_near,
                               int filter_self, cbh_vmatch* out, size_t cap, uint64_t* out_offsets);
/* .vdx v2 (VideoIndex::save_v2/load_v2/verify_v2, src/videoindex.cpp:248-429).  encode returns the file
 * size (0 = invalid input: first frame must be 0, frames strictly increasing) and writes it when it fits;
 * decode = load_v2: the frame count or a negative CBH_E_* (bad header, short data); like the reference's loader it
 * does not look at the trailer and loads a file with more than 2^24 frames up to that limit; verify = verify_v2
 * (what VideoIndex::isValid runs): header + "cbir" trailer, 1 = valid. */
size_t cbh_vdx_encode(const int32_t* frames, const uint64_t* hashes, size_t n, const char* cbird_version,
                      uint8_t* out, size_t cap);
long long cbh_vdx_decode(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap);
int cbh_vdx_verify(const uint8_t* buf, size_t len);
/* the old version-1 files (src/videoindex.cpp:41-68, 431-541: u16 frame count, u16 frame numbers, u64 hashes): decode and
 * verify above pick the version by the magic like VideoIndex::load / isValid and apply load_v1's two repairs (frame
 * numbers that wrapped past 65535; a missing frame 0); cbh_vdx_version tells which it is, cbh_vdx_encode_v1 = save_v1. */
int cbh_vdx_version(const uint8_t* buf, size_t len);
size_t cbh_vdx_encode_v1(const int32_t* frames, const uint64_t* hashes, size_t n, uint8_t* out, size_t cap);
/* frame de-dup of Media::makeVideoIndex (src/media.cpp:958-1024); keep[i]=1 for stored frames */
size_t cbh_video_dedup(const uint64_t* hashes, size_t n, int threshold, uint8_t* keep);

/* Media::makeVideoIndex (src/media.cpp:925-1037) for a decoder that delivers its frames in chunks: per frame
 * grayscale (the decoder outputs grey: a no-op, :958) -> autocrop(img, 20) -> dctHash64 (:961-962, :987-992) on the
 * device, then the near-frame filter (:994-1011), whose state -- frameNumber, window, the stored (frame, hash) lists --
 * lives in the handle between pushes.  threshold = IndexParams::videoThreshold (src/scanner.h; <= 0 stores every
 * frame), autocrop_range = 20 in cbird (< 0: no autocrop).
 *   resume(): start from an index written earlier (:929-936: the next frame is frames[n-1] + 1 and, like the first
 *             frame of a fresh run, is stored unconditionally); only before the first push.
 *   push():   n grey frames of one geometry in host memory, in decode order; push_dev(): the same in device memory
 *             (a hardware decoder's output).  Frames after MAX_FRAMES_PER_VIDEO (1 << 24, src/dctvideoindex.h:32,50)
 *             are dropped as at :1013-1016.
 *   finish(): the index as makeVideoIndex leaves it, with the last frame appended if it was not stored (:1018-1024);
 *             returns the number of entries and writes them when cap suffices.  Does not change the handle: more
 *             frames may be pushed afterwards.
 * One handle = one video, used by one thread at a time (Scanner::processVideo runs one per worker thread); handles of
 * different threads are independent (each owns its stream and buffers). */
typedef struct cbh_vindexer cbh_vindexer;
cbh_vindexer* cbh_vindexer_create(int device, int threshold, int autocrop_range);
void cbh_vindexer_destroy(cbh_vindexer*);
int cbh_vindexer_resume(cbh_vindexer*, const int32_t* frames, const uint64_t* hashes, size_t n);
int cbh_vindexer_push(cbh_vindexer*, const uint8_t* frames, size_t n, int w, int h, size_t row_stride,
                      size_t img_stride);
int cbh_vindexer_push_dev(cbh_vindexer*, const void* d_frames, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride);
long long cbh_vindexer_frames_seen(const cbh_vindexer*);   /* makeVideoIndex's frameNumber */
long long cbh_vindexer_finish(const cbh_vindexer*, int32_t* frames, uint64_t* hashes, size_t cap);

/* ---- TemplateMatcher::match's score (src/templatematcher.cpp:331-374) ------------------------------------------------
 * For n candidate patches as warpAffine left them (template-sized; channels 1, 3 = BGR or 4 = BGRA; pixels outside the
 * warped patch are 0) and ONE template image: grayscale(cand); per pixel the candidate's grey value is the mask -- where
 * it is 0 the template's pixel is zeroed too, a BGRA template is premultiplied by its alpha (and scales the candidate's
 * grey value by it) (:343-364); candHash = dctHash64(cand), tmplHash = dctHash64(tmplMasked); score = hamm64 of the two
 * (:366-371; the caller compares with tmThresh, :373).  The descriptor match in front of it is cbh_idx256_radius_match;
 * estimateRigidTransform / warpAffine between the two stay with the caller (OpenCV). */
int cbh_template_scores(const uint8_t* cands, size_t n, int w, int h, size_t cand_row_stride, size_t cand_img_stride,
                        int cand_channels, const uint8_t* tmpl, size_t tmpl_row_stride, int tmpl_channels,
                        uint64_t* cand_hashes, uint64_t* tmpl_hashes, int32_t* scores, int device);
int cbh_template_hashes_dev(const void* d_cands, size_t n, int w, int h, size_t cand_row_stride, size_t cand_img_stride,
                            int cand_channels, const void* d_tmpl, size_t tmpl_row_stride, int tmpl_channels,
                            void* d_cand_hashes, void* d_tmpl_hashes, int device, void* stream);

/* ---- CvFeaturesIndex: src/cvfeaturesindex.{h,cpp} ---------------------------------------------------
 * N x 32-byte ORB/BRIEF descriptor rows (cv::Mat CV_8U, cvfeaturesindex.h:73) + the first-row -> mediaId
 * map (_indexMap/_idMap, :77-81).  Searches are exact brute force (the reference asks a FLANN LSH index,
 * :497, which returns a subset). */
cbh_idx256* cbh_idx256_create(int device);
/* ONE CvFeaturesIndex over several GPUs / logical shards in one process (as cbh_idx64_create_sharded): sharded BY IMAGE
 * -- a media's descriptor rows stay together (runs of 16384 rows per shard in add order), the first-row -> mediaId maps
 * stay with the handle in global row numbers, every search scans all shards on their own devices and streams, rewrites
 * the shard-local rows of the records to global ones and merges them on the first device of the mask (device-to-device
 * copies inside a device, one grouped ncclAllGather between devices).  Every other cbh_idx256_* call accepts the handle
 * and returns what the one-device index returns, bit for bit (the knn tie-break is (distance, global row)). */
cbh_idx256* cbh_idx256_create_sharded(uint32_t device_mask, int shards_per_device);
int cbh_idx256_shard_count(const cbh_idx256*);
size_t cbh_idx256_shard_rows(const cbh_idx256*, int i); /* rows held by shard i */
int cbh_idx256_shard_stats(const cbh_idx256*, cbh_shard_stats* out);
void cbh_idx256_destroy(cbh_idx256*);
/* add() (:122-150): append one media's rows; n_rows == 0 is skipped ("no descriptors for ...").
 * load() is add() per row of `select media_id,... from matrix`. */
int cbh_idx256_add(cbh_idx256*, uint32_t media_id, const uint8_t* rows, size_t n_rows);
/* remove() (:152-165): the map entry's id becomes 0, descriptors stay and keep taking knn places */
int cbh_idx256_remove(cbh_idx256*, const uint32_t* ids, size_t n);
int cbh_idx256_is_loaded(const cbh_idx256*);
size_t cbh_idx256_count(const cbh_idx256*);        /* count() = _descriptors.rows (:103) */
size_t cbh_idx256_memory_usage(const cbh_idx256*); /* memoryUsage() = 2 * rows * 32 (:105-120) */
/* descriptorsForMediaId (:421-436): row range of a media; download of a row range */
int cbh_idx256_rows_of(const cbh_idx256*, uint32_t media_id, size_t* first, size_t* count);
int cbh_idx256_download_rows(const cbh_idx256*, size_t first, size_t count, uint8_t* out);
/* exact `knnSearch(needles, k)` below thresh: out_row/out_dist[nq*k] ordered (distance, row), counts[nq] =
 * rows under thresh.  nq < 2^23. */
int cbh_idx256_knn(cbh_idx256*, const uint8_t* needles, size_t nq, int k, int thresh, uint32_t* out_row,
                   uint16_t* out_dist, uint32_t* counts);
/* Multi-GPU halves of find() (shard by image): the knn table with the mediaId of every candidate row
 * (0 = removed; the first-row -> mediaId map is shard-local), and the scoring of a (merged) table -- votes per media,
 * median distance * 1000 / votes (:499-596; host code), needle i owning table rows [offsets[i], offsets[i+1]). */
int cbh_idx256_knn_media(cbh_idx256*, const uint8_t* needles, size_t nq, int k, int thresh, uint32_t* out_row,
                         uint16_t* out_dist, uint32_t* out_media, uint32_t* counts);
int cbh_cvfeatures_score(const uint32_t* media, const uint16_t* dist, const uint32_t* counts, const uint64_t* offsets,
                         size_t n_needles, int k, cbh_match* out, size_t cap, uint64_t* out_offsets);
/* find() (:438-604): knn k (reference: 10) per needle descriptor, distance < thresh (cvThresh), votes per
 * media, score = median distance * 1000 / votes; results ascending mediaId. */
int cbh_idx256_find(cbh_idx256*, const uint8_t* needle_rows, size_t n_desc, int thresh, int k, cbh_match* out,
                    size_t cap, size_t* n_out);
int cbh_idx256_find_batch(cbh_idx256*, const uint8_t* needle_rows, const uint64_t* offsets, size_t n_needles,
                          int thresh, int k, cbh_match* out, size_t cap, uint64_t* out_offsets);
/* TemplateMatcher's descriptor match (src/templatematcher.cpp:134,217): cv::BFMatcher(NORM_HAMMING).radiusMatch with
 * the index rows as the train set (load the template's descriptors with cbh_idx256_add): every (query, train) pair
 * with distance <= max_dist, grouped by query, each group in ascending (distance, train row) order (OpenCV: by
 * distance, ties unspecified).  out_first has nq + 1 entries; returns CBH_E_OVERFLOW (with out_first complete) when
 * cap is too small.  The steps after it (estimateRigidTransform's RANSAC, warpAffine) are OpenCV's and stay there. */
typedef struct {
  int32_t query_idx, train_idx, distance; /* cv::DMatch::queryIdx, trainIdx, distance */
} cbh_dmatch;
int cbh_idx256_radius_match(cbh_idx256*, const uint8_t* queries, size_t nq, int max_dist, cbh_dmatch* out,
                            size_t cap, uint64_t* out_first);

/* ---- ColorDescIndex: src/colordescindex.{h,cpp}; ColorDescriptor: src/cvutil.h:57-113 -----------------
 * A descriptor is the reference's 258-byte struct: 32 x {l,u,v,w : uint16} + numColors : uint8 (+1 pad). */
#define CBH_COLOR_DESC_BYTES 258
typedef struct cbh_color cbh_color;
cbh_color* cbh_color_create(int device);
void cbh_color_destroy(cbh_color*);
int cbh_color_add(cbh_color*, const uint32_t* ids, const void* descs, size_t n); /* load/add :123-168,201-213 */
int cbh_color_remove(cbh_color*, const uint32_t* ids, size_t n);                  /* remove :215-229 */
size_t cbh_color_count(const cbh_color*);
int cbh_color_is_loaded(const cbh_color*);                                       /* _count > 0 (:114) */
size_t cbh_color_memory_usage(const cbh_color*);                                 /* (258 + 4) * count (:118-121) */
int cbh_color_find_index_data(const cbh_color*, uint32_t id, void* out_desc);    /* :231-239; returns 1/0 */
int cbh_color_download(const cbh_color*, uint32_t* ids, void* descs, size_t cap);
/* find() (:250-278) with ColorDescriptor::distance (src/cvutil.cpp:682-749): every entry whose distance is
 * finite (both sides have colours, counts differ by <= 2) and whose id != 0, in index order,
 * score = int(1 + sum of nearest-colour distances).  Bit-exact to the reference's float arithmetic. */
int cbh_color_find(cbh_color*, const void* needle_desc, cbh_match* out, size_t cap, size_t* n_out);
/* cbh_color_find for nq needles in one pass (needle q: out[out_offsets[q] .. out_offsets[q+1]), index order) */
int cbh_color_find_all_batch(cbh_color*, const void* needle_descs, size_t nq, cbh_match* out, size_t cap,
                             uint64_t* out_offsets);
/* ColorDescriptor::distance (src/cvutil.cpp:682-749) as FLOATS, nq needles x every index entry: out[q*count + i]
 * for entry i in add order (whatever its id); FLT_MAX where the reference returns FLT_MAX (:683-684).  The int
 * scores of cbh_color_find are (int) of exactly these values. */
int cbh_color_distances(cbh_color*, const void* needle_descs, size_t nq, float* out);
/* many needles + the sort/cut of Database::searchIndex: first min(counts[q], k) matches by (score, id) */
int cbh_color_find_batch(cbh_color*, const void* needle_descs, size_t nq, int k, cbh_match* out,
                         uint32_t* counts);

/* Knobs (25).  Results never change with any of them except "color_fma".  Unknown keys return CBH_E_INVAL.
 * Which kernel serves a call:
 *   "scan_mfma"     64-bit scan on the matrix cores (k_hamm64_mfma*): 0 = never (the popcount kernel k_hamm64_scan), 1 = calls
 *                   with >= 256 needles and >= 4096 slots (default), 2 = always; 3 = as 1, and calls with thresholds <= 8 and
 *                   no needle masks whose scan would take >= 1 ms go to the bucketed join (hamm64_join.hip: multi-index
 *                   hashing -- only pairs that share one of max(4, thresh) chunk values are compared; the same records) when
 *                   its exact candidate count says it is cheaper; 4 = the join for every call it can represent (tests).
 *                   The join avoids comparisons rather than making them faster: it is opt-in, the default compares every pair
 *   "scan_mfma_pre_max" prefilter kernel or three-field 64-bit kernel: -1 (default) = per launch, by the candidate rate of the
 *                   launch's own data -- r_cand = P[popc(fold(a) ^ fold(b)) < thresh] and r_true = P[hamm64(a, b) < thresh], counted
 *                   on 2048 x 2048 sampled (slot, needle) pairs by k_fold_probe; the prefilter while r_cand - 4 r_true <=
 *                   "scan_pre_rate_e9" (a candidate costs the prefilter a re-check, a true match costs the three-field kernel
 *                   four times that); launches of < 2^31 pairs: thresholds <= 6 -- 0 = never the prefilter, t > 0 =
 *                   thresholds <= t (<= 32) take it whatever the data
 *   "scan_pre_rate_e9" that bound x 1e9 (default 125000 = 1.25e-4: where the two kernels tie, profiles/r06_adaptive_ab*.jsonl)
 *   "scan256_mfma"  256-bit scan on the matrix cores (k_hamm256_*): 0 = never (k_hamm256_scan), 1 = calls with >= 64 needle
 *                   descriptors and >= 4096 rows (default), 2 = always
 *   "scan256_small" 1 = searches with <= 512 needle descriptors (one ORB needle image) and thresholds <= 40 use the
 *                   stationary-needle kernel k_hamm256_small (default), 0 = the row-stationary kernels
 *   "hash_mfma"     256 x 256 tiles: non-zero (default 2) = k_dcthash_256_band (the 7 x 7 box filter as i8 MFMAs, its vertical
 *                   sum kept in their accumulators; rows must be 16-byte aligned, otherwise 0 is taken), 0 = k_dcthash_256
 *                   (all VALU; also what runs if the band table cannot be made)
 *   "hash_band_area" fractional resize ratios, 7 x 7 blur, <= 1920 columns: 1 (default) = k_band_area (the matrix-core blur
 *                   for any width and height; whole images and views whose vertical edges are the parent's or lie >= 8 / >= 3
 *                   columns inside it), 0 = the VALU kernels that serve every other geometry
 *   "hash_stream"   the VALU kernels: 1 (default) = k_blur_area_regs walks strips of 3..8 steps when the batch has enough
 *                   workgroups for that, k_blur_area (a workgroup per 16-row band) otherwise; 0 = always the band kernel;
 *                   v >= 2 = always strips, of v steps
 *   "hash_fuse"     k_blur_area_regs: 1 (default) = vertical INTER_AREA pass and tile inside the kernel when a workgroup per
 *                   image still fills the machine, 2 = always, 0 = never (k_tile_hash reads the rows back)
 *   "kp_lds_side"   largest keypoint square k_kp_hashes stages in LDS (default 134; larger: global-memory routine)
 *   "kp_blur_side"  largest keypoint square whose blurred copy also stays in LDS (default 112)
 *   "fdct_host_vote", "video_host_reduce"  the per-needle reductions of DctFeaturesIndex / DctVideoIndex finds: 0 (default) =
 *                   on the device for batches, on the host for a single needle; 1 = host, 2 = device
 *   "orb_retain_order" KeyPointsFilter::retainBest: 1 (default) = the survivors in the order libstdc++'s nth_element +
 *                   partition leave them (cbird's Linux builds), 0 = canonical (every tie kept, raster order)
 * The one knob that CHANGES results (within north_star's float tolerance; not the default: the shipped kernel is
 * deliberately stricter -- raw floats bitwise equal to ColorDescriptor::distance -- at a cost of 9 % of its time):
 *   "color_fma"     1 = dl^2 + du^2 + dv^2 with fused multiply-adds: distances within 1e-5 (relative) of the reference's,
 *                   int(score) can move by one at an integer boundary (2 of 188 902 on profiles/r04_knobs_area_color.json)
 * Memory:
 *   "color_create_chunk_mb" scratch one launch of ColorDescriptor::create may take, in MB (default 32768): larger batches
 *                   are worked off in chunks of that many images (1.6 MB per 256 x 192 image)
 *   "pool_keep_mb"  cached scratch that may outlive its stream, per device, in MB (default 16384; < 0: everything)
 *   "pool_live_keep_mb" what the cache of a LIVE stream may hold, in MB (0 = default: a quarter of the device's memory, at
 *                   least 16384; < 0: everything).  Beyond it the blocks freed longest ago go back to the driver once the
 *                   work queued behind them has run.  Whatever is cached stays reclaimable: an allocation the driver
 *                   refuses -- scratch or index storage alike -- gives the device's cached blocks back and is tried again
 * Multi-device handles:
 *   "shard_force_rccl" 1 = a sharded index on ONE device still sends its blocks through ncclAllGather (one rank): the
 *                   transport test of a one-GPU box (default 0)
 *   "shard_exchange" how the records of a multi-device index reach the root device: 1 = hipMemcpyPeerAsync of exactly
 *                   count_s records per shard into the root block (default: only the root consumes them), 0 = one
 *                   grouped ncclAllGather of the per-device blocks (falls back to 1, with a note in cbh_last_error, when
 *                   librccl cannot be loaded, its version does not match the headers, or ncclCommInitAll fails)
 * Fault injection (tests/test_error_paths.py; never armed by the library itself):
 *   "fault_alloc_after" n >= 0: the n-th allocation from now (0 = the next; device, pinned and scratch allocations all
 *                   count) fails once with out-of-memory and the knob disarms itself; -1 disarms
 *   "fault_alloc_sticky" 1 = once that countdown has run out, every later allocation fails too, until disarmed
 *   "fault_driver_oom" n >= 0: the n-th driver allocation of the scratch arena is refused once (its trim-and-retry path)
 *   "fault_persist_oom" n >= 0: the n-th allocation of index / workspace / table memory is refused by the driver once (the
 *                   arena gives its cached blocks of the device back, the allocation is tried again)
 *   "fault_rccl"    1 = librccl is treated as absent
 * (Rounds 1-5 carried 57 knobs, most of them A/B switches between kernel generations; the losers and their switches
 * went with round 6 -- git history and NOTES.md keep what each measured.) */
int cbh_set_tuning(const char* key, int value);
/* Read-back for tests and soak tools: "fault_alloc_after" (what is left of the countdown, -1 = disarmed or fired),
 * "fault_fired", "alloc_calls" (allocations seen since the library loaded), and the scratch arena's
 * "arena_cached_bytes", "arena_pending_bytes", "arena_live_bytes", "arena_live_blocks", "arena_trimmed_live",
 * "arena_oom_retry_stream", "arena_oom_retry_device", "arena_oom_retry_persistent", "arena_released"; "scan_pre_mask"
 * (bit t = the most recent matrix-core launch at threshold t took the prefilter kernel), "scan_joins" (calls the bucketed join has answered), "scan_probes" (candidate-rate
 * probes run so far), "scan_probe_rate_e9" / "scan_probe_true_e9" (the candidate and true-match rates the last one found for its threshold,
 * x 1e9; -1 = none yet). */
int cbh_get_tuning(const char* key, long long* value);

/* ---- measurement support ---------------------------------------------------------------- */
/* Run the scan kernel `iters` times on the index's own stream bracketed by hipEvents and
 * return the average kernel time in milliseconds (bench.py roofline leg).  */
int cbh_idx64_time_scan_dev(cbh_idx64*, const void* d_q, size_t nq, int thresh, void* d_records,
                            size_t cap, void* d_total, int iters, float* ms_avg);
int cbh_time_dcthash_dev(const void* d_imgs, size_t n, int w, int h, size_t row_stride,
                         size_t img_stride, void* d_out, int device, int iters, float* ms_avg);
/* The hardware behaviour three kernels rely on (k_hamm256_small's row tiles past the end, k_band_area's rows at the end of
 * the buffer, the prestage first look): a raw buffer load whose SCALAR offset alone carries it past the descriptor's
 * num_records -- the per-lane offset inside the range -- returns zeros and touches nothing.  Loads a 4 KB window of an
 * 8 KB allocation whose second half is poisoned, with scalar offsets inside, at and beyond the window's end; *ok = 1 iff
 * every load that starts at or past num_records returned 0 and every load inside it returned the data
 * (tests/test_boundary.py; no product call depends on this entry point). */
int cbh_selftest_buffer_range(int device, int* ok);

#ifdef __cplusplus
}
#endif
#endif /* CBIRD_HIP_H */
