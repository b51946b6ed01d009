// tools/gen_golden_opencv.cpp -- produce REAL golden vectors for dctHash64 and its pre-stages with the library cbird
// pins: OpenCV 2.4.13.7 (cbird.pri:148-152; build recipe docker/build-opencv.sh).  This image has no OpenCV, so
// nobody has run this yet -- the day someone does, tests/test_opencv_golden.py turns "parity unpinned" into a pin (or
// into a list of the stages to correct in oracle/cbird_oracle.c, oracle/cv_dct32.c and dcthash.hip).
//
//   g++ -O2 -std=c++11 tools/gen_golden_opencv.cpp -o gen_golden_opencv `pkg-config --cflags --libs opencv`   (core, imgproc, features2d)
//   ./gen_golden_opencv > opencv_hash.txt
//   python tools/opencv_golden_to_npz.py opencv_hash.txt tests/golden/opencv_hash.npz
//
// Everything OpenCV-free (the image generator, the geometry list) can be built with -DNO_OPENCV, which is how
// tests/test_opencv_golden.py checks here that this file and its Python twin (tools/opencv_golden_to_npz.py)
// generate identical inputs.
//
// The calls below are the ones the reference makes, in its order, written from its description in SURVEY.md 8(a1):
//   dctHash64        src/cvutil.cpp:435-545   blur (kernel by area) -> resize 32x32 INTER_AREA -> CV_32F -> cv::dct
//                                             -> 9x9 block, zig-zag, keep 6..69 -> mean via cv::sum -> bits 1..63
//   grayscale        src/cvutil.cpp:1265-1283 cvtColor BGR2GRAY
//   sizeLongestSide  src/cvutil.cpp:1932-1950 resize INTER_LANCZOS4
//   keypoint squares src/media.cpp:874-923    dctHash64(sub-rectangle view, inPlace = true), one after the other
//
// Output (text, one record per line):
//   V <CV_VERSION string>
//   H <w> <h> <seed> <hash:016x> <thresh:08x> <64 coefficient bit patterns :08x> <1024 tile bytes hex>
//   G <w> <h> <seed> <w*h gray bytes hex>                 (input: channels from seeds seed, seed+1, seed+2 as B, G, R)
//   L <w> <h> <seed> <size> <ow> <oh> <ow*oh bytes hex>
//   R <w> <h> <seed> <n> {<x> <y> <side> <hash:016x>}*n <sum of the image's bytes after the in-place blurs>
// ORB (src/media.cpp:859-872) and the library pieces under ColorDescriptor::create (src/cvutil.cpp:790-1099), stage by
// stage so that each recalled piece of oracle/orb_oracle.c / oracle/colordesc_oracle.c gets its own pin:
//   P <w> <h> <seed> <dw> <dh> <dw*dh bytes hex>          cv::resize(INTER_LINEAR): one pyramid step
//   B <w> <h> <seed> <w*h bytes hex>                      GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101)
//   F <w> <h> <seed> <n> {<x> <y> <score>}*n              cv::FAST(img, 20, true)
//   O <w> <h> <seed> <nfeat> <n> {<x:08x> <y:08x> <size:08x> <angle:08x> <response:08x> <octave>}*n <32n bytes hex>
//                                                         detect() then compute(), keypoints as compute() leaves them
//   A <count> {<y:08x> <x:08x> <fastAtan2:08x>}*count
//   M <cols> <rows> <cols*rows bytes hex>                 cv::ellipse(mask, RotatedRect(0.5, 0.9), 255, CV_FILLED)
//   U <n> {<b> <g> <r> <L:08x> <u:08x> <v:08x>}*n         convertTo(CV_32F) * (1/255) -> cvtColor(CV_BGR2Luv)
//   K <N> <iterations-unknown:0> <N labels> <96 centre bit patterns :08x>
//                                                         cv::kmeans(N x 3 samples, 32, (ITER|EPS, 100, 10), 1, PP) with
//                                                         theRNG().state = 0xffffffff; samples = Luv of gen_image planes
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

// ---- deterministic integer-only image generator (twin: gen_image in tools/opencv_golden_to_npz.py) ----------------
static inline uint32_t lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return s >> 8;
}
static inline int tri(int t) {  // triangle wave over 0..1023, range -256..256
  t &= 1023;
  return t < 512 ? t - 256 : 768 - t;
}
static inline int floordiv256(int v) { return v >= 0 ? v / 256 : -((-v + 255) / 256); }
static inline uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x45d9f3bu;
  x ^= x >> 16;
  x *= 0x45d9f3bu;
  x ^= x >> 16;
  return x;
}
static std::vector<uint8_t> gen_image(int w, int h, uint32_t seed) {
  uint32_t s = seed * 2654435761u + 12345u;
  int fx[4], fy[4], ph[4], amp[4];
  for (int k = 0; k < 4; ++k) {
    fx[k] = 1 + (int)(lcg(s) % 7u);
    fy[k] = 1 + (int)(lcg(s) % 7u);
    ph[k] = (int)(lcg(s) % 1024u);
    amp[k] = 10 + (int)(lcg(s) % 30u);
  }
  std::vector<uint8_t> img((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int acc = 0;
      for (int k = 0; k < 4; ++k) acc += amp[k] * tri((x * fx[k] * 1024) / w + (y * fy[k] * 1024) / h + ph[k]);
      const int noise = (int)((mix((uint32_t)(y * w + x) * 2654435761u + seed) >> 24) & 15u) - 8;
      int v = 128 + floordiv256(acc) + noise;
      img[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  return img;
}

// geometries: no blur / 3x3 / 5x5 / 7x7; copy, integer blocks, 2x2, weighted tables, the sizes whose cv::resize scale
// misses the integer by an ulp (1568) or changes a weight (3885), bilinear emulation (< 32); then a batch of 256x256
static const int kGeom[][2] = {{32, 32},   {64, 64},   {256, 256}, {96, 64},   {40, 36},   {100, 100}, {127, 129},
                               {300, 200}, {640, 480}, {1568, 64}, {3885, 33}, {31, 31},   {16, 16},   {20, 100},
                               {5, 31},    {400, 300}, {512, 512}, {1920, 1080}};
static const int kNGeom = (int)(sizeof(kGeom) / sizeof(kGeom[0]));
static const int kBatch256 = 200;  // seeds 1000..1199 at 256x256

static void put_hex(const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i) std::printf("%02x", p[i]);
}

#ifdef NO_OPENCV
// generator self-test: print checksums the Python twin must reproduce
int main() {
  for (int g = 0; g < kNGeom; ++g) {
    std::vector<uint8_t> img = gen_image(kGeom[g][0], kGeom[g][1], 100u + (uint32_t)g);
    uint64_t sum = 0, wsum = 0;
    for (size_t i = 0; i < img.size(); ++i) sum += img[i], wsum += (uint64_t)img[i] * (uint64_t)(i % 251 + 1);
    std::printf("S %d %d %u %llu %llu\n", kGeom[g][0], kGeom[g][1], 100u + (unsigned)g, (unsigned long long)sum,
                (unsigned long long)wsum);
  }
  return 0;
}
#else
#include <opencv2/core/core.hpp>
#include <opencv2/core/version.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

static const char kZigZag[81] = {0,  9,  1,  2,  10, 18, 27, 19, 11, 3,  4,  12, 20, 28, 36, 45, 37, 29, 21, 13, 5,
                                 6,  14, 22, 30, 38, 46, 54, 63, 55, 47, 39, 31, 23, 15, 7,  8,  16, 24, 32, 40, 48,
                                 56, 64, 72, 73, 65, 57, 49, 41, 33, 25, 17, 26, 34, 42, 50, 58, 66, 74, 75, 67, 59,
                                 51, 43, 35, 44, 52, 60, 68, 76, 77, 69, 61, 53, 62, 70, 78, 79, 71, 80};

// the hash of an 8-bit single-channel image or view; in_place: the blur writes into the view (keypoint squares)
static uint64_t hash_gray(cv::Mat gray, bool in_place, float* coefs64, float* thresh_out, uint8_t* tile1024) {
  const int area = gray.size().area();
  const int k = area <= 32 * 32 ? 0 : area <= 64 * 64 ? 3 : area <= 128 * 128 ? 5 : 7;
  if (k) {
    cv::Mat blurred;
    if (in_place) blurred = gray;
    cv::blur(gray, blurred, cv::Size(k, k));
    gray = blurred;
  }
  cv::resize(gray, gray, cv::Size(32, 32), 0, 0, cv::INTER_AREA);
  if (tile1024)
    for (int y = 0; y < 32; ++y) std::memcpy(tile1024 + 32 * y, gray.ptr(y), 32);
  cv::Mat freq;
  gray.convertTo(freq, CV_32F);
  cv::dct(freq, freq);
  freq = freq.rowRange(cv::Range(0, 9)).colRange(cv::Range(0, 9)).clone();
  freq = freq.reshape(1, 1);
  cv::Mat ordered = freq.clone();
  for (int i = 0; i < 81; ++i) ordered.at<float>(0, i) = freq.at<float>(0, (int)kZigZag[i]);
  cv::Mat sel = ordered.colRange(6, 70).clone();
  const float sum = float(cv::sum(sel)[0]);
  const float thresh = sum / 64;
  uint64_t hash = 0;
  const float* row = sel.ptr<float>(0);
  for (int i = 1; i < 64; ++i)
    if (row[i] > thresh) hash |= 1ULL << i;
  if (hash == 0) hash = 1;
  if (coefs64) std::memcpy(coefs64, row, 64 * sizeof(float));
  if (thresh_out) *thresh_out = thresh;
  return hash;
}

static void emit_hash(int w, int h, uint32_t seed) {
  std::vector<uint8_t> img = gen_image(w, h, seed);
  cv::Mat gray(h, w, CV_8UC1, img.data());
  float coefs[64], thresh;
  uint8_t tile[1024];
  const uint64_t hv = hash_gray(gray.clone(), false, coefs, &thresh, tile);
  uint32_t tb;
  std::memcpy(&tb, &thresh, 4);
  std::printf("H %d %d %u %016llx %08x", w, h, seed, (unsigned long long)hv, tb);
  for (int i = 0; i < 64; ++i) {
    uint32_t cb;
    std::memcpy(&cb, &coefs[i], 4);
    std::printf(" %08x", cb);
  }
  std::printf(" ");
  put_hex(tile, 1024);
  std::printf("\n");
}

int main() {
  std::printf("V %s\n", CV_VERSION);
  for (int g = 0; g < kNGeom; ++g) emit_hash(kGeom[g][0], kGeom[g][1], 100u + (uint32_t)g);
  for (int i = 0; i < kBatch256; ++i) emit_hash(256, 256, 1000u + (uint32_t)i);
  {  // grayscale: BGR planes from three seeds
    const int w = 61, h = 37;
    const uint32_t seed = 300;
    std::vector<uint8_t> b = gen_image(w, h, seed), g = gen_image(w, h, seed + 1), r = gen_image(w, h, seed + 2);
    cv::Mat bgr(h, w, CV_8UC3);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) bgr.at<cv::Vec3b>(y, x) = cv::Vec3b(b[y * w + x], g[y * w + x], r[y * w + x]);
    cv::Mat gray;
    cv::cvtColor(bgr, gray, CV_BGR2GRAY);
    std::printf("G %d %d %u ", w, h, seed);
    for (int y = 0; y < h; ++y) put_hex(gray.ptr(y), (size_t)w);
    std::printf("\n");
  }
  {  // sizeLongestSide(img, 128) with the default filter INTER_LANCZOS4
    const int w = 500, h = 300, size = 128;
    const uint32_t seed = 700;
    std::vector<uint8_t> img = gen_image(w, h, seed);
    cv::Mat src(h, w, CV_8UC1, img.data());
    const float aspect = (float)w / h;
    int ow, oh;
    if (w > h) {
      ow = size;
      oh = (int)(size / aspect);
    } else {
      oh = size;
      ow = (int)(size * aspect);
    }
    cv::Mat dst;
    cv::resize(src, dst, cv::Size(ow, oh), 0, 0, cv::INTER_LANCZOS4);
    std::printf("L %d %d %u %d %d %d ", w, h, seed, size, ow, oh);
    for (int y = 0; y < oh; ++y) put_hex(dst.ptr(y), (size_t)ow);
    std::printf("\n");
  }
  {  // keypoint squares hashed in place, in order (overlapping on purpose)
    const int w = 400, h = 300;
    const uint32_t seed = 500;
    std::vector<uint8_t> img = gen_image(w, h, seed);
    cv::Mat gray(h, w, CV_8UC1, img.data());
    const int rects[][3] = {{12, 20, 31},  {40, 33, 45},   {15, 25, 65},  {100, 60, 134}, {30, 30, 38},
                            {200, 150, 54}, {210, 160, 77}, {5, 5, 31},    {250, 100, 93}, {120, 80, 112}};
    const int n = (int)(sizeof(rects) / sizeof(rects[0]));
    std::printf("R %d %d %u %d", w, h, seed, n);
    for (int i = 0; i < n; ++i) {
      cv::Mat sub = gray(cv::Rect(rects[i][0], rects[i][1], rects[i][2], rects[i][2]));
      const uint64_t hv = hash_gray(sub, true, nullptr, nullptr, nullptr);
      std::printf(" %d %d %d %016llx", rects[i][0], rects[i][1], rects[i][2], (unsigned long long)hv);
    }
    uint64_t sum = 0;
    for (size_t i = 0; i < img.size(); ++i) sum += img[i];
    std::printf(" %llu\n", (unsigned long long)sum);
  }
  {  // ---- ORB, stage by stage -------------------------------------------------------------------------------
    const int w = 400, h = 300;
    const uint32_t seed = 900;
    std::vector<uint8_t> img = gen_image(w, h, seed);
    cv::Mat gray(h, w, CV_8UC1, img.data());
    {
      cv::Mat lvl;
      cv::resize(gray, lvl, cv::Size(333, 250), 0, 0, cv::INTER_LINEAR);
      std::printf("P %d %d %u %d %d ", w, h, seed, lvl.cols, lvl.rows);
      for (int y = 0; y < lvl.rows; ++y) put_hex(lvl.ptr(y), (size_t)lvl.cols);
      std::printf("\n");
    }
    {
      cv::Mat bl;
      cv::GaussianBlur(gray, bl, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
      std::printf("B %d %d %u ", w, h, seed);
      for (int y = 0; y < h; ++y) put_hex(bl.ptr(y), (size_t)w);
      std::printf("\n");
    }
    {
      std::vector<cv::KeyPoint> kps;
      cv::FAST(gray, kps, 20, true);
      std::printf("F %d %d %u %d", w, h, seed, (int)kps.size());
      for (size_t i = 0; i < kps.size(); ++i) std::printf(" %d %d %d", (int)kps[i].pt.x, (int)kps[i].pt.y, (int)kps[i].response);
      std::printf("\n");
    }
    {
      std::vector<cv::KeyPoint> kps;
      cv::OrbFeatureDetector detector(400, 1.2f, 12, 31, 0, 2, cv::OrbFeatureDetector::HARRIS_SCORE, 31);
      detector.detect(gray, kps);
      cv::Mat desc;
      cv::OrbDescriptorExtractor extractor;
      extractor.compute(gray, kps, desc);
      std::printf("O %d %d %u %d %d", w, h, seed, 400, (int)kps.size());
      for (size_t i = 0; i < kps.size(); ++i) {
        uint32_t b[5];
        std::memcpy(b, &kps[i].pt.x, 4), std::memcpy(b + 1, &kps[i].pt.y, 4), std::memcpy(b + 2, &kps[i].size, 4);
        std::memcpy(b + 3, &kps[i].angle, 4), std::memcpy(b + 4, &kps[i].response, 4);
        std::printf(" %08x %08x %08x %08x %08x %d", b[0], b[1], b[2], b[3], b[4], kps[i].octave);
      }
      std::printf(" ");
      for (int r = 0; r < desc.rows; ++r) put_hex(desc.ptr(r), 32);
      std::printf("\n");
    }
    {
      const float ys[] = {0.f, 1.f, 1.f, -3.f, 250.5f, -1e-3f, 7.f, -1234.f}, xs[] = {1.f, 1.f, -1.f, 0.f, -17.25f, 5.f, -7.f, -4321.f};
      std::printf("A %d", 8);
      for (int i = 0; i < 8; ++i) {
        const float a = cv::fastAtan2(ys[i], xs[i]);
        uint32_t b[3];
        std::memcpy(b, &ys[i], 4), std::memcpy(b + 1, &xs[i], 4), std::memcpy(b + 2, &a, 4);
        std::printf(" %08x %08x %08x", b[0], b[1], b[2]);
      }
      std::printf("\n");
    }
  }
  {  // ---- the library pieces under ColorDescriptor::create ------------------------------------------------
    const int dims[][2] = {{256, 192}, {192, 256}, {100, 100}, {40, 30}, {255, 131}};
    for (int d = 0; d < 5; ++d) {
      cv::Mat mask(dims[d][1], dims[d][0], CV_8UC1);
      mask = mask.setTo(0);
      cv::RotatedRect box(cv::Point2f(mask.cols * 0.5f, mask.rows * 0.5f), cv::Size2f(mask.cols * 0.9f, mask.rows * 0.9f), 0.0f);
      cv::ellipse(mask, box, 255, CV_FILLED);
      std::printf("M %d %d ", mask.cols, mask.rows);
      for (int y = 0; y < mask.rows; ++y) put_hex(mask.ptr(y), (size_t)mask.cols);
      std::printf("\n");
    }
    const int w = 120, h = 90;
    const uint32_t seed = 950;
    std::vector<uint8_t> pb = gen_image(w, h, seed), pg = gen_image(w, h, seed + 1), pr = gen_image(w, h, seed + 2);
    cv::Mat bgr(h, w, CV_8UC3);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) bgr.at<cv::Vec3b>(y, x) = cv::Vec3b(pb[y * w + x], pg[y * w + x], pr[y * w + x]);
    cv::Mat luv;
    bgr.convertTo(luv, CV_32FC3);
    luv *= 1.0 / 255.0;
    cv::cvtColor(luv, luv, CV_BGR2Luv);
    std::printf("U %d", 256);
    for (int i = 0; i < 256; ++i) {
      const int y = (i * 37) % h, x = (i * 101) % w;
      const cv::Vec3b c = bgr.at<cv::Vec3b>(y, x);
      const cv::Vec3f v = luv.at<cv::Vec3f>(y, x);
      uint32_t b[3];
      std::memcpy(b, &v[0], 4), std::memcpy(b + 1, &v[1], 4), std::memcpy(b + 2, &v[2], 4);
      std::printf(" %d %d %d %08x %08x %08x", c[0], c[1], c[2], b[0], b[1], b[2]);
    }
    std::printf("\n");
    {
      std::vector<cv::Point3f> samples;
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
          const cv::Vec3f v = luv.at<cv::Vec3f>(y, x);
          samples.push_back(cv::Point3f(v[0], v[1], v[2]));
        }
      cv::theRNG().state = 0xffffffff;  // a fresh thread's generator
      cv::Mat labels, centers;
      (void)cv::kmeans(samples, 32, labels, cvTermCriteria(CV_TERMCRIT_ITER | CV_TERMCRIT_EPS, 100, 10), 1,
                       cv::KMEANS_PP_CENTERS, centers);
      std::printf("K %d 0", (int)samples.size());
      for (int i = 0; i < labels.rows; ++i) std::printf(" %d", labels.at<int>(i));
      for (int k = 0; k < 32; ++k)
        for (int j = 0; j < 3; ++j) {
          const float c = centers.at<float>(k, j);
          uint32_t b;
          std::memcpy(&b, &c, 4);
          std::printf(" %08x", b);
        }
      std::printf("\n");
    }
  }
  return 0;
}
#endif
