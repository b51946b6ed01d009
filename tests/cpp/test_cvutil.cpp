// test_cvutil.cpp -- the dctHash64 / makeKeyPointHashes drop-ins (cbird_amd/cpp/gpu_cvutil.h) compiled against the
// mock cv::Mat, run on the MI355X and compared with values the Python test passes in (computed by the oracle).
//
//   test_cvutil <w> <h> <seed> : builds the same pseudo-random image the Python side builds (xorshift bytes),
//   prints one line per check:  name hash...
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "index.h"
#include "gpu_cvutil.h"

static uint32_t xs(uint32_t& s) {
  s ^= s << 13, s ^= s >> 17, s ^= s << 5;
  return s;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int w = atoi(argv[1]), h = atoi(argv[2]);
  uint32_t seed = uint32_t(atoi(argv[3]));
  cv::Mat img(h, w);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) img.ptr<uint8_t>(y)[x] = uint8_t(xs(seed) >> 24);

  // 1. whole image, not in place: pixels untouched
  cv::Mat copy(h, w);
  for (int y = 0; y < h; ++y) memcpy(copy.ptr<uint8_t>(y), img.ptr<uint8_t>(y), size_t(w));
  printf("whole %" PRIu64 "\n", cbird_gpu::gpuDctHash64(img));
  for (int y = 0; y < h; ++y)
    if (memcmp(copy.ptr<uint8_t>(y), img.ptr<uint8_t>(y), size_t(w))) return 3;

  // 2. a view, not in place (blur sees the parent's pixels around it), then in place (pixels change)
  const int vx = w / 5, vy = h / 4, vw = w / 2, vh = h / 2;
  cv::Mat view = img.colRange(vx, vx + vw).rowRange(vy, vy + vh);
  printf("view %" PRIu64 "\n", cbird_gpu::gpuDctHash64(view, false));
  printf("view_inplace %" PRIu64 "\n", cbird_gpu::gpuDctHash64(view, true));
  uint64_t sum = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) sum = sum * 1099511628211ull + img.ptr<uint8_t>(y)[x];
  printf("after_view_checksum %" PRIu64 "\n", sum);

  // 3. keypoint hashes on the (now partly blurred) image
  KeyPointList kps;
  for (int i = 0; i < 40; ++i) {
    const float size = 31.f * (i % 4 == 0 ? 1.f : i % 4 == 1 ? 1.2f : i % 4 == 2 ? 1.44f : 2.0736f);
    const float kx = float(xs(seed) % uint32_t(w)) + 0.25f;  // (two statements: argument evaluation order is unspecified)
    const float ky = float(xs(seed) % uint32_t(h)) + 0.5f;
    kps.push_back(cv::KeyPoint(kx, ky, size));
  }
  KeyPointHashList hashes;
  hashes.push_back(42);  // appended to, not replaced
  cbird_gpu::gpuMakeKeyPointHashes(img, kps, hashes);
  printf("kp");
  for (uint64_t v : hashes) printf(" %" PRIu64, v);
  printf("\n");
  sum = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) sum = sum * 1099511628211ull + img.ptr<uint8_t>(y)[x];
  printf("after_kp_checksum %" PRIu64 "\n", sum);
  // 4. sizeLongestSide on a fresh copy of the original pattern (img itself was blurred above)
  cv::Mat big(h, w);
  uint32_t seed2 = uint32_t(atoi(argv[3]));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) big.ptr<uint8_t>(y)[x] = uint8_t(xs(seed2) >> 24);
  cbird_gpu::gpuSizeLongestSide(big, 128);
  sum = 0;
  for (int y = 0; y < big.rows; ++y)
    for (int x = 0; x < big.cols; ++x) sum = sum * 1099511628211ull + big.ptr<uint8_t>(y)[x];
  printf("resized %d %d %" PRIu64 "\n", big.cols, big.rows, sum);
  // 5. ORB: makeKeyPoints, then makeKeyPointDescriptors on its result, as Scanner::processImage calls them; the
  //    pattern is the stand-in the Python side generates too (x = (i * 7 + 3) % 27 - 13 ...)
  if (w > 62 && h > 62) {
    int pat[1024];
    for (int i = 0; i < 1024; ++i) pat[i] = (i * 7 + (i / 4) * 3 + 3) % 27 - 13;
    cbird_gpu::gpuOrbSetPattern(pat);
    cv::Mat scene(h, w);
    uint32_t s3 = uint32_t(atoi(argv[3])) + 17u;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) scene.ptr<uint8_t>(y)[x] = uint8_t(((x / 9 + y / 7) % 2 ? 200 : 40) + int(xs(s3) % 9u));
    KeyPointList orbKp;
    cbird_gpu::gpuMakeKeyPoints(scene, 100, orbKp);
    printf("orb_n %zu\n", orbKp.size());
    KeyPointDescriptors descr;
    cbird_gpu::gpuMakeKeyPointDescriptors(scene, orbKp, descr);
    sum = 0;
    for (const cv::KeyPoint& k : orbKp) {
      uint32_t bits[5];
      memcpy(bits, &k.pt.x, 4), memcpy(bits + 1, &k.pt.y, 4), memcpy(bits + 2, &k.size, 4);
      memcpy(bits + 3, &k.angle, 4), memcpy(bits + 4, &k.response, 4);
      for (uint32_t b : bits) sum = sum * 1099511628211ull + b;
      sum = sum * 1099511628211ull + uint32_t(k.octave);
    }
    printf("orb_kp %zu %" PRIu64 "\n", orbKp.size(), sum);
    sum = 0;
    for (int r = 0; r < descr.rows; ++r)
      for (int c = 0; c < 32; ++c) sum = sum * 1099511628211ull + descr.ptr<uint8_t>(r)[c];
    printf("orb_desc %d %" PRIu64 "\n", descr.rows, sum);
  }
  // 6. ColorDescriptor::create on a BGR image built from the same generator; grey input leaves the descriptor alone
  {
    cv::Mat bgr(h, w, CV_8UC3);
    uint32_t s4 = uint32_t(atoi(argv[3])) + 99u;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        for (int c = 0; c < 3; ++c)
          bgr.ptr<uint8_t>(y)[3 * x + c] = uint8_t(((x / 23 + 2 * (y / 17) + c) % 5) * 50 + int(xs(s4) % 7u));
    ColorDescriptor cd;
    cbird_gpu::gpuColorDescriptorCreate(bgr, cd);
    sum = 0;
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(&cd);
    for (size_t i = 0; i < 257; ++i) sum = sum * 1099511628211ull + pb[i];
    printf("color %d %" PRIu64 "\n", int(cd.numColors), sum);
    ColorDescriptor untouched;
    untouched.numColors = 77;
    cbird_gpu::gpuColorDescriptorCreate(img, untouched);  // CV_8UC1
    printf("color_gray %d\n", int(untouched.numColors));
  }
  // 7. TemplateMatcher::match's scoring block: the BGR image of step 6 (regenerated) as the template, BGRA variant
  //    with a varying alpha; the candidate = the template shifted by 3 px with a 12-px undefined (black) margin
  {
    cv::Mat tmpl3(h, w, CV_8UC3), tmpl4(h, w, CV_8UC4), cand(h, w, CV_8UC3);
    uint32_t s5 = uint32_t(atoi(argv[3])) + 99u;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        for (int c = 0; c < 3; ++c) {
          const uint8_t v = uint8_t(((x / 23 + 2 * (y / 17) + c) % 5) * 50 + int(xs(s5) % 7u));
          tmpl3.ptr<uint8_t>(y)[3 * x + c] = v;
          tmpl4.ptr<uint8_t>(y)[4 * x + c] = v;
        }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        tmpl4.ptr<uint8_t>(y)[4 * x + 3] = uint8_t(255 - (x + y) % 97);
        for (int c = 0; c < 3; ++c) {
          const bool inside = x >= 12 && y >= 12 && x < w - 12 && y < h - 12;
          cand.ptr<uint8_t>(y)[3 * x + c] = inside ? tmpl3.ptr<uint8_t>(y)[3 * (x - 3) + c] : uint8_t(0);
        }
      }
    uint64_t ch = 0, th = 0;
    const int d3 = cbird_gpu::gpuTemplateScore(cand, tmpl3, &ch, &th);
    printf("tm3 %d %" PRIu64 " %" PRIu64 "\n", d3, ch, th);
    const int d4 = cbird_gpu::gpuTemplateScore(cand, tmpl4, &ch, &th);
    printf("tm4 %d %" PRIu64 " %" PRIu64 "\n", d4, ch, th);
  }
  return 0;
}
