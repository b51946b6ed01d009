// gpu_cvutil.h -- drop-ins for the two hash producers of cbird's indexer, on libcbird_hip:
//
//   uint64_t dctHash64(const cv::Mat& cvImg, bool inPlace = false)                  src/cvutil.h, src/cvutil.cpp:435-545
//   void Media::makeKeyPointHashes(const cv::Mat&, const KeyPointList&, KeyPointHashList&) const   src/media.cpp:874-923
//   void sizeLongestSide(cv::Mat& img, int size, int filter = INTER_LANCZOS4)        src/cvutil.h:251, cvutil.cpp:1932-1950
//   void Media::makeKeyPoints(const cv::Mat&, int numKeyPoints, KeyPointList&) const                 src/media.cpp:859-866
//   void Media::makeKeyPointDescriptors(const cv::Mat&, KeyPointList&, KeyPointDescriptors&) const   src/media.cpp:868-872
//   static void ColorDescriptor::create(const cv::Mat& cvImg, ColorDescriptor& desc)                 src/cvutil.cpp:790-1099
//   void Media::makeVideoIndex(VideoContext&, int threshold, VideoIndex&, const std::function<void(int)>&) const
//                                                                                                  src/media.cpp:925-1037
//   the scoring block of TemplateMatcher::match (mask, two dctHash64, hamm64)                     src/templatematcher.cpp:331-371
//
// Same arguments and effects as the originals for 8-bit single-channel images (what Scanner::processImage passes
// after grayscale(), src/scanner.cpp:859,876-889): the hash is returned, and with inPlace = true the blurred pixels
// are written back into the caller's image.  A cv::Mat that is a VIEW into a larger image (colRange/rowRange) is
// blurred with the parent's pixels around it, as cv::blur does (Mat::locateROI); the view's parent is what gets
// staged on the device.
//
// These are per-call conveniences (one image, one H2D copy per call).  The batched entry points
// cbh_dcthash_batch / cbh_process_images / cbh_keypoint_hashes are what an indexer that wants the GPU's throughput
// calls with many images at once.
//
// Build note: needs cbird's cvutil.h / media.h (OpenCV 2.4 cv::Mat, cv::KeyPoint) on the include path; in this
// repository it is compiled against tests/cpp/mock/index.h instead (tests/cpp/test_cvutil.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <vector>

#include "cbird_hip.h"

namespace cbird_gpu {

inline int& hashDevice() {  // the device the per-call helpers use
  static int dev = 0;
  return dev;
}

namespace detail {
struct ParentView {
  const uint8_t* base;  // first pixel of the parent image
  uint32_t w, h, step;
  int32_t x, y;         // the view's offset inside it
};
inline ParentView parentOf(const cv::Mat& m) {
  cv::Size whole;
  cv::Point ofs;
  m.locateROI(whole, ofs);
  ParentView p;
  p.step = uint32_t(m.step);
  p.base = m.data - size_t(ofs.y) * m.step - size_t(ofs.x);
  p.w = uint32_t(whole.width), p.h = uint32_t(whole.height);
  p.x = ofs.x, p.y = ofs.y;
  return p;
}
}  // namespace detail

// dctHash64(cvImg, inPlace) for CV_8UC1.  Colour input is converted by the caller exactly as before
// (grayscale(), cvutil.cpp:1265-1283; on the GPU: cbh_process_images).
inline uint64_t gpuDctHash64(const cv::Mat& cvImg, bool inPlace = false) {
  if (cvImg.type() != CV_8UC1 || cvImg.rows <= 0 || cvImg.cols <= 0)
    qFatal("gpuDctHash64: expected a non-empty CV_8UC1 image");
  const detail::ParentView p = detail::parentOf(cvImg);
  const size_t bytes = size_t(p.h - 1) * p.step + p.w;
  const uint64_t off = 0;
  const int32_t rect[4] = {p.x, p.y, cvImg.cols, cvImg.rows};
  const uint32_t first[2] = {0, 1};
  uint64_t hash = 0;
  // in place: the device copy of the parent comes back over the caller's pixels (only the view's rectangle differs)
  const int rc = cbh_dcthash_rects(p.base, bytes, 1, &off, &p.w, &p.h, &p.step, rect, first, inPlace ? 1 : 0, &hash,
                                   inPlace ? const_cast<uint8_t*>(p.base) : nullptr, hashDevice());
  if (rc) qFatal("gpuDctHash64: %s (%s)", cbh_strerror(rc), cbh_last_error());
  return hash;
}

// Media::makeKeyPointHashes: the hashes are appended to outHashes like the original does (push_back per rectangle);
// cvImg is modified by the in-place blurs.
inline void gpuMakeKeyPointHashes(const cv::Mat& cvImg, const KeyPointList& keyPoints, KeyPointHashList& outHashes) {
  Q_ASSERT(cvImg.type() == CV_8UC1);  // grayscale, media.cpp:877
  if (keyPoints.empty() || cvImg.rows <= 0 || cvImg.cols <= 0) return;
  const detail::ParentView p = detail::parentOf(cvImg);
  if (p.x != 0 || p.y != 0 || int(p.w) != cvImg.cols || int(p.h) != cvImg.rows)
    qFatal("gpuMakeKeyPointHashes: expected a whole image, not a view");
  std::vector<float> kp;
  kp.reserve(keyPoints.size() * 3);
  for (const cv::KeyPoint& k : keyPoints) {
    kp.push_back(k.pt.x);
    kp.push_back(k.pt.y);
    kp.push_back(k.size);
  }
  const size_t bytes = size_t(p.h - 1) * p.step + p.w;
  const uint64_t off = 0;
  const uint32_t kpFirst[2] = {0, uint32_t(keyPoints.size())};
  uint32_t outFirst[2] = {0, 0};
  std::vector<uint64_t> hashes(keyPoints.size());
  const int rc = cbh_keypoint_hashes(p.base, bytes, 1, &off, &p.w, &p.h, &p.step, kp.data(), kpFirst, hashes.data(),
                                     outFirst, const_cast<uint8_t*>(p.base), hashDevice());
  if (rc) qFatal("gpuMakeKeyPointHashes: %s (%s)", cbh_strerror(rc), cbh_last_error());
  for (uint32_t i = 0; i < outFirst[1]; ++i) outHashes.push_back(hashes[i]);
}

// sizeLongestSide(img, size) with the default INTER_LANCZOS4 filter: img is replaced by the resized image.
inline void gpuSizeLongestSide(cv::Mat& img, int size) {
  if (img.type() != CV_8UC1 || img.rows <= 0 || img.cols <= 0) qFatal("gpuSizeLongestSide: expected a CV_8UC1 image");
  int w = 0, h = 0;
  cbh_longest_side_dims(img.cols, img.rows, size, &w, &h);
  if (w == 0 || h == 0)  // the reference's own check (cvutil.cpp:1944-1946)
    throw std::invalid_argument("sizeLongestSide: computed width or height is 0, probably bad input");
  cv::Mat out(h, w, CV_8UC1);
  const int rc = cbh_size_longest_side(img.data, 1, img.cols, img.rows, size_t(img.step), 0, size, out.data, &w, &h,
                                       hashDevice());
  if (rc) qFatal("gpuSizeLongestSide: %s (%s)", cbh_strerror(rc), cbh_last_error());
  img = out;
}

// ---- ORB (cbh_orb*, cbird_amd/csrc/orb.hip) ---------------------------------------------------------------------
// rBRIEF's test pairs are OpenCV's learned table bit_pattern_31_ (modules/features2d/src/orb.cpp), an input of the
// library: hand it over once, e.g. gpuOrbSetPattern(bit_pattern_31_) from a translation unit that has the table.
inline void gpuOrbSetPattern(const int* bitPattern31 /* 256 * 4 */) {
  int8_t xy[1024];
  for (int i = 0; i < 1024; ++i) xy[i] = int8_t(bitPattern31[i]);
  const int rc = cbh_orb_set_pattern(xy);
  if (rc) qFatal("gpuOrbSetPattern: %s", cbh_strerror(rc));
}

namespace detail {
inline void wholeImage(const cv::Mat& cvImg, const char* who, ParentView* p) {
  if (cvImg.type() != CV_8UC1 || cvImg.rows <= 0 || cvImg.cols <= 0) qFatal("%s: expected a non-empty CV_8UC1 image", who);
  *p = parentOf(cvImg);
  if (p->x != 0 || p->y != 0 || int(p->w) != cvImg.cols || int(p->h) != cvImg.rows)
    qFatal("%s: expected a whole image, not a view", who);
}
}  // namespace detail

// Media::makeKeyPoints: outKeypoints is replaced, like cv::FeatureDetector::detect does
inline void gpuMakeKeyPoints(const cv::Mat& cvImg, int numKeyPoints, KeyPointList& outKeypoints) {
  detail::ParentView p;
  detail::wholeImage(cvImg, "gpuMakeKeyPoints", &p);
  const size_t bytes = size_t(p.h - 1) * p.step + p.w;
  const uint64_t off = 0;
  int cap = numKeyPoints + 64;
  std::vector<cbh_keypoint> kp;
  uint32_t count = 0;
  for (;;) {  // retainBest can keep ties beyond n: a count above the capacity asks for a second call
    kp.resize(size_t(cap));
    const int rc = cbh_orb(p.base, bytes, 1, &off, &p.w, &p.h, &p.step, numKeyPoints, cap, kp.data(), nullptr, nullptr,
                           &count, hashDevice());
    if (rc) qFatal("gpuMakeKeyPoints: %s (%s)", cbh_strerror(rc), cbh_last_error());
    if (int(count) <= cap) break;
    cap = int(count);
  }
  outKeypoints.clear();
  for (uint32_t i = 0; i < count; ++i)
    outKeypoints.push_back(cv::KeyPoint(kp[i].x, kp[i].y, kp[i].size, kp[i].angle, kp[i].response, kp[i].octave, -1));
}

// Media::makeKeyPointDescriptors: keyPoints is in/out as in the original (border filter, grouping by octave, the
// coordinate round trip); outDescriptors becomes a keyPoints.size() x 32 CV_8UC1 matrix
inline void gpuMakeKeyPointDescriptors(const cv::Mat& cvImg, KeyPointList& keyPoints, KeyPointDescriptors& outDescriptors) {
  detail::ParentView p;
  detail::wholeImage(cvImg, "gpuMakeKeyPointDescriptors", &p);
  const size_t bytes = size_t(p.h - 1) * p.step + p.w;
  const uint64_t off = 0;
  std::vector<cbh_keypoint> in(keyPoints.size()), out(keyPoints.size() + 1);
  for (size_t i = 0; i < keyPoints.size(); ++i) {
    const cv::KeyPoint& k = keyPoints[i];
    in[i] = cbh_keypoint{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave};
  }
  std::vector<uint8_t> desc((keyPoints.size() + 1) * 32);
  const uint32_t first[2] = {0, uint32_t(keyPoints.size())};
  uint32_t outFirst[2] = {0, 0};
  const int rc = cbh_orb_describe(p.base, bytes, 1, &off, &p.w, &p.h, &p.step, in.data(), first, out.data(), desc.data(),
                                  outFirst, hashDevice());
  if (rc) qFatal("gpuMakeKeyPointDescriptors: %s (%s)", cbh_strerror(rc), cbh_last_error());
  keyPoints.clear();
  outDescriptors = cv::Mat(int(outFirst[1]), 32, CV_8UC1);
  for (uint32_t i = 0; i < outFirst[1]; ++i) {
    keyPoints.push_back(cv::KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, -1));
    memcpy(outDescriptors.ptr<uint8_t>(int(i)), desc.data() + size_t(i) * 32, 32);
  }
}

// ColorDescriptor::create: desc is written only when the reference would write it (a BGR / BGRA image with at least 32
// samples brighter than L = 4); one image per call -- an indexer that wants the GPU's throughput batches them through
// cbh_color_descriptors / cbh_index_images (the clustering kernel runs one lane per image)
inline void gpuColorDescriptorCreate(const cv::Mat& cvImg, ColorDescriptor& desc) {
  if (cvImg.type() != CV_8UC3 && cvImg.type() != CV_8UC4) {
    qDebug("passed a grayscale image");
    return;
  }
  static_assert(sizeof(ColorDescriptor) == 258, "the record ColorDescIndex stores");
  const uint64_t off = 0;
  const uint32_t w = uint32_t(cvImg.cols), h = uint32_t(cvImg.rows), step = uint32_t(cvImg.step);
  uint8_t rec[258], ok = 0;
  const int rc = cbh_color_descriptors(cvImg.data, size_t(h - 1) * step + size_t(w) * size_t(cvImg.channels()), 1, &off, &w,
                                       &h, &step, cvImg.channels(), rec, &ok, hashDevice());
  if (rc) qFatal("gpuColorDescriptorCreate: %s (%s)", cbh_strerror(rc), cbh_last_error());
  if (ok) memcpy(&desc, rec, sizeof desc);
  else qWarning("not enough colors");
}

// Media::makeVideoIndex(video, threshold, outIndex, progressCb) -- src/media.cpp:925-1037.  Same resume rule
// (:929-936), same result; the frames are collected chunkFrames at a time and hashed on the device
// (autocrop(img, 20) + dctHash64), the near-frame filter runs in the library between chunks.  VideoContextT is
// cbird's VideoContext: seek(int), nextFrame(cv::Mat&), metadata().frameRate / .duration.
template <class VideoContextT>
inline void gpuMakeVideoIndex(VideoContextT& video, int threshold, VideoIndex& outIndex,
                              const std::function<void(int)>& progressCb = std::function<void(int)>(),
                              int chunkFrames = 64) {
  VideoIndex& index = outIndex;
  cbh_vindexer* ix = cbh_vindexer_create(hashDevice(), threshold, 20 /* FIXME upstream: index settings, :961 */);
  if (!ix) qFatal("gpuMakeVideoIndex: no usable device");
  if (index.frames.size() > 0 && index.frames.size() == index.hashes.size() && video.seek(index.frames.back() + 1)) {
    static_assert(sizeof(index.frames[0]) == sizeof(int32_t) && sizeof(index.hashes[0]) == sizeof(uint64_t), "");
    const int rc = cbh_vindexer_resume(ix, reinterpret_cast<const int32_t*>(index.frames.data()),
                                       reinterpret_cast<const uint64_t*>(index.hashes.data()), index.frames.size());
    if (rc != CBH_OK) {
      cbh_vindexer_destroy(ix);
      qFatal("gpuMakeVideoIndex: cannot resume from the given index");
    }
    qDebug("resuming index from frame: %d", index.frames.back() + 1);
  }
  index.hashes.clear();
  index.frames.clear();
  const int totalFrames = int(video.metadata().frameRate * video.metadata().duration);
  if (chunkFrames < 1) chunkFrames = 1;
  std::vector<uint8_t> stage;
  int cw = 0, ch = 0, held = 0;
  auto flush = [&]() {
    if (held == 0) return;
    const int rc = cbh_vindexer_push(ix, stage.data(), size_t(held), cw, ch, size_t(cw), size_t(cw) * size_t(ch));
    if (rc != CBH_OK) {
      cbh_vindexer_destroy(ix);
      qFatal("gpuMakeVideoIndex: cbh_vindexer_push failed");
    }
    held = 0;
    if (progressCb) progressCb(int(cbh_vindexer_frames_seen(ix) * 100 / std::max(totalFrames, 1)));
  };
  cv::Mat cvFrame;
  while (video.nextFrame(cvFrame)) {
    if (cvFrame.type() != CV_8UC1 || cvFrame.rows <= 0 || cvFrame.cols <= 0)
      qFatal("gpuMakeVideoIndex: the decoder is expected to output grey frames (media.cpp:959)");
    if (cvFrame.cols != cw || cvFrame.rows != ch) {  // first frame (or a mid-stream size change)
      flush();
      cw = cvFrame.cols, ch = cvFrame.rows;
      stage.resize(size_t(chunkFrames) * size_t(cw) * size_t(ch));
    }
    uint8_t* dst = stage.data() + size_t(held) * size_t(cw) * size_t(ch);
    for (int y = 0; y < ch; ++y) memcpy(dst + size_t(y) * size_t(cw), cvFrame.template ptr<uint8_t>(y), size_t(cw));
    if (++held == chunkFrames) flush();
    if (cbh_vindexer_frames_seen(ix) + held >= (1 << 24)) break;  // MAX_FRAMES_PER_VIDEO, :1013-1016
  }
  flush();
  const long long n = cbh_vindexer_finish(ix, nullptr, nullptr, 0);
  if (n > 0) {
    index.frames.resize(size_t(n));
    index.hashes.resize(size_t(n));
    cbh_vindexer_finish(ix, reinterpret_cast<int32_t*>(index.frames.data()),
                        reinterpret_cast<uint64_t*>(index.hashes.data()), size_t(n));
  }
  cbh_vindexer_destroy(ix);
  if (progressCb) progressCb(100);
}

// The scoring block of TemplateMatcher::match, src/templatematcher.cpp:331-371: `img` is the candidate patch as
// cv::warpAffine(..., tmplImg.size(), ...) left it, tmplImg the template (8-bit, 1 / 3 / 4 channels each).  Replaces
//     cv::Mat tmplMasked = tmplImg.clone(); grayscale(img, img); { the masking loop }
//     uint64_t candHash = dctHash64(img); uint64_t tmplHash = dctHash64(tmplMasked); int dist = hamm64(candHash, tmplHash);
// by   int dist = cbird_gpu::gpuTemplateScore(img, tmplImg);
// (the caller's img / tmplImg are left as they are: nothing after :371 reads them).
inline int gpuTemplateScore(const cv::Mat& img, const cv::Mat& tmplImg, uint64_t* candHash = nullptr,
                            uint64_t* tmplHash = nullptr) {
  if (img.depth() != CV_8U || tmplImg.depth() != CV_8U || img.rows != tmplImg.rows || img.cols != tmplImg.cols ||
      img.rows <= 0 || img.cols <= 0)
    qFatal("gpuTemplateScore: expected two 8-bit images of the template's size");
  uint64_t ch = 0, th = 0;
  int32_t score = 0;
  const int rc = cbh_template_scores(img.data, 1, img.cols, img.rows, size_t(img.step), 0, img.channels(), tmplImg.data,
                                     size_t(tmplImg.step), tmplImg.channels(), &ch, &th, &score, hashDevice());
  if (rc != CBH_OK) qFatal("gpuTemplateScore: cbh_template_scores failed");
  if (candHash) *candHash = ch;
  if (tmplHash) *tmplHash = th;
  return score;
}

}  // namespace cbird_gpu
