// gpu_cvutil.h -- drop-ins for the two hash producers of cbird's indexer, on libcbird_hip:
//
//   uint64_t dctHash64(const cv::Mat& cvImg, bool inPlace = false)                  src/cvutil.h, src/cvutil.cpp:435-545
//   void Media::makeKeyPointHashes(const cv::Mat&, const KeyPointList&, KeyPointHashList&) const   src/media.cpp:874-923
//   void sizeLongestSide(cv::Mat& img, int size, int filter = INTER_LANCZOS4)        src/cvutil.h:251, cvutil.cpp:1932-1950
//   void Media::makeKeyPoints(const cv::Mat&, int numKeyPoints, KeyPointList&) const                 src/media.cpp:859-866
//   void Media::makeKeyPointDescriptors(const cv::Mat&, KeyPointList&, KeyPointDescriptors&) const   src/media.cpp:868-872
//   static void ColorDescriptor::create(const cv::Mat& cvImg, ColorDescriptor& desc)                 src/cvutil.cpp:790-1099
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
#include <cstdint>
#include <cstring>
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
  for (;;) {  // ties in retainBest are never cut: a count above the capacity asks for a second call
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

}  // namespace cbird_gpu
