/* oracle/retain_stl.cpp -- TEST INFRASTRUCTURE ONLY (parity checker; never linked into the product).
 *
 * KeyPointsFilter::retainBest as OpenCV 2.4 writes it (modules/features2d/src/keypoint.cpp, restated AS RECALLED: the
 * library is not vendored in /root/reference), run on the REAL std::nth_element / std::partition of the libstdc++ this
 * image carries (GCC 11; the introselect of bits/stl_algo.h has not changed since GCC 4.9: median of first+1 / mid /
 * last-1 moved to first, unguarded Hoare partition, insertion sort below 4 elements, heap select at depth 0).  cbird's
 * Linux builds link libstdc++, so WHICH of several keypoints with equal response survive a cut, and the ORDER the
 * survivors are left in (which Media::makeKeyPointHashes inherits, /root/reference/src/media.cpp:874-923), follow from
 * this file.  It pins the selection ORDER of cbird_amd/csrc/orb.hip's `orb_retain_order = 1` path on the real
 * library; what it does not pin is retainBest's own few lines, recalled:
 *
 *     if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
 *       if (n_points == 0) { keypoints.clear(); return; }
 *       std::nth_element(begin, begin + n_points, end, KeypointResponseGreater());       // a.response > b.response
 *       float ambiguous_response = keypoints[n_points - 1].response;
 *       new_end = std::partition(begin + n_points, end, KeypointResponseGreaterThanThreshold(ambiguous_response));
 *       keypoints.resize(new_end - begin);                                               // response >= value
 *     }
 *
 * Note what that does: element n_points - 1 after nth_element is SOME element of the best n, not their minimum, so the
 * tail keeps its ties only when introselect happened to leave the minimum there.  The "every tie is kept" rule of the
 * canonical order (oracle/orb_oracle.c header, (2)) is therefore a superset of this one. */
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct Kp { /* cv::KeyPoint (2.4): pt, size, angle, response, octave, class_id -- plus where it came from */
  float x, y, size, angle, response;
  int octave, class_id;
  int32_t ref;
};
struct ResponseGreater {
  bool operator()(const Kp& a, const Kp& b) const { return a.response > b.response; }
};
struct ResponseGeThreshold {
  float value;
  bool operator()(const Kp& k) const { return k.response >= value; }
};
}  // namespace

/* resp[cnt] -> order[]: the original positions of the survivors in the order retainBest leaves them; returns how many.
 * depth_limit < 0: std::nth_element itself.  depth_limit >= 0: the same introselect entered with that depth limit
 * (std::__introselect, what nth_element calls with 2 * lg(n)) -- tests use 0..3 to walk the heap-select branch, which
 * real inputs reach only on adversarial orderings. */
extern "C" long orc_retain_best_stl(const float* resp, long cnt, int n_points, int depth_limit, int32_t* order) {
  std::vector<Kp> keypoints((size_t)cnt);
  for (long i = 0; i < cnt; ++i) keypoints[(size_t)i] = Kp{0.f, 0.f, 0.f, -1.f, resp[i], 0, -1, (int32_t)i};
  if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
    if (n_points == 0) {
      keypoints.clear();
    } else {
      if (depth_limit < 0)
        std::nth_element(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(), ResponseGreater());
      else
        std::__introselect(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(), (long)depth_limit,
                           __gnu_cxx::__ops::__iter_comp_iter(ResponseGreater()));
      const float ambiguous_response = keypoints[(size_t)n_points - 1].response;
      std::vector<Kp>::iterator new_end =
          std::partition(keypoints.begin() + n_points, keypoints.end(), ResponseGeThreshold{ambiguous_response});
      keypoints.resize((size_t)(new_end - keypoints.begin()));
    }
  }
  for (size_t i = 0; i < keypoints.size(); ++i) order[i] = keypoints[i].ref;
  return (long)keypoints.size();
}
