// vindexer.hip -- Media::makeVideoIndex (src/media.cpp:925-1037) for a decoder that hands over its frames in chunks.
//
// The reference walks a video one frame at a time: grayscale (a no-op, the decoder outputs grey) -> autocrop(img, 20)
// -> dctHash64 -> the "near frame" filter that stores a frame only when it differs from a hash seen since the last
// stored one.  Only the filter is sequential, and it works on 8 bytes per frame; everything in front of it is the
// batch hash path of prestage.hip / dcthash.hip.  So the indexer here keeps the filter's state (frame number, window,
// stored lists) in a handle and takes the frames in whatever chunks the decoder produces, from host memory or -- a
// hardware decoder writes there -- from device memory:
//
//   cbh_vindexer_create -> [cbh_vindexer_resume] -> cbh_vindexer_push{,_dev} ... -> cbh_vindexer_finish
//
// A video's frames share their geometry and, almost always, their letterbox: the hash launches go out per run of
// frames with the same kept region, which is normally one launch per chunk.
#include <algorithm>
#include <cstring>
#include <vector>

#include "cbh_index.h"

namespace {
constexpr int kMaxFramesPerVideo = 1 << 24;  // MAX_FRAMES_PER_VIDEO, src/dctvideoindex.h:32,50 (VINDEX_FRAME_BITS 24)
}

struct cbh_vindexer {
  int device = 0, threshold = 0, autocrop = 0;
  hipStream_t s = nullptr;
  uint8_t* d_src = nullptr;  // staging for host frames
  size_t src_cap = 0;
  uint64_t* d_out = nullptr;
  int* d_rects = nullptr;
  size_t n_cap = 0;
  int* h_rects = nullptr;        // pinned: the two copies back are enqueued behind the kernels and waited for ONCE
  uint64_t* h_hashes = nullptr;  // (into pageable memory each copy is its own blocking round trip)
  // makeVideoIndex's locals
  std::vector<int32_t> frames;
  std::vector<uint64_t> hashes;
  std::vector<uint64_t> window;
  int frame_number = 0;
  bool first_done = false;  // the frame after (re)start is stored unconditionally and does not enter the window
  bool full = false;        // frameNumber reached MAX_FRAMES_PER_VIDEO: "too many frames, skipping the rest"
  uint64_t last_hash = 0;
  long long near_frames = 0;
  int spec[4] = {0, 0, 0, 0}, spec_w = 0, spec_h = 0;  // the kept region the last chunk ended on
};

namespace {

int ensure(cbh_vindexer* v, size_t n) {
  if (n <= v->n_cap) return CBH_OK;
  if (v->d_out) (void)hipFree(v->d_out);
  if (v->d_rects) (void)hipFree(v->d_rects);
  if (v->h_rects) (void)hipHostFree(v->h_rects);
  if (v->h_hashes) (void)hipHostFree(v->h_hashes);
  v->d_out = nullptr, v->d_rects = nullptr, v->h_rects = nullptr, v->h_hashes = nullptr, v->n_cap = 0;
  CBH_HIP(hipMalloc(&v->d_out, n * sizeof(uint64_t)));
  CBH_HIP(hipMalloc(&v->d_rects, n * 4 * sizeof(int)));
  CBH_HIP(hipHostMalloc((void**)&v->h_rects, n * 4 * sizeof(int), hipHostMallocDefault));
  CBH_HIP(hipHostMalloc((void**)&v->h_hashes, n * sizeof(uint64_t), hipHostMallocDefault));
  v->n_cap = n;
  return CBH_OK;
}

// the per-frame body of the loop at media.cpp:958-1015 for one hash
void feed(cbh_vindexer* v, uint64_t hash) {
  if (v->full) return;
  if (!v->first_done) {  // :958-968
    v->hashes.push_back(hash);
    v->frames.push_back(v->frame_number);
    v->frame_number++;
    v->first_done = true;
    v->last_hash = hash;
    return;
  }
  if (v->threshold > 0) {  // :994-1009
    // "close != window.size()" = some hash of the window is not near this one.  Only that is asked of the window, so
    // it is kept as the SET of its hashes (a value already in it is not added again): a static scene repeats a handful
    // of hashes, and the reference's list -- every frame since the last stored one -- makes this loop quadratic there
    bool far = false, have = false;
    for (uint64_t prev : v->window) {
      const int d = __builtin_popcountll(prev ^ hash);
      if (d >= v->threshold) {
        far = true;
        break;
      }
      have |= d == 0;
    }
    if (far) {
      v->window.clear();
      v->hashes.push_back(hash);
      v->frames.push_back(v->frame_number);
    } else {
      v->near_frames++;
    }
    if (far || !have) v->window.push_back(hash);
  } else {
    v->hashes.push_back(hash);
    v->frames.push_back(v->frame_number);
  }
  v->last_hash = hash;
  v->frame_number++;
  if (v->frame_number == kMaxFramesPerVideo) v->full = true;  // :1013-1016
}

// grey frames in device memory -> hashes on the host (autocrop + dctHash64 of the kept VIEW, cvutil.cpp:1397-1401).
// The kept region of a video hardly ever changes, so the hash launch does not wait for this chunk's rectangles: it
// goes out behind the autocrop kernels with the region the previous chunk ended on (the whole frame at the start),
// rectangles and hashes come back with ONE synchronisation, and only frames whose rectangle turned out different are
// hashed again (the first chunk of a letterboxed video; a change of letterbox inside one).
int hash_chunk(cbh_vindexer* v, const uint8_t* d_gray, size_t m, int w, int h, size_t gs, size_t gi) {
  int rc = ensure(v, m);
  if (rc) return rc;
  hipStream_t s = v->s;
  int* hr = v->h_rects;
  const bool crop = v->autocrop >= 0;
  int spec[4] = {0, 0, w, h};
  if (crop && !(v->spec_w == w && v->spec_h == h) && m > 8) {
    // nothing to go by yet (the first chunk of a video): one frame from the middle of the chunk (the first ones are
    // often a fade-in) tells the region to speculate on -- 50 us once instead of hashing a letterboxed chunk twice
    rc = cbh_autocrop_dev(d_gray + (m / 2) * gi, 1, w, h, gs, gi, v->autocrop, v->d_rects, v->device, s);
    if (rc) return rc;
    CBH_HIP(hipMemcpyAsync(v->spec, v->d_rects, sizeof v->spec, hipMemcpyDeviceToHost, s));
    CBH_HIP(hipStreamSynchronize(s));
    v->spec_w = w, v->spec_h = h;
  }
  if (crop && v->spec_w == w && v->spec_h == h) memcpy(spec, v->spec, sizeof spec);
  auto launch = [&](size_t i, size_t run, const int* r, uint64_t* d_out) {
    if (r[0] == 0 && r[1] == 0 && r[2] == w && r[3] == h) return cbh::launch_dcthash(d_gray + i * gi, run, w, h, gs, gi, d_out, s);
    const cbh::HashView view{w, h, r[0], r[1]};
    return cbh::launch_dcthash(d_gray + i * gi, run, r[2] - r[0], r[3] - r[1], gs, gi, d_out, s, nullptr, &view);
  };
  if (crop) {
    rc = cbh_autocrop_dev(d_gray, m, w, h, gs, gi, v->autocrop, v->d_rects, v->device, s);
    if (rc) return rc;
  }
  rc = launch(0, m, spec, v->d_out);
  if (rc) return rc;
  if (crop) CBH_HIP(hipMemcpyAsync(hr, v->d_rects, m * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  CBH_HIP(hipMemcpyAsync(v->h_hashes, v->d_out, m * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  CBH_HIP(hipStreamSynchronize(s));
  if (!crop) return CBH_OK;
  bool again = false;
  for (size_t i = 0, run = 1; i < m; i += run) {
    const int* r = &hr[i * 4];
    run = 1;
    while (i + run < m && !memcmp(r, &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
    if (!memcmp(r, spec, sizeof spec)) continue;
    rc = launch(i, run, r, v->d_out + i);
    if (rc) return rc;
    CBH_HIP(hipMemcpyAsync(v->h_hashes + i, v->d_out + i, run * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    again = true;
  }
  if (again) CBH_HIP(hipStreamSynchronize(s));
  memcpy(v->spec, &hr[(m - 1) * 4], sizeof v->spec);
  v->spec_w = w, v->spec_h = h;
  return CBH_OK;
}

bool frames_ok(const void* p, size_t n, int w, int h, size_t row_stride, size_t img_stride) {
  return p && w > 0 && h > 0 && row_stride >= (size_t)w &&
         (n <= 1 || img_stride >= (size_t)(h - 1) * row_stride + (size_t)w);
}

}  // namespace

extern "C" {

cbh_vindexer* cbh_vindexer_create(int device, int threshold, int autocrop_range) {
  cbh::clear_last_error();
  if (!cbh::device_usable(device)) return (cbh_vindexer*)cbh::fail_handle(CBH_E_NODEVICE, "cbh_vindexer_create: no usable gfx950 device at that ordinal");
  cbh::DeviceGuard g(device);
  if (!g.ok) return (cbh_vindexer*)cbh::fail_handle(CBH_E_NODEVICE, nullptr);
  cbh_vindexer* v = new cbh_vindexer;
  v->device = device;
  v->threshold = threshold;
  v->autocrop = autocrop_range;
  hipError_t e = hipStreamCreateWithFlags(&v->s, hipStreamNonBlocking);
  if (e != hipSuccess) {
    cbh::set_last_error("hipStreamCreateWithFlags", e);
    delete v;
    return nullptr;
  }
  return v;
}

void cbh_vindexer_destroy(cbh_vindexer* v) {
  if (!v) return;
  cbh::DeviceGuard g(v->device);
  if (v->s) cbh::stream_destroy(v->s);
  for (void* p : {(void*)v->d_src, (void*)v->d_out, (void*)v->d_rects})
    if (p) (void)hipFree(p);
  if (v->h_rects) (void)hipHostFree(v->h_rects);
  if (v->h_hashes) (void)hipHostFree(v->h_hashes);
  delete v;
}

int cbh_vindexer_resume(cbh_vindexer* v, const int32_t* frames, const uint64_t* hashes, size_t n) {
  if (!v || (n && (!frames || !hashes))) return CBH_E_INVAL;
  if (v->first_done || !v->frames.empty()) return CBH_E_INVAL;  // only in front of the first push
  if (n == 0) return CBH_OK;
  for (size_t i = 0; i < n; ++i)
    if (frames[i] < 0 || (i && frames[i] <= frames[i - 1])) return CBH_E_INVAL;
  v->frames.assign(frames, frames + n);
  v->hashes.assign(hashes, hashes + n);
  v->frame_number = frames[n - 1] + 1;  // :930-933
  return CBH_OK;
}

int cbh_vindexer_push_dev(cbh_vindexer* v, const void* d_frames, size_t n, int w, int h, size_t row_stride,
                          size_t img_stride) {
  if (!v) return CBH_E_INVAL;
  if (n == 0) return CBH_OK;
  if (!frames_ok(d_frames, n, w, h, row_stride, img_stride)) return CBH_E_INVAL;
  if (v->full) return CBH_OK;
  cbh::DeviceGuard g(v->device);
  if (!g.ok) return CBH_E_NODEVICE;
  const size_t per = 16384;  // frames per hash launch group
  const size_t gi = n > 1 ? img_stride : (size_t)h * row_stride;
  for (size_t i0 = 0; i0 < n && !v->full; i0 += per) {
    const size_t m = std::min(per, n - i0);
    int rc = hash_chunk(v, (const uint8_t*)d_frames + i0 * gi, m, w, h, row_stride, gi);
    if (rc) return rc;
    for (size_t i = 0; i < m; ++i) feed(v, v->h_hashes[i]);
  }
  return CBH_OK;
}

int cbh_vindexer_push(cbh_vindexer* v, const uint8_t* frames, size_t n, int w, int h, size_t row_stride,
                      size_t img_stride) {
  if (!v) return CBH_E_INVAL;
  if (n == 0) return CBH_OK;
  if (!frames_ok(frames, n, w, h, row_stride, img_stride)) return CBH_E_INVAL;
  if (v->full) return CBH_OK;
  cbh::DeviceGuard g(v->device);
  if (!g.ok) return CBH_E_NODEVICE;
  const size_t span1 = (size_t)(h - 1) * row_stride + (size_t)w;
  const size_t stride = n > 1 ? img_stride : (size_t)h * row_stride;
  const size_t per = std::min(n, std::max<size_t>(1, ((size_t)256 << 20) / std::max(stride, span1)));
  const size_t need = (per - 1) * stride + span1;
  if (need > v->src_cap) {
    if (v->d_src) (void)hipFree(v->d_src);
    v->d_src = nullptr, v->src_cap = 0;
    CBH_HIP(hipMalloc(&v->d_src, need));
    v->src_cap = need;
  }
  for (size_t i0 = 0; i0 < n && !v->full; i0 += per) {
    const size_t m = std::min(per, n - i0);
    CBH_HIP(hipMemcpyAsync(v->d_src, frames + i0 * stride, (m - 1) * stride + span1, hipMemcpyHostToDevice, v->s));
    int rc = hash_chunk(v, v->d_src, m, w, h, row_stride, stride);
    if (rc) return rc;
    for (size_t i = 0; i < m; ++i) feed(v, v->h_hashes[i]);
  }
  return CBH_OK;
}

long long cbh_vindexer_frames_seen(const cbh_vindexer* v) {
  if (!v) return CBH_E_INVAL;
  return v->frame_number;
}

long long cbh_vindexer_finish(const cbh_vindexer* v, int32_t* frames, uint64_t* hashes, size_t cap) {
  if (!v) return CBH_E_INVAL;
  // :1017-1024 -- "always include the last frame so it can be used as a reference"
  const int last = v->frame_number - 1;
  const bool extra = !v->frames.empty() && v->frames.back() != last;
  const size_t n = v->frames.size() + (extra ? 1 : 0);
  if (n <= cap && frames && hashes) {
    if (!v->frames.empty()) {
      memcpy(frames, v->frames.data(), v->frames.size() * sizeof(int32_t));
      memcpy(hashes, v->hashes.data(), v->hashes.size() * sizeof(uint64_t));
    }
    if (extra) {
      frames[n - 1] = last;
      hashes[n - 1] = v->last_hash;
    }
  }
  return (long long)n;
}

}  // extern "C"
