// pipeline.hip -- Scanner::processImage (/root/reference/src/scanner.cpp:828-895) for a batch of decoded images of one
// geometry: the caller of every feature stage of the hot path, chained on the device.  One upload per chunk; between the
// stages only the autocrop rectangles (to pick launch geometries) and the keypoints (12 bytes each, to build the
// keypoint-hash work list) visit the host.
//
//   grayscale -> autocrop(20) -> dctHash64(view)             prestage.hip, dcthash.hip          (:859-866)
//   ColorDescriptor::create(cvColor)                          colordesc_create.hip               (:868-872)
//   sizeLongestSide(cvGray, 400)                              prestage.hip (Lanczos-4)           (:876)
//   makeKeyPoints / makeKeyPointDescriptors                   orb.hip                            (:878-884)
//   makeKeyPointHashes on the keypoints compute() left        dcthash.hip (k_kp_hashes)          (:886-889)
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "cbh_index.h"

static int index_images_one(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                            int channels, const cbh_index_params* p, uint64_t* dct_hashes, int32_t* rects,
                            int32_t* resized_dims, uint32_t* kp_counts, cbh_keypoint* kp, uint8_t* desc,
                            uint32_t* kph_counts, uint64_t* kp_hashes, uint8_t* color_descs, uint8_t* color_ok,
                            int device);

// A large batch without the colour leg is cut into a few sub-batches that run on their own host threads and streams: one
// uploads while another computes (host in / host out, 4096 BGR images 640x480, hash + ORB + keypoint hashes: 156 -> 122
// ms).  The colour leg wants ONE big batch (a lane per image), so calls that include it stay whole.
extern "C" int cbh_index_images(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                                int channels, const cbh_index_params* p, uint64_t* dct_hashes, int32_t* rects,
                                int32_t* resized_dims, uint32_t* kp_counts, cbh_keypoint* kp, uint8_t* desc,
                                uint32_t* kph_counts, uint64_t* kp_hashes, uint8_t* color_descs, uint8_t* color_ok,
                                int device) {
  static const int max_threads = [] {
    const char* e = getenv("CBH_PIPELINE_THREADS");
    const int v = e ? atoi(e) : 4;
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  const bool split_ok = p && n >= 1024 && !(p->algos & 8) && (p->algos & 6) && imgs && img_stride > 0;
  const size_t T = split_ok ? std::min<size_t>((size_t)max_threads, n / 512) : 1;
  if (T <= 1)
    return index_images_one(imgs, n, w, h, row_stride, img_stride, channels, p, dct_hashes, rects, resized_dims,
                            kp_counts, kp, desc, kph_counts, kp_hashes, color_descs, color_ok, device);
  std::vector<int> rc(T, CBH_OK);
  std::vector<std::thread> th;
  const size_t cap = p->kp_cap > 0 ? (size_t)p->kp_cap : 0;
  for (size_t t = 0; t < T; ++t) {
    const size_t a = t * n / T, b = (t + 1) * n / T;
    th.emplace_back([&, t, a, b]() {
      rc[t] = index_images_one(imgs + a * img_stride, b - a, w, h, row_stride, img_stride, channels, p,
                               dct_hashes ? dct_hashes + a : nullptr, rects ? rects + 4 * a : nullptr,
                               resized_dims ? resized_dims + 2 * a : nullptr, kp_counts ? kp_counts + a : nullptr,
                               kp ? kp + a * cap : nullptr, desc ? desc + a * cap * 32 : nullptr,
                               kph_counts ? kph_counts + a : nullptr, kp_hashes ? kp_hashes + a * cap : nullptr,
                               color_descs, color_ok, device);
    });
  }
  for (auto& x : th) x.join();
  for (int r : rc)
    if (r != CBH_OK) return r;
  return CBH_OK;
}

static int index_images_one(const uint8_t* imgs, size_t n, int w, int h, size_t row_stride, size_t img_stride,
                            int channels, const cbh_index_params* p, uint64_t* dct_hashes, int32_t* rects,
                            int32_t* resized_dims, uint32_t* kp_counts, cbh_keypoint* kp, uint8_t* desc,
                            uint32_t* kph_counts, uint64_t* kp_hashes, uint8_t* color_descs, uint8_t* color_ok,
                            int device) {
  if (!cbh::device_usable(device)) return CBH_E_NODEVICE;
  if (!p) return CBH_E_INVAL;
  const bool a_dct = p->algos & 1, a_fdct = p->algos & 2, a_orb = p->algos & 4, a_color = p->algos & 8;
  const bool feats = a_fdct || a_orb;
  const int rs = p->resize_longest_side, cap = p->kp_cap;
  if (n == 0) return CBH_OK;
  if (!imgs || w <= 0 || h <= 0 || (channels != 1 && channels != 3 && channels != 4) || row_stride < (size_t)w * channels ||
      (a_dct && !dct_hashes) || (feats && (rs < 1 || rs > 8192 || cap < 1 || p->num_features < 0 || !kp_counts || !kp)) ||
      (a_orb && !desc) || (a_fdct && (!kph_counts || !kp_hashes)) || (a_color && (!color_descs || !color_ok)))
    return CBH_E_INVAL;
  cbh::DeviceGuard g(device);
  if (!g.ok) return CBH_E_NODEVICE;
  const size_t span1 = (size_t)(h - 1) * row_stride + (size_t)w * channels;
  // chunks of up to 8 GB of input (16384 images): the colour leg runs one lane per image, so its rate grows with the
  // chunk (0.43 s for anything up to 4096 images, 0.48 s for 16384)
  size_t per_chunk = std::max<size_t>(1, ((size_t)8192 << 20) / std::max(img_stride, span1));
  per_chunk = std::min<size_t>(std::min(per_chunk, n), 16384);
  const size_t slot = feats ? (size_t)rs * rs + 16 : 0;  // room per image in the packed resize buffer
  uint8_t *d_src = nullptr, *d_gray = nullptr, *d_res = nullptr, *d_desc = nullptr, *d_cdesc = nullptr, *d_cok = nullptr;
  uint64_t *d_out = nullptr, *d_kph = nullptr;
  int* d_rects = nullptr;
  cbh_keypoint* d_kp = nullptr;
  float* d_after = nullptr;
  uint32_t* d_cnt = nullptr;
  hipStream_t s = nullptr, s2 = nullptr;
  hipEvent_t ev_up = nullptr;
  int rc = CBH_OK, color_rc = CBH_OK;
  std::thread color_thread;
  auto cleanup = [&]() {
    if (color_thread.joinable()) color_thread.join();
    if (s2) (void)hipStreamSynchronize(s2), cbh::stream_destroy(s2);
    if (ev_up) (void)hipEventDestroy(ev_up);
    if (s) (void)hipStreamSynchronize(s);
    for (void* q : {(void*)d_src, (void*)d_gray, (void*)d_res, (void*)d_desc, (void*)d_cdesc, (void*)d_cok, (void*)d_out,
                    (void*)d_kph, (void*)d_rects, (void*)d_kp, (void*)d_after, (void*)d_cnt})
      if (q) (void)cbh::free_async(q, s);  // back to the cached pool (keep_pool_memory): the next call reuses it
    if (s) (void)hipStreamSynchronize(s), cbh::stream_destroy(s);
  };
#define CBH_TRY(call)                                             \
  do {                                                            \
    hipError_t e_ = (call);                                       \
    if (e_ != hipSuccess) {                                       \
      cbh::set_last_error(#call, e_);                             \
      cleanup();                                                  \
      return e_ == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP; \
    }                                                             \
  } while (0)
  CBH_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CBH_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CBH_TRY(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
  CBH_TRY(cbh::malloc_async((void**)&d_src, (per_chunk - 1) * img_stride + span1, s));
  if (channels != 1) CBH_TRY(cbh::malloc_async((void**)&d_gray, per_chunk * (size_t)w * h, s));
  CBH_TRY(cbh::malloc_async((void**)&d_out, per_chunk * sizeof(uint64_t), s));
  CBH_TRY(cbh::malloc_async((void**)&d_rects, per_chunk * 4 * sizeof(int), s));
  if (feats) {
    CBH_TRY(cbh::malloc_async((void**)&d_res, 2 * per_chunk * slot, s));  // the chunk-wide pass + the odd images redone
    CBH_TRY(cbh::malloc_async((void**)&d_kp, per_chunk * (size_t)cap * sizeof(cbh_keypoint), s));
    CBH_TRY(cbh::malloc_async((void**)&d_cnt, per_chunk * sizeof(uint32_t), s));
    if (a_orb) {
      CBH_TRY(cbh::malloc_async((void**)&d_after, per_chunk * (size_t)cap * 2 * sizeof(float), s));
      CBH_TRY(cbh::malloc_async((void**)&d_desc, per_chunk * (size_t)cap * 32, s));
    }
    if (a_fdct) CBH_TRY(cbh::malloc_async((void**)&d_kph, per_chunk * (size_t)cap * sizeof(uint64_t), s));
  }
  if (a_color && channels != 1) {
    CBH_TRY(cbh::malloc_async((void**)&d_cdesc, per_chunk * 258, s));
    CBH_TRY(cbh::malloc_async((void**)&d_cok, per_chunk, s));
  }
  const bool trace = getenv("CBH_PIPELINE_TRACE") != nullptr;  // stage times of each chunk on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_last = now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    const auto t = now();
    fprintf(stderr, "[cbh_index_images] %-18s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  std::vector<int> hr(per_chunk * 4);
  std::vector<uint64_t> off(per_chunk), coff(per_chunk);
  std::vector<uint32_t> ws(per_chunk), hs(per_chunk), cw(per_chunk, (uint32_t)w), chh(per_chunk, (uint32_t)h),
      cst(per_chunk, (uint32_t)row_stride), cnt(per_chunk), kp_first(per_chunk + 1), out_first(per_chunk + 1);
  std::vector<cbh_keypoint> hkp;
  std::vector<float> hafter, tri;
  for (size_t i0 = 0; rc == CBH_OK && i0 < n; i0 += per_chunk) {
    const size_t m = std::min(per_chunk, n - i0);
    lap("(setup)");
    CBH_TRY(hipMemcpyAsync(d_src, imgs + i0 * img_stride, (m - 1) * img_stride + span1, hipMemcpyHostToDevice, s));
    lap("upload");
    const uint8_t* gray = d_src;
    size_t gs = row_stride, gi = img_stride;
    if (channels != 1) {
      rc = cbh_bgr2gray_dev(d_src, m, w, h, row_stride, img_stride, channels, d_gray, device, s);
      if (rc) break;
      gray = d_gray, gs = (size_t)w, gi = (size_t)w * h;
    }
    // the colour descriptor is made from the ORIGINAL colour image (scanner.cpp:868-872); grey input: "passed a
    // grayscale image", the descriptor stays empty
    // It runs on its own stream from a helper thread: the clustering kernel is one lane per image (a few dozen waves
    // with long sequential chains), so it occupies a sliver of the chip for a long time -- the other stages, with
    // their host synchronisations, run beside it.
    color_rc = CBH_OK;
    if (a_color) {
      if (channels == 1) {
        memset(color_descs + i0 * 258, 0, m * 258);
        memset(color_ok + i0, 0, m);
      } else {
        CBH_TRY(hipEventRecord(ev_up, s));
        for (size_t i = 0; i < m; ++i) coff[i] = i * img_stride;
        color_thread = std::thread([&, m, i0] {
          cbh::DeviceGuard tg(device);
          hipError_t e = hipStreamWaitEvent(s2, ev_up, 0);
          if (e == hipSuccess)
            color_rc = cbh_color_descriptors_dev(d_src, m, coff.data(), cw.data(), chh.data(), cst.data(), channels, d_cdesc,
                                                 d_cok, device, s2);
          if (e == hipSuccess && color_rc == CBH_OK &&
              ((e = hipMemcpyAsync(color_descs + i0 * 258, d_cdesc, m * 258, hipMemcpyDeviceToHost, s2)) != hipSuccess ||
               (e = hipMemcpyAsync(color_ok + i0, d_cok, m, hipMemcpyDeviceToHost, s2)) != hipSuccess ||
               (e = hipStreamSynchronize(s2)) != hipSuccess))
            ;
          if (e != hipSuccess) {
            cbh::set_last_error("index_images colour leg", e);
            color_rc = CBH_E_HIP;
          }
        });
      }
    }
    bool cropped = false;
    if (p->autocrop_range >= 0) {  // `if (_params.algos && _params.autocrop) autocrop(cvGray, 20)`
      rc = cbh_autocrop_dev(gray, m, w, h, gs, gi, p->autocrop_range, d_rects, device, s);
      if (rc) break;
      CBH_TRY(hipMemcpyAsync(hr.data(), d_rects, m * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
      CBH_TRY(hipStreamSynchronize(s));
      for (size_t i = 0; i < m; ++i)
        cropped |= hr[i * 4] != 0 || hr[i * 4 + 1] != 0 || hr[i * 4 + 2] != w || hr[i * 4 + 3] != h;
    } else {
      for (size_t i = 0; i < m; ++i) hr[i * 4] = hr[i * 4 + 1] = 0, hr[i * 4 + 2] = w, hr[i * 4 + 3] = h;
    }
    if (rects) memcpy(rects + i0 * 4, hr.data(), m * 4 * sizeof(int));
    lap("gray + autocrop");
    // Launch plan for the per-geometry stages (hash, resize).  Runs of equal kept regions take one launch each; when
    // a few odd regions are scattered among images that share one region (photos: almost all keep the whole frame),
    // it is cheaper to run that region over the WHOLE chunk once and redo the odd images one by one.
    size_t n_runs = 0;
    for (size_t i = 0, run = 1; i < m; i += run, ++n_runs) {
      run = 1;
      while (i + run < m && !memcmp(&hr[i * 4], &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
    }
    int mode[4] = {0, 0, w, h};
    size_t n_odd = 0;
    {
      std::map<std::array<int, 4>, size_t> freq;
      for (size_t i = 0; i < m; ++i) ++freq[{hr[i * 4], hr[i * 4 + 1], hr[i * 4 + 2], hr[i * 4 + 3]}];
      size_t best = 0;
      for (const auto& kv : freq)
        if (kv.second > best) best = kv.second, memcpy(mode, kv.first.data(), sizeof mode);
      n_odd = m - best;
    }
    const bool by_mode = 1 + n_odd < n_runs;
    auto is_mode = [&](size_t i) { return !memcmp(&hr[i * 4], mode, sizeof mode); };
    auto hash_run = [&](size_t i, size_t run, const int* r) {
      if (r[0] == 0 && r[1] == 0 && r[2] == w && r[3] == h) return cbh::launch_dcthash(gray + i * gi, run, w, h, gs, gi, d_out + i, s);
      const cbh::HashView view{w, h, r[0], r[1]};  // the blur sees the cropped-away margins (a cv::Mat view)
      return cbh::launch_dcthash(gray + i * gi, run, r[2] - r[0], r[3] - r[1], gs, gi, d_out + i, s, nullptr, &view);
    };
    if (a_dct) {
      if (!cropped) {
        rc = cbh::launch_dcthash(gray, m, w, h, gs, gi, d_out, s);
      } else if (by_mode) {
        rc = hash_run(0, m, mode);
        for (size_t i = 0; i < m && rc == CBH_OK; ++i)
          if (!is_mode(i)) rc = hash_run(i, 1, &hr[i * 4]);
      } else {
        for (size_t i = 0, run = 1; i < m && rc == CBH_OK; i += run) {
          run = 1;
          while (i + run < m && !memcmp(&hr[i * 4], &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
          rc = hash_run(i, run, &hr[i * 4]);
        }
      }
      if (rc) break;
      CBH_TRY(hipMemcpyAsync(dct_hashes + i0, d_out, m * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    }
    lap("dct hash");
    if (!feats) {
      CBH_TRY(hipStreamSynchronize(s));
      if (color_thread.joinable()) color_thread.join();
      rc = color_rc;
      continue;
    }
    // sizeLongestSide(cvGray, 400) of each kept region with the same launch plan; the resized images are packed back
    // to back (ORB and the keypoint hashes take per-image offsets)
    size_t cur = 0;
    auto resize_run = [&](size_t i, size_t run, const int* r) {
      int dw = 0, dh = 0;
      cbh_longest_side_dims(r[2] - r[0], r[3] - r[1], rs, &dw, &dh);
      if (dw <= 0 || dh <= 0 || dw > rs || dh > rs) dw = dh = 0;  // the reference throws: no features for this image
      cur = (cur + 15) & ~(size_t)15;
      for (size_t t = 0; t < run; ++t) {
        ws[i + t] = (uint32_t)dw, hs[i + t] = (uint32_t)dh;
        off[i + t] = cur + t * (size_t)dw * dh;
      }
      if (dw == 0) return (int)CBH_OK;
      const int r2 = cbh_resize_lanczos4_dev(gray + i * gi + (size_t)r[1] * gs + r[0], run, r[2] - r[0], r[3] - r[1], gs, gi,
                                             dw, dh, d_res + cur, device, s);
      cur += run * (size_t)dw * dh;
      return r2;
    };
    if (by_mode) {
      rc = resize_run(0, m, mode);
      for (size_t i = 0; i < m && rc == CBH_OK; ++i)
        if (!is_mode(i)) rc = resize_run(i, 1, &hr[i * 4]);  // its own slot behind the chunk-wide pass
    } else {
      for (size_t i = 0, run = 1; i < m && rc == CBH_OK; i += run) {
        run = 1;
        while (i + run < m && !memcmp(&hr[i * 4], &hr[(i + run) * 4], 4 * sizeof(int))) ++run;
        rc = resize_run(i, run, &hr[i * 4]);
      }
    }
    if (rc) break;
    lap("resize");
    if (resized_dims)
      for (size_t i = 0; i < m; ++i) resized_dims[2 * (i0 + i)] = (int)ws[i], resized_dims[2 * (i0 + i) + 1] = (int)hs[i];
    // images the resize rejected get a 1x1 stand-in geometry: no pyramid level, no keypoints
    std::vector<uint32_t> ow(ws.begin(), ws.begin() + m), oh(hs.begin(), hs.begin() + m);
    for (size_t i = 0; i < m; ++i)
      if (ow[i] == 0) ow[i] = oh[i] = 1, off[i] = 0;
    rc = cbh_orb_dev(d_res, m, off.data(), ow.data(), oh.data(), ow.data(), p->num_features, cap, d_kp, a_orb ? d_after : nullptr,
                     a_orb ? d_desc : nullptr, d_cnt, device, s);
    if (rc) break;
    lap("orb");
    hkp.resize(m * (size_t)cap);
    CBH_TRY(hipMemcpyAsync(cnt.data(), d_cnt, m * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    CBH_TRY(hipMemcpyAsync(hkp.data(), d_kp, m * (size_t)cap * sizeof(cbh_keypoint), hipMemcpyDeviceToHost, s));
    if (a_orb) {
      hafter.resize(m * (size_t)cap * 2);
      CBH_TRY(hipMemcpyAsync(hafter.data(), d_after, m * (size_t)cap * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
      CBH_TRY(hipMemcpyAsync(desc + i0 * (size_t)cap * 32, d_desc, m * (size_t)cap * 32, hipMemcpyDeviceToHost, s));
    }
    CBH_TRY(hipStreamSynchronize(s));
    lap("orb download");
    // the keypoint list as processImage holds it after makeKeyPointDescriptors (which rewrites pt) -- or as detected
    for (size_t i = 0; i < m; ++i) {
      kp_counts[i0 + i] = cnt[i];
      const size_t c = std::min<size_t>(cnt[i], (size_t)cap);
      for (size_t j = 0; j < c; ++j) {
        cbh_keypoint k = hkp[i * cap + j];
        if (a_orb) k.x = hafter[2 * (i * cap + j)], k.y = hafter[2 * (i * cap + j) + 1];
        kp[(i0 + i) * cap + j] = k;
      }
    }
    lap("keypoint lists");
    if (a_fdct) {
      tri.clear();
      kp_first[0] = 0;
      for (size_t i = 0; i < m; ++i) {
        const size_t c = std::min<size_t>(cnt[i], (size_t)cap);
        for (size_t j = 0; j < c; ++j) {
          const cbh_keypoint& k = kp[(i0 + i) * cap + j];
          tri.push_back(k.x), tri.push_back(k.y), tri.push_back(k.size);
        }
        kp_first[i + 1] = (uint32_t)(tri.size() / 3);
      }
      lap("kp triples");
      rc = cbh_keypoint_hashes_dev(d_res, m, off.data(), ow.data(), oh.data(), ow.data(), tri.data(), kp_first.data(), d_kph,
                                   out_first.data(), device, s);
      if (rc) break;
      lap("kp hashes");
      const size_t total = out_first[m];
      std::vector<uint64_t> hh(std::max<size_t>(total, 1));
      if (total) CBH_TRY(hipMemcpyAsync(hh.data(), d_kph, total * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      CBH_TRY(hipStreamSynchronize(s));
      for (size_t i = 0; i < m; ++i) {
        const uint32_t c = out_first[i + 1] - out_first[i];
        kph_counts[i0 + i] = c;
        memcpy(kp_hashes + (i0 + i) * cap, hh.data() + out_first[i], (size_t)c * sizeof(uint64_t));
      }
    }
    if (color_thread.joinable()) color_thread.join();
    if (rc == CBH_OK) rc = color_rc;
  }
#undef CBH_TRY
  cleanup();
  return rc;
}
