// video.hip -- DctVideoIndex (src/dctvideoindex.{h,cpp}) on top of the 64-bit scan, the .vdx v2 codec
// (src/videoindex.cpp:271-429) and the frame de-dup of Media::makeVideoIndex (src/media.cpp:958-1024).
//
// The reference keeps per video a list of (frame number, dct hash) in <id>.vdx and builds, lazily, one
// search structure over all of them: RadixMap<VideoTreeIndex{idx:24, frame:24}> (dctvideoindex.h:37-43,
// buildTree :113-170).  Here the structure is a cbh_idx64 whose payload is the entry position; the
// (video index, frame) pair of an entry stays on the host.  Searches are exact (the reference's
// vradix=0 behaviour, used by its own test unit/testdctvideoindex.cpp:24); vradix>0 only ever returns
// a subset of this.  findVideo's reductions (closest frame per video, adjacency scoring) run on the device
// (reduce.hip); the host versions below remain as a second implementation (knob "video_host_reduce").
#include <map>
#include <unordered_map>

#include "cbh_index.h"

struct cbh_vidx {
  int device = 0;
  uint32_t device_mask = 0;  // != 0: the search structure is a sharded cbh_idx64 (cbh_vidx_create_sharded)
  int shards_per_device = 1;
  struct Video {
    uint32_t media_id;
    std::vector<int32_t> frames;
    std::vector<uint64_t> hashes;
  };
  std::vector<Video> videos;  // _mediaId order (load: ascending id, add: appended)
  // built state (buildTree)
  bool built = false;
  int built_skip = 0;
  cbh_idx64* idx = nullptr;
  std::vector<uint32_t> evidx;   // entry -> video index
  std::vector<int32_t> eframe;   // entry -> frame number
  // device copies for the on-device reduction (reduce.hip): entry -> video index / frame, video index -> mediaId
  uint32_t* d_evidx = nullptr;
  int32_t* d_eframe = nullptr;
  uint32_t* d_vmedia = nullptr;
  std::mutex build_mu;           // QMutex _mutex (dctvideoindex.cpp:118)
  // 0 = exact search.  > 0 = RadixMap-compatible: a needle hash only sees the entries of its bucket
  // (hash >> 1) & (2^radix - 1) (src/tree/radix.h:135-141), like `-p.vradix N` in the reference.
  unsigned radix = 0;
};

namespace {

// insertHashes (dctvideoindex.cpp:61-111) for every video, then upload
int build(cbh_vidx* v, int skip) {
  std::lock_guard<std::mutex> lk(v->build_mu);
  if (v->built) return CBH_OK;  // `if (_tree) return;` -- a later skipFrames does not rebuild (:116)
  std::vector<uint64_t> hashes;
  std::vector<uint32_t> ids;
  v->evidx.clear();
  v->eframe.clear();
  for (size_t vi = 0; vi < v->videos.size() && vi < (1u << 24); ++vi) {
    const auto& vid = v->videos[vi];
    if (vid.frames.empty()) continue;
    const int lastFrame = vid.frames.back();
    for (size_t j = 0; j < vid.hashes.size(); ++j) {
      const uint64_t h = vid.hashes[j];
      const int pc = __builtin_popcountll(h);
      if (pc < 5 || 64 - pc < 5) continue;  // insufficient detail (:89)
      const int frame = vid.frames[j];
      if (skip && lastFrame / 2 > skip) {
        if (frame < skip || frame > lastFrame - skip) continue;  // (:93-95)
      }
      hashes.push_back(h);
      v->evidx.push_back((uint32_t)vi);
      v->eframe.push_back(frame);
      ids.push_back((uint32_t)hashes.size());  // payload = entry position + 1 (never 0)
    }
  }
  if (v->idx) cbh_idx64_destroy(v->idx);
  v->idx = v->device_mask ? cbh_idx64_create_sharded(v->device_mask, v->shards_per_device) : cbh_idx64_create(v->device);
  if (!v->idx) return CBH_E_NODEVICE;
  int rc = cbh_idx64_load(v->idx, hashes.data(), ids.data(), hashes.size());
  if (rc) return rc;
  {
    DeviceGuard g(v->device);
    if (!g.ok) return CBH_E_NODEVICE;
    for (void* p : {(void*)v->d_evidx, (void*)v->d_eframe, (void*)v->d_vmedia})
      if (p) (void)hipFree(p);
    v->d_evidx = nullptr, v->d_eframe = nullptr, v->d_vmedia = nullptr;
    const size_t ne = std::max<size_t>(1, v->evidx.size()), nv = std::max<size_t>(1, v->videos.size());
    std::vector<uint32_t> vmedia(v->videos.size());
    for (size_t i = 0; i < v->videos.size(); ++i) vmedia[i] = v->videos[i].media_id;
    CBH_HIP(hipMalloc(&v->d_evidx, ne * 4));
    CBH_HIP(hipMalloc(&v->d_eframe, ne * 4));
    CBH_HIP(hipMalloc(&v->d_vmedia, nv * 4));
    if (!v->evidx.empty()) {
      CBH_HIP(hipMemcpy(v->d_evidx, v->evidx.data(), v->evidx.size() * 4, hipMemcpyHostToDevice));
      CBH_HIP(hipMemcpy(v->d_eframe, v->eframe.data(), v->eframe.size() * 4, hipMemcpyHostToDevice));
    }
    if (!vmedia.empty()) CBH_HIP(hipMemcpy(v->d_vmedia, vmedia.data(), vmedia.size() * 4, hipMemcpyHostToDevice));
  }
  v->built = true;
  v->built_skip = skip;
  return CBH_OK;
}

void invalidate(cbh_vidx* v) {
  std::lock_guard<std::mutex> lk(v->build_mu);
  v->built = false;
}

// sorted records of `nq` needle hashes against the built index, copied to the host
int scan_to_host(cbh_vidx* v, const uint64_t* q, size_t nq, int thresh, std::vector<cbh_record>* recs) {
  recs->clear();
  cbh_idx64* idx = v->idx;
  if (nq == 0 || idx->n == 0 || thresh <= 0) return CBH_OK;
  if (nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  Workspace* ws = L.ws;
  if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, nq))) return rc;
  CBH_HIP(hipMemcpyAsync(ws->d_q, q, nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  std::vector<uint64_t> masks;
  if (v->radix) {  // equal bucket <=> equal bits 1..radix
    masks.assign(nq, ((1ull << v->radix) - 1) << 1);
    if ((rc = Workspace::grow(&ws->d_qmask, &ws->qmask_cap, nq))) return rc;
    CBH_HIP(hipMemcpyAsync(ws->d_qmask, masks.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice, ws->stream));
  }
  unsigned long long total = 0;
  rc = scan_all(idx, ws, ws->d_q, nq, thresh, ws->stream, &total, 0, v->radix ? ws->d_qmask : nullptr);
  if (rc) return rc;
  if ((rc = ws->ensure_sort())) return rc;
  rc = launch_sort_records(ws->d_rec, ws->d_alt, (size_t)total, nq, ws->d_tmp, ws->tmp_bytes, ws->stream);
  if (rc) return rc;
  recs->resize((size_t)total);
  if (total)
    CBH_HIP(hipMemcpyAsync(recs->data(), ws->d_rec, total * sizeof(cbh_record), hipMemcpyDeviceToHost,
                           ws->stream));
  CBH_HIP(hipStreamSynchronize(ws->stream));
  return CBH_OK;
}

struct Range {
  int src, dst;
};

// scoring of one needle from its (needle frame -> closest entry per media) candidates (:595-654)
void score_needle(std::map<uint32_t, std::vector<Range>>& cand, int min_matched, int min_near,
                  std::vector<cbh_vmatch>* out) {
  const int frameMargin = 15;
  for (auto& kv : cand) {
    auto& ranges = kv.second;
    std::sort(ranges.begin(), ranges.end(), [](const Range& a, const Range& b) { return a.src < b.src; });
    int numAdjacent = 0, lastFrame = 0;
    for (const Range& r : ranges) {
      if (abs(r.dst - lastFrame) < frameMargin) numAdjacent++;
      lastFrame = r.dst;
    }
    const int num = (int)ranges.size();
    const int percentNear = numAdjacent * 100 / num;
    if (num < min_matched) continue;
    if (percentNear < min_near) continue;
    cbh_vmatch m;
    m.id = kv.first;
    m.score = 100 - percentNear;
    m.src_in = ranges.front().src;
    m.dst_in = ranges.front().dst;
    m.len = std::max(ranges.back().src - m.src_in, ranges.back().dst - m.dst_in);
    out->push_back(m);
  }
}

struct VNeedle {
  size_t begin, end;  // range in the concatenated needle arrays
  uint32_t id;
};

// scan -> group per needle frame (topk.hip) -> K8 on the device (reduce.hip): closest frame per video, adjacency
// scoring, gates.  Only the final matches come back.
int reduce_on_device(cbh_vidx* v, const std::vector<uint64_t>& q, const std::vector<int32_t>& qframe,
                     const std::vector<uint32_t>& qneedle, const std::vector<VNeedle>& needles, int thresh,
                     int min_matched, int min_near, int filter_self, std::vector<std::vector<cbh_vmatch>>* results) {
  cbh_idx64* idx = v->idx;
  const size_t nq = q.size();
  if (nq == 0 || idx->n == 0 || thresh <= 0) return CBH_OK;
  if (nq > CBH_MAX_QUERIES_PER_CALL) return CBH_E_INVAL;
  DeviceGuard g(idx->device);
  if (!g.ok) return CBH_E_NODEVICE;
  int rc;
  WsLease L(idx, &rc);
  if (!L.ws) return rc;
  Workspace* ws = L.ws;
  hipStream_t s = ws->stream;
  if ((rc = Workspace::grow(&ws->d_q, &ws->q_cap, nq))) return rc;
  CBH_HIP(hipMemcpyAsync(ws->d_q, q.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  std::vector<uint64_t> masks;
  if (v->radix) {  // equal bucket <=> equal bits 1..radix
    masks.assign(nq, ((1ull << v->radix) - 1) << 1);
    if ((rc = Workspace::grow(&ws->d_qmask, &ws->qmask_cap, nq))) return rc;
    CBH_HIP(hipMemcpyAsync(ws->d_qmask, masks.data(), nq * sizeof(uint64_t), hipMemcpyHostToDevice, s));
  }
  unsigned long long total = 0;
  rc = scan_all(idx, ws, ws->d_q, nq, thresh, s, &total, 0, v->radix ? ws->d_qmask : nullptr);
  if (rc) return rc;
  if (total == 0) return CBH_OK;
  if (total >= (1ull << 32)) return CBH_E_OVERFLOW;
  std::vector<uint32_t> nid(needles.size());
  for (size_t i = 0; i < needles.size(); ++i) nid[i] = needles[i].id;
  void* scratch = nullptr;
  uint32_t *d_qneedle = nullptr, *d_nid = nullptr;
  int32_t* d_qframe = nullptr;
  const size_t sbytes = topk_scratch_bytes(nq, (size_t)total);
  hipError_t e = cbh::malloc_async(&scratch, sbytes + 16, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_qneedle, nq * 4, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_qframe, nq * 4, s);
  if (e == hipSuccess) e = cbh::malloc_async((void**)&d_nid, nid.size() * 4, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_qneedle, qneedle.data(), nq * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_qframe, qframe.data(), nq * 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_nid, nid.data(), nid.size() * 4, hipMemcpyHostToDevice, s);
  std::vector<cbh_nvmatch> flat;
  if (e == hipSuccess) {
    const unsigned* d_off = nullptr;
    const unsigned long long* d_seg = nullptr;
    unsigned* d_status = (unsigned*)((char*)scratch + sbytes);
    rc = topk_scratch_init(scratch, nq, s);
    if (!rc) rc = launch_records_group(ws->d_total, 1, 0, (size_t)total, nq, d_status, scratch, &d_off, &d_seg, s);
    if (!rc)
      rc = launch_video_reduce(d_off, d_seg, (size_t)total, nq, v->d_evidx, v->d_eframe, v->d_vmedia, d_qneedle, d_qframe,
                               d_nid, filter_self, min_matched, min_near, &flat, s);
  }
  for (void* p : {scratch, (void*)d_qneedle, (void*)d_qframe, (void*)d_nid})
    if (p) (void)cbh::free_async(p, s);
  CBH_HIP(e);
  if (rc) return rc;
  std::sort(flat.begin(), flat.end(), [](const cbh_nvmatch& a, const cbh_nvmatch& b) {
    return a.needle != b.needle ? a.needle < b.needle : a.m.id < b.m.id;  // std::map order: ascending mediaId
  });
  for (const cbh_nvmatch& m : flat) (*results)[m.needle].push_back(m.m);
  return CBH_OK;
}

int find_videos(cbh_vidx* v, const int32_t* frames, const uint64_t* hashes, const std::vector<VNeedle>& needles,
                int thresh, int skip, int min_matched, int min_near, int filter_self,
                std::vector<std::vector<cbh_vmatch>>* results) {
  results->assign(needles.size(), {});
  int rc = build(v, skip);
  if (rc) return rc;
  // needle frames that survive the (unconditional) trim (:431), concatenated as scan queries
  std::vector<uint64_t> q;
  std::vector<int32_t> qframe;
  std::vector<uint32_t> qneedle;
  for (size_t k = 0; k < needles.size(); ++k) {
    const VNeedle& nd = needles[k];
    if (nd.end <= nd.begin) continue;
    const int lastFrame = frames[nd.end - 1];
    for (size_t i = nd.begin; i < nd.end; ++i) {
      if (frames[i] < skip || frames[i] > lastFrame - skip) continue;
      q.push_back(hashes[i]);
      qframe.push_back(frames[i]);
      qneedle.push_back((uint32_t)k);
    }
  }
  if (g_video_host_reduce == 2 || (g_video_host_reduce == 0 && needles.size() > 1))
    return reduce_on_device(v, q, qframe, qneedle, needles, thresh, min_matched, min_near,
                                                    filter_self, results);
  std::vector<cbh_record> recs;
  rc = scan_to_host(v, q.data(), q.size(), thresh, &recs);
  if (rc) return rc;
  // records are ordered (needle frame, distance, entry position): the first record of a media inside
  // one needle frame is its closest entry, earliest in scan order among equals (:499-502)
  std::vector<std::map<uint32_t, std::vector<Range>>> cand(needles.size());
  std::unordered_map<uint32_t, char> seen;
  size_t i = 0;
  while (i < recs.size()) {
    const uint32_t qi = CBH_REC_QUERY(recs[i]);
    const uint32_t k = qneedle[qi];
    seen.clear();
    for (; i < recs.size() && CBH_REC_QUERY(recs[i]) == qi; ++i) {
      const uint32_t pos = CBH_REC_ID(recs[i]) - 1;
      const uint32_t id = v->videos[v->evidx[pos]].media_id;
      if (filter_self && id == needles[k].id) continue;
      if (seen.emplace(id, 1).second) cand[k][id].push_back(Range{qframe[qi], v->eframe[pos]});
    }
  }
  for (size_t k = 0; k < needles.size(); ++k) score_needle(cand[k], min_matched, min_near, &(*results)[k]);
  return CBH_OK;
}

}  // namespace

extern "C" {

cbh_vidx* cbh_vidx_create(int device) {
  clear_last_error();
  if (!device_usable(device)) return (cbh_vidx*)fail_handle(CBH_E_NODEVICE, "cbh_vidx_create: no usable gfx950 device at that ordinal");
  cbh_vidx* v = new (std::nothrow) cbh_vidx;
  if (!v) return (cbh_vidx*)fail_handle(CBH_E_NOMEM, "cbh_vidx_create: host allocation failed");
  v->device = device;
  return v;
}

cbh_vidx* cbh_vidx_create_sharded(uint32_t device_mask, int shards_per_device) {
  cbh_idx64* probe = cbh_idx64_create_sharded(device_mask, shards_per_device);  // validates the mask
  if (!probe) return nullptr;  // (code set by the probe)
  cbh_vidx* v = new (std::nothrow) cbh_vidx;
  if (v) {
    v->device = probe->device;
    v->device_mask = device_mask;
    v->shards_per_device = shards_per_device;
  }
  cbh_idx64_destroy(probe);
  if (!v) return (cbh_vidx*)fail_handle(CBH_E_NOMEM, "cbh_vidx_create_sharded: host allocation failed");
  return v;
}

/* RadixMap(params.videoRadix) (dctvideoindex.cpp:128): radix is limited like the reference's constructor does
 * (radix.h:105-112: 30 - ceil(log2(sizeof(Bucket) + sizeof(Bucket*))) = 24 with its two-vector Bucket) */
int cbh_vidx_set_radix(cbh_vidx* v, int radix) {
  if (!v || radix < 0) return CBH_E_INVAL;
  v->radix = (unsigned)std::min(radix, 24);
  return CBH_OK;
}

void cbh_vidx_destroy(cbh_vidx* v) {
  if (!v) return;
  cbh::combiner_drop(v);  // combine.hip: the queue of cbh_*_find_coalesced callers
  if (v->idx) cbh_idx64_destroy(v->idx);
  for (void* p : {(void*)v->d_evidx, (void*)v->d_eframe, (void*)v->d_vmedia})
    if (p) (void)hipFree(p);
  delete v;
}

int cbh_vidx_add_video(cbh_vidx* v, uint32_t media_id, const int32_t* frames, const uint64_t* hashes,
                       size_t n) {
  if (!v || (n && (!frames || !hashes))) return CBH_E_INVAL;
  cbh_vidx::Video vid;
  vid.media_id = media_id;
  vid.frames.assign(frames, frames + n);
  vid.hashes.assign(hashes, hashes + n);
  v->videos.push_back(std::move(vid));
  invalidate(v);  // add(): `delete _tree` (:250-254)
  return CBH_OK;
}

int cbh_vidx_remove(cbh_vidx* v, const uint32_t* media_ids, size_t n) {
  if (!v || (n && !media_ids)) return CBH_E_INVAL;
  std::vector<uint32_t> rm(media_ids, media_ids + n);
  std::sort(rm.begin(), rm.end());
  std::vector<cbh_vidx::Video> keep;
  for (auto& vid : v->videos)
    if (!std::binary_search(rm.begin(), rm.end(), vid.media_id)) keep.push_back(std::move(vid));
  v->videos.swap(keep);
  invalidate(v);  // remove(): ids dropped, tree deleted (:256-275)
  return CBH_OK;
}

size_t cbh_vidx_count(const cbh_vidx* v) { return v ? v->videos.size() : 0; }  // _mediaId.size() (:55-59)

// memoryUsage() (:57-59): nothing before buildTree; then 8 + 6 bytes per entry (hash_t + packed VideoTreeIndex)
size_t cbh_vidx_memory_usage(const cbh_vidx* v) {
  if (!v) return 0;
  std::lock_guard<std::mutex> lk(const_cast<cbh_vidx*>(v)->build_mu);
  return v->built ? v->evidx.size() * 14u : 0;
}

size_t cbh_vidx_entries(cbh_vidx* v, int skip_frames) {
  if (!v || build(v, skip_frames)) return 0;
  return v->evidx.size();
}

int cbh_vidx_find_frame(cbh_vidx* v, uint64_t hash, int thresh, int skip_frames, int src_in,
                        cbh_vmatch* out, size_t cap, size_t* n_out) {
  if (!v || !n_out || (cap && !out)) return CBH_E_INVAL;
  *n_out = 0;
  if (hash == 0) return CBH_OK;  // "needle has no dct hash" (:331-335)
  int rc = build(v, skip_frames);
  if (rc) return rc;
  std::vector<cbh_record> recs;
  rc = scan_to_host(v, &hash, 1, thresh, &recs);
  if (rc) return rc;
  // nearest frame per video, ascending video index (QMap<mediaid_t mediaIndex, Match>, :346-356)
  std::map<uint32_t, std::pair<int, int>> nearest;
  for (cbh_record r : recs) {
    const uint32_t pos = CBH_REC_ID(r) - 1;
    const uint32_t vi = v->evidx[pos];
    if (!nearest.count(vi)) nearest[vi] = {CBH_REC_DIST(r), v->eframe[pos]};  // records ascend (dist, pos)
  }
  if (src_in < 0) src_in = 0;
  size_t m = 0;
  for (auto& kv : nearest) {
    if (m < cap) {
      out[m].id = v->videos[kv.first].media_id;
      out[m].score = kv.second.first;
      out[m].src_in = src_in;
      out[m].dst_in = kv.second.second;
      out[m].len = 1;
    }
    ++m;
  }
  *n_out = m;
  return CBH_OK;
}

int cbh_vidx_find_video(cbh_vidx* v, const int32_t* frames, const uint64_t* hashes, size_t n,
                        uint32_t needle_id, int thresh, int skip_frames, int min_frames_matched,
                        int min_frames_near, int filter_self, cbh_vmatch* out, size_t cap, size_t* n_out) {
  if (!v || !n_out || (cap && !out) || (n && (!frames || !hashes))) return CBH_E_INVAL;
  *n_out = 0;
  std::vector<VNeedle> nd{{0, n, needle_id}};
  std::vector<std::vector<cbh_vmatch>> res;
  int rc = find_videos(v, frames, hashes, nd, thresh, skip_frames, min_frames_matched, min_frames_near,
                       filter_self, &res);
  if (rc) return rc;
  *n_out = res[0].size();
  for (size_t i = 0; i < res[0].size() && i < cap; ++i) out[i] = res[0][i];
  return CBH_OK;
}

int cbh_vidx_find_videos_batch(cbh_vidx* v, const int32_t* frames, const uint64_t* hashes,
                               const uint64_t* offsets, const uint32_t* needle_ids, size_t n_needles,
                               int thresh, int skip_frames, int min_frames_matched, int min_frames_near,
                               int filter_self, cbh_vmatch* out, size_t cap, uint64_t* out_offsets) {
  if (!v || !offsets || !needle_ids || !out_offsets || (cap && !out)) return CBH_E_INVAL;
  std::vector<VNeedle> nd(n_needles);
  for (size_t i = 0; i < n_needles; ++i) {
    if (offsets[i + 1] < offsets[i]) return CBH_E_INVAL;
    nd[i] = VNeedle{(size_t)offsets[i], (size_t)offsets[i + 1], needle_ids[i]};
  }
  std::vector<std::vector<cbh_vmatch>> res;
  int rc = find_videos(v, frames, hashes, nd, thresh, skip_frames, min_frames_matched, min_frames_near,
                       filter_self, &res);
  if (rc) return rc;
  uint64_t pos = 0;
  for (size_t i = 0; i < n_needles; ++i) {
    out_offsets[i] = pos;
    for (auto& m : res[i]) {
      if (pos < cap) out[pos] = m;
      ++pos;
    }
  }
  out_offsets[n_needles] = pos;
  return pos > cap ? CBH_E_OVERFLOW : CBH_OK;
}

/* ---- .vdx v2 codec (host) ------------------------------------------------------------------- */

// Layout (src/videoindex.cpp:271-337 fixes the bytes, nothing else): text header line, u32 length of the frame-number
// block, that block -- one byte for frame 0, then every gap to the next stored frame as a base-128 number, least
// significant group first, bit 7 set on every group but the last -- zero padding up to an 8-byte file offset, the u64
// hashes, the four characters "cbir".  Two passes: size everything, then write straight into the caller's buffer.
size_t cbh_vdx_encode(const int32_t* frames, const uint64_t* hashes, size_t n, const char* version,
                      uint8_t* out, size_t cap) {
  char header[256];
  const size_t hl = (size_t)snprintf(header, sizeof header, "cbird video index:%s:%d:%d:%d:%d:%zu:\n",
                                     version ? version : "0.8.1", 2, 1 /* QSysInfo::LittleEndian */, 1, 8, n);
  if (n == 0) {  // an empty index is its header
    if (out && hl <= cap) memcpy(out, header, hl);
    return hl;
  }
  if (!frames || !hashes || frames[0] != 0) return 0;  // the format has no place for a first frame other than 0
  size_t groups = 1;                                     // (frame 0's byte)
  for (size_t i = 1; i < n; ++i) {
    const long long gap = (long long)frames[i] - frames[i - 1];
    if (gap < 1) return 0;  // frame numbers must rise
    for (unsigned long long g = (unsigned long long)gap; g; g >>= 7) ++groups;
  }
  if (groups > 0xffffffffull) return 0;
  const size_t pad = (8 - (hl + 4 + groups) % 8) % 8;
  const size_t total = hl + 4 + groups + pad + n * 8 + 4;
  if (!out || total > cap) return total;
  uint8_t* w = out;
  memcpy(w, header, hl), w += hl;
  const uint32_t len = (uint32_t)groups;
  memcpy(w, &len, 4), w += 4;
  *w++ = 0;
  for (size_t i = 1; i < n; ++i)
    for (unsigned long long g = (unsigned long long)((long long)frames[i] - frames[i - 1]); g;) {
      const uint8_t low = (uint8_t)(g & 0x7f);
      g >>= 7;
      *w++ = (uint8_t)(low | (g ? 0x80 : 0x00));
    }
  memset(w, 0, pad), w += pad;
  memcpy(w, hashes, n * 8), w += n * 8;
  memcpy(w, "cbir", 4);
  return total;
}

// header of a v2 file: fields of the first line, as checkHeader_v2 (:214-246) accepts them
static int vdx_header(const uint8_t* buf, size_t len, size_t* hdr_len, size_t* num_frames) {
  size_t nl = 0;
  while (nl < len && nl < 255 && buf[nl] != '\n') ++nl;
  if (nl >= len || buf[nl] != '\n') return CBH_E_INVAL;
  std::vector<std::string> f;
  {
    std::string cur;
    for (size_t i = 0; i <= nl; ++i) {
      if (i == nl || buf[i] == ':') {
        f.push_back(cur);
        cur.clear();
      } else
        cur.push_back((char)buf[i]);
    }
  }
  // "a:b:c:d:e:f:g:" + "\n" -> 8 fields in the reference's split(':') (the last is the newline)
  if (f.size() != 8 || f[0] != "cbird video index") return CBH_E_INVAL;
  if (atoi(f[2].c_str()) != 2 || atoi(f[4].c_str()) != 1 || atoi(f[5].c_str()) != 8) return CBH_E_UNSUPPORTED;
  if (atoi(f[3].c_str()) != 1) return CBH_E_UNSUPPORTED;  // other endianness (:242-245)
  *num_frames = strtoul(f[6].c_str(), nullptr, 10);
  *hdr_len = nl + 1;
  return CBH_OK;
}

/* VideoIndex::load_v2 (:350-429).  Like the reference's loader it does NOT look at the "cbir" trailer (that is
 * verify_v2's job, cbh_vdx_verify below), and a file claiming more than MAX_FRAMES_PER_VIDEO frames is loaded up to
 * that limit (:366-370, :395) instead of being rejected. */
static long long vdx_decode_v2(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap) {
  if (!buf) return CBH_E_INVAL;
  size_t hdr = 0, numFrames = 0;
  int rc = vdx_header(buf, len, &hdr, &numFrames);
  if (rc) return rc;
  if (numFrames == 0) return 0;
  bool reduced = false;
  if (numFrames > (1u << 24)) {  // MAX_FRAMES_PER_VIDEO
    numFrames = 1u << 24;
    reduced = true;
  }
  if (numFrames > cap) return CBH_E_OVERFLOW;
  if (hdr + 4 > len) return CBH_E_INVAL;
  uint32_t packedLen;
  memcpy(&packedLen, buf + hdr, 4);
  if (packedLen < numFrames || hdr + 4 + (size_t)packedLen > len) return CBH_E_INVAL;
  int frame = 0, jump = 0, shift = 0;
  size_t nfr = 0;
  for (uint32_t i = 0; i < packedLen; ++i) {
    const uint8_t byte = buf[hdr + 4 + i];
    if (0 == (byte & 0x80)) {
      frame += jump | (byte << shift);
      jump = 0;
      shift = 0;
      if (nfr < numFrames && frames) frames[nfr] = frame;
      ++nfr;
      if (reduced && nfr == numFrames) break;  // (:395)
    } else {
      jump |= (byte & 0x7F) << shift;
      shift += 7;
    }
  }
  if (jump || nfr != numFrames) return CBH_E_INVAL;
  const size_t here = hdr + 4 + packedLen;
  size_t pad = 8 - (here % 8);
  if (pad == 8) pad = 0;
  if (here + pad + numFrames * 8 > len) return CBH_E_INVAL;  // "hashes": short read
  if (hashes) memcpy(hashes, buf + here + pad, numFrames * 8);
  return (long long)numFrames;
}

/* VideoIndex::getVersion (src/videoindex.cpp:41-68): a file that starts with "cbird" is version 2, anything else --
 * including a file too short to tell -- the old version 1 (u16 frame count, u16 frame numbers, u64 hashes; "limited to
 * 65k frames/videos"). */
int cbh_vdx_version(const uint8_t* buf, size_t len) {
  if (!buf || len < 5) return 1;
  return memcmp(buf, "cbird", 5) == 0 ? 2 : 1;
}

/* VideoIndex::load_v1 (:478-541) with its two repairs: frame numbers that wrapped past 65535 (an old writer's bug) cut
 * the index there (:503-516), and an index whose first frame is not 0 gets a frame 0 with hash 0 in front (:530-535) */
static long long vdx_decode_v1(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap) {
  if (len < 2) return CBH_E_INVAL;  // "header"
  uint16_t numFrames;
  memcpy(&numFrames, buf, 2);
  if (numFrames == 0) return 0;
  const size_t orig = numFrames;
  if (2 + 2 * orig > len) return CBH_E_INVAL;  // "frame numbers"
  std::vector<int32_t> f(orig);
  size_t n = orig;
  uint16_t last = 0;
  for (size_t i = 0; i < orig; ++i) {
    uint16_t frame;
    memcpy(&frame, buf + 2 + 2 * i, 2);
    if (frame < last) {
      if (last > 65000) {  // probably wrapped due to having too many frames
        if (last != UINT16_MAX) {
          f[i] = UINT16_MAX;
          i++;
        }
        n = i;
        break;
      }
      return CBH_E_INVAL;  // non-sequential frame number (corrupt file?)
    }
    last = frame;
    f[i] = frame;
  }
  if (2 + 2 * orig + 8 * n > len) return CBH_E_INVAL;  // "hashes"
  const bool fix0 = n && f[0] != 0;
  if (n + (fix0 ? 1 : 0) > cap) return CBH_E_OVERFLOW;
  const size_t o = fix0 ? 1 : 0;
  if (frames) {
    if (fix0) frames[0] = 0;
    memcpy(frames + o, f.data(), n * sizeof(int32_t));
  }
  if (hashes) {
    if (fix0) hashes[0] = 0;
    memcpy(hashes + o, buf + 2 + 2 * orig, n * 8);
  }
  return (long long)(n + o);
}

/* VideoIndex::load (:70-90): version 1 or 2 by the magic; a negative CBH_E_* where the reference's loader fails (it
 * then leaves the index empty) */
long long cbh_vdx_decode(const uint8_t* buf, size_t len, int32_t* frames, uint64_t* hashes, size_t cap) {
  if (!buf) return CBH_E_INVAL;
  return cbh_vdx_version(buf, len) == 2 ? vdx_decode_v2(buf, len, frames, hashes, cap)
                                        : vdx_decode_v1(buf, len, frames, hashes, cap);
}

/* VideoIndex::save_v1 (:448-476), for files an old cbird still has to read: at most INT16_MAX frames, frame numbers
 * up to 65535 (the rest is dropped, as there).  Returns the size, and writes the file when it fits. */
size_t cbh_vdx_encode_v1(const int32_t* frames, const uint64_t* hashes, size_t n, uint8_t* out, size_t cap) {
  if (n && (!frames || !hashes)) return 0;
  size_t numFrames = n < (size_t)INT16_MAX ? n : (size_t)INT16_MAX;
  for (size_t i = 0; i < numFrames; ++i)
    if (frames[i] > UINT16_MAX || frames[i] < 0) {
      numFrames = i;
      break;
    }
  const size_t size = 2 + 10 * numFrames;
  if (out && size <= cap) {
    const uint16_t nf = (uint16_t)numFrames;
    memcpy(out, &nf, 2);
    for (size_t i = 0; i < numFrames; ++i) {
      const uint16_t fr = (uint16_t)frames[i];
      memcpy(out + 2 + 2 * i, &fr, 2);
    }
    if (numFrames) memcpy(out + 2 + 2 * numFrames, hashes, 8 * numFrames);
  }
  return size;
}

/* VideoIndex::isValid (:92-103): verify_v2 (:248-269: header fields + the "cbir" trailer at the end of the file; a file
 * with 0 frames is valid without one) or verify_v1 (:431-446: the size the frame count implies).  1 = valid, 0 = not. */
int cbh_vdx_verify(const uint8_t* buf, size_t len) {
  if (!buf) return 0;
  if (cbh_vdx_version(buf, len) == 1) {
    if (len < 2) return 0;
    uint16_t numFrames;
    memcpy(&numFrames, buf, 2);
    return len == 2 + (size_t)10 * numFrames;
  }
  size_t hdr = 0, numFrames = 0;
  if (vdx_header(buf, len, &hdr, &numFrames)) return 0;
  if (numFrames == 0) return 1;
  return len >= hdr + 4 && memcmp(buf + len - 4, "cbir", 4) == 0;
}

/* Media::makeVideoIndex frame de-dup (src/media.cpp:958-1024) over a sequence of frame hashes */
size_t cbh_video_dedup(const uint64_t* hashes, size_t n, int threshold, uint8_t* keep) {
  if (n == 0 || !hashes || !keep) return 0;
  std::vector<uint64_t> window;
  size_t kept = 1;
  keep[0] = 1;
  for (size_t i = 1; i < n; ++i) {
    keep[i] = 0;
    if (threshold > 0) {
      // the window as the set of its hashes: only "is any of them far" is asked of it (vindexer.hip, feed)
      bool far = false, have = false;
      for (uint64_t prev : window) {
        const int d = __builtin_popcountll(prev ^ hashes[i]);
        if (d >= threshold) {
          far = true;
          break;
        }
        have |= d == 0;
      }
      if (far) {
        window.clear();
        keep[i] = 1;
      }
      if (far || !have) window.push_back(hashes[i]);
    } else
      keep[i] = 1;
    kept += keep[i];
  }
  if (!keep[n - 1]) {
    keep[n - 1] = 1;
    ++kept;
  }
  return kept;
}

}  // extern "C"
