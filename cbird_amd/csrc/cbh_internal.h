// cbh_internal.h -- shared declarations between the host shim and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "cbird_hip.h"

namespace cbh {

void combiner_drop(const void* handle);  // combine.hip
void set_last_error(const char* where, hipError_t e);
void set_last_error_text(const char* text);  // non-HIP failures (RCCL)
// The thread's error state is per call for the entry points that return a handle: clear_last_error() on entry, and every
// NULL return goes through fail_handle(code, why) so that cbh_last_error_code() names THIS failure, never an older one.
void clear_last_error();
void set_last_error_code(int code);  // keeps the text (a successful fall-back leaves its note, not an error code)
void* fail_handle(int code, const char* why);  // sets code + text (text kept if `why` is null and a text exists), returns nullptr

// ---- stream-ordered scratch memory (cbird_hip.hip) ------------------------------------------------------------------
// Every kernel launcher takes its scratch with malloc_async(&p, bytes, stream) and gives it back with
// free_async(p, stream) right behind the last kernel that uses it (the contract of hipMallocAsync / hipFreeAsync).
// The source is the library's own arena: hipMalloc'ed blocks cached per (device, stream), reused only by the stream that
// freed them; bounded ("pool_live_keep_mb" per live stream, "pool_keep_mb" for blocks that outlive theirs, 32 stream
// caches, cbh_trim).  (ROCm's hipMallocAsync pools hand out memory that is still in use on this stack:
// tools/ubench/pool_cross_stream.hip.)
hipError_t malloc_async(void** p, size_t bytes, hipStream_t s);
hipError_t free_async(void* p, hipStream_t s);
// for streams the library creates itself: synchronise, hand the cached blocks to the device's orphan list, destroy
void stream_destroy(hipStream_t s);
void set_pool_keep_mb(int mb);
void set_pool_live_keep_mb(int mb);
int trim_pools(int device, unsigned long long* released_bytes);
// The scratch of one launcher: every block taken through get() goes back with free_async on the same stream when the
// guard leaves scope -- behind the kernels that were queued, and on every early return (a later allocation that fails,
// a launch error) as well as at the end.
struct Scratch {
  hipStream_t s;
  void* blocks[16];
  int n = 0;
  explicit Scratch(hipStream_t stream) : s(stream) {}
  Scratch(const Scratch&) = delete;
  Scratch& operator=(const Scratch&) = delete;
  template <class T>
  hipError_t get(T** p, size_t bytes) {
    void* q = nullptr;
    *p = nullptr;
    if (n >= 16) return hipErrorInvalidValue;
    hipError_t e = malloc_async(&q, bytes, s);
    if (e != hipSuccess) return e;
    blocks[n++] = q;
    *p = static_cast<T*>(q);
    return hipSuccess;
  }
  ~Scratch() {
    while (n > 0) (void)free_async(blocks[--n], s);
  }
};

// ---- fault injection (cbird_hip.hip; tests/test_error_paths.py) ------------------------------------------------------
// Every allocation the library makes -- scratch through malloc_async, index / table memory through hipMalloc, pinned
// words through hipHostMalloc -- first passes fault_gate().  cbh_set_tuning("fault_alloc_after", n) arms it: the n-th
// allocation from now (0 = the next one) fails ONCE with hipErrorOutOfMemory, then the gate disarms itself;
// cbh_get_tuning("fault_alloc_after") reads what is left (-1 = disarmed or fired).  The arena's own calls into the
// driver are written (hipMalloc)(...) and pass "fault_driver_oom" instead (the trim-and-retry path).
hipError_t fault_gate();
hipError_t fault_gate_driver();
void set_fault_alloc_after(int n);
void set_fault_driver_oom(int n);
void set_fault_alloc_sticky(int v);
void set_fault_rccl(int v);  // sharded.hip: 1 = behave as if librccl could not be loaded
long get_fault_alloc_after();
unsigned long get_fault_fired();
unsigned long get_alloc_calls();
int arena_counter(const char* name, long long* value);
hipError_t persistent_malloc(void** p, size_t bytes);
void set_fault_persist_oom(int n);  // "fault_persist_oom": the n-th persistent allocation is refused by the "driver" once
template <class T>
static inline hipError_t gated_malloc(T** p, size_t bytes) {
  hipError_t e = fault_gate();
  if (e != hipSuccess) {
    *p = nullptr;
    return e;
  }
  return persistent_malloc((void**)p, bytes);  // cbird_hip.hip: gives the scratch arena's caches back before it fails
}
template <class T>
static inline hipError_t gated_host_malloc(T** p, size_t bytes, unsigned flags = hipHostMallocDefault) {
  hipError_t e = fault_gate();
  if (e != hipSuccess) {
    *p = nullptr;
    return e;
  }
  return (hipHostMalloc)(p, bytes, flags);
}
#define hipMalloc(...) ::cbh::gated_malloc(__VA_ARGS__)
#define hipHostMalloc(...) ::cbh::gated_host_malloc(__VA_ARGS__)

#define CBH_HIP(call)                          \
  do {                                         \
    hipError_t e_ = (call);                    \
    if (e_ != hipSuccess) {                    \
      ::cbh::set_last_error(#call, e_);        \
      return e_ == hipErrorOutOfMemory ? CBH_E_NOMEM : CBH_E_HIP; \
    }                                          \
  } while (0)

// ---- hamm64_scan.hip ------------------------------------------------------------------
// Appends one record per (query j, slot i) with popc(q[j]^hashes[i]) < thresh, ids[i] != 0,
// q[j] != 0.  *d_total += number of such pairs; records with slot index >= cap are dropped.
// d_qmask (optional, one u64 per query): additionally require ((q[j] ^ hashes[i]) & qmask[j]) == 0 -- the
// reference's approximate structures only compare a needle with the entries that share its low bits
// (HammingTree leaf, src/tree/hammingtree.h:244-252; RadixMap bucket, src/tree/radix.h:135-141).
int launch_hamm64_scan(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n,
                       const uint64_t* d_q, size_t nq, int thresh, cbh_record* d_rec, size_t cap,
                       unsigned long long* d_total, hipStream_t stream, unsigned flags = 0,
                       const uint64_t* d_qmask = nullptr, const void* qx_given = nullptr);
enum { SCAN_KEEP_ID0 = 1u,    // also emit slots whose id is 0 (DctFeaturesIndex top-10 cut)
       SCAN_PRE_GIVEN = 2u,   // matrix-core scan: the prefilter / three-field choice was made by the caller (a sharded
       SCAN_PRE_VALUE = 4u,   // handle probes once for all its shards) -- SCAN_PRE_VALUE says which
       SCAN_SIBLINGS_SHIFT = 8 }; // bits 8..15: launches against the same needles running side by side on the device (0 = alone)

// ---- the lone needle (Engine::query / -similar-to: one find() at a time) ---------------------------------------------
// One kernel launch and no copies: the needle travels as a kernel argument, matches go straight into a pinned, coherent
// host block, the last workgroup to finish publishes the count and a sequence number that the host polls for
// (hipStreamSynchronize costs ~6.5 us even on a drained stream; the plain path's needle upload, counter reset, two
// read-backs and synchronisation were 26 us per find, ~25 us PER SHARD on a sharded handle).
struct LoneBlock {
  volatile unsigned long long done;  // the call's sequence number, written last
  unsigned long long count;          // matches found (may exceed kRecs: then the caller takes the general path)
  static constexpr unsigned kRecs = 512;
  cbh_record recs[kRecs];            // dist << 32 | id (needle index 0), unordered
};
// d_state: two zeroed words of device memory owned by the caller's workspace (the kernel leaves them zeroed)
int launch_find_one(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n, uint64_t q, int thresh, unsigned* d_state,
                    LoneBlock* h_block, unsigned long long seq, hipStream_t stream);
// spin until the block carries `seq` (falls back to a stream synchronisation after ~2 s); CBH_OK / CBH_E_HIP
int wait_find_one(const LoneBlock* h_block, unsigned long long seq, hipStream_t stream);

// ---- hamm64_mfma.hip: the same scan on the matrix cores (FP4 sign dot products) --------
int launch_hamm64_scan_mfma(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n,
                            const uint64_t* d_q, size_t nq, int thresh, cbh_record* d_rec,
                            size_t cap, unsigned long long* d_total, hipStream_t stream,
                            unsigned flags = 0, const uint64_t* d_qmask = nullptr, const void* qx_given = nullptr);
// the needles of a call in the matrix-core kernels' operand layout, made ONCE for several launches against the same needles
// on one device (the shards of a sharded handle): *qx = malloc_async on `stream`, to be handed to every launch as qx_given
// (launches on other streams wait for an event of `stream`) and given back with free_async once they have all finished
int expand_needles_for_scan(const uint64_t* d_q, size_t nq, hipStream_t stream, void** qx);
bool scan_mfma_wanted(size_t n, size_t nq, int thresh);
int get_scan_mfma();  // the "scan_mfma" knob: 0 popcount kernel, 1 as shipped, 2 matrix-core scan forced, 3 + the bucketed join where
                      // its candidate count says so, 4 the join forced wherever it can represent the call
// ---- hamm64_join.hip: the same search as a bucketed join (multi-index hashing), thresholds <= 8 ----------------------
bool scan_join_possible(size_t n, size_t nq, int thresh, unsigned flags, const uint64_t* d_qmask);
long long get_scan_joins();  // calls the join has answered so far (cbh_get_tuning "scan_joins")
// CBH_OK = done; CBH_E_UNSUPPORTED = the scan is cheaper for this call (decided from the exact candidate count against
// scan_ms_estimate unless `force`): nothing written, the caller scans
int launch_hamm64_join(const uint64_t* d_hashes, const uint32_t* d_ids, size_t n, const uint64_t* d_q, size_t nq,
                       int thresh, cbh_record* d_rec, size_t cap, unsigned long long* d_total, hipStream_t stream,
                       unsigned flags, bool force, double scan_ms_estimate);
unsigned scan_pre_flags(const uint64_t* d_hashes, size_t n, size_t n_total, const uint64_t* d_q, size_t nq, int thresh,
                        hipStream_t stream);  // SCAN_PRE_GIVEN | SCAN_PRE_VALUE, probed once for a sharded call
void set_scan_mfma(int on);  // <0 = keep; 2 = force for any size
void set_scan_pre_max(int t);   // -1 = prefilter or three-field kernel by the launch's candidate rate (default), 0 = never the
                                // prefilter, t > 0 = thresholds <= t take it whatever the data
void set_scan_pre_rate(int e9); // candidate rate x 1e9 up to which the prefilter kernel is taken ("scan_pre_rate_e9")
long long get_scan_pre_mask();  // bit t = the most recent matrix-core launch at threshold t took the prefilter kernel
long long get_scan_probes();    // candidate-rate probes run so far
long long get_scan_probe_rate_e9();  // what the last one found for its threshold, x 1e9 (-1: none yet): candidates ...
long long get_scan_probe_true_e9();  // ... and true matches

// ---- hamm256_mfma.hip: 256-bit threshold scan on the matrix cores -----------------------
int launch_scan256_mfma(const uint8_t* d_rows, size_t n, const uint8_t* d_q, size_t nq, int thresh,
                        unsigned long long* d_rec, size_t cap, unsigned long long* d_total,
                        hipStream_t stream);
bool scan256_mfma_wanted(size_t n, size_t nq, int thresh);
void set_scan256_small(int v);  // stationary-needle kernel for <= 512 needle descriptors: 0 / 1
void set_scan256_mfma(int on);  // <0 = keep; 2 = force for any size



// ---- setters behind cbh_set_tuning (include/cbird_hip.h documents every knob) ---------------------------------------
int g_hash_mfma_set(int v);        // dcthash.hip "hash_mfma": 256 x 256 tiles on k_dcthash_256_band (non-zero) or k_dcthash_256 (0)
void set_hash_band_area(int v);    // dcthash.hip "hash_band_area": k_band_area for fractional ratios up to 1920 columns (1) or never (0)
void set_hash_stream(int v);       // dcthash.hip "hash_stream": k_blur_area_regs on strips -- 0 never, 1 by batch size, >= 2 always, of v steps
void set_hash_fuse(int v);         // dcthash.hip "hash_fuse": vertical INTER_AREA pass + tile inside k_blur_area_regs (0 never, 1 auto, 2 always)
void set_kp_blur_side(int v);      // kphash.hip "kp_blur_side": largest keypoint square whose blurred copy stays in LDS (default 112)
void set_kp_lds_side(int v);       // kphash.hip "kp_lds_side": largest keypoint square processed in LDS (default 134)
extern int g_fdct_host_vote, g_video_host_reduce;  // fdct.hip: 0 = device for batches / host for one needle, 1 = host, 2 = device
void set_orb_retain_order(int v);  // orb.hip: 1 (default) retainBest in libstdc++'s order, 0 canonical (ties kept, raster order)
void set_cd_chunk_mb(int v);       // colordesc_create.hip "color_create_chunk_mb": MB of scratch one launch may take
void set_color_fma(int on);        // color.hip "color_fma": fused squares in k_color_dist3 (default off: not bit-identical)

// ---- records.hip ----------------------------------------------------------------------
// Ascending u64 sort of n records in place (uses d_alt as the ping-pong buffer and d_tmp as
// scratch; sizes from sort_records_scratch_bytes).
size_t sort_records_scratch_bytes(size_t n);
int sort_keys64_db(unsigned long long* d_keys, unsigned long long* d_alt, size_t n, unsigned end_bit, void* d_tmp,
                   size_t tmp_bytes, hipStream_t stream, unsigned long long** sorted);  // any 64-bit keys, double buffer
int launch_sort_records(cbh_record* d_rec, cbh_record* d_alt, size_t n, size_t nq, void* d_tmp,
                        size_t tmp_bytes, hipStream_t stream);
int launch_select_records(const cbh_record* d_sorted, size_t n, size_t nq, int k, cbh_match* d_out,
                          uint32_t* d_counts, hipStream_t stream);
int launch_remove_ids(uint64_t* d_hashes, uint32_t* d_ids, size_t n, const uint32_t* d_sorted_rm,
                      size_t n_rm, hipStream_t stream, int zero_hash = 1);

// ---- topk.hip: K4 counting select over { count, records[cap] } blocks ---------------------------
constexpr int kTopkMaxK = 64;  // larger cuts take the radix sort (records.hip)
size_t topk_scratch_bytes(size_t nq, size_t total_cap);
int topk_scratch_init(void* d_scratch, size_t nq, hipStream_t stream);
int launch_records_group(const unsigned long long* d_blocks, unsigned nb, size_t stride, size_t cap, size_t nq,
                         unsigned* d_status, void* d_scratch, const unsigned** d_off, const unsigned long long** d_seg,
                         hipStream_t stream);
int launch_records_topk(const unsigned long long* d_blocks, unsigned nb, size_t stride, size_t cap, size_t nq, int k,
                        cbh_match* d_out, uint32_t* d_counts, unsigned* d_status, void* d_scratch,
                        hipStream_t stream);

// ---- reduce.hip: K5 (fdct votes) and K8 (video closest-frame + adjacency) on the device ---------
struct cbh_nmatch {   // one DctFeaturesIndex result of needle image `needle`
  uint32_t needle, id;
  int32_t score;
};
struct cbh_nvmatch {  // one DctVideoIndex::findVideo result of needle video `needle`
  uint32_t needle;
  cbh_vmatch m;
};
int sort_keys_u64(unsigned long long* d_keys, size_t n, int end_bit, hipStream_t s);      // (records.hip: in place, arena scratch)
int sort_pairs_u64_u32(unsigned long long* d_keys, uint32_t* d_vals, size_t n, int end_bit, hipStream_t s);
int launch_fdct_vote(const cbh_match* d_top, const uint32_t* d_counts, const uint32_t* d_qneedle, size_t nq, int k,
                     const uint32_t* d_needle_id, size_t n_needles, std::vector<cbh_nmatch>* h_out, hipStream_t s);
int launch_video_reduce(const unsigned* d_off, const unsigned long long* d_seg, size_t total, size_t nq,
                        const uint32_t* d_evidx, const int32_t* d_eframe, const uint32_t* d_vmedia,
                        const uint32_t* d_qneedle, const int32_t* d_qframe, const uint32_t* d_needle_id, int filter_self,
                        int min_matched, int min_near, std::vector<cbh_nvmatch>* h_out, hipStream_t s);

// ---- dcthash.hip (whole images), kphash.hip (rectangles, keypoint squares) -------------
// view: the images are w x h sub-rectangles at (ox, oy) of pw x ph parents starting at d_imgs -- cv::blur on a
// cv::Mat view takes its border pixels from the parent (dctHash64 after autocrop(), src/cvutil.cpp:1397-1401)
struct HashView {
  int pw, ph, ox, oy;
};
int launch_dcthash(const uint8_t* d_imgs, size_t n, int w, int h, size_t row_stride,
                   size_t img_stride, uint64_t* d_out, hipStream_t stream,
                   uint8_t* d_tiles = nullptr, const HashView* view = nullptr);
// rectangles of images hashed one after the other, optionally in place (Media::makeKeyPointHashes); also the path of
// images with a side < 32
struct RectImageDesc {
  unsigned long long off;  // first byte of the image in the batch buffer
  int w, h;
  unsigned row_stride;
  unsigned first, count;   // its rectangles: rects[4*first .. 4*(first+count))
};
int launch_keypoint_hashes(uint8_t* d_base, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                           const uint32_t* img_h, const uint32_t* img_row_stride, const float* kp,
                           const uint32_t* kp_first, uint64_t* d_out, uint32_t* out_first, hipStream_t stream);
int launch_rect_hashes(uint8_t* d_base, const std::vector<RectImageDesc>& images, const std::vector<int>& rects,
                       int write_back, uint64_t* d_out, hipStream_t stream, uint8_t* d_tiles = nullptr);

// ---- orb.hip: ORB keypoints + rBRIEF descriptors (Media::makeKeyPoints / makeKeyPointDescriptors) ------------------
int orb_set_pattern(const int8_t* xy);
int launch_orb(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w, const uint32_t* img_h,
               const uint32_t* img_row_stride, int nfeatures, int kp_cap, cbh_keypoint* d_kp, float* d_kp_after,
               uint8_t* d_desc, uint32_t* d_counts, hipStream_t s);
int launch_orb_describe(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                        const uint32_t* img_h, const uint32_t* img_row_stride, const cbh_keypoint* kp,
                        const uint32_t* kp_first, cbh_keypoint* out_kp, uint8_t* out_desc, uint32_t* out_first,
                        hipStream_t s);

// ---- colordesc_create.hip: ColorDescriptor::create for a batch --------------------------------------------------
int launch_color_descriptors(const uint8_t* d_imgs, size_t n, const uint64_t* img_off, const uint32_t* img_w,
                             const uint32_t* img_h, const uint32_t* img_row_stride, int channels, uint8_t* d_descs,
                             uint8_t* d_ok, hipStream_t s);

}  // namespace cbh
