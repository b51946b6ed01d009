// cbh_shard.h -- internal: the exchange step shared by the sharded 64-bit index (sharded.hip) and the sharded
// CvFeaturesIndex (idx256.hip).  Not part of the C-ABI.
//
// Every shard has scanned its rows into its own record buffer and the host knows the R counts.  exchange() brings the
// records of all shards into one destination array on the root device, shard after shard:
//   inside a device   device-to-device copies of exactly count_s records
//   between devices   level 1: a device's shards concatenate into ONE block B_d = { count_d, records }; level 2: one
//                     grouped ncclAllGather of the B_d, sized to the fullest device (1 + max count words) -- librccl
//                     called directly, one communicator per device, all in this process (ncclCommInitAll), loaded with
//                     dlopen on first use; level 3: the root compacts the D gathered blocks into the destination.
//                     (the "shard_exchange" = 0 form).  The default, "shard_exchange" = 1, is hipMemcpyPeerAsync of
//                     exactly count_s records straight into the destination: only the root consumes the records, an
//                     all-gather would move D times the bytes.  Also the fallback when no communicator can be made.
// With one device the collective is skipped unless "shard_force_rccl" = 1 (transport test of a one-GPU box).
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "cbh_internal.h"

namespace cbh {

struct XBuf {  // a grow-only device buffer on the device that was current when it grew
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t b) {
    if (b <= bytes) return CBH_OK;
    if (p) (void)hipFree(p);
    p = nullptr, bytes = 0;
    CBH_HIP(hipMalloc(&p, b));
    bytes = b;
    return CBH_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, bytes = 0;
  }
};

struct ShardPart {
  int dev_pos = 0;                             // position of the shard's device in ShardComm::devices
  hipStream_t stream = nullptr;                // the shard's stream (its scan ran there)
  const unsigned long long* d_rec = nullptr;   // its records
  unsigned long long count = 0;                // how many (<= what d_rec holds)
  // optional: d_rec[-1] is the count word written by the scan kernel and the block has room for own_cap records
  // (a single shard per device then sends its block as it stands)
  const unsigned long long* own_block = nullptr;
  size_t own_cap = 0;
  hipEvent_t ev = nullptr;                     // an event of the shard's device, recorded by exchange()
  XBuf* x = nullptr;                           // two exchange buffers x[0], x[1] on the shard's device
  unsigned long long* h_word = nullptr;        // one pinned word (the device's count travels through it)
};

struct ShardComm {
  std::vector<int> devices;  // distinct devices in mask order; devices[0] = root
  int per_device = 1;
  uint32_t mask = 0;
  std::mutex coll_mu;  // a communicator takes one grouped call at a time
  std::vector<void*> comms;  // ncclComm_t, one per device, created with the first exchange that needs them
  bool comms_tried = false;
  std::atomic<uint64_t> n_scans{0}, n_rescans{0}, n_collectives{0}, n_peer_copies{0}, n_local_copies{0};
  std::atomic<uint64_t> n_fallbacks{0};  // exchanges that wanted the collective and went by copies (no communicator)

  // device list of a mask (every named device must be a usable gfx950); false when the mask is unusable
  bool init(uint32_t device_mask, int shards_per_device);
  size_t shard_count() const { return devices.size() * (size_t)per_device; }
  int device_of_shard(size_t s) const { return devices[s / (size_t)per_device]; }
  int dev_pos_of_shard(size_t s) const { return (int)(s / (size_t)per_device); }
  // parts in shard order (shards of a device adjacent).  On return root_stream waits for everything that lands in
  // d_dst; the shard streams may still be busy with their side of the collective.
  // went_collective (optional): whether the records travelled through ncclAllGather (false: by copies)
  // root_prefilled (copies only): records that already sit at the start of d_dst (the root device's shards appended them
  // in place, sharded_scan_all); the parts land behind them
  int exchange(std::vector<ShardPart>& parts, hipStream_t root_stream, unsigned long long* d_dst,
               bool* went_collective = nullptr, unsigned long long root_prefilled = 0);
  void destroy_comms();
};

void set_shard_force_rccl(int v);
void set_shard_exchange(int v);

}  // namespace cbh
