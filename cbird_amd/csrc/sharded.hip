// sharded.hip -- one index over several shards / several GPUs, inside ONE process, behind the same C-ABI handle.
//
// cbird is one process: Engine::Engine registers each Index once (src/engine.cpp:38-45) and Database::similar fans
// find() out from a thread pool (src/database.cpp:1400-1432).  A drop-in that wants all GPUs of the node therefore has
// to shard INSIDE the handle -- cbh_idx64_create_sharded(device_mask, shards_per_device) returns a cbh_idx64 that every
// other entry point accepts -- not in a torchrun harness (cbird_amd/dist.py keeps that form for bench.py --gpus N).
//
// Layout: the haystack is row-sharded, shard s of R owns the slots [s*n/R, (s+1)*n/R) of a load in load
// order (add() appends to the emptiest shard and the global order is kept as a segment list); needles are replicated.
// A threshold search over a union of shards is the union of the per-shard results, so there is exactly one sharded
// primitive: scan_all (cbh_index.h) -- "all records of these needles, in the root workspace's { count, records }
// block".  Everything above it (the K4 cut, fdct votes, video reduce, searchIndex escalation, the coalescer's self-join)
// runs unchanged on the root device over the merged block.
//
//   scan      every shard scans its slots for all needles on its own device and stream, into its own block
//             { u64 count; records[cap] } (needles reach the other devices by peer copies behind an event);
//             the host reads the R counts (the one synchronisation scan_all always had); only a shard whose block
//             overflowed grows it and scans again
//   exchange  ShardComm::exchange (cbh_shard.h; shared with the sharded CvFeaturesIndex, idx256.hip) --
//             inside a device: device-to-device copies of exactly count_s records behind each other;
//             between devices: ONE grouped ncclAllGather of the per-device blocks, sized to the fullest device
//             (1 + max count words) -- librccl called directly (ncclCommInitAll, one communicator per device, all
//             in this process), found with dlopen so that a single-GPU user never loads it;
//             "shard_exchange" = 1 replaces the collective by hipMemcpyPeerAsync straight into the root block
//   merge     the D gathered blocks are compacted into the root block; word 0 = total
//
// One GPU per box is all this pool offers, so the inter-device leg runs here with one rank ("shard_force_rccl" = 1:
// a one-device index still goes through ncclAllGather), and everything else -- ragged shards, overflow-redo, removal,
// order of a merged result -- with R logical shards on one device (tests/test_sharded_capi.py, tests/cpp).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "cbh_index.h"

namespace cbh {

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;  // when it could not be loaded
};

// librccl.so.1: the copy already mapped in this process if there is one (PyTorch ships its own), else the system's
Rccl* rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      x->handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (x->handle) break;
    }
    if (!x->handle)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        x->handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (x->handle) break;
      }
    if (!x->handle) {
      const char* e = dlerror();
      x->why = e ? e : "librccl not found";
      return x;
    }
#define CBH_SYM(field, name)                                         \
  x->field = reinterpret_cast<decltype(x->field)>(dlsym(x->handle, name)); \
  if (!x->field) x->why = std::string("librccl lacks ") + name
    CBH_SYM(CommInitAll, "ncclCommInitAll");
    CBH_SYM(CommDestroy, "ncclCommDestroy");
    CBH_SYM(AllGather, "ncclAllGather");
    CBH_SYM(GroupStart, "ncclGroupStart");
    CBH_SYM(GroupEnd, "ncclGroupEnd");
    CBH_SYM(GetErrorString, "ncclGetErrorString");
    CBH_SYM(GetVersion, "ncclGetVersion");
#undef CBH_SYM
    // the copy found may be the one another library of this process brought along (PyTorch ships its own): use it only
    // when it is the API generation this file was compiled against (same major version as <rccl/rccl.h>)
    int v = 0;
    if (x->why.empty() && (x->GetVersion(&v) != ncclSuccess || v / 10000 != NCCL_VERSION_CODE / 10000))
      x->why = "librccl version " + std::to_string(v) + " does not match the headers (" + std::to_string(NCCL_VERSION_CODE) + ")";
    return x;
  }();
  return r;
}

int g_force_rccl = 0;  // "shard_force_rccl": the collective also at one device (transport test on a one-GPU box)
// "shard_exchange": 1 = copies of exactly count_s records into the root block (default: only the root device consumes
// the records, an all-gather would put D times the bytes on the links), 0 = grouped ncclAllGather of the device blocks
int g_exchange = 1;
int g_fault_rccl = 0;  // "fault_rccl": librccl treated as absent

int ensure_comms(ShardComm* C) {  // under coll_mu
  if (!C->comms.empty()) return CBH_OK;
  if (C->comms_tried) return CBH_E_UNSUPPORTED;
  C->comms_tried = true;
  if (g_fault_rccl) {
    set_last_error_text("RCCL unavailable: fault_rccl");
    return CBH_E_UNSUPPORTED;
  }
  Rccl* r = rccl();
  if (!r->handle || !r->why.empty()) {
    set_last_error_text(("RCCL unavailable: " + r->why).c_str());
    return CBH_E_UNSUPPORTED;
  }
  std::vector<ncclComm_t> c(C->devices.size());
  ncclResult_t e = r->CommInitAll(c.data(), (int)C->devices.size(), C->devices.data());
  if (e != ncclSuccess) {
    set_last_error_text((std::string("ncclCommInitAll: ") + r->GetErrorString(e)).c_str());
    return CBH_E_HIP;
  }
  for (ncclComm_t x : c) C->comms.push_back((void*)x);
  return CBH_OK;
}

}  // namespace

void set_shard_force_rccl(int v) { g_force_rccl = v; }
void set_shard_exchange(int v) { g_exchange = v; }
void set_fault_rccl(int v) { g_fault_rccl = v; }

bool ShardComm::init(uint32_t device_mask, int shards_per_device) {
  if (device_mask == 0 || shards_per_device < 0 || shards_per_device > 64) return false;
  devices.clear();
  for (int d = 0; d < 32; ++d)
    if (device_mask & (1u << d)) {
      if (!device_usable(d)) return false;  // a device of the mask is not there: no silent narrowing
      devices.push_back(d);
    }
  per_device = std::max(1, shards_per_device);
  mask = device_mask;
  // direct xGMI copies where the platform allows them (RCCL opens its own).  A refusal is not an error: only copies
  // cross devices (no kernel dereferences a peer pointer), and hipMemcpyPeerAsync stages through the host without it.
  if (devices.size() > 1)
    for (int a : devices) {
      DeviceGuard g(a);
      for (int b : devices)
        if (a != b) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b, 0);
        }
      (void)hipGetLastError();  // "already enabled" is not an error worth keeping
    }
  return true;
}

void ShardComm::destroy_comms() {
  if (comms.empty()) return;
  Rccl* r = rccl();
  for (size_t d = 0; d < comms.size(); ++d) {
    DeviceGuard g(devices[d]);
    (void)r->CommDestroy((ncclComm_t)comms[d]);
  }
  comms.clear();
}

int ShardComm::exchange(std::vector<ShardPart>& parts, hipStream_t root_stream, unsigned long long* d_dst,
                        bool* went_collective, unsigned long long root_prefilled) {
  const size_t R = parts.size(), D = devices.size();
  const int root = devices[0];
  bool collective = g_exchange == 0 && (D > 1 || g_force_rccl);
  if (collective) {
    // no communicator (librccl absent or of another generation, ncclCommInitAll refused): the copies below give the
    // same block; said once per index in cbh_last_error and on stderr, counted in cbh_shard_stats.collective_fallbacks
    std::lock_guard<std::mutex> lk(coll_mu);
    if (ensure_comms(this) != CBH_OK) {
      collective = false;
      if (!n_fallbacks.fetch_add(1))
        fprintf(stderr, "cbird_hip: sharded index exchanges by device copies, not ncclAllGather (%s)\n", cbh_last_error());
      set_last_error_code(CBH_OK);  // absorbed: the note stays in cbh_last_error, the call has not failed
    }
  }
  // per-device totals, the place of every shard inside its device's run, and of every device in the destination
  std::vector<unsigned long long> dev_total(D, 0), shard_off(R, 0), dev_off(D, 0);
  dev_total[0] = root_prefilled;  // (copies only: records the root device's shards appended in place, ahead of everything)
  for (size_t s = 0; s < R; ++s) {
    shard_off[s] = dev_total[(size_t)parts[s].dev_pos];
    dev_total[(size_t)parts[s].dev_pos] += parts[s].count;
  }
  for (size_t d = 1; d < D; ++d) dev_off[d] = dev_off[d - 1] + dev_total[d - 1];
  std::vector<size_t> first_of(D, R);  // first shard of a device: its stream carries the device's part of a collective
  for (size_t s = R; s-- > 0;) first_of[(size_t)parts[s].dev_pos] = s;
  int rc;
  if (went_collective) *went_collective = collective;
  if (!collective) {
    // every shard copies exactly its records to their final place on the root device
    for (size_t s = 0; s < R; ++s) {
      ShardPart& P = parts[s];
      if (!P.count) continue;
      const int dev = devices[(size_t)P.dev_pos];
      DeviceGuard g(dev);
      if (!g.ok) return CBH_E_NODEVICE;
      unsigned long long* dst = d_dst + dev_off[(size_t)P.dev_pos] + shard_off[s];
      if (dev == root) {
        CBH_HIP(hipMemcpyAsync(dst, P.d_rec, P.count * 8, hipMemcpyDeviceToDevice, P.stream));
        n_local_copies++;
      } else {
        CBH_HIP(hipMemcpyPeerAsync(dst, root, P.d_rec, dev, P.count * 8, P.stream));
        n_peer_copies++;
      }
      CBH_HIP(hipEventRecord(P.ev, P.stream));
    }
    DeviceGuard g(root);
    for (size_t s = 0; s < R; ++s)
      if (parts[s].count) CBH_HIP(hipStreamWaitEvent(root_stream, parts[s].ev, 0));
    return CBH_OK;
  }
  unsigned long long m = 0;
  for (size_t d = 0; d < D; ++d) m = std::max(m, dev_total[d]);
  const size_t words = 1 + (size_t)m;
  std::vector<const void*> send(D, nullptr);
  std::vector<void*> recv(D, nullptr);
  for (size_t d = 0; d < D; ++d) {
    const size_t f = first_of[d];
    if (f == R) return CBH_E_INVAL;  // a device without a shard
    ShardPart& F = parts[f];
    DeviceGuard g(devices[d]);
    if (!g.ok) return CBH_E_NODEVICE;
    if ((rc = F.x[1].ensure(D * words * 8))) return rc;
    recv[d] = F.x[1].p;
    if (per_device == 1 && F.own_block && F.own_cap + 1 >= words) {
      send[d] = F.own_block;  // a single shard's block is the device block as it stands: word 0 = count (written by
      continue;               // the scan kernel), and it is at least `words` long
    }
    if ((rc = F.x[0].ensure(words * 8))) return rc;
    unsigned long long* B = (unsigned long long*)F.x[0].p;
    send[d] = B;
    *F.h_word = dev_total[d];
    CBH_HIP(hipMemcpyAsync(B, F.h_word, 8, hipMemcpyHostToDevice, F.stream));
    for (size_t s = 0; s < R; ++s) {
      ShardPart& P = parts[s];
      if ((size_t)P.dev_pos != d || !P.count) continue;
      CBH_HIP(hipMemcpyAsync(B + 1 + shard_off[s], P.d_rec, P.count * 8, hipMemcpyDeviceToDevice, P.stream));
      n_local_copies++;
      if (s != f) {
        CBH_HIP(hipEventRecord(P.ev, P.stream));
        CBH_HIP(hipStreamWaitEvent(F.stream, P.ev, 0));
      }
    }
  }
  {
    std::lock_guard<std::mutex> lk(coll_mu);
    if (comms.empty()) return CBH_E_UNSUPPORTED;  // (destroyed under our feet: an index being torn down)
    Rccl* r = rccl();
    ncclResult_t e = r->GroupStart();
    for (size_t d = 0; d < D && e == ncclSuccess; ++d) {
      DeviceGuard g(devices[d]);
      e = r->AllGather(send[d], recv[d], words, ncclUint64, (ncclComm_t)comms[d], parts[first_of[d]].stream);
    }
    ncclResult_t e2 = r->GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) {
      set_last_error_text((std::string("ncclAllGather: ") + r->GetErrorString(e)).c_str());
      return CBH_E_HIP;
    }
    n_collectives++;
  }
  DeviceGuard g(root);
  ShardPart& Rt = parts[first_of[0]];
  const unsigned long long* G = (const unsigned long long*)Rt.x[1].p;
  for (size_t d = 0; d < D; ++d)
    if (dev_total[d])
      CBH_HIP(hipMemcpyAsync(d_dst + dev_off[d], G + d * words + 1, dev_total[d] * 8, hipMemcpyDeviceToDevice, Rt.stream));
  CBH_HIP(hipEventRecord(Rt.ev, Rt.stream));
  CBH_HIP(hipStreamWaitEvent(root_stream, Rt.ev, 0));
  return CBH_OK;
}

struct ShardSet {
  ShardComm comm;
  std::vector<cbh_idx64*> child;  // plain single-device indexes; child[s]->device == comm.device_of_shard(s)
  // global slot order: segment g covers parent slots [global, global+len) = child[shard] slots [local, local+len)
  struct Seg {
    uint32_t shard;
    size_t local, global, len;
  };
  std::vector<Seg> segs;
};

void shardset_free(ShardSet* S) {
  if (!S) return;
  S->comm.destroy_comms();
  for (cbh_idx64* c : S->child) cbh_idx64_destroy(c);
  delete S;
}

namespace {

constexpr size_t kKeepXBufBytes = (size_t)64 << 20;  // exchange buffers above this go back after the call
constexpr size_t kKeepShardRecs = (size_t)1 << 22;   // a shard's record block above this (32 MB) likewise

// leases of one call: a workspace per shard, given back (idle) at the end
struct ShardLeases {
  ShardSet* S;
  std::vector<Workspace*> ws;
  std::vector<char> idle;  // the call has already seen this shard's stream drained (see sharded_scan_all's end)
  explicit ShardLeases(ShardSet* s) : S(s), ws(s->child.size(), nullptr), idle(s->child.size(), 0) {}
  int acquire_all() {
    for (size_t s = 0; s < ws.size(); ++s) {
      DeviceGuard g(S->child[s]->device);
      if (!g.ok) return CBH_E_NODEVICE;
      int rc = CBH_OK;
      ws[s] = S->child[s]->acquire(&rc);
      if (!ws[s]) return rc ? rc : CBH_E_NOMEM;
    }
    return CBH_OK;
  }
  ~ShardLeases() {
    for (size_t s = 0; s < ws.size(); ++s)
      if (ws[s]) {
        DeviceGuard g(S->child[s]->device);
        // its buffers must be idle when the next call takes it (6.5 us per synchronisation even on a drained stream: a
        // lone find() on 8 shards spent 52 of its 230 us here)
        if (!idle[s]) (void)hipStreamSynchronize(ws[s]->stream);
        // one large result (a self-join attempt) must not leave every shard holding a block of that size for good
        ws[s]->shrink_records(std::max<size_t>(S->child[s]->rec_cap_default, kKeepShardRecs));
        for (XBuf& x : ws[s]->x)
          if (x.bytes > kKeepXBufBytes) x.release();
        S->child[s]->give_back(ws[s]);
      }
  }
};

}  // namespace

int sharded_scan_all(cbh_idx64* idx, Workspace* ws, const uint64_t* d_q, size_t nq, int thresh, hipStream_t stream,
                     unsigned long long* total, unsigned flags, const uint64_t* d_qmask, size_t max_records) {
  ShardSet* S = idx->shards;
  ShardComm& C = S->comm;
  const size_t R = S->child.size();
  const int root = idx->device;
  *total = 0;
  // the root block must exist whatever happens (consumers read word 0)
  int rc = ws->ensure_records(std::min(std::max<size_t>(idx->rec_cap_default, 1024), std::max<size_t>(max_records, 1024)));
  if (rc) return rc;
  if (nq == 0 || idx->n == 0 || thresh <= 0) {
    CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), stream));
    CBH_HIP(hipStreamSynchronize(stream));
    return CBH_OK;
  }
  // (declared ahead of the leases: given back after their destructor has drained every shard stream that may still read it
  // -- on the paths that succeed the root stream has waited for those shards and nothing is synchronised again)
  void* qx_root = nullptr;
  struct QxFree {
    void*& p;
    hipStream_t st;
    int dev;
    ~QxFree() {
      if (p) {
        DeviceGuard g(dev);
        (void)free_async(p, st);
      }
    }
  } qx_free{qx_root, stream, root};
  ShardLeases L(S);
  if ((rc = L.acquire_all())) return rc;
  // ---- scan: every shard, its own device and stream ----
  std::vector<const uint64_t*> q_of(R, d_q), mask_of(R, d_qmask);
  std::vector<unsigned long long> count(R, 0);
  std::vector<char> todo(R, 1);
  for (size_t s = 0; s < R; ++s) todo[s] = S->child[s]->n != 0;
  // Shards that live on the root device append STRAIGHT into the root block through its one counter (the scan kernels
  // append with one atomic per wave and flush: eight kernels on a counter cost what one does): no block of their own, no
  // count read-back and stream synchronisation per shard, no copy into place -- per threshold a handle over 8 shards of one
  // device spent ~140 us between its scans and its cut on exactly those (tools/ab/sharded_trace.py).  Not under the
  // collective exchange, whose send buffers are the devices' own blocks.
  const bool collective_mode = g_exchange == 0 && (C.devices.size() > 1 || g_force_rccl);
  std::vector<char> direct(R, 0);
  bool any_direct = false;
  for (size_t s = 0; s < R; ++s) {
    direct[s] = !collective_mode && todo[s] && S->child[s]->device == root;
    any_direct = any_direct || direct[s];
  }
  unsigned long long direct_count = 0;
  float scan_ms = 0.f;
  // kernel timing for cbh_idx64_get_stats only where it can matter: a handful of needles is launch-bound, and every
  // HIP call counts there (a lone find() on 8 shards: ~12 calls per shard from this one thread)
  const bool timed = nq >= 256;
  // prefilter or three-field kernel: one probe for the whole call, on the slots of a shard that lives where the needles
  // are (every shard probing for itself cost a stream synchronisation per shard and threshold); and ONE expansion of the
  // needles into the matrix-core operand layout for all the shards of the root device (48 bytes per needle: eight of them
  // per threshold were 3.9 ms of kernel time beside the scans)
  for (size_t s = 0; s < R; ++s) {
    cbh_idx64* c = S->child[s];
    if (c->device == root && c->n != 0 && scan_mfma_wanted(c->n, nq, thresh)) {
      DeviceGuard g(root);
      flags |= scan_pre_flags(c->d_hashes, c->n, idx->n, d_q, nq, thresh, stream);
      if ((rc = expand_needles_for_scan(d_q, nq, stream, &qx_root))) return rc;
      break;
    }
  }
  for (int attempt = 0; attempt < 4; ++attempt) {
    bool any = false, any_direct_now = false;
    for (size_t s = 0; s < R; ++s) any_direct_now = any_direct_now || (todo[s] && direct[s]);
    {
      DeviceGuard g(root);
      if (any_direct_now) CBH_HIP(hipMemsetAsync(ws->d_total, 0, sizeof(unsigned long long), stream));
      // the needles (and masks, the expansion, the zeroed counter) are complete on the root device
      if (attempt == 0 || any_direct_now) CBH_HIP(hipEventRecord(ws->ev0, stream));
    }
    for (size_t s = 0; s < R; ++s) {
      if (!todo[s]) continue;
      any = true;
      cbh_idx64* c = S->child[s];
      Workspace* cw = L.ws[s];
      DeviceGuard g(c->device);
      if (!g.ok) return CBH_E_NODEVICE;
      hipStream_t cs = cw->stream;
      if (attempt == 0 || direct[s]) CBH_HIP(hipStreamWaitEvent(cs, ws->ev0, 0));
      if (attempt == 0) {
        if (c->device != root) {  // replicate the needles: one peer copy per shard and call (8 B per needle)
          if ((rc = Workspace::grow(&cw->d_q, &cw->q_cap, nq))) return rc;
          CBH_HIP(hipMemcpyPeerAsync(cw->d_q, c->device, d_q, root, nq * sizeof(uint64_t), cs));
          q_of[s] = cw->d_q;
          if (d_qmask) {
            if ((rc = Workspace::grow(&cw->d_qmask, &cw->qmask_cap, nq))) return rc;
            CBH_HIP(hipMemcpyPeerAsync(cw->d_qmask, c->device, d_qmask, root, nq * sizeof(uint64_t), cs));
            mask_of[s] = cw->d_qmask;
          }
          C.n_peer_copies++;
        }
        if (!direct[s] && (rc = cw->ensure_records(std::max<size_t>(c->rec_cap_default, 1024)))) return rc;
      }
      if (!direct[s]) CBH_HIP(hipMemsetAsync(cw->d_total, 0, sizeof(unsigned long long), cs));
      if (timed) CBH_HIP(hipEventRecord(cw->ev0, cs));
      rc = launch_hamm64_scan(c->d_hashes, c->d_ids, c->n, q_of[s], nq, thresh, direct[s] ? ws->d_rec : cw->d_rec,
                              direct[s] ? ws->rec_cap : cw->rec_cap, direct[s] ? ws->d_total : cw->d_total, cs,
                              flags | ((unsigned)std::min(C.per_device, 255) << SCAN_SIBLINGS_SHIFT), mask_of[s],
                              c->device == root ? qx_root : nullptr);
      if (rc) return rc;
      if (timed || direct[s]) CBH_HIP(hipEventRecord(cw->ev1, cs));
      if (!direct[s])
        CBH_HIP(hipMemcpyAsync(cw->h_total, cw->d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
      C.n_scans++;
      if (attempt) C.n_rescans++;
    }
    if (!any) break;
    float worst = 0.f;
    if (any_direct_now) {  // one wait for all of them, on the root stream
      DeviceGuard g(root);
      for (size_t s = 0; s < R; ++s)
        if (todo[s] && direct[s]) CBH_HIP(hipStreamWaitEvent(stream, L.ws[s]->ev1, 0));
      CBH_HIP(hipMemcpyAsync(ws->h_total, ws->d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
      CBH_HIP(hipStreamSynchronize(stream));
      direct_count = *ws->h_total;
      for (size_t s = 0; s < R; ++s) {
        if (!todo[s] || !direct[s]) continue;
        float ms = 0.f;
        if (timed && hipEventElapsedTime(&ms, L.ws[s]->ev0, L.ws[s]->ev1) == hipSuccess) worst = std::max(worst, ms);
        todo[s] = 0;
        L.idle[s] = 1;  // (its stream's last operation is the event the root stream waited for, and that stream is drained)
      }
    }
    for (size_t s = 0; s < R; ++s) {
      if (!todo[s]) continue;
      cbh_idx64* c = S->child[s];
      Workspace* cw = L.ws[s];
      DeviceGuard g(c->device);
      CBH_HIP(hipStreamSynchronize(cw->stream));
      count[s] = *cw->h_total;
      float ms = 0.f;
      if (timed && hipEventElapsedTime(&ms, cw->ev0, cw->ev1) == hipSuccess) worst = std::max(worst, ms);
      todo[s] = 0;
      if (count[s] > cw->rec_cap) {  // this shard alone grows its block and scans again
        unsigned long long others = direct_count;
        for (size_t o = 0; o < R; ++o)
          if (o != s) others += count[o];
        if (others + count[s] > max_records && others + count[s] > ws->rec_cap) {  // the merged result cannot fit anyway
          *total = others + count[s];
          return CBH_E_OVERFLOW;
        }
        rc = cw->ensure_records((size_t)count[s] + 1024);
        if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
        todo[s] = 1;
      }
    }
    scan_ms += worst;  // the shards run side by side: a round costs what its slowest shard costs
    bool pending = false;
    for (size_t s = 0; s < R; ++s) pending = pending || todo[s];
    if (pending) continue;
    // every shard has reported: does the root block hold the whole result?
    unsigned long long sum = direct_count;
    for (size_t s = 0; s < R; ++s) sum += count[s];
    if (sum <= ws->rec_cap) break;
    *total = sum;
    if (sum > max_records) return CBH_E_OVERFLOW;  // nobody grows for it
    {
      DeviceGuard g(root);
      CBH_HIP(hipStreamSynchronize(stream));
      rc = ws->ensure_records((size_t)sum + 1024);  // (a new block: what the root device's shards appended is gone)
      if (rc) return rc == CBH_E_NOMEM ? CBH_E_OVERFLOW : rc;
    }
    if (!any_direct) break;
    for (size_t s = 0; s < R; ++s) todo[s] = direct[s];
    direct_count = 0;
  }
  for (size_t s = 0; s < R; ++s)
    if (todo[s]) return CBH_E_OVERFLOW;
  {
    std::lock_guard<std::mutex> lk(idx->stats_mu);
    idx->stats.scan_launches += 1;
    idx->stats.scan_pairs += (uint64_t)idx->n * (uint64_t)nq;
    idx->stats.scan_ms += (double)scan_ms;
  }
  unsigned long long sum = direct_count, remote = 0;
  for (size_t s = 0; s < R; ++s) remote += count[s];
  sum += remote;
  *total = sum;
  if (sum > ws->rec_cap) return CBH_E_OVERFLOW;
  if (!remote && any_direct) {
    // all of it is in place, the counter holds the sum, and the root stream was drained when the count was read
    for (size_t s = 0; s < R; ++s) L.idle[s] = 1;
    return CBH_OK;
  }
  // ---- exchange: the other shards' records into the root block, behind what the root device's appended ----
  std::vector<ShardPart> parts(R);
  for (size_t s = 0; s < R; ++s) {
    Workspace* cw = L.ws[s];
    parts[s].dev_pos = C.dev_pos_of_shard(s);
    parts[s].stream = cw->stream;
    parts[s].d_rec = reinterpret_cast<const unsigned long long*>(cw->d_rec);
    parts[s].count = count[s];
    parts[s].own_block = cw->d_total;  // { count, records[rec_cap] }
    parts[s].own_cap = cw->rec_cap;
    parts[s].ev = cw->ev1;
    parts[s].x = cw->x;
    parts[s].h_word = cw->h_total;
  }
  bool by_collective = true;
  if ((rc = C.exchange(parts, stream, reinterpret_cast<unsigned long long*>(ws->d_rec), &by_collective, direct_count)))
    return rc;
  {
    DeviceGuard g(root);
    *ws->h_total = sum;
    CBH_HIP(hipMemcpyAsync(ws->d_total, ws->h_total, sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    CBH_HIP(hipStreamSynchronize(stream));  // scan_all's contract: the block is complete on return
  }
  // Copies as the exchange: a shard stream's last operation is its count read-back (synchronised above) or its record copy
  // + event, which `stream` waited for before the synchronisation just done -- every shard stream is drained.  (A collective
  // leaves the peers' halves of the all-gather possibly in flight: those streams are synchronised when the leases end.)
  if (!by_collective)
    for (size_t s = 0; s < R; ++s) L.idle[s] = 1;
  return CBH_OK;
}

// The lone needle on a sharded handle (cbh_idx64_find): one launch_find_one per shard, issued back to back from this thread,
// then one poll per shard -- no needle copies, no event waits, no counter resets, no read-backs, no stream synchronisation
// (what made a find() on 8 shards cost 203 us against 26 on the plain index).  *fits = every shard's matches fitted its
// LoneBlock (then *recs holds them all, unordered); otherwise the caller takes the general path.
int sharded_find_one(cbh_idx64* idx, uint64_t q, int thresh, std::vector<cbh_record>* recs, bool* fits) {
  ShardSet* S = idx->shards;
  const size_t R = S->child.size();
  *fits = false;
  recs->clear();
  ShardLeases L(S);
  int rc = L.acquire_all();
  if (rc) return rc;
  std::vector<unsigned long long> seq(R, 0);
  for (size_t s = 0; s < R; ++s) {
    cbh_idx64* c = S->child[s];
    if (c->n == 0) continue;
    DeviceGuard g(c->device);
    if (!g.ok) return CBH_E_NODEVICE;
    Workspace* cw = L.ws[s];
    if ((rc = cw->ensure_lone())) return rc;
    seq[s] = ++cw->lone_seq;
    if ((rc = launch_find_one(c->d_hashes, c->d_ids, c->n, q, thresh, cw->d_lone, cw->h_lone, seq[s], cw->stream))) return rc;
    S->comm.n_scans++;
  }
  bool all_fit = true;
  for (size_t s = 0; s < R; ++s) {
    if (!seq[s]) {
      L.idle[s] = 1;
      continue;
    }
    Workspace* cw = L.ws[s];
    DeviceGuard g(S->child[s]->device);
    if ((rc = wait_find_one(cw->h_lone, seq[s], cw->stream))) return rc;
    L.idle[s] = 1;  // its kernel has published its result: nothing of this call is left on the stream
    const unsigned long long t = cw->h_lone->count;
    if (t > LoneBlock::kRecs) all_fit = false;
    else if (all_fit) recs->insert(recs->end(), cw->h_lone->recs, cw->h_lone->recs + t);
  }
  *fits = all_fit;
  if (!all_fit) recs->clear();
  return CBH_OK;
}

int sharded_download(const cbh_idx64* idx, uint64_t* hashes, uint32_t* ids, size_t cap) {
  const ShardSet* S = idx->shards;
  for (const ShardSet::Seg& g : S->segs) {
    if (g.global >= cap) continue;
    const size_t m = std::min(g.len, cap - g.global);
    const cbh_idx64* c = S->child[g.shard];
    DeviceGuard dg(c->device);
    if (!dg.ok) return CBH_E_NODEVICE;
    if (hashes) CBH_HIP(hipMemcpy(hashes + g.global, c->d_hashes + g.local, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (ids) CBH_HIP(hipMemcpy(ids + g.global, c->d_ids + g.local, m * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return CBH_OK;
}

// load: shard s of R takes [s*n/R, (s+1)*n/R) (cbird_amd/dist.py shard_range); src on the host, or on the
// root device (load_dev)
int sharded_load(cbh_idx64* idx, const void* hashes, const void* ids, size_t n, bool on_device, hipStream_t stream) {
  ShardSet* S = idx->shards;
  const size_t R = S->child.size();
  S->segs.clear();
  idx->n = 0;
  if (n && (!hashes || !ids)) return CBH_E_INVAL;
  if (n > 0xfffffff0ull) return CBH_E_INVAL;
  if (on_device && stream) CBH_HIP(hipStreamSynchronize(stream));  // the source arrays are complete
  for (size_t s = 0; s < R; ++s) {
    const size_t a = s * n / R, b = (s + 1) * n / R;
    cbh_idx64* c = S->child[s];
    const uint64_t* h = (const uint64_t*)hashes + a;
    const uint32_t* i = (const uint32_t*)ids + a;
    int rc;
    if (!on_device) {
      rc = cbh_idx64_load(c, h, i, b - a);
    } else if (c->device == idx->device) {
      rc = cbh_idx64_load_dev(c, h, i, b - a, nullptr);
    } else {
      DeviceGuard g(c->device);
      if (!g.ok) return CBH_E_NODEVICE;
      c->n = 0;
      c->generation++;
      c->loaded = true;
      rc = c->reserve(b - a);
      if (!rc && b > a) {
        CBH_HIP(hipMemcpyPeer(c->d_hashes, c->device, h, idx->device, (b - a) * sizeof(uint64_t)));
        CBH_HIP(hipMemcpyPeer(c->d_ids, c->device, i, idx->device, (b - a) * sizeof(uint32_t)));
        c->n = b - a;
      }
    }
    if (rc) return rc;
    if (b > a) S->segs.push_back(ShardSet::Seg{(uint32_t)s, 0, a, b - a});
  }
  idx->n = n;
  return CBH_OK;
}

// add(): the reference appends and rebuilds its tree (src/dcthashindex.cpp:158-173); here the batch goes to the shard
// that holds the fewest slots, the parent's slot order continues
int sharded_add(cbh_idx64* idx, const uint64_t* hashes, const uint32_t* ids, size_t n) {
  ShardSet* S = idx->shards;
  if (n == 0) return CBH_OK;
  if (!hashes || !ids) return CBH_E_INVAL;
  if (idx->n + n > 0xfffffff0ull) return CBH_E_INVAL;
  size_t best = 0;
  for (size_t s = 1; s < S->child.size(); ++s)
    if (S->child[s]->n < S->child[best]->n) best = s;
  cbh_idx64* c = S->child[best];
  const size_t local = c->n;
  int rc = c->loaded ? cbh_idx64_add(c, hashes, ids, n) : cbh_idx64_load(c, hashes, ids, n);
  if (rc) return rc;
  if (!S->segs.empty() && S->segs.back().shard == best && S->segs.back().local + S->segs.back().len == local &&
      S->segs.back().global + S->segs.back().len == idx->n)
    S->segs.back().len += n;
  else
    S->segs.push_back(ShardSet::Seg{(uint32_t)best, local, idx->n, n});
  idx->n += n;
  return CBH_OK;
}

int sharded_remove(cbh_idx64* idx, const uint32_t* ids, size_t n, int zero_hash) {
  for (cbh_idx64* c : idx->shards->child) {
    int rc = zero_hash ? cbh_idx64_remove(c, ids, n) : cbh_idx64_remove_ids_only(c, ids, n);
    if (rc) return rc;
  }
  return CBH_OK;
}

}  // namespace cbh

extern "C" {

cbh_idx64* cbh_idx64_create_sharded(uint32_t device_mask, int shards_per_device) {
  clear_last_error();
  cbh_idx64* idx = new (std::nothrow) cbh_idx64;
  ShardSet* S = new (std::nothrow) ShardSet;
  if (!idx || !S) {
    delete idx;
    delete S;
    return (cbh_idx64*)fail_handle(CBH_E_NOMEM, "cbh_idx64_create_sharded: host allocation failed");
  }
  if (!S->comm.init(device_mask, shards_per_device)) {
    delete idx;
    delete S;
    return (cbh_idx64*)fail_handle(CBH_E_INVAL, "cbh_idx64_create_sharded: empty mask, a device of the mask is not usable, or shards_per_device out of range");
  }
  idx->device = S->comm.devices[0];
  idx->shards = S;
  const size_t R = S->comm.shard_count();
  for (size_t s = 0; s < R; ++s) {
    cbh_idx64* c = cbh_idx64_create(S->comm.device_of_shard(s));
    if (!c) {
      cbh_idx64_destroy(idx);
      return nullptr;
    }
    c->rec_cap_default = std::max<size_t>(65536, idx->rec_cap_default / R);
    S->child.push_back(c);
  }
  return idx;
}

uint32_t cbh_idx64_device_mask(const cbh_idx64* idx) {
  return !idx ? 0 : idx->shards ? idx->shards->comm.mask : (1u << idx->device);
}

int cbh_idx64_shards_per_device(const cbh_idx64* idx) { return !idx ? 0 : idx->shards ? idx->shards->comm.per_device : 1; }

int cbh_idx64_shard_count(const cbh_idx64* idx) { return !idx ? 0 : idx->shards ? (int)idx->shards->child.size() : 1; }

cbh_idx64* cbh_idx64_shard(cbh_idx64* idx, int i) {
  if (!idx) return nullptr;
  if (!idx->shards) return i == 0 ? idx : nullptr;
  if (i < 0 || (size_t)i >= idx->shards->child.size()) return nullptr;
  return idx->shards->child[(size_t)i];
}

int cbh_idx64_shard_stats(const cbh_idx64* idx, cbh_shard_stats* out) {
  if (!idx || !out) return CBH_E_INVAL;
  memset(out, 0, sizeof *out);
  if (!idx->shards) {
    out->shards = 1, out->devices = 1;
    return CBH_OK;
  }
  const ShardSet* S = idx->shards;
  out->shards = (uint32_t)S->child.size();
  out->devices = (uint32_t)S->comm.devices.size();
  out->device_mask = S->comm.mask;
  out->scans = S->comm.n_scans.load();
  out->rescans = S->comm.n_rescans.load();
  out->collectives = S->comm.n_collectives.load();
  out->peer_copies = S->comm.n_peer_copies.load();
  out->local_copies = S->comm.n_local_copies.load();
  out->collective_fallbacks = S->comm.n_fallbacks.load();
  out->segments = S->segs.size();
  return CBH_OK;
}

}  // extern "C"
