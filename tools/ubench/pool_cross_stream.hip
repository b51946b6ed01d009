// Does ROCm's stream-ordered allocator hand a block that stream A freed (hipFreeAsync behind A's last kernel) to
// stream B while A's kernels are still running?  Round 2 blamed exactly that for wrong match counts in the pipelined
// threshold sweep and worked around it with one hipMemPool_t per stream (cbh_internal.h) without a minimal
// reproducer.  This is the reproducer: nothing but hipMallocAsync / kernels / hipFreeAsync on S streams, issued the
// way the library issues them (one host thread walking the streams round robin, or one host thread per stream), with
// a long-running owner check inside every allocation's lifetime:
//
//     p = hipMallocAsync(size, s);  fill<<<s>>>(p, tag);  spin<<<s>>>(~50 us);  verify<<<s>>>(p, tag);  hipFreeAsync(p, s)
//
// A block reused across streams before its owner is done shows up as a foreign tag in verify.  Modes:
//   pool 0  the device's default pool (what plain hipMallocAsync uses)
//   pool 1  ... with ReleaseThreshold = UINT64_MAX (what the library set on it: keep_pool_memory)
//   pool 2  one explicit pool per stream (the round-2 workaround)
//   pool 3  control: no allocator, one hipMalloc'ed buffer per stream
// Build: hipcc --offload-arch=gfx950 -O2 -pthread -o pool_cross_stream pool_cross_stream.hip
// Run:   pool_cross_stream [streams=4] [iterations=4000] [host threads: 0 = one for all, 1 = one per stream]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__global__ void k_fill(uint32_t* p, size_t n, uint32_t tag) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag;
}
__global__ void k_spin(long long cycles, unsigned* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {
  }
  if (cycles < 0) *sink = 1;
}
__global__ void k_verify(const uint32_t* p, size_t n, uint32_t tag, unsigned long long* errors, uint32_t* first_bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != tag) {
      if (atomicAdd(errors, 1ull) == 0) {
        first_bad[0] = tag;
        first_bad[1] = p[i];
      }
    }
}

struct Ctx {
  int pool_mode;
  std::vector<hipStream_t> streams;
  std::vector<hipMemPool_t> pools;
  unsigned long long* d_err;
  uint32_t* d_bad;
  unsigned* d_sink;
  std::vector<uint32_t*> fixed;
};

static void one_alloc_cycle(Ctx& c, int s, int it, uint32_t& rng) {
  static const size_t sizes[] = {4096, 65536, 1 << 20, 3 << 20, 24 << 20};
  rng = rng * 1664525u + 1013904223u;
  const size_t bytes = sizes[(rng >> 24) % 5];
  const size_t n = bytes / 4;
  const uint32_t tag = ((uint32_t)s << 24) | ((uint32_t)it & 0xffffff);
  uint32_t* p = nullptr;
  hipStream_t st = c.streams[s];
  if (c.pool_mode == 3)
    p = c.fixed[s];  // control: no allocator at all, one hipMalloc'ed buffer per stream (validates this harness)
  else if (c.pool_mode == 2)
    CK(hipMallocFromPoolAsync((void**)&p, bytes, c.pools[s], st));
  else
    CK(hipMallocAsync((void**)&p, bytes, st));
  const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
  hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st, p, n, tag);
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, (long long)(5000 + (rng >> 8) % 5000), c.d_sink);  // 100 MHz clock
  hipLaunchKernelGGL(k_verify, dim3(grid), dim3(256), 0, st, p, n, tag, c.d_err, c.d_bad);
  CK(hipGetLastError());
  if (c.pool_mode != 3) CK(hipFreeAsync(p, st));
}

static unsigned long long run(int pool_mode, int S, int iters, int per_stream_threads) {
  Ctx c;
  c.pool_mode = pool_mode;
  c.streams.resize(S);
  c.pools.assign(S, nullptr);
  for (int s = 0; s < S; ++s) CK(hipStreamCreateWithFlags(&c.streams[s], hipStreamNonBlocking));
  hipMemPool_t def = nullptr;
  CK(hipDeviceGetDefaultMemPool(&def, 0));
  uint64_t thr = pool_mode == 1 ? ~0ull : 0ull;
  CK(hipMemPoolSetAttribute(def, hipMemPoolAttrReleaseThreshold, &thr));
  if (pool_mode == 2)
    for (int s = 0; s < S; ++s) {
      hipMemPoolProps props = {};
      props.allocType = hipMemAllocationTypePinned;
      props.handleTypes = hipMemHandleTypeNone;
      props.location.type = hipMemLocationTypeDevice;
      props.location.id = 0;
      CK(hipMemPoolCreate(&c.pools[s], &props));
      uint64_t keep = ~0ull;
      CK(hipMemPoolSetAttribute(c.pools[s], hipMemPoolAttrReleaseThreshold, &keep));
    }
  c.fixed.assign(S, nullptr);
  if (pool_mode == 3)
    for (int s = 0; s < S; ++s) CK(hipMalloc(&c.fixed[s], 24 << 20));
  CK(hipMalloc(&c.d_err, 8));
  CK(hipMalloc(&c.d_bad, 8));
  CK(hipMalloc(&c.d_sink, 4));
  CK(hipMemset(c.d_err, 0, 8));
  CK(hipMemset(c.d_bad, 0, 8));
  if (per_stream_threads) {
    std::vector<std::thread> th;
    for (int s = 0; s < S; ++s)
      th.emplace_back([&, s] {
        CK(hipSetDevice(0));
        uint32_t rng = 12345u + 977u * (uint32_t)s;
        for (int it = 0; it < iters; ++it) one_alloc_cycle(c, s, it, rng);
      });
    for (auto& t : th) t.join();
  } else {
    uint32_t rng = 424242u;
    for (int it = 0; it < iters; ++it)
      for (int s = 0; s < S; ++s) one_alloc_cycle(c, s, it, rng);
  }
  CK(hipDeviceSynchronize());
  unsigned long long err = 0;
  uint32_t bad[2] = {0, 0};
  CK(hipMemcpy(&err, c.d_err, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(bad, c.d_bad, 8, hipMemcpyDeviceToHost));
  uint64_t reserved = 0, used = 0;
  (void)hipMemPoolGetAttribute(def, hipMemPoolAttrReservedMemHigh, &reserved);
  (void)hipMemPoolGetAttribute(def, hipMemPoolAttrUsedMemHigh, &used);
  printf("{\"pool_mode\": %d, \"streams\": %d, \"iterations\": %d, \"host_threads\": %d, \"foreign_words\": %llu, "
         "\"first_bad\": [\"0x%08x\", \"0x%08x\"], \"default_pool_reserved_high\": %llu, \"default_pool_used_high\": %llu}\n",
         pool_mode, S, iters, per_stream_threads ? S : 1, err, bad[0], bad[1], (unsigned long long)reserved,
         (unsigned long long)used);
  for (int s = 0; s < S; ++s) CK(hipStreamDestroy(c.streams[s]));
  for (int s = 0; s < S; ++s)
    if (c.pools[s]) CK(hipMemPoolDestroy(c.pools[s]));
  for (int s = 0; s < S; ++s)
    if (c.fixed[s]) CK(hipFree(c.fixed[s]));
  CK(hipFree(c.d_err));
  CK(hipFree(c.d_bad));
  CK(hipFree(c.d_sink));
  return err;
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 4;
  const int iters = argc > 2 ? atoi(argv[2]) : 4000;
  const int threads = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipSetDevice(0));
  unsigned long long total = 0;
  for (int mode = 0; mode < 4; ++mode) total += run(mode, S, iters, threads);
  return total ? 1 : 0;
}
