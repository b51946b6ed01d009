// What does HBM deliver for k_dcthash_256's access pattern?  n images of 64 KB; variants of who reads what.
// Build: hipcc --offload-arch=gfx950 -O3 -o read_pattern read_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// A: as k_dcthash_256 -- half-wave per image, lane owns 8 bytes of every row, PF rows in flight, 8 images per workgroup
template <int PF>
__global__ __launch_bounds__(256) void k_a(const unsigned char* __restrict__ imgs, unsigned n, uint64_t* out) {
  const unsigned img = blockIdx.x * 8 + (threadIdx.x >> 5), l32 = threadIdx.x & 31;
  if (img >= n) return;
  const unsigned char* p = imgs + (size_t)img * 65536 + l32 * 8;
  uint2 r[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) r[j] = *reinterpret_cast<const uint2*>(p + j * 256);
  unsigned acc = 0;
  for (int s0 = 0; s0 < 256; s0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      acc += r[j].x ^ r[j].y;
      const int nx = s0 + j + PF < 256 ? s0 + j + PF : 255;
      r[j] = *reinterpret_cast<const uint2*>(p + nx * 256);
    }
  }
  if (acc == 0x12345678u) out[img] = acc;
}
// B: wave per image, lane owns 16 bytes: a wave load covers 4 rows (1 KB)
template <int PF>
__global__ __launch_bounds__(256) void k_b(const unsigned char* __restrict__ imgs, unsigned n, uint64_t* out) {
  const unsigned img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (img >= n) return;
  const unsigned char* p = imgs + (size_t)img * 65536 + lane * 16;
  uint4 r[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) r[j] = *reinterpret_cast<const uint4*>(p + j * 1024);
  unsigned acc = 0;
  for (int s0 = 0; s0 < 64; s0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      acc += r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
      const int nx = s0 + j + PF < 64 ? s0 + j + PF : 63;
      r[j] = *reinterpret_cast<const uint4*>(p + nx * 1024);
    }
  }
  if (acc == 0x12345678u) out[img] = acc;
}
// C: plain grid-stride streaming read of the whole buffer, 16 B per lane (the ceiling)
__global__ __launch_bounds__(256) void k_c(const uint4* __restrict__ p, size_t n16, uint64_t* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const unsigned n = argc > 1 ? atoi(argv[1]) : 400000;
  unsigned char* d;
  uint64_t* o;
  hipMalloc(&d, (size_t)n * 65536);
  hipMalloc(&o, (size_t)n * 8);
  hipMemset(d, 1, (size_t)n * 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, best, (double)n * 65536 / best * 1e-6);
  };
  run("A half-wave/image 8B/lane PF=7 (kernel's)", [&] { hipLaunchKernelGGL(k_a<7>, dim3((n + 7) / 8), dim3(256), 0, 0, d, n, o); });
  run("A half-wave/image 8B/lane PF=14", [&] { hipLaunchKernelGGL(k_a<14>, dim3((n + 7) / 8), dim3(256), 0, 0, d, n, o); });
  run("A half-wave/image 8B/lane PF=4", [&] { hipLaunchKernelGGL(k_a<4>, dim3((n + 7) / 8), dim3(256), 0, 0, d, n, o); });
  run("B wave/image 16B/lane PF=4", [&] { hipLaunchKernelGGL(k_b<4>, dim3((n + 3) / 4), dim3(256), 0, 0, d, n, o); });
  run("B wave/image 16B/lane PF=8", [&] { hipLaunchKernelGGL(k_b<8>, dim3((n + 3) / 4), dim3(256), 0, 0, d, n, o); });
  run("C grid-stride 16B/lane, 256x8 blocks", [&] { hipLaunchKernelGGL(k_c, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)d, (size_t)n * 4096, o); });
  run("C grid-stride 16B/lane, 256x32 blocks", [&] { hipLaunchKernelGGL(k_c, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)d, (size_t)n * 4096, o); });
  return 0;
}
