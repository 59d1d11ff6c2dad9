// pcie_duplex_probe: what a host-to-device transfer of one coalesced endpoint write (3 MiB) gets while a scatter kernel
// is storing a drain into pinned host memory -- the two directions of a stream through the endpoint vtable.  The
// host-to-device side as (a) a kernel that loads from pinned host memory (what k_copy does with the send buffers) and
// (b) hipMemcpyAsync (a copy engine); the device-to-host side as a kernel with G workgroups (what k_rx_apply does with the
// receive window), or as hipMemcpyAsync.  Prints the time of the H2D transfer alone and under each kind of D2H load.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/pcie_duplex_probe tools/pcie_duplex_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("error: %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid-strided 16-byte copy, 16 loads in flight per lane (a 16 KiB tile per wave, like the product's copy kernels)
__global__ __launch_bounds__(256) void k_move(u32x4* dst, const u32x4* src, size_t units) {
  const size_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t t = wave * 1024; t < units; t += nwaves * 1024) {
    u32x4 a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const size_t u = t + lane + 64 * k;
      a[k] = u < units ? __builtin_nontemporal_load(src + u) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const size_t u = t + lane + 64 * k;
      if (u < units) dst[u] = a[k];
    }
  }
}

// the same towards the host with other store shapes: U 16-byte units per lane in flight, then wait for the acknowledgements
// (U = 16 is k_move); NT: non-temporal stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_store_paced(u32x4* dst, const u32x4* src, size_t units) {
  const size_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t t = wave * 64 * U; t < units; t += nwaves * 64 * U) {
    u32x4 a[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const size_t u = t + lane + 64 * k;
      a[k] = u < units ? __builtin_nontemporal_load(src + u) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      const size_t u = t + lane + 64 * k;
      if (u < units) {
        if (NT) __builtin_nontemporal_store(a[k], dst + u);
        else dst[u] = a[k];
      }
    }
    __builtin_amdgcn_s_waitcnt(0);   // every store of the tile acknowledged before the next tile's
  }
}

int main() {
  const size_t H2D = 3u << 20, D2H = 64u << 20;
  uint8_t *h_src, *h_dst, *d_a, *d_b;
  CK(hipHostMalloc((void**)&h_src, H2D, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc((void**)&h_dst, D2H, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipMalloc((void**)&d_a, H2D));
  CK(hipMalloc((void**)&d_b, D2H));
  CK(hipMemset(d_b, 3, D2H));
  hipStream_t s_up, s_down;
  CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  // D2H load kinds: 0 none, > 0 kernel with that many workgroups, -1 hipMemcpyAsync
  auto start_load = [&](int kind) {
    if (kind >= 100000) {  // paced variants: 100000 + variant * 10000 + workgroups
      const int v = (kind - 100000) / 10000, wg = kind % 10000;
      u32x4* d = (u32x4*)h_dst; const u32x4* sr = (const u32x4*)d_b; const size_t n = D2H / 16;
      if (v == 0) hipLaunchKernelGGL((k_store_paced<1, false>), dim3(wg), dim3(256), 0, s_down, d, sr, n);
      else if (v == 1) hipLaunchKernelGGL((k_store_paced<4, false>), dim3(wg), dim3(256), 0, s_down, d, sr, n);
      else if (v == 2) hipLaunchKernelGGL((k_store_paced<16, true>), dim3(wg), dim3(256), 0, s_down, d, sr, n);
      else hipLaunchKernelGGL((k_store_paced<1, true>), dim3(wg), dim3(256), 0, s_down, d, sr, n);
    } else if (kind > 0) hipLaunchKernelGGL(k_move, dim3(kind), dim3(256), 0, s_down, (u32x4*)h_dst, (const u32x4*)d_b, D2H / 16);
    else if (kind < 0) CK(hipMemcpyAsync(h_dst, d_b, D2H, hipMemcpyDeviceToHost, s_down));
  };
  auto h2d = [&](int how, int blocks) {
    if (how == 0) hipLaunchKernelGGL(k_move, dim3(blocks), dim3(256), 0, s_up, (u32x4*)d_a, (const u32x4*)h_src, H2D / 16);
    else CK(hipMemcpyAsync(d_a, h_src, H2D, hipMemcpyHostToDevice, s_up));
  };
  auto measure = [&](const char* what, int how, int blocks, int load_kind) {
    double best = 1e9, sum = 0, dsum = 0;
    const int reps = 6;
    for (int r = 0; r < reps + 1; r++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, s_down));
      start_load(load_kind);
      CK(hipEventRecord(f1, s_down));
      // (let the load get going: ~100 us of host time)
      if (load_kind) for (volatile int spin = 0; spin < 30000; spin++) {}
      CK(hipEventRecord(e0, s_up));
      for (int k = 0; k < 4; k++) h2d(how, blocks);   // four transfers back to back: 12 MiB
      CK(hipEventRecord(e1, s_up));
      CK(hipDeviceSynchronize());
      float ms = 0, dms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipEventElapsedTime(&dms, f0, f1));
      if (r == 0) continue;
      const double us = 1e3 * ms / 4;
      if (us < best) best = us;
      sum += us;
      dsum += dms;
    }
    printf("%-44s H2D 3 MiB: %7.1f us mean (%6.1f best) = %5.1f GB/s", what, sum / reps, best, H2D / (sum / reps) / 1e3);
    if (load_kind) printf("   | D2H 64 MiB %7.1f us = %5.1f GB/s", 1e3 * dsum / reps, D2H / (1e3 * dsum / reps) / 1e3);
    printf("\n");
  };
  char name[128];
  for (int how = 0; how < 2; how++) {
    const int blocks = 64;
    for (int load : {0, 8, 32, 128, 768, -1}) {
      snprintf(name, sizeof(name), "H2D by %s, D2H load: %s%d", how ? "copy engine" : "kernel (64 wg)",
               load == 0 ? "none " : load < 0 ? "copy engine " : "kernel, workgroups ", load < 0 ? 0 : load);
      measure(name, how, blocks, load);
    }
  }
  {
    const char* vn[4] = {"1 unit/lane, acked", "4 units/lane, acked", "16 units nt, acked", "1 unit nt, acked"};
    for (int v = 0; v < 4; v++)
      for (int wg : {2, 4, 8, 32}) {
        snprintf(name, sizeof(name), "H2D kernel; D2H %s, %d wg", vn[v], wg);
        measure(name, 0, 64, 100000 + v * 10000 + wg);
      }
  }
  for (int blocks : {4, 16, 256, 768}) {
    snprintf(name, sizeof(name), "H2D by kernel (%d wg), no load", blocks);
    measure(name, 0, blocks, 0);
  }
  return 0;
}
