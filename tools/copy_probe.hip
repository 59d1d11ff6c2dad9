// copy_probe: the scatter kernel's tile machinery (csrc/grdma_devfn.h) on a synthetic steady-state drain -- per HTTP/2
// frame a 9-byte record and a 16384-byte record in the ring, delivered as a 256-byte slice (9 + 247 bytes) and a
// 16137-byte slice, the ring cleared behind -- outside the pipeline: what grid size, tile-to-wave mapping and number of
// tiles in flight per wave do to the time of ONE launch, next to hipMemcpyDtoD of the same bytes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I grpc-rdma_amd/csrc -o tools/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "grdma_devfn.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("error: %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// V0: the product's kernel body (k_rx_apply without the credit epilogue)
__global__ __launch_bounds__(COPY_THREADS) void k_v0(const grdma_plan* plan) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  run_plan<256, true>(plan, wave, nwaves, lane);
}

// V1: a wave takes G CONSECUTIVE segments (one tile each) and has the loads of all of them in flight before the first
// store: small segments (<= 1 KiB) through one 16-byte unit per lane, large ones through sixteen.
template <int AUX_LD, int U>
struct tile_regs {
  u32x4 a[U + 1];
  uint8_t hb, tb;
  uint32_t head, units, tail, shift, n;
  uint64_t dst, src;
};
template <int AUX_LD, int U>
__device__ __forceinline__ void tile_load(tile_regs<AUX_LD, U>& R, uint64_t dst, uint64_t src, uint32_t n, int lane) {
  uint32_t head = (uint32_t)((16 - (dst & 15)) & 15);
  if (head > n) head = n;
  const uint32_t n2 = n - head;
  R.head = head;
  R.units = n2 >> 4;
  R.tail = n2 & 15;
  R.n = n;
  R.dst = dst;
  R.src = src;
  const uint64_t s2 = src + head;
  R.shift = (uint32_t)(s2 & 15);
  const uint32_t nblk = R.units ? R.units + (R.shift ? 1u : 0u) : 0u;
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(uni64(s2 & ~15ull), uni32(nblk * 16));
  const __amdgpu_buffer_rsrc_t rsb = mk_rsrc(uni64(src), uni32(n));
#pragma unroll
  for (int k = 0; k < U; k++) R.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * k) * 16, 0, AUX_LD);
  R.a[U] = __builtin_amdgcn_raw_buffer_load_b128(rs, 64 * U * 16, 0, AUX_LD);
  const uint32_t tail_off = head + (R.units << 4);
  R.hb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < head ? lane : n, 0, AUX_LD);
  R.tb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < R.tail ? tail_off + lane : n, 0, AUX_LD);
}
template <int AUX_LD, int AUX_ST, int U>
__device__ __forceinline__ void tile_store(tile_regs<AUX_LD, U>& R, int lane) {
  const uint64_t d2 = R.dst + R.head;
  const __amdgpu_buffer_rsrc_t rd = mk_rsrc(uni64(d2), uni32(R.units * 16));
  const __amdgpu_buffer_rsrc_t rdb = mk_rsrc(uni64(R.dst), uni32(R.n));
  const __amdgpu_buffer_rsrc_t rsb = mk_rsrc(uni64(R.src), uni32(R.n));
  const uint32_t tail_off = R.head + (R.units << 4);
  if (R.shift == 0) {
#pragma unroll
    for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(R.a[k], rd, (lane + 64 * k) * 16, 0, AUX_ST);
  } else {
    u32x4 r_cur = dpp_rol1(R.a[0]);
#pragma unroll
    for (int k = 0; k < U; k++) {
      const u32x4 r_next = dpp_rol1(R.a[k + 1]);
      const u32x4 b = lane == 63 ? r_next : r_cur;
      __builtin_amdgcn_raw_buffer_store_b128(funnel16(R.a[k], b, R.shift), rd, (lane + 64 * k) * 16, 0, AUX_ST);
      r_cur = r_next;
    }
  }
  __builtin_amdgcn_raw_buffer_store_b8(R.hb, rdb, (uint32_t)lane < R.head ? lane : R.n, 0, AUX_ST);
  __builtin_amdgcn_raw_buffer_store_b8(R.tb, rdb, (uint32_t)lane < R.tail ? tail_off + lane : R.n, 0, AUX_ST);
  // zero behind
  const uint64_t zs = (R.src + 15) & ~15ull, ze = (R.src + R.n) & ~15ull;
  if (ze > zs) {
    const uint32_t zu = (uint32_t)((ze - zs) >> 4);
    const __amdgpu_buffer_rsrc_t rz = mk_rsrc(uni64(zs), uni32(zu * 16));
#pragma unroll
    for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rz, (lane + 64 * k) * 16, 0, AUX_ST);
    const uint32_t e0 = (uint32_t)(zs - R.src), e1 = (uint32_t)(R.src + R.n - ze);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e0 ? lane : R.n, 0, AUX_ST);
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e1 ? (uint32_t)(ze - R.src) + lane : R.n, 0, AUX_ST);
  } else {
    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, lane, 0, AUX_ST);
  }
}

// a wave takes three consecutive segments: two small ones and a large one in any order are the steady state; anything
// else (two large ones in a group) falls back to one after the other
__global__ __launch_bounds__(COPY_THREADS) void k_v1(const grdma_plan* plan) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  const uint32_t nsegs = plan->nsegs;
  for (uint32_t g0 = wave * 3; g0 < nsegs; g0 += nwaves * 3) {
    grdma_seg sg[3];
#pragma unroll
    for (int i = 0; i < 3; i++) sg[i] = plan->segs[g0 + i < nsegs ? g0 + i : nsegs - 1];
    const uint32_t cnt = nsegs - g0 < 3 ? nsegs - g0 : 3;
    // classify (wave-uniform)
    int big = -1, nbig = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
      if ((uint32_t)i < cnt && sg[i].len > 1024) { big = i; nbig++; }
    if (nbig <= 1) {
      tile_regs<2, 1> S0, S1;
      tile_regs<2, 16> B;
      int s0 = -1, s1 = -1;
#pragma unroll
      for (int i = 0; i < 3; i++)
        if ((uint32_t)i < cnt && i != big) { if (s0 < 0) s0 = i; else s1 = i; }
      if (big >= 0) tile_load<2, 16>(B, sg[big].dst, sg[big].src, (uint32_t)sg[big].len, lane);
      if (s0 >= 0) tile_load<2, 1>(S0, sg[s0].dst, sg[s0].src, (uint32_t)sg[s0].len, lane);
      if (s1 >= 0) tile_load<2, 1>(S1, sg[s1].dst, sg[s1].src, (uint32_t)sg[s1].len, lane);
      if (big >= 0) tile_store<2, 0, 16>(B, lane);
      if (s0 >= 0) tile_store<2, 0, 1>(S0, lane);
      if (s1 >= 0) tile_store<2, 0, 1>(S1, lane);
    } else {
      for (uint32_t i = 0; i < cnt; i++)
        wave_move_tile<2, 0, true, 16>(sg[i].dst, sg[i].src, (uint32_t)sg[i].len, lane);
    }
  }
}

// ideal: wave w copies bytes [w * 16 KiB, ...) of a contiguous range and clears the source (no plan at all)
__global__ __launch_bounds__(COPY_THREADS) void k_ideal(uint64_t dst, uint64_t src, uint64_t n) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint64_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  for (uint64_t o = wave * 16384; o < n; o += nwaves * 16384) {
    const uint64_t m = n - o < 16384 ? n - o : 16384;
    wave_move_tile<2, 0, true, 16>(dst + o, src + o, (uint32_t)m, lane);
  }
}

// ideal with other cache policies for the stores (AUX_ST: 0 default, 2 nt, 16 sc1) and tile sizes
template <int AUX_LD, int AUX_ST, int U>
__global__ __launch_bounds__(COPY_THREADS) void k_ideal_v(uint64_t dst, uint64_t src, uint64_t n) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint64_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  const uint64_t T = 1024ull * U;
  for (uint64_t o = wave * T; o < n; o += nwaves * T) {
    const uint64_t m = n - o < T ? n - o : T;
    wave_move_tile<AUX_LD, AUX_ST, true, U>(dst + o, src + o, (uint32_t)m, lane);
  }
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 1820;
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  const uint64_t R = 256ull << 20;
  uint8_t *ring, *arena;
  grdma_plan* d_plan;
  CK(hipMalloc((void**)&ring, R));
  CK(hipMalloc((void**)&arena, R));
  CK(hipMalloc((void**)&d_plan, sizeof(grdma_plan)));
  CK(hipMemset(ring, 1, R));
  // W windows of ring and arena, one per launch in turn: 3 x (79 + 79 MB) does not fit the 256 MiB Infinity Cache, so the
  // launches are fed from HBM as in the pipeline (COPY_PROBE_WINDOWS=1: one window, cache-resident after the first launch)
  const int W = getenv("COPY_PROBE_WINDOWS") ? atoi(getenv("COPY_PROBE_WINDOWS")) : 3;
  const uint64_t stride = 85ull << 20;
  grdma_plan* d_plans[3];
  d_plans[0] = d_plan;
  for (int k = 1; k < W; k++) CK(hipMalloc((void**)&d_plans[k], sizeof(grdma_plan)));
  std::vector<uint8_t> hp(sizeof(grdma_plan));
  grdma_plan* P = reinterpret_cast<grdma_plan*>(hp.data());
  uint64_t total = 0;
  uint32_t ns = 0;
  for (int win = 0; win < W; win++) {
  memset(P, 0, sizeof(grdma_plan));
  uint64_t x = 4096 + win * stride, o = win * stride;
  total = 0;
  ns = 0;
  auto seg = [&](uint64_t d, uint64_t s, uint64_t len) {
    P->segs[ns].dst = (uint64_t)arena + d;
    P->segs[ns].src = (uint64_t)ring + s;
    P->segs[ns].len = len;
    P->segs[ns].flags = GRDMA_SEG_ZERO_SRC;
    P->tile_prefix[ns] = ns;
    ns++;
    total += len;
  };
  for (int f = 0; f < frames; f++) {
    // record A: 8 + 9 + 7 + 8 = 32 bytes; record B: 8 + 16384 + 8
    const uint64_t a_pay = x + 8, b_pay = x + 32 + 8;
    seg(o, a_pay, 9);
    seg(o + 9, b_pay, 247);
    o += 256;
    seg(o, b_pay + 247, 16384 - 247);
    o += 16384 - 247;
    o = (o + 15) & ~15ull;  // (the next read's slice starts at a 16-byte boundary of the arena)
    x += 32 + 16400;
  }
  P->nsegs = ns;
  P->ntiles = ns;
  P->tile_prefix[ns] = ns;
  P->bytes = total;
  P->tile_bytes = 16384;
  P->tag_base = (uint64_t)ring;
  P->tag_mask = ~0ull;
  CK(hipMemcpy(d_plans[win], P, sizeof(grdma_plan), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double alg = 3.0 * (double)total;  // read + write + clear
  printf("frames %d segments %u payload %.2f MB algorithmic %.2f MB (read + write + clear)\n", frames, ns, total / 1e6, alg / 1e6);
  auto timeit = [&](const char* name, int blocks, auto&& launch) {
    for (int i = 0; i < 10; i++) launch(i % W);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch(i % W);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("%-12s blocks %5d  %7.2f us per launch  %6.2f TB/s\n", name, blocks, us, alg / us / 1e6);
  };
  for (int blocks : {512, 768, 1024, 1280, 1366, 1536, 2048, 2730, 4096})
    timeit("v0", blocks, [&](int w) { hipLaunchKernelGGL(k_v0, dim3(blocks), dim3(COPY_THREADS), 0, 0, (const grdma_plan*)d_plans[w]); });
  for (int blocks : {256, 455, 512, 768, 1024, 1366})
    timeit("v1", blocks, [&](int w) { hipLaunchKernelGGL(k_v1, dim3(blocks), dim3(COPY_THREADS), 0, 0, (const grdma_plan*)d_plans[w]); });
  for (int blocks : {256, 455, 512, 768, 1024, 2048})
    timeit("ideal", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
  // where the tiles' edges fall in the destination: +8 -- a record's payload behind its 8-byte header: every 16 KiB tile
  // starts and ends inside a line and moves its first and last 8 bytes as byte stores; +64 / +16: 16-byte units only, but
  // the 128-byte lines at the tile edges are shared by two waves; src + 8: the realigning path (DPP) on aligned stores
  for (int blocks : {768, 1024}) {
    timeit("ideal dst+8", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride + 8, (uint64_t)ring + w * stride, total); });
    timeit("ideal dst+16", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride + 16, (uint64_t)ring + w * stride, total); });
    timeit("ideal dst+64", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride + 64, (uint64_t)ring + w * stride, total); });
    timeit("ideal src+8", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride + 8, total); });
    timeit("ideal both+8", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride + 8, (uint64_t)ring + w * stride + 8, total); });
    timeit("ideal aligned", blocks, [&](int w) { hipLaunchKernelGGL(k_ideal, dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
  }
  for (int blocks : {512, 1024, 2048}) {
    timeit("ideal st=nt", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<2, 2, 16>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
    timeit("ideal ld=0", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<0, 0, 16>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
    timeit("ideal sc1", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<2, 16, 16>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
    timeit("ideal 8K", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<2, 0, 8>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
    timeit("ideal 4K", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<2, 0, 4>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
    timeit("ideal 8Knt", blocks, [&](int w) { hipLaunchKernelGGL((k_ideal_v<2, 2, 8>), dim3(blocks), dim3(COPY_THREADS), 0, 0, (uint64_t)arena + w * stride, (uint64_t)ring + w * stride, total); });
  }
  {
    for (int i = 0; i < 5; i++) CK(hipMemcpyAsync(arena, ring, total, hipMemcpyDeviceToDevice, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) CK(hipMemcpyAsync(arena, ring, total, hipMemcpyDeviceToDevice, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("memcpyDtoD  %7.2f us per copy of %.2f MB  %6.2f TB/s (read + write)\n", us, total / 1e6, 2.0 * total / us / 1e6);
  }
  return 0;
}
