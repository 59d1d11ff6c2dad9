// Device-side helpers shared by the kernels (inline, header-only).
#ifndef GRDMA_DEVFN_H
#define GRDMA_DEVFN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"

#define PLAN_THREADS 256
#define COPY_THREADS 256

__device__ __forceinline__ uint64_t round_up8(uint64_t v) { return (v + 7ull) & ~7ull; }
__device__ __forceinline__ uint64_t round_down8(uint64_t v) { return v & ~7ull; }
__device__ __forceinline__ uint64_t enc_size(uint64_t pay) { return 16ull + round_up8(pay); }
// CalculateWritableSize, ring_buffer.h:185-189
__device__ __forceinline__ uint64_t writable_of(uint64_t space) {
  return space > GRDMA_RESERVED ? round_down8(space - GRDMA_RESERVED) : 0ull;
}
__device__ __forceinline__ uint64_t sat_sub(uint64_t a, uint64_t b) { return a > b ? a - b : 0ull; }

// Tag words are written by another agent (the wire kernel of a peer, a NIC):
// relaxed agent-scope atomic loads bypass the per-CU L1 (never refreshed by other
// writers) and are served by L2 / memory.  A ring registered for NIC writes must
// be allocated uncached (fine-grained), where the same load reaches memory.
__device__ __forceinline__ uint64_t ld_tag(const uint8_t* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Inclusive wave64 prefix sum of a 32-bit value on the DPP network (row shifts
// 1,2,4,8, then row_bcast:15 / row_bcast:31 across the four 16-lane rows): six
// VALU instructions, no LDS traffic.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

// Exclusive scan of one value per thread over a 256-thread block; returns the
// exclusive prefix and the block total (via *total).
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* wave_sums,
                                                    uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t incl = wave_incl_scan(v, lane);
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PLAN_THREADS / 64; w++) {
    uint64_t s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}


#endif  // GRDMA_DEVFN_H
