// Device-side helpers shared by the kernels (inline, header-only).
#ifndef GRDMA_DEVFN_H
#define GRDMA_DEVFN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"

#define PLAN_THREADS 256
#define COPY_THREADS 256

// Every vector-memory operation this wave has issued is acknowledged (loads returned, stores accepted by
// the memory system): what orders "the bytes" before "the word that publishes them" inside one wave.
// (A macro so that the host emulation of tests/cc/wave_emu.h can say what it means there.)
#ifndef GRDMA_WAIT_VMEM
#define GRDMA_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
// ... and every LOAD of the wave has returned as well, scalar ones included
#ifndef GRDMA_WAIT_LOADS
#define GRDMA_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
// A point every lane of the wave reaches before any lane goes on.  Nothing on the GPU, where a wave's
// lanes execute an instruction together; the host emulation (one coroutine per lane) meets here, so that a
// value one lane stores behind this point is not seen by a lane that has yet to load it in front of it.
#ifndef GRDMA_WAVE_CONVERGE
#define GRDMA_WAVE_CONVERGE()
#endif
// Lane l of a wave loads 64-bit word l of a 512-byte block another agent writes line by line (payload words, then
// the line's stamp in its last word).  On the GPU the lanes' loads of one 64-byte line are one request and the line
// arrives as a unit, so a lane that sees the stamp has neighbours that see that line's payload.  (The host emulation
// loads word by word and reads a line's stamp before its payload to keep that property.)
#ifndef GRDMA_WAVE_LOAD_LINES
#define GRDMA_WAVE_LOAD_LINES(base, lane) __hip_atomic_load(&(base)[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#endif


// Profiling stamps of the latency paths (bit 16 of grdma_tx_op::inline_copy / grdma_rx_op::inline_apply, bit 2 of the
// resident kernels' flags; host: GRDMA_PROFILE_TICKS=1).  Off by default: the clocks are scalar-memory reads that
// return out of order with LDS traffic -- a wait for an LDS word behind one waits for the clock too -- and a dozen of
// them cost the 64-byte round trip two microseconds (profiles/r05_rtt_notes.txt).
__device__ __forceinline__ uint64_t prof_time(bool on) { return on ? __builtin_amdgcn_s_memtime() : 0ull; }
__device__ __forceinline__ uint64_t prof_realtime(bool on) { return on ? __builtin_amdgcn_s_memrealtime() : 0ull; }
#define GRDMA_OP_PROFILE 16u

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

// ---- segment-list byte mover: tiles, funnel-shift realignment, zero-behind ----
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 funnel16(u32x4 a, u32x4 b, unsigned shift) {
  // bytes [shift, shift+16) of the 32-byte little-endian concatenation a|b
  uint64_t q0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
  uint64_t q1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
  uint64_t q2 = (uint64_t)b.x | ((uint64_t)b.y << 32);
  uint64_t q3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
  if (shift & 8) {
    q0 = q1;
    q1 = q2;
    q2 = q3;
  }
  unsigned s = (shift & 7) * 8;
  uint64_t o0 = q0, o1 = q1;
  if (s) {
    o0 = (q0 >> s) | (q1 << (64 - s));
    o1 = (q1 >> s) | (q2 << (64 - s));
  }
  u32x4 o;
  o.x = (uint32_t)o0;
  o.y = (uint32_t)(o0 >> 32);
  o.z = (uint32_t)o1;
  o.w = (uint32_t)(o1 >> 32);
  return o;
}

// One wave moves n (<= GRDMA_TILE_BYTES) bytes src -> dst, any alignment.
__device__ __forceinline__ void wave_copy_tile(uint8_t* dst, const uint8_t* src, uint64_t n,
                                               int lane) {
  uint64_t head = (16 - ((uint64_t)dst & 15)) & 15;
  if (head > n) head = n;
  if ((uint64_t)lane < head) dst[lane] = src ? src[lane] : 0;
  dst += head;
  if (src) src += head;
  n -= head;
  const uint64_t units = n >> 4;
  const unsigned shift = (unsigned)((uint64_t)src & 15);
  const u32x4* sa = reinterpret_cast<const u32x4*>((uint64_t)src & ~15ull);
  u32x4* da = reinterpret_cast<u32x4*>(dst);
  if (src == nullptr) {
    for (uint64_t u = lane; u < units; u += 64) da[u] = u32x4{0, 0, 0, 0};
  } else if (shift == 0) {
#pragma unroll 4
    for (uint64_t u = lane; u < units; u += 64) da[u] = __builtin_nontemporal_load(sa + u);
  } else {
#pragma unroll 4
    for (uint64_t u = lane; u < units; u += 64) {
      u32x4 a = __builtin_nontemporal_load(sa + u);
      u32x4 b = __builtin_nontemporal_load(sa + u + 1);
      da[u] = funnel16(a, b, shift);
    }
  }
  const uint64_t tail = n & 15;
  if ((uint64_t)lane < tail) {
    uint64_t o = (units << 4) + lane;
    dst[o] = src ? src[o] : 0;
  }
}

// Global-memory pointers: loads and stores through these are global_* instructions the
// compiler may reorder freely against LDS traffic (a generic pointer could point into LDS).
typedef const __attribute__((address_space(1))) u32x4* g_u32x4_c;
typedef __attribute__((address_space(1))) u32x4* g_u32x4;

// wave_copy_tile for tiles whose source and destination are known to be global memory
// (HBM or mapped host memory), n <= GRDMA_TILE_BYTES.  All loads of the tile -- at most
// four 16-byte units per lane -- are issued before the first store: a load -> wait ->
// store -> load chain costs one memory round trip per unit.
__device__ __forceinline__ void wave_copy_tile_g(uint8_t* dst, const uint8_t* src, uint64_t n,
                                                 int lane) {
  if (n > GRDMA_TILE_BYTES) {  // (not produced by the planners: tiles are cut at GRDMA_TILE_BYTES)
    wave_copy_tile(dst, src, n, lane);
    return;
  }
  uint64_t head = (16 - ((uint64_t)dst & 15)) & 15;
  if (head > n) head = n;
  const uint64_t n2 = n - head;
  const uint64_t units = n2 >> 4;
  const uint64_t tail = n2 & 15;
  // edge bytes first (their loads join the batch below)
  uint8_t hb = 0, tb = 0;
  if (src) {
    if ((uint64_t)lane < head) hb = src[lane];
    if ((uint64_t)lane < tail) tb = src[head + (units << 4) + lane];
  }
  uint8_t* d2 = dst + head;
  g_u32x4 da = (g_u32x4)(uint64_t)d2;
  if (src == nullptr) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t u = (uint64_t)lane + 64u * k;
      if (u < units) da[u] = u32x4{0, 0, 0, 0};
    }
  } else if (units) {
    const uint8_t* s2 = src + head;
    const unsigned shift = (unsigned)((uint64_t)s2 & 15);
    g_u32x4_c sa = (g_u32x4_c)((uint64_t)s2 & ~15ull);
    u32x4 a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t u = (uint64_t)lane + 64u * k;
      a[k] = __builtin_nontemporal_load(sa + (u < units ? u : units - 1));
    }
    if (shift == 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint64_t u = (uint64_t)lane + 64u * k;
        if (u < units) da[u] = a[k];
      }
    } else {
      // unaligned source: unit u needs blocks u and u + 1.  Block u + 1 is what the next
      // lane holds, so it comes over the DPP network (wave_shl:1); lane 63 fetches its own.
      u32x4 e[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint64_t u = (uint64_t)lane + 64u * k + 1;
        e[k] = a[k];
        if (lane == 63) e[k] = __builtin_nontemporal_load(sa + (u < units ? u : units));
      }
      // the last active lane of the tile needs block `units` too
      const uint64_t lastu = units - 1;
      u32x4 last_b = u32x4{0, 0, 0, 0};
      if ((uint64_t)lane == (lastu & 63)) last_b = __builtin_nontemporal_load(sa + units);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint64_t u = (uint64_t)lane + 64u * k;
        u32x4 b;
        b.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k].x, 0x130, 0xf, 0xf, false);
        b.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k].y, 0x130, 0xf, 0xf, false);
        b.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k].z, 0x130, 0xf, 0xf, false);
        b.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[k].w, 0x130, 0xf, 0xf, false);
        if (lane == 63) b = e[k];
        if (u == lastu) b = last_b;
        if (u < units) da[u] = funnel16(a[k], b, shift);
      }
    }
  }
  if ((uint64_t)lane < head) dst[lane] = hb;
  if ((uint64_t)lane < tail) d2[(units << 4) + lane] = tb;
}

// wave-uniform copy of a 64-bit value, provably uniform to the compiler (descriptor inputs)
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(uint64_t base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, (int)bytes, 0x00020000);
}


// ---------------------------------------------------------------------------------------------
// One wave moves n <= GRDMA_TILE_BYTES bytes src -> dst, any alignment, optionally clearing the source
// behind itself (reader zero-fill, ring_buffer.cc:160,164).  Destination-aligned 16-byte units;
// an unaligned source is realigned in registers (unit u needs source blocks u and u + 1; block
// u + 1 is what the next lane holds: wave_rol over the DPP network).  All loads -- up to nine
// 16-byte blocks per lane plus the edge bytes -- are in flight before the first store.  Loads
// and stores go through buffer descriptors sized to the tile: lanes beyond the tile read zero
// and their stores are dropped by the bounds check, so there is no per-lane branching.  AUX_LD /
// AUX_ST are the cache policies of the buffer instructions: 16 = sc1 (write-through stores, L1-
// bypassing loads: what a hand-off inside a resident kernel needs), 2 = nt (streaming), 0 = default.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 dpp_rol1(u32x4 v) {
  u32x4 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x134, 0xf, 0xf, false);  // wave_rol:1
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x134, 0xf, 0xf, false);
  r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x134, 0xf, 0xf, false);
  r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x134, 0xf, 0xf, false);
  return r;
}

// U: 16-byte units per lane = loads in flight (a tile is U KiB: n <= U * 1024)
template <int AUX_LD, int AUX_ST, bool ZERO, int U = GRDMA_TILE_BYTES / 1024>
__device__ __forceinline__ void wave_move_tile(uint64_t dst, uint64_t src, uint32_t n, int lane) {
  uint32_t head = (uint32_t)((16 - (dst & 15)) & 15);
  if (head > n) head = n;
  const uint32_t n2 = n - head;
  const uint32_t units = n2 >> 4, tail = n2 & 15;
  const uint64_t d2 = dst + head, s2 = src + head;
  const uint32_t shift = (uint32_t)(s2 & 15);
  const uint32_t nblk = units ? units + (shift ? 1u : 0u) : 0u;
  // (descriptor inputs must be provably wave-uniform: readfirstlane them)
  const __amdgpu_buffer_rsrc_t rs = mk_rsrc(uni64(s2 & ~15ull), uni32(nblk * 16));
  const __amdgpu_buffer_rsrc_t rd = mk_rsrc(uni64(d2), uni32(units * 16));
  const __amdgpu_buffer_rsrc_t rsb = mk_rsrc(uni64(src), uni32(n));
  const __amdgpu_buffer_rsrc_t rdb = mk_rsrc(uni64(dst), uni32(n));
  const uint32_t tail_off = head + (units << 4);
  u32x4 a[U + 1];
#pragma unroll
  for (int k = 0; k < U; k++) a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * k) * 16, 0, AUX_LD);
  a[U] = __builtin_amdgcn_raw_buffer_load_b128(rs, 64 * U * 16, 0, AUX_LD);  // block 64 U: same for every lane
  // edge bytes (out-of-range lanes read 0 and store nothing)
  const uint8_t hb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < head ? lane : n, 0, AUX_LD);
  const uint8_t tb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < tail ? tail_off + lane : n, 0, AUX_LD);
  if (shift == 0) {
#pragma unroll
    for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(a[k], rd, (lane + 64 * k) * 16, 0, AUX_ST);
  } else {
    // unit u needs blocks u and u + 1: block u + 1 sits in the next lane (lane 63: in lane 0's
    // next register)
    u32x4 r_cur = dpp_rol1(a[0]);
#pragma unroll
    for (int k = 0; k < U; k++) {
      const u32x4 r_next = dpp_rol1(a[k + 1]);
      const u32x4 b = lane == 63 ? r_next : r_cur;
      __builtin_amdgcn_raw_buffer_store_b128(funnel16(a[k], b, shift), rd, (lane + 64 * k) * 16, 0, AUX_ST);
      r_cur = r_next;
    }
  }
  __builtin_amdgcn_raw_buffer_store_b8(hb, rdb, (uint32_t)lane < head ? lane : n, 0, AUX_ST);
  __builtin_amdgcn_raw_buffer_store_b8(tb, rdb, (uint32_t)lane < tail ? tail_off + lane : n, 0, AUX_ST);
  if (ZERO) {
    // Every load of the tile has returned: its data fed the stores above, and a store cannot
    // issue before its operands arrived.  So the source may be overwritten right away.
    const uint64_t zs = (src + 15) & ~15ull, ze = (src + n) & ~15ull;
    if (ze > zs) {
      const uint32_t zu = (uint32_t)((ze - zs) >> 4);
      const __amdgpu_buffer_rsrc_t rz = mk_rsrc(uni64(zs), uni32(zu * 16));
#pragma unroll
      for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rz, (lane + 64 * k) * 16, 0, AUX_ST);
      const uint32_t e0 = (uint32_t)(zs - src), e1 = (uint32_t)(src + n - ze);  // < 16 each
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e0 ? lane : n, 0, AUX_ST);
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e1 ? (uint32_t)(ze - src) + lane : n, 0, AUX_ST);
    } else {
      // fewer than 31 bytes, no whole aligned block inside: bytes only
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, lane, 0, AUX_ST);
    }
  }
}


// Reader zero-fill (ring_buffer.cc:160,164): clear exactly [p, p+n).
__device__ __forceinline__ void wave_zero_tile(uint8_t* p, uint64_t n, int lane) {
  uint64_t head = (16 - ((uint64_t)p & 15)) & 15;
  if (head > n) head = n;
  if ((uint64_t)lane < head) p[lane] = 0;
  p += head;
  n -= head;
  const uint64_t units = n >> 4;
  u32x4* q = reinterpret_cast<u32x4*>(p);
  for (uint64_t u = lane; u < units; u += 64) q[u] = u32x4{0, 0, 0, 0};
  const uint64_t tail = n & 15;
  if ((uint64_t)lane < tail) p[(units << 4) + lane] = 0;
}

// ---------------------------------------------------------------------------------------------
// Words handed from one workgroup to another INSIDE one kernel (the drain workgroups of k_plan_pair_mw hand the promised
// credit to its Send workgroups; the entries a declining drain's general planner rewrites): the workgroups of a launch sit on different XCDs, whose L2s do not see each other's lines
// until a kernel ends.  The producer's stores are write-through (sc1: at the memory side once acknowledged), it waits
// for their acknowledgement (GRDMA_WAIT_VMEM) before it publishes; the consumer's loads bypass its L2 (sc1).  The same
// recipe as the tables of the link engine of rounds 2 - 4 were.  WT / SC1 = false: plain stores and loads (the consumer is
// a later kernel).  The host emulation has one coherent memory: plain accesses there.
// ---------------------------------------------------------------------------------------------
template <bool WT>
__device__ __forceinline__ void xwg_st32(uint32_t* p, uint32_t v) {
  if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool WT>
__device__ __forceinline__ void xwg_st64(uint64_t* p, uint64_t v) {
  if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool SC1>
__device__ __forceinline__ uint32_t xwg_ld32(const uint32_t* p) {
  return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool SC1>
__device__ __forceinline__ uint64_t xwg_ld64(const uint64_t* p) {
  return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// one 32-byte segment descriptor (two 16-byte write-through stores when WT: 8-byte sc1 stores cost 2.7 x per byte)
template <bool WT>
__device__ __forceinline__ void xwg_put_seg(grdma_seg* slot, uint64_t dst, uint64_t src, uint64_t len, uint64_t flags) {
#ifndef GRDMA_WAVE_EMU
  if (WT) {
    typedef __attribute__((address_space(1))) u32x4 g_q;
    g_q* q = (g_q*)(uint64_t)slot;
    const u32x4 a = {(uint32_t)dst, (uint32_t)(dst >> 32), (uint32_t)src, (uint32_t)(src >> 32)};
    const u32x4 b = {(uint32_t)len, (uint32_t)(len >> 32), (uint32_t)flags, (uint32_t)(flags >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q), "v"(a) : "memory");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(q + 1), "v"(b) : "memory");
    return;
  }
#endif
  *slot = grdma_seg{dst, src, len, flags};
}
// the header words of a plan as one 64-byte line (the copy kernels' waves take them from LDS)
struct plan_hdr {
  uint32_t nsegs, ntiles;
  uint64_t tag_base, tag_mask;
  uint32_t tile_bytes;
};
__device__ __forceinline__ plan_hdr plan_hdr_of(const uint32_t* w) {
  static_assert(offsetof(grdma_plan, nsegs) == 0 && offsetof(grdma_plan, ntiles) == 4 && offsetof(grdma_plan, tag_base) == 16 &&
                    offsetof(grdma_plan, tag_mask) == 24 && offsetof(grdma_plan, tile_bytes) == 32 && offsetof(grdma_plan, ready) == 36,
                "plan header layout");
  return plan_hdr{w[0], w[1], (uint64_t)w[4] | ((uint64_t)w[5] << 32), (uint64_t)w[6] | ((uint64_t)w[7] << 32), w[8]};
}
// Promised credit (k_plan_pair_mw): grdma_plan::promise is the hand-over word between the workgroup that
// commits the drain of a launch and the Send workgroups of the SAME launch.  0: nothing yet; state 1: the drain is
// committed -- the promise is KEPT, and the word carries it (whether the drain posts a credit, and its head); state 2: a
// Send workgroup's bounded wait ran out first -- the promise is GIVEN UP for this launch.  Both transitions start from 0
// and are compare-and-swaps, so the word takes exactly one value per launch and every Send workgroup, whenever it
// looks, comes away with the same answer: a wait that runs out in one workgroup cannot split the Send between
// workgroups that price with the promise and workgroups that price without it (ADVICE r4 / VERDICT r5).
// (round 6) The word IS the message.  Through round 5 it was a flag over the drain's result block: the committer wrote
// its L2 back before raising it (an agent-scope release), a Send workgroup invalidated its own L2 after seeing it (an
// acquire) and then fetched two words of the block -- 6.6 us between the drain's last stamp and the Send's "priced"
// stamp at the reference's default knobs (profiles/r06_plan_phases.txt), most of it those two cache operations.  A
// device-scope atomic is carried out at the memory side whichever XCD issues it: no fence, no second round trip.
// Every participant -- the committer and each Send workgroup -- counts itself out in grdma_plan::promise_done once it is
// through with the word; the last of the H + 1 to do so zeroes both for the next launch (nobody touches them after it:
// a committer that comes late, behind Send workgroups that all gave up, finds 2, leaves it, and is the one that clears).
#define GRDMA_PROMISE_KEPT 1ull
#define GRDMA_PROMISE_GIVEN_UP 2ull
#define GRDMA_PROMISE_RAN_OUT (1ull << 63)  // (in promise_wait's return value only: this workgroup's wait is the one that ran out)
__device__ __forceinline__ void promise_leave(grdma_plan* plan, uint32_t participants) {
  const uint32_t prev = __hip_atomic_fetch_add(&plan->promise_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (prev + 1 == participants) {
    __hip_atomic_store(&plan->promise_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&plan->promise, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the committing drain workgroup, thread 0, with what the drain's scatter is going to post
__device__ __forceinline__ uint32_t promise_keep(grdma_plan* plan, uint32_t participants, uint64_t credit_sent, uint64_t credit_head) {
  unsigned long long seen = 0;
  const unsigned long long word = GRDMA_PROMISE_KEPT | (credit_sent ? 4ull : 0ull) | (credit_head << 3);
  const bool kept = __hip_atomic_compare_exchange_strong((unsigned long long*)&plan->promise, &seen, word, __ATOMIC_RELAXED,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  promise_leave(plan, participants);
  return kept ? 1u : 0u;
}
// a Send workgroup (all threads call it): polls at most `bound` times; returns the word -- state 1 = kept, 2 = given up
// (by this workgroup or by another one: uniformly for the launch)
__device__ __forceinline__ uint64_t promise_wait(grdma_plan* plan, uint32_t bound, uint32_t participants) {
  __shared__ uint64_t s_word;
  if (threadIdx.x == 0) {
    unsigned long long v = 0;
    for (uint32_t spins = 0; spins < bound; spins++) {
      v = __hip_atomic_load((unsigned long long*)&plan->promise, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (v == 0) {  // the wait ran out -- a bug or a stalled drain, not a state: give the promise up for everybody
      unsigned long long seen = 0;
      v = __hip_atomic_compare_exchange_strong((unsigned long long*)&plan->promise, &seen, (unsigned long long)GRDMA_PROMISE_GIVEN_UP,
                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
              ? (GRDMA_PROMISE_GIVEN_UP | GRDMA_PROMISE_RAN_OUT)
              : seen;
    }
    promise_leave(plan, participants);
    s_word = v;
  }
  __syncthreads();
  return s_word;
}

// Every workgroup stages the tile prefix in LDS (one coalesced load).  A wave takes a
// CONTIGUOUS run of tiles: one LDS binary search finds the segment of its first tile, the
// following tiles walk forward through the staged prefix (a 16 KiB record is four tiles
// of the same segment: one search and one descriptor fetch instead of four).  Plans with
// more segments than LDS slots are staged SAMPLED (every 2nd / 4th entry, the stride a
// power of two): the search then lands on a window of `stride` entries, which the wave
// fetches with one global load (lane k takes entry k of the window) and resolves with a
// ballot, and the walk reads the next prefix entry from memory.
// LDS_N: slots of the staged prefix -- 8192 in the copy kernels (32 KB, occupancy is
// bound by registers there), 1024 where a plan workgroup runs its own small plan inline.
// One tile of segment sg (bytes [off, off + n)), plus the record tags the segment carries.
// cache policies of the plan tiles' loads / stores (AUX_LD / AUX_ST of wave_move_tile)
#ifndef GRDMA_PLAN_LD
#define GRDMA_PLAN_LD 2
#endif
#ifndef GRDMA_PLAN_ST
#define GRDMA_PLAN_ST 0
#endif
// ... and of the tiles of a SCATTER (segments that clear their source): what they store -- the delivered slices and the
// zeros behind them -- is not read again by any kernel of the round (the arena belongs to the transport, the ring's next
// writer overwrites whole records), unlike what the gather and the wire store: non-temporal, so that 126 MiB of
// scatter output per round do not push the staging buffer and the ring window (what the wire and the next scatter read)
// out of the Infinity Cache.  value 540 -> 565 GiB/s, eight alternations on one box (profiles/r05_scatter_store_policy.txt;
// the plan-free probe: tools/copy_probe, 46.5 -> 42.7 us).  Round 3 had tried nt for ALL plan stores: slower.
#ifndef GRDMA_PLAN_ST_ZERO
#define GRDMA_PLAN_ST_ZERO 2
#endif
// ... and what a scatter LOADS is the ring window the wire has just written: it is in the Infinity Cache now that the
// scatter's own output no longer pushes it out, and a default-policy load takes it from there where a streaming one
// did not (copy launch 56.7 -> 51.0 us, value 576 -> 601, five alternations: profiles/r05_scatter_store_policy.txt;
// default loads for the gather and the wire as well: +3 % instead of +4.5 %, the wire itself 21.5 -> 23 us)
#ifndef GRDMA_PLAN_LD_ZERO
#define GRDMA_PLAN_LD_ZERO 0
#endif
// the record tags a segment carries (header in front of its first tile, padding + footer behind its last one)
__device__ __forceinline__ void plan_tags(const grdma_seg& sg_in, uint64_t off, uint64_t n, uint64_t tag_base, uint64_t tm, int lane) {
  const grdma_seg sg = sg_in;  // (by value: a select between two fields of a referenced struct pins it to memory)
  if (sg.flags & (GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR)) {
    const bool wr = (sg.flags & GRDMA_SEG_TAG_WRITE) != 0;
    const uint64_t v_dst = sg.dst, v_src = sg.src;
    const uint64_t side = wr ? v_dst : v_src;
    uint8_t* const tb = reinterpret_cast<uint8_t*>(tag_base);
    if ((sg.flags & GRDMA_SEG_TAG_HDR) && off == 0 && lane == 0)
      *reinterpret_cast<uint64_t*>(tb + ((side - 8 - tag_base) & tm)) = wr ? (sg.flags >> GRDMA_SEG_TAG_LEN_SHIFT) : 0;
    if ((sg.flags & GRDMA_SEG_TAG_FTR) && off + n == sg.len) {
      const uint64_t e = (side + sg.len - tag_base) & tm;  // first byte behind the payload
      const uint64_t pad = (0 - e) & 7;
      if ((uint64_t)lane < pad) tb[e + lane] = 0;
      if (lane == 8) *reinterpret_cast<uint64_t*>(tb + ((e + pad) & tm)) = wr ? GRDMA_FOOTER : 0;
    }
  }
}

template <uint32_t TILE>
__device__ __forceinline__ void plan_tile(const grdma_seg& sg, uint64_t off, uint64_t n, uint64_t tag_base, uint64_t tm, int lane) {
  // (nontemporal loads: payload streams through once; plain stores: the next kernel of the
  // round reads what this one wrote out of the Infinity Cache)
  if (sg.src == 0) wave_zero_tile(reinterpret_cast<uint8_t*>(sg.dst + off), n, lane);
  else if (sg.flags & GRDMA_SEG_ZERO_SRC) wave_move_tile<GRDMA_PLAN_LD_ZERO, GRDMA_PLAN_ST_ZERO, true, (int)(TILE / 1024)>(sg.dst + off, sg.src + off, (uint32_t)n, lane);
  else wave_move_tile<GRDMA_PLAN_LD, GRDMA_PLAN_ST, false, (int)(TILE / 1024)>(sg.dst + off, sg.src + off, (uint32_t)n, lane);
  plan_tags(sg, off, n, tag_base, tm, lane);
}

// A segment of at most 64 bytes -- the 9-byte frame-header slices and records of an HTTP/2 stream: HALF the segments
// of a streaming round, each of which used to take a wave's whole pass (descriptor, load, store, tags: four dependent
// steps for nine bytes) -- moved one byte per lane in two halves, so that it can ride along with the 16 KiB tile the
// same wave moves next: its load is issued in front of that tile's loads, its stores behind that tile's stores.
#define GRDMA_TINY_MAX 64u
__device__ __forceinline__ uint8_t tiny_load(const grdma_seg& sg, int lane) {
  const uint8_t* src = reinterpret_cast<const uint8_t*>(sg.src);
  return (src != nullptr && (uint64_t)lane < sg.len) ? __builtin_nontemporal_load(src + lane) : (uint8_t)0;
}
__device__ __forceinline__ void tiny_store(const grdma_seg& sg, uint8_t v, uint64_t tag_base, uint64_t tm, int lane) {
  if ((uint64_t)lane < sg.len) {
    reinterpret_cast<uint8_t*>(sg.dst)[lane] = v;
    if (sg.src != 0 && (sg.flags & GRDMA_SEG_ZERO_SRC)) reinterpret_cast<uint8_t*>(sg.src)[lane] = 0;
  }
  plan_tags(sg, 0, sg.len, tag_base, tm, lane);
}

// a segment descriptor of the plan (SC1: laid out by another workgroup of this launch, see xwg_put_seg)
template <bool SC1>
__device__ __forceinline__ grdma_seg plan_seg(const grdma_plan* plan, uint32_t i) {
  if (!SC1) return plan->segs[i];
  const uint64_t* w = reinterpret_cast<const uint64_t*>(&plan->segs[i]);
  return grdma_seg{xwg_ld64<true>(w), xwg_ld64<true>(w + 1), xwg_ld64<true>(w + 2), xwg_ld64<true>(w + 3)};
}

template <uint32_t LDS_N, bool CONTIG, uint32_t TILE, bool SC1 = false>
__device__ __forceinline__ void run_plan_tiles(const grdma_plan* plan, uint32_t wave,
                                               uint32_t nwaves, int lane, const plan_hdr* hdr = nullptr) {
  __shared__ uint32_t s_prefix[LDS_N + 1];
  // (hdr: the header as plan_wait_ready fetched it)
  const uint32_t nsegs = hdr ? hdr->nsegs : xwg_ld32<SC1>(&plan->nsegs);
  const uint32_t ntiles = hdr ? hdr->ntiles : xwg_ld32<SC1>(&plan->ntiles);
  const uint64_t tag_base = hdr ? hdr->tag_base : xwg_ld64<SC1>(&plan->tag_base), tm = hdr ? hdr->tag_mask : xwg_ld64<SC1>(&plan->tag_mask);
  // In flight together with the header: the segment of tile `wave`, should the plan turn out to
  // hold one tile per segment (the steady state of both the gather and the scatter: tile t then IS
  // segment t -- no prefix staging, no search, no barrier, one memory round trip less).
  // (a wave takes segments 2 w and 2 w + 1, then 2 (w + nwaves), ...: a tiny segment rides along with its neighbour.
  // The two 32-byte descriptors of a pass travel as ONE register: lane l holds dword l & 15 of the pair -- a single
  // 64-byte load, in flight together with the plan header for the first pass and while the pass in front moves its
  // bytes for the others -- and are read out into scalar registers when the pass begins.)
  static_assert(sizeof(grdma_seg) == 32, "a pair of descriptors is sixteen dwords");
  const uint32_t w2 = 2 * wave;
  const uint32_t* const seg_words = reinterpret_cast<const uint32_t*>(plan->segs);
  uint32_t pairw = xwg_ld32<SC1>(&seg_words[(size_t)(w2 + 1 < GRDMA_MAX_SEGS ? w2 : 0) * 8 + (lane & 15)]);
  if (ntiles == nsegs) {  // (uniform over the workgroup; every segment has at least one tile)
    for (uint32_t t = w2; t < ntiles; t += 2 * nwaves) {
      auto word = [&](int k) -> uint64_t {  // 64-bit word k of the pair, wave-uniform
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)pairw, 2 * k + 1) << 32) |
               (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)pairw, 2 * k);
      };
      const grdma_seg cur0 = {word(0), word(1), word(2), word(3)};
      const grdma_seg cur1 = {word(4), word(5), word(6), word(7)};
      const uint32_t tn = t + 2 * nwaves;
      pairw = xwg_ld32<SC1>(&seg_words[(size_t)(tn + 1 < GRDMA_MAX_SEGS && tn < ntiles ? tn : t) * 8 + (lane & 15)]);
      const bool two = t + 1 < ntiles;
      const bool tiny0 = cur0.len <= GRDMA_TINY_MAX, tiny1 = two && cur1.len <= GRDMA_TINY_MAX;
      uint8_t b0 = 0, b1 = 0;
      if (tiny0) b0 = tiny_load(cur0, lane);
      if (tiny1) b1 = tiny_load(cur1, lane);
      if (!tiny0) plan_tile<TILE>(cur0, 0, cur0.len, tag_base, tm, lane);
      if (two && !tiny1) plan_tile<TILE>(cur1, 0, cur1.len, tag_base, tm, lane);
      if (tiny0) tiny_store(cur0, b0, tag_base, tm, lane);
      if (tiny1) tiny_store(cur1, b1, tag_base, tm, lane);
    }
    return;
  }
  uint32_t shift = 0;  // entry k of s_prefix is tile_prefix[k << shift]
  while (((nsegs >> shift) + 1) > LDS_N) shift++;
  const uint32_t nsamp = (nsegs >> shift) + 1;  // samples 0 .. nsegs >> shift
  for (uint32_t i = threadIdx.x; i < nsamp; i += blockDim.x) s_prefix[i] = xwg_ld32<SC1>(&plan->tile_prefix[i << shift]);
  __syncthreads();
  // CONTIG: tiles [wave * per, (wave + 1) * per); otherwise wave, wave + nwaves, ... (each
  // located by its own search)
  const uint32_t per = CONTIG ? (ntiles + nwaves - 1) / nwaves : 1;
  uint32_t t = CONTIG ? wave * per : wave;
  uint32_t tend = t + per < ntiles ? t + per : ntiles;
  if (t >= tend) return;
 next_run:
  uint32_t seg, p0, pnext;
  {
    uint32_t lo = 0, hi = nsamp;  // invariant: sample[lo] <= t, (hi == nsamp or sample[hi] > t)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    seg = lo << shift;
    p0 = s_prefix[lo];
    if (shift) {
      // entries seg .. seg + stride - 1 (those that exist): the last one that is <= t
      const uint32_t k = seg + (uint32_t)lane;
      const bool in = (uint32_t)lane < (1u << shift) && k < nsegs;
      const uint32_t pk = in ? xwg_ld32<SC1>(&plan->tile_prefix[k]) : 0xFFFFFFFFu;
      const uint64_t le = __ballot(in && pk <= t);
      const int last = 63 - __builtin_clzll(le);  // lane 0 always qualifies (pk == p0 <= t)
      seg += (uint32_t)last;
      p0 = __shfl(pk, last, 64);
    }
    pnext = shift ? xwg_ld32<SC1>(&plan->tile_prefix[seg + 1]) : s_prefix[seg + 1];
  }
  grdma_seg sg = plan_seg<SC1>(plan, seg);
  for (;; ) {
    const uint64_t off = (uint64_t)(t - p0) * TILE;
    uint64_t n = sg.len - off;
    if (n > TILE) n = TILE;
    plan_tile<TILE>(sg, off, n, tag_base, tm, lane);
    if (++t >= tend) {
      if (CONTIG) break;
      t += nwaves - 1;
      if (t >= ntiles) break;
      tend = t + 1;
      goto next_run;
    }
    if (t >= pnext) {
      // next segment that owns a tile (t < ntiles = prefix[nsegs], so the walk ends inside the plan)
      do {
        seg++;
        p0 = pnext;
        pnext = shift ? xwg_ld32<SC1>(&plan->tile_prefix[seg + 1]) : s_prefix[seg + 1];
      } while (t >= pnext);
      sg = plan_seg<SC1>(plan, seg);
    }
  }
}


// The plan says which tile size it was laid out for.
template <uint32_t LDS_N, bool CONTIG = true, bool SC1 = false>
__device__ __forceinline__ void run_plan(const grdma_plan* plan, uint32_t wave, uint32_t nwaves, int lane, const plan_hdr* hdr = nullptr) {
  if ((hdr ? hdr->tile_bytes : xwg_ld32<SC1>(&plan->tile_bytes)) == 16384u) run_plan_tiles<LDS_N, CONTIG, 16384u, SC1>(plan, wave, nwaves, lane, hdr);
  else run_plan_tiles<LDS_N, CONTIG, 8192u, SC1>(plan, wave, nwaves, lane, hdr);
}

#endif  // GRDMA_DEVFN_H
