// k_link: the persistent link engine (see grdma_link.h for the roles and the hand-off rules).
//
// Parity: the engine is deterministic and produces exactly what the reference's loops produce
// when they run one after the other -- one Send from the rdma_flush cursor, then endpoint reads
// until one would block (the schedule tests/test_gpu_stream_job.py drives the CPU oracle with):
//  * the receiver drains Send by Send, each drain ending in the read that finds nothing and
//    keeps its slice (rdma_bp_posix.cc:241-243);
//  * the sender runs ahead of the receiver only while that cannot change a record: a Send that
//    would be cut by the peer's credit (free space < staging budget) is priced again once the
//    receiver has caught up and every credit report of the earlier rounds is in.
// So records, delivered slices, credit reports and final state equal the sequential execution,
// while gather, wire and scatter of neighbouring Sends overlap in time.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_link.h"

namespace {

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define LK_AUX_SC1 16  // cache policy of the buffer builtins: sc1 = write-through / L1-bypassing
#define MINRD GRDMA_MIN_READ_SLICE

// every shared word is a GLOBAL-address-space agent-scope access (never flat: a flat access also
// counts against the LDS counter and is waited for with lgkmcnt)
typedef __attribute__((address_space(1))) uint64_t gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ uint64_t ldw(const uint64_t* p) { return __hip_atomic_load((const gu64*)(uint64_t)p, RLX_AGENT); }
__device__ __forceinline__ void stw(uint64_t* p, uint64_t v) { __hip_atomic_store((gu64*)(uint64_t)p, v, RLX_AGENT); }
__device__ __forceinline__ uint32_t ldw32(const uint32_t* p) { return __hip_atomic_load((const gu32*)(uint64_t)p, RLX_AGENT); }
__device__ __forceinline__ void stw32(uint32_t* p, uint32_t v) { __hip_atomic_store((gu32*)(uint64_t)p, v, RLX_AGENT); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave-uniform copy of a 64-bit value, provably uniform to the compiler (descriptor inputs)
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(uint64_t base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Bounded spin: polls cond() (relaxed loads), leaves on the abort word or on the wall clock.
template <typename F>
__device__ __forceinline__ bool lk_spin(lk_ctl* L, uint64_t limit_ticks, F cond) {
  uint32_t n = 0;
  uint64_t t0 = 0;
  for (;;) {
    if (cond()) return true;
    if (ldw(&L->abort.v) != 0) return false;
    if ((++n & 127u) == 0) {
      const uint64_t now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > limit_ticks) {
        stw(&L->abort.v, LK_ERR_TIMEOUT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// ---------------------------------------------------------------------------------------------
// One wave moves n <= LK_TILE bytes src -> dst, any alignment, optionally clearing the source
// behind itself (reader zero-fill, ring_buffer.cc:160,164).  Destination-aligned 16-byte units;
// an unaligned source is realigned in registers (unit u needs source blocks u and u + 1; block
// u + 1 is what the next lane holds: wave_rol over the DPP network).  All loads -- up to nine
// 16-byte blocks per lane plus the edge bytes -- are in flight before the first store.  Loads
// and stores go through buffer descriptors sized to the tile: lanes beyond the tile read zero
// and their stores are dropped by the bounds check, so there is no per-lane branching, and the
// sc1 policy makes every store a write-through one (visible to other CUs once acknowledged).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 dpp_rol1(u32x4 v) {
  u32x4 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x134, 0xf, 0xf, false);  // wave_rol:1
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x134, 0xf, 0xf, false);
  r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x134, 0xf, 0xf, false);
  r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x134, 0xf, 0xf, false);
  return r;
}

template <bool ZERO>
__device__ __forceinline__ void lk_move_tile(uint64_t dst, uint64_t src, uint32_t n, int lane) {
  constexpr int U = LK_TILE / 1024;  // 16-byte units per lane
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
  for (int k = 0; k < U; k++) a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (lane + 64 * k) * 16, 0, LK_AUX_SC1);
  a[U] = __builtin_amdgcn_raw_buffer_load_b128(rs, 64 * U * 16, 0, LK_AUX_SC1);  // block 64 U: same for every lane
  // edge bytes (out-of-range lanes read 0 and store nothing)
  const uint8_t hb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < head ? lane : n, 0, LK_AUX_SC1);
  const uint8_t tb = __builtin_amdgcn_raw_buffer_load_b8(rsb, (uint32_t)lane < tail ? tail_off + lane : n, 0, LK_AUX_SC1);
  if (shift == 0) {
#pragma unroll
    for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(a[k], rd, (lane + 64 * k) * 16, 0, LK_AUX_SC1);
  } else {
    // unit u needs blocks u and u + 1: block u + 1 sits in the next lane (lane 63: in lane 0's
    // next register)
    u32x4 r_cur = dpp_rol1(a[0]);
#pragma unroll
    for (int k = 0; k < U; k++) {
      const u32x4 r_next = dpp_rol1(a[k + 1]);
      const u32x4 b = lane == 63 ? r_next : r_cur;
      __builtin_amdgcn_raw_buffer_store_b128(funnel16(a[k], b, shift), rd, (lane + 64 * k) * 16, 0, LK_AUX_SC1);
      r_cur = r_next;
    }
  }
  __builtin_amdgcn_raw_buffer_store_b8(hb, rdb, (uint32_t)lane < head ? lane : n, 0, LK_AUX_SC1);
  __builtin_amdgcn_raw_buffer_store_b8(tb, rdb, (uint32_t)lane < tail ? tail_off + lane : n, 0, LK_AUX_SC1);
  if (ZERO) {
    // Every load of the tile has returned: its data fed the stores above, and a store cannot
    // issue before its operands arrived.  So the source may be overwritten right away.
    const uint64_t zs = (src + 15) & ~15ull, ze = (src + n) & ~15ull;
    if (ze > zs) {
      const uint32_t zu = (uint32_t)((ze - zs) >> 4);
      const __amdgpu_buffer_rsrc_t rz = mk_rsrc(uni64(zs), uni32(zu * 16));
#pragma unroll
      for (int k = 0; k < U; k++) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rz, (lane + 64 * k) * 16, 0, LK_AUX_SC1);
      const uint32_t e0 = (uint32_t)(zs - src), e1 = (uint32_t)(src + n - ze);  // < 16 each
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e0 ? lane : n, 0, LK_AUX_SC1);
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, (uint32_t)lane < e1 ? (uint32_t)(ze - src) + lane : n, 0, LK_AUX_SC1);
    } else {
      // fewer than 31 bytes, no whole aligned block inside: bytes only
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rsb, lane, 0, LK_AUX_SC1);
    }
  }
}

// One table entry: up to two tiles, plus the record tags the entry carries
// (GRDMA_SEG_TAG_*: AppendHeader / AppendFooter on the send side, ring_buffer.h:84-99; the
// clearing of header, padding and footer on the receive side, ring_buffer.cc:146,173-180).
template <bool ZERO>
__device__ __forceinline__ void lk_run_entry(uint64_t e_dst, uint64_t e_src, uint32_t e_len, uint32_t e_flags,
                                             uint64_t e_aux, uint64_t tag_base, uint64_t tag_mask, int lane) {
#pragma unroll 1
  for (uint32_t off = 0; off < e_len; off += LK_TILE) {
    const uint32_t n = e_len - off < LK_TILE ? e_len - off : LK_TILE;
    lk_move_tile<ZERO>(e_dst + off, e_src + off, n, lane);
  }
  if (e_flags & (GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR)) {
    const bool wr = (e_flags & GRDMA_SEG_TAG_WRITE) != 0;
    const uint64_t side = ZERO ? e_src : e_dst;  // the ring side: source of a scatter, destination of a gather
    uint8_t* const tb = reinterpret_cast<uint8_t*>(tag_base);
    if ((e_flags & GRDMA_SEG_TAG_HDR) && lane == 0)
      stw(reinterpret_cast<uint64_t*>(tb + ((side - 8 - tag_base) & tag_mask)), wr ? e_aux : 0);
    if (e_flags & GRDMA_SEG_TAG_FTR) {
      const uint64_t end = (side + e_len - tag_base) & tag_mask;  // first byte behind the payload
      const uint32_t pad = (uint32_t)((0 - end) & 7);
      const __amdgpu_buffer_rsrc_t rp = mk_rsrc(uni64(tag_base + end), uni32(pad));
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rp, lane, 0, LK_AUX_SC1);
      if (lane == 8) stw(reinterpret_cast<uint64_t*>(tb + ((end + pad) & tag_mask)), wr ? GRDMA_FOOTER : 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// workers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lk_worker(lk_ctl* L, int stage, uint32_t w, uint32_t W, uint64_t ticks, int lane) {
  uint64_t tag_base = 0, tag_mask = ~0ull;
  if (stage == LK_SCATTER) {
    tag_base = (uint64_t)L->rx->ring;
    tag_mask = L->rx->cap - 1;
  } else if (stage == LK_GATHER && L->direct) {
    tag_base = (uint64_t)L->tx->peer_ring;
    tag_mask = L->tx->cap - 1;
  }
  const uint64_t* const pub = &L->published[stage].v;
  const uint64_t* const closed = &L->closed[stage].v;
  const uint64_t* const abort_w = &L->abort.v;
  const __amdgpu_buffer_rsrc_t rtab = mk_rsrc(uni64((uint64_t)L->tab[stage]), (uint32_t)(sizeof(lk_entry) * LK_TABLE_CAP));
  uint64_t pub_seen = 0;  // published count read last (entries below it need no new poll)
  for (uint64_t lap = 0;; lap++) {
    // Entry ownership rotates by one wave per lap: a periodic mix of small and large entries (a
    // 9-byte frame header in front of every 16 KiB payload) is spread over all waves whatever W is.
    const uint64_t e = lap * W + (w + W - (uint32_t)(lap % W)) % W;
    if (pub_seen <= e) {
      uint32_t n = 0;
      uint64_t t0 = 0;
      for (;;) {
        pub_seen = ldw(pub);
        if (pub_seen > e) break;
        if (ldw(closed) != 0) {  // published is final once closed is set: read it again
          pub_seen = ldw(pub);
          if (pub_seen > e) break;
          return;
        }
        if (ldw(abort_w) != 0) return;
        if ((++n & 127u) == 0) {
          const uint64_t now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > ticks) {
            stw(&L->abort.v, LK_ERR_TIMEOUT);
            return;
          }
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    // the whole 32-byte entry in one round trip (same address in every lane)
    const uint32_t eoff = (uint32_t)(e & (LK_TABLE_CAP - 1)) * (uint32_t)sizeof(lk_entry);
    const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff, 0, LK_AUX_SC1);
    const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff + 16, 0, LK_AUX_SC1);
    const uint64_t e_dst = (uint64_t)q0.x | ((uint64_t)q0.y << 32), e_src = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
    const uint32_t e_len = q1.x, e_flags = q1.y;
    const uint64_t e_aux = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
    const uint32_t slot = (e_flags >> 8) & 0xFFu;
    if (stage == LK_WIRE) {
      // staging is complete when every gather entry of this Send has been counted in
      const uint32_t* g = &L->done_tx[LK_GATHER][slot].v;
      const uint32_t need = (uint32_t)e_aux;
      if (!lk_spin(L, ticks, [&]() { return ldw32(g) >= need; })) return;
    }
    if (stage == LK_SCATTER) lk_run_entry<true>(e_dst, e_src, e_len, e_flags, e_aux, tag_base, tag_mask, lane);
    else lk_run_entry<false>(e_dst, e_src, e_len, e_flags, e_aux, tag_base, tag_mask, lane);
    drain();  // my write-through stores are acknowledged: the entry may be counted
    if (lane == 0) {
      uint32_t* d = stage == LK_SCATTER ? &L->done_rx[slot].v : &L->done_tx[stage][slot].v;
      __hip_atomic_fetch_add((gu32*)(uint64_t)d, 1u, RLX_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// entry emission: every lane brings up to NP pieces {dst, src, len, flags, aux}; pieces longer
// than LK_ENTRY_MAX are cut.  lk_count() first (table space), then lk_emit().
// ---------------------------------------------------------------------------------------------
struct lk_piece {
  uint64_t dst, src;
  uint64_t len;
  uint32_t flags;
  uint64_t aux;
};

__device__ __forceinline__ uint32_t lk_sub_entries(uint64_t len) {
  return (uint32_t)((len + LK_ENTRY_MAX - 1) / LK_ENTRY_MAX);
}

__device__ __forceinline__ void lk_store_entry(lk_entry* tab, uint64_t idx, uint64_t dst, uint64_t src,
                                               uint32_t len, uint32_t flags, uint64_t aux) {
  uint64_t* q = reinterpret_cast<uint64_t*>(tab + (idx & (LK_TABLE_CAP - 1)));
  stw(q, dst);
  stw(q + 1, src);
  stw(q + 2, (uint64_t)len | ((uint64_t)flags << 32));
  stw(q + 3, aux);  // (8-byte write-through stores; a reader takes the entry only after the published count covers it)
}

template <int NP>
__device__ __forceinline__ uint32_t lk_count(const lk_piece (&pc)[NP]) {
  uint32_t mine = 0;
#pragma unroll
  for (int p = 0; p < NP; p++) mine += lk_sub_entries(pc[p].len);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
  return mine;
}

template <int NP>
__device__ __forceinline__ void lk_emit(lk_entry* tab, uint64_t base, const lk_piece (&pc)[NP], uint32_t slot,
                                        int lane) {
  bool big = false;
  uint32_t cnt = 0;
#pragma unroll
  for (int p = 0; p < NP; p++) {
    big |= pc[p].len > LK_ENTRY_MAX;
    cnt += pc[p].len ? 1u : 0u;
  }
  const uint32_t sl = slot << 8;
  if (__ballot(big) == 0) {
    // the common case: one entry per piece, every lane stores its own
    const uint32_t incl = wave_incl_scan_u32(cnt);
    uint64_t at = base + incl - cnt;
#pragma unroll
    for (int p = 0; p < NP; p++)
      if (pc[p].len) {
        lk_store_entry(tab, at, pc[p].dst, pc[p].src, (uint32_t)pc[p].len, pc[p].flags | sl, pc[p].aux);
        at++;
      }
    return;
  }
  // some piece is longer than an entry: lane by lane, the wave cuts each piece together
  uint64_t at = base;
  const uint64_t any = __ballot(cnt != 0);
  for (uint64_t m = any; m; m &= m - 1) {
    const int r = __builtin_ctzll(m);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const uint64_t plen = __shfl(pc[p].len, r, 64);
      if (plen == 0) continue;  // uniform
      const uint64_t pdst = __shfl(pc[p].dst, r, 64), psrc = __shfl(pc[p].src, r, 64), paux = __shfl(pc[p].aux, r, 64);
      const uint32_t pfl = __shfl(pc[p].flags, r, 64);
      const uint32_t nsub = lk_sub_entries(plen);
      for (uint32_t j0 = 0; j0 < nsub; j0 += 64) {
        const uint32_t j = j0 + lane;
        if (j < nsub) {
          const uint64_t o = (uint64_t)j * LK_ENTRY_MAX;
          const uint32_t len = plen - o < LK_ENTRY_MAX ? (uint32_t)(plen - o) : LK_ENTRY_MAX;
          uint32_t fl = pfl & ~(uint32_t)(GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR);
          if (j == 0) fl |= pfl & GRDMA_SEG_TAG_HDR;
          if (j == nsub - 1) fl |= pfl & GRDMA_SEG_TAG_FTR;
          lk_store_entry(tab, at + j, pdst + o, psrc + o, len, fl | sl, paux);
        }
      }
      at += nsub;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TX leader: PairPollable::Send (pair.cc:645-734) + the rdma_flush cursor
// (rdma_bp_posix.cc:476-493), one Send per iteration, 64 records priced per step:
// enc_i = 16 + round_up8(len_i) prefix-summed on the DPP network, every record tests its own
// budget pay_i = min(len_i, W(S - st_i), W(free0 - st_i)) assuming the earlier ones went out
// whole, a ballot finds the first short record -- where the reference's loop stops
// (SURVEY.md Appendix A.4).
// ---------------------------------------------------------------------------------------------
__device__ void lk_tx_leader(lk_ctl* L, uint64_t ticks, int lane) {
  grdma_conn* c = L->tx;
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap;
  uint32_t max_sge = c->max_sge;
  if (max_sge > GRDMA_TX_MAX_RECORDS - 1) max_sge = GRDMA_TX_MAX_RECORDS - 1;
  const bool direct = L->direct != 0;
  const uint32_t B = L->n_staging;
  uint8_t* const peer_ring = c->peer_ring;
  const grdma_sge* const slices = L->slices;
  const uint64_t nslices = L->nslices;
  lk_entry* const gtab = L->tab[LK_GATHER];
  lk_entry* const wtab = L->tab[LK_WIRE];

  uint64_t tail = c->remote_tail;
  uint64_t idx = 0, bidx = 0, remaining = L->total_bytes;  // rdma_write: outgoing_byte_idx = 0
  uint64_t k = 0, gpub = 0, wpub = 0, retired = 0, gfloor = 0, wfloor = 0;
  uint64_t total_written = 0, records = 0, rounds = 0;
  uint32_t partial = (uint32_t)c->partial_write, last_records = 0;
  // in-flight Sends, one per lane: lane j keeps what the Send in slot j published
  uint32_t my_ng = 0, my_nw = 0;
  uint64_t wait_slot = 0, wait_credit = 0;
  bool failed = false;

  // the slice table is read one 64-record chunk ahead of the pricing: the loads are issued as
  // soon as the next cursor is known and have landed by the time the chunk is priced
  uint64_t pf_base = ~0ull, pf_len = 0, pf_ptr = 0;
  auto prefetch = [&](uint64_t base) {
    const uint64_t i = base + lane;
    pf_len = 0;
    pf_ptr = 0;
    if (i < nslices) {
      const grdma_sge g = slices[i];
      pf_len = g.len;
      pf_ptr = (uint64_t)g.ptr;
    }
    pf_base = base;
  };

  auto retire_oldest = [&]() -> bool {
    // the oldest Send in flight has left its staging buffer: gathered and on the wire
    const int rs = (int)(retired % LK_SLOTS);
    const uint32_t ng = __shfl(my_ng, rs, 64), nw = __shfl(my_nw, rs, 64);
    const uint32_t* dg = &L->done_tx[LK_GATHER][rs].v;
    const uint32_t* dw = &L->done_tx[LK_WIRE][rs].v;
    const uint64_t t0 = wall_clock64();
    if (!lk_spin(L, ticks, [&]() { return ldw32(dg) >= ng && ldw32(dw) >= nw; })) return false;
    wait_slot += wall_clock64() - t0;
    retired++;
    gfloor += ng;
    wfloor += nw;
    return true;
  };

  if (c->status == GRDMA_PAIR_CONNECTED) {
    while (idx < nslices && !failed) {
      const uint32_t slot = (uint32_t)(k % LK_SLOTS);
      // a staging buffer and a counter slot must be free; the receiver must have read the
      // descriptor that last lived in this slot
      const uint64_t max_inflight = direct ? (uint64_t)(LK_SLOTS - 2) : (B < LK_SLOTS - 2 ? B : LK_SLOTS - 2);
      while (k - retired >= max_inflight)
        if (!retire_oldest()) { failed = true; break; }
      if (failed) break;
      if (k >= LK_SLOTS) {
        const uint64_t need = k - LK_SLOTS + 1;
        if (!lk_spin(L, ticks, [&]() { return ldw(&L->rx_sends_seen.v) >= need; })) { failed = true; break; }
      }
      uint8_t* const sbuf = direct ? nullptr : L->staging[k % B];
      uint64_t st_base, nrec, sent, whole_records;
      uint32_t ents;
      uint64_t short_pay_total;
      for (;;) {  // pricing attempts of Send k
        // how far the receiver is, then the credit it granted (get_remote_head(), pair.h:229-233):
        // in this order, so that "caught up" implies every credit report is in
        const uint64_t rounds_done = ldw(&L->rx_rounds_done.v);
        const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t free0 = cap - ((tail + cap - rhead) & mask);
        const uint64_t room0 = S < free0 ? S : free0;
        st_base = 0; nrec = 0; sent = 0; whole_records = 0; ents = 0; short_pay_total = 0;
        bool shorted = false;
        for (bool stop = false; !stop;) {
          const uint64_t i = idx + nrec + lane, rec_no = nrec + lane;
          const bool valid = i < nslices && rec_no < max_sge;
          if (pf_base != idx + nrec) prefetch(idx + nrec);
          uint64_t len = 0, src = 0;
          if (valid) {
            len = pf_len;
            src = pf_ptr;
            if (rec_no == 0) {  // the first slice continues at outgoing_byte_idx
              len = sat_sub(len, bidx);
              src += bidx;
            }
          }
          // (clamped so that sums cannot overflow; anything above 2 * cap cannot fit anyway)
          const uint32_t enc = valid ? (uint32_t)enc_size(len < (cap << 1) ? len : (cap << 1)) : 0;
          // rings in the engine are at most 256 MiB, 64 records of at most 2 * cap: 32 bits do
          const uint32_t incl = wave_incl_scan_u32(enc);
          const uint64_t st = st_base + incl - enc;
          const bool shortf = valid && (len == 0 || len > writable_of(sat_sub(room0, st)));
          const uint64_t bm = __ballot(shortf);
          const uint32_t nv = (uint32_t)__builtin_popcountll(__ballot(valid));
          const uint32_t take = bm ? (uint32_t)__builtin_ctzll(bm) : nv;
          uint64_t short_pay = 0;
          if (bm) {
            const int f = __builtin_ctzll(bm);
            const uint64_t lf = __shfl(len, f, 64), stf = __shfl(st, f, 64);
            const uint64_t a = writable_of(sat_sub(S, stf)), b = writable_of(sat_sub(free0, stf));
            short_pay = lf;
            if (a < short_pay) short_pay = a;
            if (b < short_pay) short_pay = b;
            shorted = true;
          }
          const uint64_t my_pay = (uint32_t)lane < take ? len : ((bm && (uint32_t)lane == take) ? short_pay : 0);
          // the record of this lane as copy pieces: the payload goes behind the 8-byte header at
          // staging + st (or straight into the peer ring, where it may cross the ring end)
          lk_piece pc[2];
          pc[0] = {0, 0, 0, 0, 0};
          pc[1] = {0, 0, 0, 0, 0};
          if (my_pay) {
            const uint32_t tagw = (uint32_t)GRDMA_SEG_TAG_WRITE;
            if (direct) {
              const uint64_t pay_off = (tail + st + 8) & mask;
              if (pay_off + my_pay > cap) {
                const uint64_t l1 = cap - pay_off;
                pc[0] = {(uint64_t)(peer_ring + pay_off), src, l1, tagw | (uint32_t)GRDMA_SEG_TAG_HDR, my_pay};
                pc[1] = {(uint64_t)peer_ring, src + l1, my_pay - l1, tagw | (uint32_t)GRDMA_SEG_TAG_FTR, my_pay};
              } else {
                pc[0] = {(uint64_t)(peer_ring + pay_off), src, my_pay,
                         tagw | (uint32_t)(GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR), my_pay};
              }
            } else {
              pc[0] = {(uint64_t)(sbuf + st + 8), src, my_pay, tagw | (uint32_t)(GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR), my_pay};
            }
          }
          const uint32_t n_new = lk_count<2>(pc);
          while (gpub + ents + n_new - gfloor > LK_TABLE_CAP)
            if (!retire_oldest()) { failed = true; break; }
          if (failed) break;
          lk_emit<2>(gtab, gpub + ents, pc, slot, lane);
          ents += n_new;
          const uint64_t whole_enc = take ? (uint64_t)__shfl(incl, (int)take - 1, 64) : 0;
          st_base += whole_enc + (short_pay ? enc_size(short_pay) : 0);
          nrec += take + (short_pay ? 1u : 0u);
          whole_records += take;
          short_pay_total = short_pay;
          uint64_t s = my_pay;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
          sent += s;
          stop = bm != 0 || nv < 64 || nrec >= max_sge;
          // next chunk of this Send, or the first chunk of the next one (a Send that is priced
          // again starts where this one did: the synchronous path above covers that)
          prefetch(stop ? idx + whole_records : idx + nrec);
        }
        if (failed) break;
        // A Send the peer's credit cut short (or left empty) is only final when the receiver
        // has drained every earlier round and posted its credit reports -- the state the
        // sequential loop would have priced it in.  Otherwise wait for the receiver and price again.
        const bool credit_limited = (shorted && free0 < S) || nrec == 0;
        if (!credit_limited || rounds_done >= k) break;
        const uint64_t t0 = wall_clock64();
        if (!lk_spin(L, ticks, [&]() {
              return ldw(&L->rx_rounds_done.v) != rounds_done ||
                     __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != rhead;
            })) { failed = true; break; }
        wait_credit += wall_clock64() - t0;
      }
      if (failed) break;
      if (nrec == 0) {  // nothing in flight, all credit in, and still nothing fits: a zero-length slice
        stw(&L->abort.v, LK_ERR_NO_PROGRESS);
        failed = true;
        break;
      }
      const uint64_t staged = st_base;
      // the wire: the <= 2 RDMA WRITEs of GetWriteRequests (ring_buffer.cc:261-330), cut into entries
      uint32_t nw = 0;
      if (!direct) {
        const uint64_t seg1 = staged < cap - tail ? staged : cap - tail;
        const uint32_t n1 = lk_sub_entries(seg1), n2 = lk_sub_entries(staged - seg1);
        nw = n1 + n2;
        while (wpub + nw - wfloor > LK_TABLE_CAP)
          if (!retire_oldest()) { failed = true; break; }
        if (failed) break;
        for (uint32_t j0 = 0; j0 < nw; j0 += 64) {
          const uint32_t j = j0 + lane;
          if (j < nw) {
            const bool second = j >= n1;
            const uint64_t o = (uint64_t)(second ? j - n1 : j) * LK_ENTRY_MAX;
            const uint64_t seglen = second ? staged - seg1 : seg1;
            const uint32_t len = seglen - o < LK_ENTRY_MAX ? (uint32_t)(seglen - o) : LK_ENTRY_MAX;
            const uint64_t so = second ? seg1 + o : o;
            const uint64_t dst = (uint64_t)peer_ring + (second ? o : tail + o);
            lk_store_entry(wtab, wpub + j, dst, (uint64_t)(sbuf + so), len, slot << 8, ents);
          }
        }
      }
      // the counters of this slot start again at zero; then the descriptor; then -- all of it
      // acknowledged -- the publication
      stw32(&L->done_tx[LK_GATHER][slot].v, 0);
      stw32(&L->done_tx[LK_WIRE][slot].v, 0);
      lk_send_desc* d = &L->sends[slot];
      stw(&d->staged, staged);
      stw(reinterpret_cast<uint64_t*>(&d->n_gather), (uint64_t)ents | ((uint64_t)nw << 32));
      stw(&d->records, nrec);
      stw(&d->seq, k + 1);
      drain();
      gpub += ents;
      wpub += nw;
      stw(&L->published[LK_GATHER].v, gpub);
      if (!direct) stw(&L->published[LK_WIRE].v, wpub);
      stw(&L->sends_pub.v, k + 1);
      if (lane == (int)slot) {
        my_ng = ents;
        my_nw = nw;
      }
      // bookkeeping of Send() and the rdma_flush cursor walk (rdma_bp_posix.cc:480-493)
      tail = (tail + staged) & mask;
      const uint64_t offered = remaining;
      remaining -= sent;
      partial = sent < offered ? 1u : 0u;  // pair.cc:709
      total_written += sent;
      records += nrec;
      last_records = (uint32_t)nrec;
      rounds++;
      if (short_pay_total) bidx = (whole_records == 0 ? bidx : 0) + short_pay_total;
      else if (whole_records != 0) bidx = 0;
      idx += whole_records;
      k++;
    }
  }
  // no more entries: let the workers run dry and leave
  drain();
  stw(&L->closed[LK_GATHER].v, 1);
  stw(&L->closed[LK_WIRE].v, 1);
  stw(&L->tx_done.v, 1);
  if (lane == 0) {
    c->remote_tail = tail;
    c->partial_write = partial;
    c->total_written += total_written;
    c->tx_records += records;
    c->tx_last_records = last_records;
    c->tx_rounds += rounds;
    c->tx_slice_idx = idx;
    c->tx_byte_idx = bidx;
    c->tx_remaining = remaining;
    L->res_sends = k;
    L->res_entries[LK_GATHER] = gpub;
    L->res_entries[LK_WIRE] = wpub;
    L->res_wait_ticks[0] = wait_slot;
    L->res_wait_ticks[1] = wait_credit;
  }
}

// ---------------------------------------------------------------------------------------------
// RX leader: see the file header.  The wave tier of k_rx_plan (grdma_rx_plan.hip) made
// incremental: 64 speculative probes of the record chain per memory round trip, 64 whole records
// replayed through the endpoint-read state machine per step on the DPP network, scalar reads for
// partial records and tails -- bounded by what the wire has completely delivered.
// ---------------------------------------------------------------------------------------------
#define LK_CHAIN_CAP 384

struct lk_walker {
  const uint8_t* ring;
  uint64_t cap;
  uint64_t pos;     // ring offset of the first unverified record
  uint64_t e0;      // encoded size of the record at pos when its header is already known
  uint64_t h2, h1;  // encoded sizes of the two records before pos (0 = unknown)
  uint64_t limit;   // bytes behind pos that have completely landed
};

// One probe round: LK_PROBE_GROUPS x 64 speculative probes per memory round trip.  Lane j of
// group g loads the tag words at the offset the chain reaches after 64 g + j records if the
// last two sizes keep alternating; every probe checks its own link and the footer in front of
// it; ballots give the verified prefix.  Stores the payload sizes in chain[0..v), returns v.
#define LK_PROBE_GROUPS 4
__device__ __forceinline__ uint32_t lk_chain_round(lk_walker* w, uint64_t* chain, int lane) {
  constexpr int K = LK_PROBE_GROUPS;
  const uint64_t cap = w->cap, mask = cap - 1;
  const uint64_t lim = w->limit;
  if (lim == 0) return 0;
  const uint64_t H2 = w->e0 ? w->h1 : w->h2;
  const uint64_t H1 = w->e0 ? w->e0 : w->h1;
  const uint64_t A = H2 ? H2 : H1, B = H1;  // predicted sizes alternate A, B, A, ...
  const uint64_t e0 = w->e0;
  auto rel_of = [&](uint64_t j) -> uint64_t {
    uint64_t base = 0, k = j;
    if (e0) {
      if (j == 0) return 0;
      base = e0;
      k = j - 1;
    }
    return base + (k >> 1) * (A + B) + ((k & 1) ? A : 0);
  };
  const bool have_pattern = (A != 0);
  uint64_t hdr[K], prev[K], rel[K], reln[K];
  bool ph[K], pp[K];
#pragma unroll
  for (int g = 0; g < K; g++) {
    const uint64_t idx = (uint64_t)(64 * g + lane);
    rel[g] = have_pattern ? rel_of(idx) : 0;
    reln[g] = have_pattern ? rel_of(idx + 1) : 0;
    // only records that end inside what has landed are looked at
    ph[g] = (idx == 0) || (have_pattern && reln[g] <= lim);
    pp[g] = idx > 0 && have_pattern && rel[g] <= lim;
  }
  // all 2 K tag loads of a lane are in flight together
#pragma unroll
  for (int g = 0; g < K; g++) {
    const uint64_t my_pos = (w->pos + rel[g]) & mask;
    hdr[g] = ph[g] ? ld_tag(w->ring + my_pos) : 0;
    prev[g] = pp[g] ? ld_tag(w->ring + ((my_pos + cap - 8) & mask)) : 0;  // footer of the record before
  }
  uint32_t v = 64 * K;
  uint64_t enc_v = 0, rel_v = 0, enc_l1 = 0, enc_l2 = 0;
  bool v_valid = false;
  uint64_t foot_next0 = 0;  // bit 0 of the NEXT group's footer ballot = footer of this group's record 63
  // (groups are examined from the last to the first so that each knows its successor's first footer)
  uint64_t m_good[K], m_valid[K];
#pragma unroll
  for (int g = K - 1; g >= 0; g--) {
    const bool valid = ph[g] && hdr[g] != 0 && hdr[g] <= cap - GRDMA_RESERVED;
    const uint64_t enc = 16 + round_up8(hdr[g]);
    const bool link_ok = valid && have_pattern && !(g == K - 1 && lane == 63) && enc == reln[g] - rel[g];
    const uint64_t m_link = __ballot(link_ok);
    const uint64_t m_fp = __ballot(pp[g] && prev[g] == GRDMA_FOOTER);
    const uint64_t m_foot = (m_fp >> 1) | (foot_next0 << 63);  // bit j: footer of record 64 g + j
    foot_next0 = m_fp & 1;
    m_good[g] = m_link & m_foot;
    m_valid[g] = __ballot(valid);
  }
#pragma unroll
  for (int g = K - 1; g >= 0; g--)
    if (m_good[g] != ~0ull) v = 64 * g + (uint32_t)__builtin_ctzll(~m_good[g]);
  // sizes of the verified records; what the walker needs from records v - 2, v - 1 and v
#pragma unroll
  for (int g = 0; g < K; g++) {
    const uint32_t idx = 64 * g + lane;
    if (idx < v) chain[idx] = hdr[g];
    const uint64_t enc = 16 + round_up8(hdr[g]);
    const uint32_t lo = 64 * g;
    if (v >= lo && v < lo + 64) {
      enc_v = __shfl(enc, (int)(v - lo), 64);
      rel_v = __shfl(rel[g], (int)(v - lo), 64);
      v_valid = (m_valid[g] >> (v - lo)) & 1;
    }
    if (v >= 1 && v - 1 >= lo && v - 1 < lo + 64) enc_l1 = __shfl(enc, (int)(v - 1 - lo), 64);
    if (v >= 2 && v - 2 >= lo && v - 2 < lo + 64) enc_l2 = __shfl(enc, (int)(v - 2 - lo), 64);
  }
  if (v == 64 * K) {  // (cannot happen: the last probe never links; kept for the arithmetic below)
    v = 64 * K - 1;
  }
  if (v >= 2) {
    w->h2 = enc_l2;
    w->h1 = enc_l1;
  } else if (v == 1) {
    w->h2 = w->h1;
    w->h1 = enc_l1;
  }
  w->pos = (w->pos + rel_v) & mask;
  w->limit = lim - rel_v;
  // record v is the first unverified one: when its header was read and is a record header, its
  // exact footer position is probed next round (as probe 0)
  w->e0 = v_valid ? enc_v : 0;
  return v;
}

__device__ __forceinline__ uint32_t lk_read_space_after(uint64_t n, uint32_t s) {
  if (s == 0) return n >= MINRD ? 0 : (uint32_t)(MINRD - n);
  if (n < s) return s - (uint32_t)n;
  if (n == s) return 0;
  const uint64_t r = n - s;
  return r >= MINRD ? 0 : (uint32_t)(MINRD - r);
}

struct lk_rec_plan {
  uint64_t c1, c2;    // bytes of the (at most) two Recv steps
  uint64_t sl0, sl1;  // lengths of the slices completed by this record, in order (0 = none)
  uint32_t sl_cnt;
};

__device__ __forceinline__ lk_rec_plan lk_replay_record(uint64_t n, uint32_t s_in) {
  lk_rec_plan r;
  r.c1 = n;
  r.c2 = 0;
  r.sl0 = r.sl1 = 0;
  if (s_in == 0) {
    if (n >= MINRD) r.sl0 = n;
  } else if (n <= s_in) {
    if (n == s_in) r.sl0 = MINRD;
  } else {
    r.c1 = s_in;
    r.c2 = n - s_in;
    r.sl0 = MINRD;
    if (r.c2 >= MINRD) r.sl1 = r.c2;
  }
  r.sl_cnt = (r.sl0 ? 1u : 0u) + (r.sl1 ? 1u : 0u);
  return r;
}

__device__ __forceinline__ void lk_split_step(uint64_t pay, uint64_t off, uint64_t len, uint64_t cap,
                                              uint64_t* o0, uint64_t* l0, uint64_t* o1, uint64_t* l1) {
  const uint64_t p0 = (pay + off) & (cap - 1);
  const uint64_t first = len < cap - p0 ? len : cap - p0;
  *o0 = p0;
  *l0 = first;
  *o1 = 0;
  *l1 = len - first;
}

__device__ __forceinline__ uint64_t lk_al16(uint64_t v) { return (v + 15) & ~15ull; }

__device__ void lk_rx_leader(lk_ctl* L, uint64_t ticks, uint64_t* s_chain, int lane) {
  grdma_conn* c = L->rx;
  uint8_t* const ring = c->ring;
  const uint64_t cap = c->cap, mask = cap - 1;
  const bool direct = L->direct != 0;
  lk_entry* const tab = L->tab[LK_SCATTER];
  uint8_t* const arena = L->arena;
  const uint64_t arena_cap = L->arena_cap;
  grdma_slice_out* const out_slices = L->out_slices;
  const uint64_t max_slices = L->slices_cap < (uint64_t)GRDMA_MAX_SLICES << 20 ? L->slices_cap : (uint64_t)GRDMA_MAX_SLICES << 20;

  // reader state (ring_buffer.h:203-205, pair.h:169, rdma_bp_posix.cc:63)
  uint64_t head = c->head, mh = c->moving_head, remain = c->remain, irs = c->internal_read_size;
  uint64_t leftover = c->leftover_cap;
  uint64_t nslices = 0, a_off = 0, bytes = 0, records = 0, credit_msgs = 0, credit_head = 0, rounds_with_data = 0;
  // chunks: emission batches with a completion counter each.  Lane j keeps the chunk in slot j.
  uint64_t spub = 0, sfloor = 0;            // scatter entries published / known complete
  uint64_t chunks = 0, chunks_retired = 0;
  uint32_t my_n = 0, my_flags = 0;          // flags: 1 = post a credit report after it, 2 = ends a round
  uint64_t my_credit = 0;
  uint64_t rounds_done = 0;                 // Sends drained with zero-fill complete and credits posted
  uint64_t wait_data = 0, wait_table = 0;
  bool failed = false;

  lk_walker w = {ring, cap, head, 0, c->rx_h2, c->rx_h1, 0};
  uint32_t chain_n = 0, chain_i = 0;

  // Publication is lazy: a sealed chunk's entries become visible to the scatter waves at the
  // next point where this wave has waited for memory anyway (its probe loads: everything issued
  // before them has been acknowledged by then), so that no step pays a round trip of its own.
  uint64_t spub_visible = 0;
  auto flush_publish = [&]() {
    if (spub_visible != spub) {
      drain();  // entries, slice table and my own tag clears are acknowledged
      stw(&L->published[LK_SCATTER].v, spub);
      spub_visible = spub;
    }
  };
  // retire finished chunks in order; post what they allow (updateStatus(), pair.cc:624-641: the
  // report must not overtake the copy-out and the zero-fill of the bytes it frees)
  auto service = [&](bool block) -> bool {
    if (block) flush_publish();
    while (chunks_retired < chunks) {
      const int s = (int)(chunks_retired % LK_RSLOTS);
      const uint32_t n = __shfl(my_n, s, 64);
      const uint32_t* d = &L->done_rx[s].v;
      if (ldw32(d) < n) {
        if (!block) return true;
        const uint64_t t0 = wall_clock64();
        if (!lk_spin(L, ticks, [&]() { return ldw32(d) >= n; })) return false;
        wait_table += wall_clock64() - t0;
      }
      const uint32_t fl = __shfl(my_flags, s, 64);
      const uint64_t ch = __shfl(my_credit, s, 64);
      if (fl & 1u) {
        grdma_status_report* ps = c->peer_status;
        if (ps != nullptr) __hip_atomic_store(&ps->remote_head, ch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (fl & 2u) {
        rounds_done++;
        drain();
        stw(&L->rx_rounds_done.v, rounds_done);
      }
      sfloor += n;
      chunks_retired++;
      block = false;  // one blocking step is enough for the caller to make room
    }
    return true;
  };

  // seal the entries emitted since the last chunk (possibly none) as one chunk
  uint32_t pend_entries = 0;
  auto close_chunk = [&](uint32_t flags, uint64_t credit_value) -> bool {
    if (pend_entries == 0 && flags == 0) return true;
    const int s = (int)(chunks % LK_RSLOTS);
    if (lane == s) {
      my_n = pend_entries;
      my_flags = flags;
      my_credit = credit_value;
    }
    spub += pend_entries;
    chunks++;
    pend_entries = 0;
    return true;
  };
  // a chunk slot's counter must be zero before its first entry is published (the store is
  // ordered before the publication by the drain in flush_publish)
  auto open_chunk = [&]() -> bool {
    while (chunks - chunks_retired >= LK_RSLOTS)
      if (!service(true)) return false;
    stw32(&L->done_rx[chunks % LK_RSLOTS].v, 0);
    return true;
  };
  auto table_room = [&](uint32_t need) -> bool {
    while (spub + pend_entries + need - sfloor > LK_TABLE_CAP) {
      // my own unsealed entries cannot complete: seal them first
      if (pend_entries && chunks_retired == chunks) {
        if (!close_chunk(0, 0) || !open_chunk()) return false;
        continue;
      }
      if (!service(true)) return false;
    }
    return true;
  };

  uint64_t credit_seen = 0;  // credit_msgs at the last chunk boundary

  uint32_t stall = 0;  // consecutive probe rounds that verified nothing (two are normal: header, then footer)
  auto refill = [&]() {
    while (chain_i == chain_n && w.limit > 0) {
      chain_n = lk_chain_round(&w, s_chain, lane);
      chain_i = 0;
      flush_publish();
      stall = chain_n ? 0 : stall + 1;
      if ((chain_n == 0 && w.e0 == 0) || stall > 3) {  // the ring does not hold what the descriptor promised
        stw(&L->abort.v, LK_ERR_CORRUPT);
        failed = true;
        return;
      }
    }
  };
  auto top_up = [&]() {
    while (chain_n - chain_i < 64 && w.limit > 0 && !failed) {
      const uint32_t kq = chain_n - chain_i;
      uint64_t keep = 0;
      if ((uint32_t)lane < kq) keep = s_chain[chain_i + lane];
      if ((uint32_t)lane < kq) s_chain[lane] = keep;
      chain_i = 0;
      chain_n = kq;
      const uint32_t got = lk_chain_round(&w, s_chain + kq, lane);
      chain_n += got;
      flush_publish();
      stall = got ? 0 : stall + 1;
      if ((got == 0 && w.e0 == 0) || stall > 3) {
        stw(&L->abort.v, LK_ERR_CORRUPT);
        failed = true;
      }
    }
  };
  auto next_ready = [&]() -> uint64_t {
    refill();
    return chain_i < chain_n ? s_chain[chain_i] : 0;
  };

  // PairPollable::Recv -> RingBufferPollable::Read(dst, capacity) (pair.cc:264-286,
  // ring_buffer.cc:122-191); returns the bytes copied.
  auto recv_step = [&](uint64_t dst, uint64_t capacity) -> uint64_t {
    uint64_t avail = remain;
    if (avail == 0) avail = next_ready();
    const uint64_t cpy = avail < capacity ? avail : capacity;
    if (cpy == 0 || failed) return 0;
    const uint64_t prev_mh = mh;
    if (remain == 0) {  // open the record, ring_buffer.cc:133-146
      if (lane == 0) stw(reinterpret_cast<uint64_t*>(ring + head), 0);  // clear header
      mh = (head + 8) & mask;
      head = (head + 16 + round_up8(avail)) & mask;
      records++;
      chain_i++;
    }
    const uint64_t l1 = cpy < cap - mh ? cpy : cap - mh;
    lk_piece pc[2];
    pc[0] = {0, 0, 0, 0, 0};
    pc[1] = {0, 0, 0, 0, 0};
    if (lane == 0) {
      pc[0] = {dst, (uint64_t)(ring + mh), l1, (uint32_t)GRDMA_SEG_ZERO_SRC, 0};
      if (cpy > l1) pc[1] = {dst + l1, (uint64_t)ring, cpy - l1, (uint32_t)GRDMA_SEG_ZERO_SRC, 0};
    }
    const uint32_t need = lk_count<2>(pc);
    if (!table_room(need)) { failed = true; return 0; }
    lk_emit<2>(tab, spub + pend_entries, pc, (uint32_t)(chunks % LK_RSLOTS), lane);
    pend_entries += need;
    mh = (mh + cpy) & mask;
    remain = avail - cpy;
    if (remain == 0) {  // finish the record, ring_buffer.cc:169-182
      const uint64_t pad_end = round_up8(mh);
      const __amdgpu_buffer_rsrc_t rp = mk_rsrc(uni64((uint64_t)ring + mh), uni32((uint32_t)(pad_end - mh)));
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)0, rp, lane, 0, LK_AUX_SC1);          // clear padded space
      if (lane == 0) stw(reinterpret_cast<uint64_t*>(ring + (pad_end & mask)), 0);        // clear footer
      mh = pad_end & mask;
      mh = (mh + 8) & mask;
    }
    const uint64_t consumed = (mh + cap - prev_mh) & mask;
    irs += consumed;  // credit return every cap / 2 consumed bytes, pair.cc:276-284
    if (irs >= cap / 2) {
      credit_head = mh;
      credit_msgs++;
      irs = 0;
    }
    return cpy;
  };

  // 64 whole records per step, one lane per record
  auto fast_chunk = [&]() -> uint32_t {
    top_up();
    if (failed) return 0;
    uint32_t kq = chain_n - chain_i;
    if (kq == 0) return 0;
    if (kq > 64) kq = 64;
    if (nslices + 128 > max_slices) return 0;
    const bool act0 = (uint32_t)lane < kq;
    const uint64_t n = act0 ? s_chain[chain_i + lane] : 0;
    const uint64_t resets = __ballot(act0 && n >= 2 * MINRD - 1);
    const uint64_t below = resets & ((1ull << lane) - 1ull);
    const uint32_t from = below ? (64 - __builtin_clzll(below)) : 0;
    uint32_t s_in = 0;
    for (uint32_t i = from; i < (uint32_t)lane && act0; i++) s_in = lk_read_space_after(s_chain[chain_i + i], s_in);
    const uint32_t s_out = lk_read_space_after(n, s_in);
    const uint64_t clean = __ballot(act0 && s_out == 0);
    if (clean == 0) return 0;
    const uint32_t cnt = 64 - __builtin_clzll(clean);
    const bool act = (uint32_t)lane < cnt;
    const uint32_t enc = act ? (uint32_t)(16 + round_up8(n)) : 0;
    lk_rec_plan rp = lk_replay_record(act ? n : 0, act ? s_in : 0);
    if (!act) { rp.c1 = rp.c2 = 0; rp.sl_cnt = 0; rp.sl0 = rp.sl1 = 0; }
    const uint32_t done_bytes = (uint32_t)(lk_al16(rp.sl0) + lk_al16(rp.sl1));
    const uint32_t i_enc = wave_incl_scan_u32(enc);
    const uint32_t i_bytes = wave_incl_scan_u32(done_bytes);
    const uint32_t i_n = wave_incl_scan_u32(act ? (uint32_t)n : 0);
    const uint64_t tot_n = __shfl(i_n, 63, 64);
    if (a_off + tot_n + 32ull * cnt + 512 > arena_cap) return 0;
    const uint64_t x_enc = i_enc - enc, x_bytes = i_bytes - done_bytes;
    const uint64_t pos = (head + x_enc) & mask;
    const uint64_t pay = (pos + 8) & mask;
    const uint64_t A = a_off + x_bytes;
    const uint64_t filled = s_in ? MINRD - s_in : 0;
    uint64_t o0, l0, o1, l1, o2, l2, o3, l3;
    lk_split_step(pay, 0, rp.c1, cap, &o0, &l0, &o1, &l1);
    lk_split_step(pay, rp.c1, rp.c2, cap, &o2, &l2, &o3, &l3);
    const uint32_t packed = rp.sl_cnt;
    const uint32_t i_packed = wave_incl_scan_u32(packed);
    const uint64_t x_slices = i_packed - rp.sl_cnt;
    // header, padding and footer (ring_buffer.cc:146,173-180) are cleared by the scatter waves of
    // the record's first / last piece
    lk_piece pc[4];
    {
      uint64_t dst = (uint64_t)arena + A + filled;  // the steps of one record are contiguous in the arena
      const int last_piece = l3 ? 3 : (l2 ? 2 : (l1 ? 1 : 0));
      const uint64_t offs[4] = {o0, o1, o2, o3}, lens[4] = {l0, l1, l2, l3};
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const uint32_t fl = (uint32_t)GRDMA_SEG_ZERO_SRC | (p == 0 ? (uint32_t)GRDMA_SEG_TAG_HDR : 0u) |
                            (p == last_piece ? (uint32_t)GRDMA_SEG_TAG_FTR : 0u);
        pc[p] = {dst, (uint64_t)(ring + offs[p]), act ? lens[p] : 0, fl, 0};
        dst += lens[p];
      }
    }
    const uint32_t need = lk_count<4>(pc);
    if (!table_room(need)) { failed = true; return 0; }
    lk_emit<4>(tab, spub + pend_entries, pc, (uint32_t)(chunks % LK_RSLOTS), lane);
    pend_entries += need;
    if (act) {
      uint64_t so = A;
      uint64_t xs = nslices + x_slices;
      if (rp.sl0) {
        out_slices[xs].off = so;
        out_slices[xs].len = rp.sl0;
        xs++;
        so += lk_al16(rp.sl0);
      }
      if (rp.sl1) {
        out_slices[xs].off = so;
        out_slices[xs].len = rp.sl1;
      }
    }
    // credit accounting over the Recv steps (pair.cc:276-284)
    const uint64_t pad_foot = round_up8(n) - n + 8;
    const uint64_t cons2 = act && rp.c2 ? rp.c2 + pad_foot : 0;
    const uint64_t mh1 = rp.c2 == 0 ? (pos + enc) & mask : (pay + rp.c1) & mask;
    const uint64_t mh2 = (pos + enc) & mask;
    const uint64_t C2 = i_enc;
    const uint64_t C1 = C2 - cons2;
    const uint64_t Ctot = __shfl(i_enc, 63, 64);
    uint64_t base = 0, thr = cap / 2 - irs;
    bool crossed = false;
    for (;;) {
      const uint64_t hit = __ballot(act && C2 >= thr);
      if (hit == 0) break;
      const int f = __builtin_ctzll(hit);
      const uint64_t fC1 = __shfl(C1, f, 64), fC2 = __shfl(C2, f, 64);
      const uint64_t fmh1 = __shfl(mh1, f, 64), fmh2 = __shfl(mh2, f, 64);
      const bool first = fC1 >= thr;
      credit_head = first ? fmh1 : fmh2;
      base = first ? fC1 : fC2;
      credit_msgs++;
      crossed = true;
      thr = base + cap / 2;
    }
    irs = crossed ? Ctot - base : irs + Ctot;
    head = (head + Ctot) & mask;
    mh = head;
    bytes += tot_n;
    records += cnt;
    nslices += __shfl(i_packed, 63, 64);
    a_off += __shfl(i_bytes, 63, 64);
    chain_i += cnt;
    return cnt;
  };

  // ---- one round per Send ---------------------------------------------------------------------
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  for (uint64_t r = 0; connected && !failed; r++) {
    // wait for Send r (or the end of the job)
    bool have = false;
    flush_publish();
    {
      const uint64_t t0 = wall_clock64();
      const bool ok = lk_spin(L, ticks, [&]() {
        service(false);
        if (ldw(&L->sends_pub.v) > r) { have = true; return true; }
        if (ldw(&L->tx_done.v) != 0) { have = ldw(&L->sends_pub.v) > r; return true; }
        return false;
      });
      wait_data += wall_clock64() - t0;
      if (!ok) { failed = true; break; }
    }
    if (!have) break;
    const uint32_t slot = (uint32_t)(r % LK_SLOTS);
    const lk_send_desc* d = &L->sends[slot];
    const uint64_t staged = ldw(&d->staged);
    const uint64_t nn = ldw(reinterpret_cast<const uint64_t*>(&d->n_gather));
    const uint32_t ng = (uint32_t)nn, nw = (uint32_t)(nn >> 32);
    if (ldw(&d->seq) != r + 1) { stw(&L->abort.v, LK_ERR_CORRUPT); failed = true; break; }
    {
      // the loop-back wire is a parallel copy: unlike an RC queue pair it does not deliver a
      // record's footer after its payload, so the receiver waits for the whole Send to land
      const uint32_t* dn = direct ? &L->done_tx[LK_GATHER][slot].v : &L->done_tx[LK_WIRE][slot].v;
      const uint32_t need = direct ? ng : nw;
      const uint64_t t0 = wall_clock64();
      const bool ok = lk_spin(L, ticks, [&]() {
        service(false);
        return ldw32(dn) >= need;
      });
      wait_data += wall_clock64() - t0;
      if (!ok) { failed = true; break; }
    }
    stw(&L->rx_sends_seen.v, r + 1);
    w.limit += staged;

    // drain: endpoint reads until one would block
    const uint64_t nslices0 = nslices;
    if (!open_chunk()) { failed = true; break; }
    for (;;) {
      if (failed) break;
      if (nslices >= max_slices) { stw(&L->abort.v, LK_ERR_SLICES); failed = true; break; }
      const bool clean = remain == 0 && leftover == 0;
      if (clean && fast_chunk() > 0) {
        // one chunk per step: its scatter starts while the next 64 records are being walked
        const bool cr = credit_msgs != credit_seen;
        credit_seen = credit_msgs;
        if (!close_chunk(cr ? 1u : 0u, credit_head) || !open_chunk()) { failed = true; break; }
        continue;
      }
      if (failed) break;
      // rdma_continue_read, rdma_bp_posix.cc:306-317
      uint64_t readable = remain;
      if (readable == 0) readable = next_ready();
      if (failed) break;
      const uint64_t alloc = leftover ? leftover : (readable > MINRD ? readable : MINRD);
      if (a_off + alloc > arena_cap) { stw(&L->abort.v, LK_ERR_ARENA); failed = true; break; }
      uint64_t total = 0;
      while (total < alloc) {  // rdma_do_read loop, rdma_bp_posix.cc:195-277
        const uint64_t n = recv_step((uint64_t)(arena + a_off + total), alloc - total);
        if (n == 0) break;
        total += n;
      }
      if (failed) break;
      if (total == 0) {  // nothing ready: notify_on_read, the slice stays allocated (:241-243)
        leftover = alloc;
        break;
      }
      leftover = alloc - total;  // grpc_slice_buffer_trim_end -> last_read_buffer
      if (lane == 0) {
        out_slices[nslices].off = a_off;
        out_slices[nslices].len = total;
      }
      nslices++;
      bytes += total;
      a_off = lk_al16(a_off + total);
    }
    if (failed) break;
    if (nslices != nslices0) rounds_with_data++;
    {
      const bool cr = credit_msgs != credit_seen;
      credit_seen = credit_msgs;
      if (!close_chunk((cr ? 1u : 0u) | 2u, credit_head)) { failed = true; break; }
    }
  }
  // everything sealed: publish, wait for the scatter waves, post the last reports
  flush_publish();
  while (!failed && chunks_retired < chunks)
    if (!service(true)) failed = true;
  drain();
  stw(&L->closed[LK_SCATTER].v, 1);
  if (lane == 0) {
    c->head = head;
    c->moving_head = mh;
    c->remain = remain;
    c->internal_read_size = irs;
    c->leftover_cap = leftover;
    c->total_read += bytes;
    c->credit_msgs += credit_msgs;
    c->rx_records += records;
    c->rx_rounds += rounds_with_data;
    c->rx_arena_off = a_off;
    c->rx_slice_idx = nslices;
    if (credit_msgs) c->status_send.remote_head = credit_head;
    c->rx_h1 = (uint32_t)w.h1;
    c->rx_h2 = (uint32_t)w.h2;
    L->res_chunks = chunks;
    L->res_entries[LK_SCATTER] = spub;
    L->res_wait_ticks[2] = wait_data;
    L->res_wait_ticks[3] = wait_table;
  }
}

// ---------------------------------------------------------------------------------------------
// k_link: grid (team, links).  Block 0 of a team is the sender's leader, block 1 the receiver's;
// the waves of the other blocks are workers, dealt to the three stages by position.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LK_THREADS) void k_link(lk_ctl* const* ctls, uint64_t timeout_ticks) {
  lk_ctl* L = ctls[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  __shared__ uint64_t s_chain[LK_CHAIN_CAP];
  if (blockIdx.x == 0) {
    if (wave == 0) lk_tx_leader(L, timeout_ticks, lane);
    return;
  }
  if (blockIdx.x == 1) {
    if (wave == 0) lk_rx_leader(L, timeout_ticks, s_chain, lane);
    return;
  }
  const uint32_t ww = (blockIdx.x - 2) * (LK_THREADS / 64) + wave;
  const uint32_t n0 = L->nwaves[LK_GATHER], n1 = L->nwaves[LK_WIRE], n2 = L->nwaves[LK_SCATTER];
  if (ww < n0) lk_worker(L, LK_GATHER, ww, n0, timeout_ticks, lane);
  else if (ww < n0 + n1) lk_worker(L, LK_WIRE, ww - n0, n1, timeout_ticks, lane);
  else if (ww < n0 + n1 + n2) lk_worker(L, LK_SCATTER, ww - n0 - n1, n2, timeout_ticks, lane);
}

}  // namespace

extern "C" {

__attribute__((visibility("hidden"))) hipError_t grdma_launch_link(lk_ctl* const* d_ctls, uint32_t nlinks, uint32_t team,
                                                                   uint64_t timeout_ticks, hipStream_t s) {
  if (nlinks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_link, dim3(team, nlinks), dim3(LK_THREADS), 0, s, d_ctls, timeout_ticks);
  return hipGetLastError();
}

// workgroups of k_link that are resident at once on the current device (the team sizes are cut
// to fit: a workgroup that is admitted but not resident would starve the others' spins)
__attribute__((visibility("hidden"))) uint32_t grdma_link_resident_blocks(void) {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_link, LK_THREADS, 0) != hipSuccess || per <= 0) return 0;
  return (uint32_t)(per * cus);
}

}  // extern "C"
