// k_link: the persistent link engine (see grdma_link.h for the roles and the hand-off rules).
//
// Parity: the engine is deterministic and produces exactly what the reference's loops produce
// when they run one after the other -- one Send from the rdma_flush cursor, then endpoint reads
// until one would block (the schedule tests/test_gpu_stream_job.py drives the CPU oracle with):
//  * the receiver drains Send by Send, each drain ending in the read that finds nothing and
//    keeps its slice (rdma_bp_posix.cc:241-243);
//  * in that schedule the receiver has drained everything before the next Send is priced, so the
//    peer's free space is cap - (bytes consumed since the last credit report) > cap / 2 = the
//    staging budget: the credit NEVER cuts a record there.  The sender therefore prices every
//    Send against the staging budget alone -- no waiting, any number of Sends ahead -- and only
//    holds a Send's WIRE back until the credit that has physically arrived covers its bytes.
// So records, delivered slices, credit reports and final state equal the sequential execution,
// while pricing, gather, wire, ring walk and scatter of neighbouring Sends overlap in time.
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

// Lanes of one wave that hand data to each other through memory (LDS tables, a word one lane
// stores and all lanes read) execute in lockstep, but the COMPILER reasons per thread: a load may
// be hoisted above another lane's store.  A wavefront-scope fence is a no-op in hardware and
// keeps the program order of the memory operations around it.
__device__ __forceinline__ void lk_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// first abort wins: remember where and with what
__device__ __forceinline__ void lk_abort(lk_ctl* L, uint64_t code, uint64_t site, uint64_t a, uint64_t b, uint64_t c2,
                                         uint64_t d) {
  if (__hip_atomic_load((const gu64*)(uint64_t)&L->abort.v, RLX_AGENT) == 0) {
    L->res_dbg[0] = code; L->res_dbg[1] = site; L->res_dbg[2] = a; L->res_dbg[3] = b; L->res_dbg[4] = c2; L->res_dbg[5] = d;
  }
  stw(&L->abort.v, code);
}

__device__ __forceinline__ void lk_trace(lk_ctl* L, int who, uint32_t& n, uint64_t tag, uint64_t arg, int lane) {
  if (lane == 0 && n < 192) L->trace[who][n] = (tag << 56) | ((arg & 0xFFFFull) << 40) | (wall_clock64() & 0xFFFFFFFFFFull);
  n++;
}

__device__ __forceinline__ void lk_publish(lk_ctl* L, int stage, uint64_t v) {
#pragma unroll
  for (uint32_t r = 0; r < LK_REPL; r++) stw(&L->published[stage][r].v, v);
}

// Bounded spin: polls cond() (relaxed loads), leaves on the abort word or on the wall clock.
template <typename F>
__device__ __forceinline__ bool lk_spin(lk_ctl* L, uint64_t limit_ticks, F cond) {
  uint32_t n = 0;
  uint64_t t0 = 0;
  for (;;) {
    if (cond()) return true;
    if ((++n & 15u) == 0) {
      if (ldw(&L->abort.v) != 0) return false;
      if ((n & 127u) == 0) {
        const uint64_t now = wall_clock64();
        if (t0 == 0) t0 = now;
        else if (now - t0 > limit_ticks) {
          stw(&L->abort.v, LK_ERR_TIMEOUT);
          return false;
        }
      }
    }
    if (n < 16u) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(20);
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
    wave_move_tile<LK_AUX_SC1, LK_AUX_SC1, ZERO, (int)(LK_TILE / 1024)>(e_dst + off, e_src + off, n, lane);
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
  const uint64_t* const pub = &L->published[stage][w % LK_REPL].v;
  const uint64_t* const closed = &L->closed[stage].v;
  const uint64_t* const abort_w = &L->abort.v;
  const __amdgpu_buffer_rsrc_t rtab = mk_rsrc(uni64((uint64_t)L->tab[stage]), (uint32_t)(sizeof(lk_entry) * LK_TABLE_CAP));
  uint64_t pub_seen = 0;  // published count read last (entries below it need no new poll)
  uint32_t trn = 0;
  const bool tracing = w == 0;
  bool pre = false;
  u32x4 p0 = {0, 0, 0, 0}, p1 = {0, 0, 0, 0};
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
        if ((++n & 7u) == 0) {   // the rare words are looked at every 8th poll only
          if (ldw(closed) != 0) {  // published is final once closed is set: read it again
            pub_seen = ldw(pub);
            if (pub_seen > e) break;
            if (tracing && lane == 0) L->trace_n[2 + stage] = trn < 192 ? trn : 192;
            return;
          }
          if (ldw(abort_w) != 0) return;
          if ((n & 127u) == 0) {
            const uint64_t now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > ticks) {
              stw(&L->abort.v, LK_ERR_TIMEOUT);
              return;
            }
          }
        }
        if (n < 8u) __builtin_amdgcn_s_sleep(6);
        else __builtin_amdgcn_s_sleep(24);
      }
    }
    // the whole 32-byte entry in one round trip (same address in every lane) -- unless it came
    // with the previous entry's data already
    u32x4 q0, q1;
    if (pre) {
      q0 = p0;
      q1 = p1;
    } else {
      const uint32_t eoff = (uint32_t)(e & (LK_TABLE_CAP - 1)) * (uint32_t)sizeof(lk_entry);
      q0 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff, 0, LK_AUX_SC1);
      q1 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff + 16, 0, LK_AUX_SC1);
    }
    {
      // my next entry, if it is published already: fetched under this entry's copy
      const uint64_t e2 = (lap + 1) * W + (w + W - (uint32_t)((lap + 1) % W)) % W;
      pre = pub_seen > e2;
      if (pre) {
        const uint32_t eoff2 = (uint32_t)(e2 & (LK_TABLE_CAP - 1)) * (uint32_t)sizeof(lk_entry);
        p0 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff2, 0, LK_AUX_SC1);
        p1 = __builtin_amdgcn_raw_buffer_load_b128(rtab, eoff2 + 16, 0, LK_AUX_SC1);
      }
    }
    const uint64_t e_dst = (uint64_t)q0.x | ((uint64_t)q0.y << 32), e_src = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
    const uint32_t e_len = q1.x, e_flags = q1.y;
    const uint64_t e_aux = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
    const uint32_t slot = (e_flags >> 8) & 0xFFu;
    if (tracing) lk_trace(L, 2 + stage, trn, 9, e, lane);
    if (stage == LK_SCATTER) lk_run_entry<true>(e_dst, e_src, e_len, e_flags, e_aux, tag_base, tag_mask, lane);
    else lk_run_entry<false>(e_dst, e_src, e_len, e_flags, e_aux, tag_base, tag_mask, lane);
    drain();  // my write-through stores are acknowledged: the entry may be counted
    if (lane == 0) {
      uint32_t* d = stage == LK_SCATTER ? &L->done_rx[slot].v : &L->done_tx[stage][slot].v;
      __hip_atomic_fetch_add((gu32*)(uint64_t)d, 1u, RLX_AGENT);
    }
    if (tracing) lk_trace(L, 2 + stage, trn, 11, e, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// entry emission: every lane brings up to NP pieces {dst, src, len, flags, aux}; pieces longer
// than LK_ENTRY_MAX are cut.  lk_count() first (table space), then lk_emit().
// ---------------------------------------------------------------------------------------------
// 16-byte write-through (sc1) store to a global address: the buffer builtins need a descriptor, so
// this one is inline asm; the trailing s_nop keeps the data registers intact until the store
// has read them (the compiler does not model the instruction)
__device__ __forceinline__ void lk_st16(__attribute__((address_space(1))) u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

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
  // two 16-byte write-through stores (8-byte sc1 stores cost 2.7x per byte and one fabric
  // write each); a reader takes the entry only after the published count covers it
  typedef __attribute__((address_space(1))) u32x4 g_q;
  g_q* q = (g_q*)(uint64_t)(tab + (idx & (LK_TABLE_CAP - 1)));
  const u32x4 a = {(uint32_t)dst, (uint32_t)(dst >> 32), (uint32_t)src, (uint32_t)(src >> 32)};
  const u32x4 b = {len, flags, (uint32_t)aux, (uint32_t)(aux >> 32)};
  lk_st16(q, a);
  lk_st16(q + 1, b);
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
// Leaders.  Both run on ONE wavefront (no cross-wave barriers inside a resident kernel), but
// every lane owns a contiguous RUN of up to 16 records per step, kept in LDS: the per-record
// work is a short serial loop in registers, and the wave-wide steps (DPP prefix sums, ballots)
// are paid once per 1024 records instead of once per 64.
// ---------------------------------------------------------------------------------------------
#define LK_RUN 16                  // records per lane per step
#define LK_STEP (64 * LK_RUN)      // records per step
#define LKP(i) ((i) + ((i) >> 4))  // LDS index padding: runs of 16 would otherwise share banks
#define LK_RX_STEP 4096            // records per step of the receiver (16 per thread of its leader block)
#define LK_QCAP 5120               // verified-record queue of the receiver (>= LK_RX_STEP + one probe round)
#define LK_PROBE_GROUPS 8          // x 64 speculative probes per memory round trip

struct lk_tx_lds {
  uint32_t len[LKP(LK_STEP) + 2];  // priced length of each slice of the step (clamped)
  uint32_t st[LKP(LK_STEP) + 2];   // where its record starts in the staging buffer
  uint64_t ptr[LKP(LK_STEP) + 2];
};
// The receiver's leader block works as one wave that drives and three that help: the driver posts
// a command in LDS, all four waves do their quarter of the records, each helper drains its stores
// and counts itself done.  (LDS words with workgroup-scope atomics; no s_barrier: the driver's
// control flow is far from uniform across the block.)
enum { LK_CMD_EXIT = 1, LK_CMD_STATE = 2, LK_CMD_TOTALS = 3, LK_CMD_EMIT = 4, LK_CMD_COUNT = 5 };
struct lk_rx_cmd {
  uint32_t seq;        // bumped last by the driver
  uint32_t type;
  uint32_t done;       // helpers that finished the current command
  uint32_t m, per, cnt, q_head, slot, head32;
  uint32_t clean_max;  // LK_CMD_STATE: max over threads of "records up to here end clean"
  uint64_t at0, xs0, a0;
  // LK_CMD_TOTALS: what each wave's quarter of the records makes (then, for LK_CMD_EMIT, the
  // exclusive prefixes over the waves)
  uint32_t w_bytes[4], w_sl[4], w_ent[4], w_n[4];
};
struct lk_rx_lds {
  lk_rx_cmd cmd;
  uint32_t t_enc[256], t_bytes[256], t_sl[256], t_ent[256], t_n[256];  // per-thread totals, then exclusive prefixes
  uint32_t hist[1024];               // LK_HCAP encoded sizes, the walker's history ring
  uint32_t q[LKP(LK_QCAP) + 2];      // payload sizes of verified records, [q_head, q_tail)
  uint32_t xenc[LKP(LK_RX_STEP) + 2];  // ring offset of each record of the step behind `head`
  uint16_t sin[LKP(LK_RX_STEP) + 2];   // space left in the open 256-byte read when the record starts
};
union lk_lds {
  lk_tx_lds tx;
  lk_rx_lds rx;
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

// ---------------------------------------------------------------------------------------------
// TX leader: PairPollable::Send (pair.cc:645-734) + the rdma_flush cursor
// (rdma_bp_posix.cc:476-493), one Send per iteration.  All records of a step are priced at once:
// enc_i = 16 + round_up8(len_i) prefix-summed, every record tests its own budget
// pay_i = min(len_i, W(S - st_i), W(free0 - st_i)) assuming the earlier ones went out whole, a
// ballot finds the first short record -- where the reference's loop stops.
// ---------------------------------------------------------------------------------------------
__device__ void lk_tx_leader(lk_ctl* L, uint64_t ticks, lk_tx_lds* D, int lane) {
  __builtin_amdgcn_s_setprio(3);  // the copy wave that shares this SIMD takes the slots I leave
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
  uint64_t my_staged = 0;            // encoded bytes of that Send (what its wire puts into the ring)
  // credit gate: Sends [gated, k) are priced (and gathered) but their ring writes are still held
  // back; rel_tail = ring offset behind everything released so far
  uint64_t gated = 0, rel_tail = c->remote_tail, rel_pub = 0;
  uint64_t wait_slot = 0, wait_credit = 0, t_price = 0, t_pub = 0;
  uint64_t tph[4] = {0, 0, 0, 0};  // profiling aid: load, price, count, emit
  uint32_t trn = 0;
  const uint64_t t_begin = wall_clock64();
  bool failed = false;

  // release the ring writes of priced Sends, in order, as far as the credit that has ARRIVED
  // covers them (the ring never fills up completely, so head == tail always means empty).
  // What is released: the wire entries -- or, when records are built in the peer ring itself,
  // the gather entries.
  const int gate_stage = L->direct ? LK_GATHER : LK_WIRE;
  auto release_ring_writes = [&]() {
    bool any = false;
    while (gated < k) {
      const int s = (int)(gated % LK_SLOTS);
      const uint64_t need = __shfl(my_staged, s, 64);
      const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const uint64_t freeb = cap - ((rel_tail + cap - rhead) & mask);
      if (freeb < need + 8) break;  // (every record of the Send passed W(free - st) >= pay against an even larger free space)
      // the wire reads staging: every gather entry of the Send must have been counted in.  (Checked
      // here by one wave, not by the wire waves: hundreds of pollers on a word that is also the
      // target of the gather waves' atomic adds starve those adds for tens of microseconds.)
      if (!direct && ldw32(&L->done_tx[LK_GATHER][s].v) < __shfl(my_ng, s, 64)) break;
      rel_pub += direct ? __shfl(my_ng, s, 64) : __shfl(my_nw, s, 64);
      rel_tail = (rel_tail + need) & mask;
      gated++;
      any = true;
    }
    if (any) {
      lk_publish(L, gate_stage, rel_pub);
      lk_trace(L, 0, trn, 3, gated, lane);
    }
  };

  auto retire_oldest = [&]() -> bool {
    // the oldest Send in flight has left its staging buffer: gathered and on the wire
    const int rs = (int)(retired % LK_SLOTS);
    const uint32_t ng = __shfl(my_ng, rs, 64), nw = __shfl(my_nw, rs, 64);
    const uint32_t* dg = &L->done_tx[LK_GATHER][rs].v;
    const uint32_t* dw = &L->done_tx[LK_WIRE][rs].v;
    const uint64_t t0 = wall_clock64();
    if (!lk_spin(L, ticks, [&]() {
          release_ring_writes();
          return ldw32(dg) >= ng && ldw32(dw) >= nw;
        })) return false;
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
        if (!lk_spin(L, ticks, [&]() {
              release_ring_writes();
              return ldw(&L->rx_sends_seen.v) >= need;
            })) { failed = true; break; }
      }
      uint8_t* const sbuf = direct ? nullptr : L->staging[k % B];
      uint64_t st_base = 0, nrec = 0, sent = 0, whole_records = 0, short_pay_total = 0;
      uint32_t ents = 0;
      const uint64_t tp0 = wall_clock64();
      release_ring_writes();
      {
        // priced against the staging budget alone (see the file header): free0 > S in the
        // sequential schedule, so min(S, free0) = S and W(free0 - st) >= W(S - st)
        const uint64_t room0 = S;
        // lengths are clamped for pricing: whatever exceeds the budget is short anyway
        const uint32_t clampv = (uint32_t)(room0 + 64);
        for (bool stop = false; !stop && !failed;) {
          if (nrec != 0) release_ring_writes();  // (a Send whose gather has finished meanwhile goes onto the wire)
          // ---- this step's slices -> LDS (striped loads, all in flight together)
          const uint64_t first = idx + nrec;
          uint64_t m64 = nslices - first;
          if (m64 > max_sge - nrec) m64 = max_sge - nrec;
          if (m64 > LK_STEP) m64 = LK_STEP;
          const uint32_t m = (uint32_t)m64;
          const uint64_t tq0 = wall_clock64();
          {
            grdma_sge g[LK_RUN];
#pragma unroll
            for (int r = 0; r < LK_RUN; r++) {
              const uint32_t i = r * 64 + lane;
              g[r] = slices[first + (i < m ? i : m - 1)];
            }
#pragma unroll
            for (int r = 0; r < LK_RUN; r++) {
              const uint32_t i = r * 64 + lane;
              uint64_t len = g[r].len, ptr = (uint64_t)g[r].ptr;
              if (nrec == 0 && i == 0) {  // the first slice continues at outgoing_byte_idx
                len = sat_sub(len, bidx);
                ptr += bidx;
              }
              if (i < m) {
                D->len[LKP(i)] = len < clampv ? (uint32_t)len : clampv;
                D->ptr[LKP(i)] = ptr;
              }
            }
          }
          lk_wave_sync();  // the table is written striped and read in runs
          const uint32_t per = (m + 63) / 64;
          const uint32_t k0 = lane * per, k1 = k0 + per < m ? k0 + per : m;
          const uint64_t tq1 = wall_clock64();
          // ---- where each run starts in the staging buffer (units of 8 bytes, saturating:
          // a run that alone exceeds the budget ends the Send inside it)
          uint64_t chunk = 0;
          for (uint32_t q = k0; q < k1; q++) chunk += enc_size(D->len[LKP(q)]);
          const uint32_t chunk8 = (uint32_t)((chunk < (uint64_t)clampv + 64 ? chunk : (uint64_t)clampv + 64) >> 3);
          const uint32_t incl8 = wave_incl_scan_u32(chunk8);
          const uint64_t st_run = st_base + ((uint64_t)(incl8 - chunk8) << 3);
          // ---- first short record of my run
          uint32_t my_short = 0xFFFFFFFFu;
          uint64_t st_short = 0;
          {
            uint64_t st = st_run;
            for (uint32_t q = k0; q < k1; q++) {
              const uint32_t l = D->len[LKP(q)];
              // a zero payload ends the send exactly like the reference's `break`
              if (l == 0 || l > writable_of(sat_sub(room0, st))) {
                my_short = q;
                st_short = st;
                break;
              }
              st += enc_size(l);
            }
          }
          const uint64_t bm = __ballot(my_short != 0xFFFFFFFFu);
          uint32_t take = m;
          uint64_t short_pay = 0;
          if (bm) {
            const int f = __builtin_ctzll(bm);
            take = __shfl(my_short, f, 64);
            const uint64_t stf = __shfl(st_short, f, 64);
            const uint64_t lf = D->len[LKP(take)];
            const uint64_t a = writable_of(sat_sub(S, stf));
            short_pay = lf < a ? lf : a;
          }
          const uint32_t nr = take + (short_pay ? 1u : 0u);  // records of this step
          const uint64_t tq2 = wall_clock64();
          // ---- entries: count, make room, emit
          uint32_t my_cnt = 0, my_sent = 0, my_enc = 0;
          for (uint32_t q = k0; q < k1 && q < nr; q++) {
            const uint32_t pay = q < take ? D->len[LKP(q)] : (uint32_t)short_pay;
            D->st[LKP(q)] = (uint32_t)st_run + my_enc;
            my_cnt += lk_sub_entries(pay);
            my_sent += pay;
            my_enc += (uint32_t)enc_size(pay);
          }
          if (direct) {
            // a record that crosses the ring end is cut there: one more entry (at most one record per Send)
            uint64_t st = st_run;
            for (uint32_t q = k0; q < k1 && q < nr; q++) {
              const uint32_t pay = q < take ? D->len[LKP(q)] : (uint32_t)short_pay;
              const uint64_t po = (tail + st + 8) & mask;
              if (po + pay > cap) my_cnt += lk_sub_entries(cap - po) + lk_sub_entries(pay - (cap - po)) - lk_sub_entries(pay);
              st += enc_size(pay);
            }
          }
          const uint32_t i_cnt = wave_incl_scan_u32(my_cnt);
          const uint32_t n_new = (uint32_t)__builtin_amdgcn_readlane((int)i_cnt, 63);
          while (gpub + ents + n_new - gfloor > LK_TABLE_CAP)
            if (!retire_oldest()) { failed = true; break; }
          if (failed) break;
          const uint64_t tq3 = wall_clock64();
          if (n_new == nr && !direct) {
            // the common case, one entry per record: dealt to the lanes round-robin, so that
            // neighbouring lanes write neighbouring table entries (coalesced 16-byte stores)
            lk_wave_sync();
            const uint32_t fl = (uint32_t)(GRDMA_SEG_TAG_WRITE | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR) | (slot << 8);
#pragma unroll 4
            for (uint32_t q = lane; q < nr; q += 64) {
              const uint32_t pay = q < take ? D->len[LKP(q)] : (uint32_t)short_pay;
              lk_store_entry(gtab, gpub + ents + q, (uint64_t)(sbuf + D->st[LKP(q)] + 8), D->ptr[LKP(q)], pay, fl, pay);
            }
          } else {
            uint64_t st = st_run;
            uint64_t at = gpub + ents + i_cnt - my_cnt;
            const uint32_t tagw = (uint32_t)GRDMA_SEG_TAG_WRITE | (slot << 8);
            for (uint32_t q = k0; q < k1 && q < nr; q++) {
              const uint32_t pay = q < take ? D->len[LKP(q)] : (uint32_t)short_pay;
              const uint64_t src = D->ptr[LKP(q)];
              // the payload goes behind the 8-byte header at staging + st (or straight into the
              // peer ring, where it may cross the ring end)
              uint64_t dst0;
              uint32_t l0 = pay, l1 = 0;
              if (direct) {
                const uint64_t po = (tail + st + 8) & mask;
                dst0 = (uint64_t)(peer_ring + po);
                if (po + pay > cap) {
                  l0 = (uint32_t)(cap - po);
                  l1 = pay - l0;
                }
              } else {
                dst0 = (uint64_t)(sbuf + st + 8);
              }
              // piece 0 (and piece 1 behind the ring end), each cut into entries
              for (int pi = 0; pi < 2; pi++) {
                const uint32_t pl = pi ? l1 : l0;
                if (pl == 0) continue;
                const uint64_t pd = pi ? (uint64_t)peer_ring : dst0, ps = pi ? src + l0 : src;
                const uint32_t nsub = lk_sub_entries(pl);
                for (uint32_t j = 0; j < nsub; j++) {
                  const uint32_t o = j * LK_ENTRY_MAX;
                  const uint32_t len = pl - o < LK_ENTRY_MAX ? pl - o : LK_ENTRY_MAX;
                  uint32_t fl = tagw;
                  if (pi == 0 && j == 0) fl |= (uint32_t)GRDMA_SEG_TAG_HDR;
                  if ((pi == 1 || l1 == 0) && j == nsub - 1) fl |= (uint32_t)GRDMA_SEG_TAG_FTR;
                  lk_store_entry(gtab, at, pd + o, ps + o, len, fl, pay);
                  at++;
                }
              }
              st += enc_size(pay);
            }
          }
          ents += n_new;
          lk_wave_sync();  // (the next step overwrites the table)
          {
            const uint64_t tq4 = wall_clock64();
            tph[0] += tq1 - tq0; tph[1] += tq2 - tq1; tph[2] += tq3 - tq2; tph[3] += tq4 - tq3;
          }
          st_base += wave_sum_u32(my_enc);
          sent += wave_sum_u32(my_sent);
          nrec += nr;
          whole_records += take;
          short_pay_total = short_pay;
          stop = bm != 0 || nrec >= max_sge || idx + nrec >= nslices;
        }
      }
      if (failed) break;
      if (nrec == 0) {  // nothing fits an empty staging buffer: a zero-length slice at the cursor
        stw(&L->abort.v, LK_ERR_NO_PROGRESS);
        failed = true;
        break;
      }
      const uint64_t staged = st_base;
      const uint64_t tp1 = wall_clock64();
      t_price += tp1 - tp0;
      lk_trace(L, 0, trn, 1, k, lane);
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
      if (lane == (int)slot) {
        my_ng = ents;
        my_nw = nw;
        my_staged = staged;
      }
      if (!direct) lk_publish(L, LK_GATHER, gpub);  // staging is mine: gather right away
      stw(&L->sends_pub.v, k + 1);
      lk_trace(L, 0, trn, 2, k, lane);
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
      release_ring_writes();
      t_pub += wall_clock64() - tp1;
    }
    // every Send is priced: release the rest as the credit comes in
    if (!failed && gated < k) {
      const uint64_t t0 = wall_clock64();
      if (!lk_spin(L, ticks, [&]() {
            release_ring_writes();
            return gated >= k;
          })) failed = true;
      wait_credit += wall_clock64() - t0;
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
    L->res_prof[0] = t_price;
    L->res_prof[1] = t_pub;
    L->res_prof[2] = wall_clock64() - t_begin;
    for (int q = 0; q < 4; q++) L->res_tx_phases[q] = tph[q];
    L->trace_n[0] = trn < 192 ? trn : 192;
  }
}

// ---------------------------------------------------------------------------------------------
// RX leader: see the file header.  Speculative probes of the record chain (LK_PROBE_GROUPS x 64
// per memory round trip) fill a queue of verified record sizes; up to 1024 whole records per
// step are replayed through the endpoint-read state machine (a record of >= 511 bytes resets the
// state, so every lane recovers the incoming state of its run by looking back to the nearest
// such record); partial records and tails take scalar reads.  Everything is bounded by what the
// wire has completely delivered.
// ---------------------------------------------------------------------------------------------
#define LK_HCAP 1024  // encoded sizes of the last verified records (the predictor's history)

struct lk_walker {
  const uint8_t* ring;
  uint64_t cap;
  uint64_t pos;       // ring offset of the first unverified record
  uint64_t e0;        // encoded size of the record at pos when its header is already known
  uint64_t limit;     // bytes behind pos that have completely landed
  uint64_t hcount;    // records verified so far (history index of the record at pos)
  uint32_t period;    // detected period of the record sizes (0: none, the last two sizes alternate)
  uint64_t retry_at;  // no new period search before this many records were verified
};

// The record chain is a linked list, but on a gRPC connection the sizes repeat: inside a message
// a 9-byte frame header alternates with a 16 KiB payload, and messages of one size repeat the
// whole run.  Finds the smallest period P <= 510 for which the newest P + 4 sizes (the header
// just read at pos included) equal the ones P earlier; lane-parallel over the candidates.
__device__ __forceinline__ uint32_t lk_find_period(const uint32_t* hist, uint64_t tt, int lane) {
  const uint64_t have = tt < LK_HCAP ? tt : LK_HCAP;
  uint32_t best = 0xFFFFFFFFu;
  for (uint32_t c = 0; c < 8; c++) {
    const uint32_t P = c * 64 + lane + 1;
    if (2ull * P + 4 > have) continue;
    bool ok = true;
    for (uint32_t j = 0; j < P + 4 && ok; j++)
      ok = hist[(tt - 1 - j) % LK_HCAP] == hist[(tt - 1 - j - P) % LK_HCAP];
    if (ok && P < best) best = P;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(best, d, 64);
    best = o < best ? o : best;
  }
  return best == 0xFFFFFFFFu ? 0 : best;
}

// One probe round.  Probe i = 64 g + lane loads the tag words at the offset the chain reaches
// after i records if the sizes keep repeating with the known period; every probe checks its own
// link and the footer in front of it; ballots give the verified prefix.  Appends the payload
// sizes to q[LKP(qt + 0 .. v)), returns v.
__device__ __forceinline__ uint32_t lk_chain_round(lk_walker* w, uint32_t* q, uint32_t qt, uint32_t* hist, int lane) {
  constexpr int K = LK_PROBE_GROUPS;
  const uint64_t cap = w->cap, mask = cap - 1;
  const uint64_t lim = w->limit;
  if (lim == 0) return 0;
  const uint64_t e0 = w->e0;
  const uint64_t t = w->hcount;
  // a header already read counts as the newest history entry
  if (e0 && lane == 0) hist[t % LK_HCAP] = (uint32_t)e0;
  lk_wave_sync();
  const uint64_t tt = t + (e0 ? 1 : 0);
  const uint32_t i0 = e0 ? 1u : 0u;
  uint32_t P = w->period;
  if (P == 0 || P > tt) P = tt >= 2 ? 2u : (uint32_t)tt;
  const bool have_pattern = P != 0;
  // predicted encoded size of probe i, then its offset behind pos: prefix sums over the 8 groups
  // (in units of 8 bytes, saturated just above the limit: a probe beyond the limit is not used)
  const uint32_t sat8 = (uint32_t)((lim >> 3) + 2);
  uint64_t rel[K], reln[K];
  {
    uint64_t carry = 0;
#pragma unroll
    for (int g = 0; g < K; g++) {
      const uint32_t i = 64 * g + lane;
      uint32_t pe = 0;
      if (i < i0) pe = (uint32_t)e0;
      else if (have_pattern) pe = hist[(tt - P + ((i - i0) % P)) % LK_HCAP];
      uint32_t p8 = pe >> 3;
      if (p8 > sat8) p8 = sat8;
      const uint32_t incl = wave_incl_scan_u32(p8);
      rel[g] = (carry + incl - p8) << 3;
      reln[g] = (carry + incl) << 3;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  uint64_t hdr[K], prev[K];
  // all 2 K tag loads of a lane are in flight together: unconditional (a masked offset is always
  // inside the ring), the conditions are applied to what comes back
#pragma unroll
  for (int g = 0; g < K; g++) {
    const uint64_t my_pos = (w->pos + rel[g]) & mask & ~7ull;
    hdr[g] = ld_tag(w->ring + my_pos);
    prev[g] = ld_tag(w->ring + ((my_pos + cap - 8) & mask));  // footer of the record before
  }
  uint32_t v = 64 * K - 1;
  uint64_t foot_next0 = 0;  // footer of this group's record 63 = first footer bit of the next group
  uint64_t m_good[K], m_valid[K];
#pragma unroll
  for (int g = K - 1; g >= 0; g--) {
    const uint32_t i = 64 * g + lane;
    const bool known = i < i0 || have_pattern;  // this probe's size is predicted (or known)
    // only records that end inside what has landed are looked at
    const bool ph = (i == 0) || (known && reln[g] <= lim);
    const bool pp = i > 0 && rel[g] <= lim && (i - 1 < i0 || have_pattern);
    const bool valid = ph && hdr[g] != 0 && hdr[g] <= cap - GRDMA_RESERVED;
    const uint64_t enc = 16 + round_up8(hdr[g]);
    const bool link_ok = valid && known && !(g == K - 1 && lane == 63) && enc == reln[g] - rel[g] && reln[g] <= lim;
    const uint64_t m_link = __ballot(link_ok);
    const uint64_t m_fp = __ballot(pp && prev[g] == GRDMA_FOOTER);
    const uint64_t m_foot = (m_fp >> 1) | (foot_next0 << 63);  // bit j: footer of record 64 g + j
    foot_next0 = m_fp & 1;
    m_good[g] = m_link & m_foot;
    m_valid[g] = __ballot(valid);
  }
#pragma unroll
  for (int g = K - 1; g >= 0; g--)
    if (m_good[g] != ~0ull) v = 64 * g + (uint32_t)__builtin_ctzll(~m_good[g]);
  // sizes of the verified records; what the walker needs from record v
  uint64_t enc_v = 0, rel_v = 0;
  bool v_valid = false;
#pragma unroll
  for (int g = 0; g < K; g++) {
    const uint32_t i = 64 * g + lane;
    const uint32_t enc = 16 + (uint32_t)round_up8(hdr[g] & 0xFFFFFFFFull);
    if (i < v) {
      q[LKP(qt + i)] = (uint32_t)hdr[g];
      hist[(t + i) % LK_HCAP] = enc;
    }
    const uint32_t lo = 64 * g;
    if (v >= lo && v < lo + 64) {  // uniform
      enc_v = (uint32_t)__builtin_amdgcn_readlane((int)enc, (int)(v - lo));
      rel_v = (uint64_t)__builtin_amdgcn_readlane((int)(uint32_t)(rel[g] >> 3), (int)(v - lo)) << 3;
      v_valid = (m_valid[g] >> (v - lo)) & 1;
    }
  }
  lk_wave_sync();
  w->pos = (w->pos + rel_v) & mask;
  w->limit = lim - rel_v;
  w->hcount = t + v;
  // record v is the first unverified one: when its header was read and is a record header, its
  // exact footer position is probed next round (as probe 0)
  w->e0 = v_valid ? enc_v : 0;
  // a round that stopped on a size it did not predict, with more data behind it: look for a
  // (longer) period, unless a search failed recently
  if (w->e0 && v < 64 * K - 1 && w->limit > 0 && w->hcount >= w->retry_at) {
    lk_wave_sync();
    if (lane == 0) hist[w->hcount % LK_HCAP] = (uint32_t)w->e0;
    lk_wave_sync();
    const uint32_t np = lk_find_period(hist, w->hcount + 1, lane);
    w->period = np;
    if (np == 0) w->retry_at = w->hcount + 256;
  }
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
  uint32_t c1, c2;    // bytes of the (at most) two Recv steps
  uint32_t sl0, sl1;  // lengths of the slices completed by this record, in order (0 = none)
  uint32_t sl_cnt;
};

__device__ __forceinline__ lk_rec_plan lk_replay_record(uint32_t n, uint32_t s_in) {
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

__device__ __forceinline__ uint64_t lk_al16(uint64_t v) { return (v + 15) & ~15ull; }
__device__ __forceinline__ uint32_t lk_al16_32(uint32_t v) { return (v + 15u) & ~15u; }

// the ring pieces of one Recv step of `len` bytes at payload offset `off`: (o0, l0) and, when the
// step crosses the ring end, (0, l1); then cut into entries
__device__ __forceinline__ uint32_t lk_step_entries(uint32_t pay, uint32_t off, uint32_t len, uint32_t cap) {
  if (len == 0) return 0;
  const uint32_t p0 = (pay + off) & (cap - 1);
  const uint32_t first = len < cap - p0 ? len : cap - p0;
  return lk_sub_entries(first) + lk_sub_entries(len - first);
}

// ---- the block-wide parts of a bulk step (the command's arguments come by value: the driver
// passes what it posted, a helper what it read behind its acquire of the sequence word).
//
// STATE: thread T of 256 owns the contiguous run [T * per, (T + 1) * per) of the m queued records:
// incoming read state of every record (look back to the nearest record that resets it, >= 511
// bytes), the last record after which the state is clean, and the encoded bytes of the run.
__device__ __forceinline__ void lk_rx_part_state(lk_rx_lds* D, const lk_rx_cmd& cm, uint32_t T) {
  const uint32_t m = cm.m, per = cm.per, qh = cm.q_head;
  const uint32_t k0 = T * per < m ? T * per : m, k1 = k0 + per < m ? k0 + per : m;
  uint32_t t_enc = 0, last_clean = 0;
  if (k0 < k1) {
    uint32_t j = k0;
    while (j > 0 && D->q[LKP(qh + j - 1)] < 2 * MINRD - 1) j--;
    uint32_t sp = 0;
    for (; j < k0; j++) sp = lk_read_space_after(D->q[LKP(qh + j)], sp);
    for (uint32_t q = k0; q < k1; q++) {
      const uint32_t n = D->q[LKP(qh + q)];
      D->sin[LKP(q)] = (uint16_t)sp;
      sp = lk_read_space_after(n, sp);
      if (sp == 0) last_clean = q + 1;
      t_enc += 16u + (uint32_t)round_up8(n);
    }
  }
  D->t_enc[T] = t_enc;
  if (last_clean) __hip_atomic_fetch_max(&D->cmd.clean_max, last_clean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// XENC: ring offset of every record behind `head`, from the exclusive prefix of the runs in t_enc
__device__ __forceinline__ void lk_rx_part_xenc(lk_rx_lds* D, const lk_rx_cmd& cm, uint32_t T) {
  const uint32_t m = cm.m, per = cm.per, qh = cm.q_head;
  const uint32_t k0 = T * per < m ? T * per : m, k1 = k0 + per < m ? k0 + per : m;
  uint32_t x = D->t_enc[T];
  for (uint32_t q = k0; q < k1; q++) {
    D->xenc[LKP(q)] = x;
    x += 16u + (uint32_t)round_up8(D->q[LKP(qh + q)]);
  }
  if (k1 == m && k0 < k1) D->xenc[LKP(m)] = x;
}

// the range of the cnt processed records that wave w works on, lane per record: contiguous
// quarters, multiples of 64 (neighbouring lanes then write neighbouring table entries)
__device__ __forceinline__ void lk_rx_wave_range(uint32_t cnt, uint32_t w, uint32_t* beg, uint32_t* end) {
  const uint32_t wchunk = (((cnt + 3) / 4) + 63u) & ~63u;
  *beg = w * wchunk < cnt ? w * wchunk : cnt;
  *end = *beg + wchunk < cnt ? *beg + wchunk : cnt;
}

// what one record makes: entries (pieces of at most LK_ENTRY_MAX bytes, cut at the ring end)
__device__ __forceinline__ uint32_t lk_rec_entries(const lk_rec_plan& rp, uint32_t pay, uint32_t cap32) {
  return lk_step_entries(pay, 0, rp.c1, cap32) + lk_step_entries(pay, rp.c1, rp.c2, cap32);
}

// TOTALS: per wave, over its range
__device__ __forceinline__ void lk_rx_part_totals(lk_rx_lds* D, const lk_rx_cmd& cm, uint32_t w, int lane, uint32_t cap32) {
  uint32_t wbeg, wend;
  lk_rx_wave_range(cm.cnt, w, &wbeg, &wend);
  const uint32_t qh = cm.q_head, head32 = cm.head32, mask32 = cap32 - 1;
  uint32_t t_bytes = 0, t_pk = 0, t_n = 0;  // t_pk: slices | entries << 12 (per lane: <= 16 records of a 4096-record step per wave... summed below in 32 bits)
  uint32_t t_sl = 0, t_ent = 0;
  for (uint32_t k = wbeg + lane; k < wend; k += 64) {
    const uint32_t n = D->q[LKP(qh + k)];
    const lk_rec_plan rp = lk_replay_record(n, D->sin[LKP(k)]);
    const uint32_t pay = (head32 + D->xenc[LKP(k)] + 8u) & mask32;
    t_ent += lk_rec_entries(rp, pay, cap32);
    t_sl += rp.sl_cnt;
    t_bytes += lk_al16_32(rp.sl0) + lk_al16_32(rp.sl1);
    t_n += n;
  }
  (void)t_pk;
  t_bytes = wave_sum_u32(t_bytes);
  t_sl = wave_sum_u32(t_sl);
  t_ent = wave_sum_u32(t_ent);
  t_n = wave_sum_u32(t_n);
  if (lane == 0) {
    D->cmd.w_bytes[w] = t_bytes;
    D->cmd.w_sl[w] = t_sl;
    D->cmd.w_ent[w] = t_ent;
    D->cmd.w_n[w] = t_n;
  }
}

// EMIT: entries and slices, 64 records per iteration, one lane per record; the running offsets
// come from three DPP prefix sums per iteration
__device__ __forceinline__ void lk_rx_part_emit(lk_rx_lds* D, const lk_rx_cmd& cm, uint32_t w, int lane, lk_entry* tab,
                                                uint8_t* ring, uint8_t* arena, grdma_slice_out* out_slices, uint32_t cap32) {
  uint32_t wbeg, wend;
  lk_rx_wave_range(cm.cnt, w, &wbeg, &wend);
  const uint32_t qh = cm.q_head, head32 = cm.head32, mask32 = cap32 - 1;
  uint64_t c_ent = cm.at0 + cm.w_ent[w];    // running prefixes of this wave
  uint64_t c_sl = cm.xs0 + cm.w_sl[w];
  uint64_t c_bytes = cm.a0 + cm.w_bytes[w];
  const uint32_t slot = cm.slot << 8;
  for (uint32_t base = wbeg; base < wend; base += 64) {
    const uint32_t k = base + lane;
    const bool act = k < wend;
    const uint32_t n = act ? D->q[LKP(qh + k)] : 0;
    const uint32_t s_in = act ? D->sin[LKP(k)] : 0;
    lk_rec_plan rp = lk_replay_record(n, s_in);
    if (!act) { rp.c1 = rp.c2 = 0; rp.sl_cnt = 0; rp.sl0 = rp.sl1 = 0; }
    const uint32_t pay = (head32 + (act ? D->xenc[LKP(k)] : 0) + 8u) & mask32;
    const uint32_t my_ent = act ? lk_rec_entries(rp, pay, cap32) : 0;
    const uint32_t my_bytes = lk_al16_32(rp.sl0) + lk_al16_32(rp.sl1);
    const uint32_t i_ent = wave_incl_scan_u32(my_ent);
    const uint32_t i_sl = wave_incl_scan_u32(rp.sl_cnt);
    const uint32_t i_bytes = wave_incl_scan_u32(my_bytes);
    if (act) {
      uint64_t at = c_ent + i_ent - my_ent;
      uint64_t xs = c_sl + i_sl - rp.sl_cnt;
      const uint64_t A = c_bytes + i_bytes - my_bytes;  // start of the open / next slice
      const uint32_t filled = s_in ? MINRD - s_in : 0;
      // the steps of one record are contiguous in the arena: step 1 fills the open 256-byte
      // slice exactly, step 2 starts the next slice right behind it.  Header, padding and
      // footer (ring_buffer.cc:146,173-180) are cleared by the scatter waves of the record's
      // first / last entry.
      uint64_t dst = (uint64_t)arena + A + filled;
      uint32_t emitted = 0;
      for (int stp = 0; stp < 2; stp++) {
        const uint32_t off = stp ? rp.c1 : 0, len = stp ? rp.c2 : rp.c1;
        if (len == 0) continue;
        const uint32_t p0 = (pay + off) & mask32;
        const uint32_t first = len < cap32 - p0 ? len : cap32 - p0;
        for (int pi = 0; pi < 2; pi++) {
          const uint32_t pl = pi ? len - first : first;
          if (pl == 0) continue;
          const uint64_t src = (uint64_t)ring + (pi ? 0u : p0);
          const uint32_t nsub = lk_sub_entries(pl);
          for (uint32_t j = 0; j < nsub; j++) {
            const uint32_t o = j * LK_ENTRY_MAX;
            const uint32_t l = pl - o < LK_ENTRY_MAX ? pl - o : LK_ENTRY_MAX;
            uint32_t fl = (uint32_t)GRDMA_SEG_ZERO_SRC | slot;
            if (emitted == 0) fl |= (uint32_t)GRDMA_SEG_TAG_HDR;
            if (emitted == my_ent - 1) fl |= (uint32_t)GRDMA_SEG_TAG_FTR;
            lk_store_entry(tab, at, dst + o, src + o, l, fl, 0);
            at++;
            emitted++;
          }
          dst += pl;
        }
      }
      uint64_t sof = A;
      if (rp.sl0) {
        out_slices[xs].off = sof;
        out_slices[xs].len = rp.sl0;
        xs++;
        sof += lk_al16_32(rp.sl0);
      }
      if (rp.sl1) {
        out_slices[xs].off = sof;
        out_slices[xs].len = rp.sl1;
      }
    }
    c_ent += (uint32_t)__builtin_amdgcn_readlane((int)i_ent, 63);
    c_sl += (uint32_t)__builtin_amdgcn_readlane((int)i_sl, 63);
    c_bytes += (uint32_t)__builtin_amdgcn_readlane((int)i_bytes, 63);
  }
}

#define LK_WG_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

// waves 1..3 of the receiver's leader block
__device__ void lk_rx_helper(lk_ctl* L, lk_rx_lds* D, uint32_t wave, int lane) {
  __builtin_amdgcn_s_setprio(3);
  const uint32_t T = wave * 64 + lane;
  uint8_t* const ring = L->rx->ring;
  const uint32_t cap32 = (uint32_t)L->rx->cap;
  uint32_t seen = 0;
  for (;;) {
    uint32_t sq;
    uint32_t spins = 0;
    while ((sq = __hip_atomic_load(&D->cmd.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == seen) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 0xFFFFu) == 0 && ldw(&L->abort.v) != 0) return;
    }
    seen = sq;
    const lk_rx_cmd cm = D->cmd;  // (behind the acquire)
    const uint32_t type = cm.type;
    if (type == LK_CMD_EXIT) return;
    if (type == LK_CMD_STATE) lk_rx_part_state(D, cm, T);
    else if (type == LK_CMD_COUNT) lk_rx_part_xenc(D, cm, T);
    else if (type == LK_CMD_TOTALS) lk_rx_part_totals(D, cm, wave, lane, cap32);
    else if (type == LK_CMD_EMIT) lk_rx_part_emit(D, cm, wave, lane, L->tab[LK_SCATTER], ring, L->arena, L->out_slices, cap32);
    drain();  // my stores (entries, slice table) are acknowledged before I count as done
    if (lane == 0) __hip_atomic_fetch_add(&D->cmd.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

__device__ void lk_rx_leader(lk_ctl* L, uint64_t ticks, lk_rx_lds* D, int lane) {
  __builtin_amdgcn_s_setprio(3);
  grdma_conn* c = L->rx;
  uint8_t* const ring = c->ring;
  const uint64_t cap = c->cap, mask = cap - 1;
  const uint32_t cap32 = (uint32_t)cap, mask32 = (uint32_t)mask;  // (rings in the engine are at most 256 MiB)
  const bool direct = L->direct != 0;
  lk_entry* const tab = L->tab[LK_SCATTER];
  uint8_t* const arena = L->arena;
  const uint64_t arena_cap = L->arena_cap;
  grdma_slice_out* const out_slices = L->out_slices;
  const uint64_t max_slices = L->slices_cap;

  // reader state (ring_buffer.h:203-205, pair.h:169, rdma_bp_posix.cc:63)
  uint64_t head = c->head, mh = c->moving_head, remain = c->remain, irs = c->internal_read_size;
  uint64_t leftover = c->leftover_cap;
  uint64_t nslices = 0, a_off = 0, bytes = 0, records = 0, credit_msgs = 0, credit_head = 0, rounds_with_data = 0;
  // chunks: emission batches with a completion counter each.  Lane j keeps the chunk in slot j.
  uint64_t spub = 0, sfloor = 0;            // scatter entries sealed / known complete
  uint64_t chunks = 0, chunks_retired = 0;
  uint32_t my_n = 0, my_flags = 0;          // flags: 1 = post a credit report after it, 2 = ends a round
  uint64_t my_credit = 0;
  uint64_t rounds_done = 0;                 // Sends drained with zero-fill complete and credits posted
  uint64_t wait_data = 0, wait_table = 0, t_walk = 0, t_fast = 0, t_scalar = 0, t_emit = 0;
  uint32_t trn = 0;
  const uint64_t t_begin = wall_clock64();
  bool failed = false;

  lk_walker w = {ring, cap, head, 0, 0, 0, 0, 0};
  // (the sizes of the two newest records of an earlier call seed the history)
  if (c->rx_h1) {
    if (lane == 0) {
      D->hist[0] = c->rx_h2 ? c->rx_h2 : c->rx_h1;
      D->hist[1] = c->rx_h1;
    }
    lk_wave_sync();
    w.hcount = 2;
  }
  uint32_t q_head = 0, q_tail = 0;  // verified records not yet consumed: D->q[LKP(q_head .. q_tail))

  // Publication is lazy: a sealed chunk's entries become visible to the scatter waves at the
  // next point where this wave has waited for memory anyway (its probe loads: everything issued
  // before them has been acknowledged by then), so that no step pays a round trip of its own.
  uint64_t spub_visible = 0;
  auto flush_publish = [&]() {
    if (spub_visible != spub) {
      drain();  // entries, slice table and my own tag clears are acknowledged
      lk_publish(L, LK_SCATTER, spub);
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
        if (c->peer_line != nullptr && lane == 0)  // the sender's host-visible state line (grdma_hostline)
          __hip_atomic_store(&c->peer_line->remote_head, ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lk_trace(L, 1, trn, 7, chunks_retired, lane);
      }
      if (fl & 2u) {
        rounds_done++;
        drain();
        stw(&L->rx_rounds_done.v, rounds_done);
        lk_trace(L, 1, trn, 8, rounds_done, lane);
      }
      sfloor += n;
      chunks_retired++;
      block = false;  // one blocking step is enough for the caller to make room
    }
    return true;
  };

  // seal the entries emitted since the last chunk (possibly none) as one chunk
  uint32_t pend_entries = 0;
  auto close_chunk = [&](uint32_t flags, uint64_t credit_value) {
    if (pend_entries == 0 && flags == 0) return;
    const int s = (int)(chunks % LK_RSLOTS);
    if (lane == s) {
      my_n = pend_entries;
      my_flags = flags;
      my_credit = credit_value;
    }
    spub += pend_entries;
    chunks++;
    pend_entries = 0;
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
        close_chunk(0, 0);
        if (!open_chunk()) return false;
        continue;
      }
      if (!service(true)) return false;
    }
    return true;
  };

  uint64_t credit_seen = 0;  // credit_msgs at the last chunk boundary

  // keep at least `want` verified records queued while more have landed
  uint32_t stall = 0;  // consecutive probe rounds that verified nothing (two are normal: header, then footer)
  auto fill = [&](uint32_t want) {
    while (q_tail - q_head < want && w.limit > 0 && !failed) {
      if (q_tail + 64 * LK_PROBE_GROUPS > LK_QCAP) {
        // move the queued records to the front (lane-parallel, through registers)
        const uint32_t nq = q_tail - q_head;
        for (uint32_t b0 = 0; b0 < nq; b0 += 64) {
          const uint32_t i = b0 + lane;
          const uint32_t v = i < nq ? D->q[LKP(q_head + i)] : 0;
          if (i < nq) D->q[LKP(i)] = v;  // (b0 + lane < q_head + b0 + lane: never overwrites an unread slot of a later pass)
        }
        q_head = 0;
        q_tail = nq;
        lk_wave_sync();
      }
      const uint64_t tw0 = wall_clock64();
      const uint32_t got = lk_chain_round(&w, D->q, q_tail, D->hist, lane);
      t_walk += wall_clock64() - tw0;
      q_tail += got;
      flush_publish();
      stall = got ? 0 : stall + 1;
      if ((got == 0 && w.e0 == 0) || stall > 3) {  // the ring does not hold what the descriptor promised
        stw(&L->abort.v, LK_ERR_CORRUPT);
        failed = true;
      }
    }
  };
  auto next_ready = [&]() -> uint64_t {
    fill(1);
    return q_head < q_tail ? D->q[LKP(q_head)] : 0;
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
      q_head++;
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

  // post a command to the helper waves, do my own quarter, wait for theirs.  The command lives
  // in (uniform) registers; lane 0 copies it to LDS for the helpers.
  lk_rx_cmd cm;
  cm.seq = 0; cm.type = 0; cm.done = 0; cm.m = cm.per = cm.cnt = cm.q_head = cm.slot = cm.head32 = cm.clean_max = 0;
  cm.at0 = cm.xs0 = cm.a0 = 0;
  for (int q = 0; q < 4; q++) cm.w_bytes[q] = cm.w_sl[q] = cm.w_ent[q] = cm.w_n[q] = 0;
  auto run_cmd = [&](uint32_t type) {
    cm.type = type;
    cm.seq++;
    lk_wave_sync();  // the tables the helpers are about to read were written by all my lanes
    if (lane == 0) {
      D->cmd.type = type;
      D->cmd.m = cm.m; D->cmd.per = cm.per; D->cmd.cnt = cm.cnt; D->cmd.q_head = cm.q_head;
      D->cmd.slot = cm.slot; D->cmd.head32 = cm.head32;
      D->cmd.at0 = cm.at0; D->cmd.xs0 = cm.xs0; D->cmd.a0 = cm.a0;
      if (type == LK_CMD_EMIT)
        for (int q = 0; q < 4; q++) {
          D->cmd.w_bytes[q] = cm.w_bytes[q]; D->cmd.w_sl[q] = cm.w_sl[q]; D->cmd.w_ent[q] = cm.w_ent[q];
        }
      if (type == LK_CMD_STATE) D->cmd.clean_max = 0;
      __hip_atomic_store(&D->cmd.done, 0u, LK_WG_RLX);
      __hip_atomic_store(&D->cmd.seq, cm.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const uint32_t T = lane;
    if (type == LK_CMD_STATE) lk_rx_part_state(D, cm, T);
    else if (type == LK_CMD_COUNT) lk_rx_part_xenc(D, cm, T);
    else if (type == LK_CMD_TOTALS) lk_rx_part_totals(D, cm, 0, lane, cap32);
    else if (type == LK_CMD_EMIT) lk_rx_part_emit(D, cm, 0, lane, tab, ring, arena, out_slices, cap32);
    uint32_t spins = 0;
    while (__hip_atomic_load(&D->cmd.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 3u) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 0xFFFFu) == 0 && ldw(&L->abort.v) != 0) {  // a helper left on the abort word
        failed = true;
        break;
      }
    }
    lk_wave_sync();  // what my own lanes wrote is read across lanes next
  };
  // exclusive prefix over the 256 per-thread values of one array (lane l holds threads 4 l .. 4 l + 3);
  // returns the total
  auto scan256 = [&](uint32_t* arr) -> uint32_t {
    // (thread T's value sits at arr[T]; the order of the threads is the order of the records)
    lk_wave_sync();
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      v[r] = arr[4 * lane + r];
      sum += v[r];
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    uint32_t x = incl - sum;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      arr[4 * lane + r] = x;
      x += v[r];
    }
    lk_wave_sync();
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  };

  // Up to 4096 whole records per step, worked on by the four waves of this block: thread T of 256
  // owns records [T * per, (T + 1) * per).
  auto bulk_step = [&]() -> uint32_t {
    fill(LK_RX_STEP);
    if (failed) return 0;
    uint32_t m = q_tail - q_head;
    if (m == 0) return 0;
    if (m > LK_RX_STEP) m = LK_RX_STEP;
    cm.m = m;
    cm.per = (m + 255) / 256;
    cm.q_head = q_head;
    cm.head32 = (uint32_t)head;
    run_cmd(LK_CMD_STATE);
    // records [0, cnt) are processed; the state ends clean
    const uint32_t cnt = __hip_atomic_load(&D->cmd.clean_max, LK_WG_RLX);
    if (cnt == 0) return 0;
    cm.cnt = cnt;
    scan256(D->t_enc);
    run_cmd(LK_CMD_COUNT);   // ring offsets of all m records
    run_cmd(LK_CMD_TOTALS);  // per wave: arena bytes, slices, entries, payload
    lk_wave_sync();
    const uint32_t Ctot = D->xenc[LKP(cnt)];
    uint32_t tot_bytes = 0, tot_sl = 0, tot_ent = 0, tot_n = 0;
    for (int q = 0; q < 4; q++) {
      const uint32_t wb = D->cmd.w_bytes[q], ws = D->cmd.w_sl[q], we = D->cmd.w_ent[q];
      cm.w_bytes[q] = tot_bytes; cm.w_sl[q] = tot_sl; cm.w_ent[q] = tot_ent;  // exclusive over the waves
      tot_bytes += wb; tot_sl += ws; tot_ent += we; tot_n += D->cmd.w_n[q];
    }
    if (nslices + tot_sl > max_slices) { lk_abort(L, LK_ERR_SLICES, 1, nslices, tot_sl, max_slices, cnt); failed = true; return 0; }
    if (a_off + tot_bytes + 512 > arena_cap) { lk_abort(L, LK_ERR_ARENA, 2, a_off, tot_bytes, cnt, m); failed = true; return 0; }
    if (!table_room(tot_ent)) { failed = true; return 0; }
    const uint64_t te0 = wall_clock64();
    cm.at0 = spub + pend_entries;
    cm.xs0 = nslices;
    cm.a0 = a_off;
    cm.slot = (uint32_t)(chunks % LK_RSLOTS);
    run_cmd(LK_CMD_EMIT);
    pend_entries += tot_ent;
    t_emit += wall_clock64() - te0;
    // ---- credit accounting over the Recv steps (pair.cc:276-284): the records whose running
    // consumption reaches cap / 2 (uniform search over the offsets in LDS)
    {
      const uint64_t T = cap / 2;
      uint64_t base = 0, thr = T - irs;
      bool crossed = false;
      while ((uint64_t)Ctot >= thr) {
        uint32_t lo = 0, hi = cnt - 1;  // first record whose consumption after its last step reaches thr
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          const uint32_t nm = D->q[LKP(q_head + mid)];
          if ((uint64_t)D->xenc[LKP(mid)] + 16u + round_up8(nm) >= thr) hi = mid; else lo = mid + 1;
        }
        const uint32_t n = D->q[LKP(q_head + lo)];
        const uint32_t e = 16u + (uint32_t)round_up8(n);
        const lk_rec_plan rp = lk_replay_record(n, D->sin[LKP(lo)]);
        const uint64_t C2 = (uint64_t)D->xenc[LKP(lo)] + e;
        const uint64_t cons2 = rp.c2 ? rp.c2 + (round_up8(n) - n + 8) : 0;
        const uint64_t C1 = C2 - cons2;
        const uint64_t pos = (head + D->xenc[LKP(lo)]) & mask;
        if (rp.c2 && C1 >= thr) {  // crossed after the first step of a two-step record
          credit_head = (pos + 8 + rp.c1) & mask;
          base = C1;
        } else {
          credit_head = (pos + e) & mask;
          base = C2;
        }
        credit_msgs++;
        crossed = true;
        thr = base + T;
      }
      irs = crossed ? (uint64_t)Ctot - base : irs + Ctot;
    }
    head = (head + Ctot) & mask;
    mh = head;
    bytes += tot_n;
    records += cnt;
    nslices += tot_sl;
    a_off += tot_bytes;
    q_head += cnt;
    return cnt;
  };

  // ---- one round per Send ---------------------------------------------------------------------
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  for (uint64_t r = 0; connected && !failed; r++) {
    // wait for Send r (or the end of the job)
    bool have = false;
    flush_publish();
    lk_trace(L, 1, trn, 4, r, lane);
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
    lk_trace(L, 1, trn, 5, r, lane);

    // drain: endpoint reads until one would block
    const uint64_t nslices0 = nslices;
    if (!open_chunk()) { failed = true; break; }
    for (;;) {
      if (failed) break;
      if (nslices >= max_slices) { stw(&L->abort.v, LK_ERR_SLICES); failed = true; break; }
      const bool clean = remain == 0 && leftover == 0;
      const uint64_t ts0 = wall_clock64();
      if (clean && bulk_step() > 0) {
        t_fast += wall_clock64() - ts0;
        // one chunk per step: its scatter starts while the next records are being walked
        const bool cr = credit_msgs != credit_seen;
        credit_seen = credit_msgs;
        close_chunk(cr ? 1u : 0u, credit_head);
        if (!open_chunk()) { failed = true; break; }
        continue;
      }
      if (failed) break;
      // rdma_continue_read, rdma_bp_posix.cc:306-317
      uint64_t readable = remain;
      if (readable == 0) readable = next_ready();
      if (failed) break;
      const uint64_t alloc = leftover ? leftover : (readable > MINRD ? readable : MINRD);
      if (a_off + alloc > arena_cap) { lk_abort(L, LK_ERR_ARENA, 3, a_off, alloc, readable, leftover); failed = true; break; }
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
      t_scalar += wall_clock64() - ts0;
    }
    if (failed) break;
    if (nslices != nslices0) rounds_with_data++;
    lk_trace(L, 1, trn, 6, r, lane);
    {
      const bool cr = credit_msgs != credit_seen;
      credit_seen = credit_msgs;
      close_chunk((cr ? 1u : 0u) | 2u, credit_head);
    }
  }
  // the helper waves may leave
  cm.seq++;
  if (lane == 0) {
    D->cmd.type = LK_CMD_EXIT;
    __hip_atomic_store(&D->cmd.seq, cm.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    if (c->line != nullptr) {  // what HasMessage() on the host compares with the arrival report
      c->line->rx_head = head;
      c->line->rx_remain = remain;
    }
    c->internal_read_size = irs;
    c->leftover_cap = leftover;
    c->total_read += bytes;
    c->credit_msgs += credit_msgs;
    c->rx_records += records;
    c->rx_rounds += rounds_with_data;
    c->rx_arena_off = a_off;
    c->rx_slice_idx = nslices;
    if (credit_msgs) c->status_send.remote_head = credit_head;
    c->rx_h1 = w.hcount >= 1 ? D->hist[(w.hcount - 1) % LK_HCAP] : 0;
    c->rx_h2 = w.hcount >= 2 ? D->hist[(w.hcount - 2) % LK_HCAP] : 0;
    L->res_chunks = chunks;
    L->res_entries[LK_SCATTER] = spub;
    L->res_wait_ticks[2] = wait_data;
    L->res_wait_ticks[3] = wait_table;
    L->res_prof[3] = t_walk;
    L->res_prof[4] = t_fast;
    L->res_prof[5] = t_scalar;
    L->res_prof[6] = wall_clock64() - t_begin;
    L->res_prof[7] = t_emit;
    L->trace_n[1] = trn < 192 ? trn : 192;
  }
}

// ---------------------------------------------------------------------------------------------
// k_link: grid (team, links).  Block 0 of a team is the sender's leader, block 1 the receiver's;
// the waves of the other blocks are workers, dealt to the three stages by position.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LK_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_link(lk_ctl* const* ctls, uint64_t timeout_ticks) {
  lk_ctl* L = ctls[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  __shared__ lk_lds s_lds;  // the leaders' tables (workers use no LDS)
  if (blockIdx.x == 0) {
    if (wave == 0) lk_tx_leader(L, timeout_ticks, &s_lds.tx, lane);
    return;
  }
  if (blockIdx.x == 1) {
    if (threadIdx.x == 0) {
      s_lds.rx.cmd.seq = 0;
      s_lds.rx.cmd.done = 0;
    }
    __syncthreads();  // (the only barrier of the block: before the roles part ways)
    if (wave == 0) lk_rx_leader(L, timeout_ticks, &s_lds.rx, lane);
    else lk_rx_helper(L, &s_lds.rx, wave, lane);
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
