// k_rx_plan: message-ready detection (K3), record-chain walk, endpoint_read
// replay and credit accounting (K5) for one drain of a connection's ring.
//
// What the reference does one record at a time on a CPU thread
// (RingBufferPollable::GetReadableSize / Read, src/core/lib/ibverbs/ring_buffer.cc:67-191;
// PairPollable::Recv, pair.cc:264-286; rdma_continue_read / rdma_do_read,
// src/core/lib/iomgr/rdma_bp_posix.cc:180-326) is done here in three tiers, all
// producing exactly the slices, ring state and credit reports the reference
// would:
//
//  bulk    256 threads, up to 4096 records per pass.  The record chain is a
//          linked list, but on a gRPC connection it is periodic (one message =
//          a fixed run of frame-header and payload slices), so the sizes of the
//          last records are kept per connection, the period is detected, every
//          predicted header/footer pair is probed at once (one memory round trip
//          for the whole ring instead of one per record) and the first
//          mismatch bounds the verified prefix.  The endpoint-read state machine
//          is then replayed data-parallel: a record of >= 511 bytes resets the
//          state, so each thread recovers its incoming state by looking back
//          to the nearest such record; offsets come from block prefix sums.
//  wave    one wavefront, 64 probes per round trip with period-2 prediction and
//          __ballot() verification, 64 records replayed per step on the DPP
//          network.  Used while no period is known and around irregularities.
//  scalar  one endpoint_read at a time (partial reads, retained slices, tails).
//
// A bulk pass that stops short of its prediction hands what it already knows to the wave
// tier (rx_state::hand_*): the records it verified but could not take (they end inside an
// open read) are queued in s_chain, and a zero header at the first unverified position is
// the would-block -- the wave tier does not walk those records a second time.
//
// In front of the tiers, in latency mode only: the EXPRESS drain -- a small unary message
// (<= 8 tiny records that fit the open read) handled by one wavefront in three memory
// round trips, no plan, no LDS tables (see "express drain" in rx_plan_body).
//
// Record tags (header / padding / footer words) are not cleared here: the segments carry
// GRDMA_SEG_TAG_* flags and the scatter waves of k_rx_apply do it (grdma_dev.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"
#include "grdma_tx_body.h"
#include "grdma_rx_fast.h"
#include "grdma_rx_multi.h"
#include "grdma_rx_hint.h"
#include "grdma_tx_multi.h"
#include "grdma_tx_fast.h"

namespace {

#define CHAIN_CAP 128
#define BULK_MAX 4096
#define MINRD GRDMA_MIN_READ_SLICE

__device__ unsigned long long g_express_drains = 0;  // diagnostics: drains served by the express path
// profiling aid (grdma_rx_express_ticks): ticks of an express drain's phases, summed -- {state loaded, records known,
// payload loaded, stores issued, stores acknowledged, commit: counters loaded, commit: stores issued, released, count}
__device__ unsigned long long g_rx_express_ticks[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // ... of which the records never touched the ring (grdma_ct_hint)

// Does the express drain take a message of T payload bytes (first record n0) when the connection's open read has
// leftover0 bytes of room (0 = none open)?  1: one read takes it all (the open read, or a fresh one of max(256, n0));
// 2: the open read is smaller than the message -- it completes with leftover0 bytes, rdma_continue_read allocates
// max(256, readable) -- readable = what is left of the record the first read stopped in (GetReadableSize is the length of
// the message at the head, ring_buffer.cc; `rem1`, the caller's) -- and that read takes the rest (rdma_bp_posix.cc:306-326;
// two slices; a rest beyond it would need a third read: not this path's case); 0: not the express path's case.
// The arena must hold every read, the one that then finds nothing included.  Shared by the express drain and by the
// watcher's single-wave drain (rxw_fast), which must agree.
__device__ __forceinline__ int express_fits(uint64_t leftover0, uint64_t n0, uint64_t T, uint64_t max_slices, uint64_t a_off0,
                                            uint64_t arena_cap, uint64_t rem1, bool credit_due) {
  if (T > 512 || max_slices < 1) return 0;
  const uint64_t alloc = leftover0 ? leftover0 : (n0 > MINRD ? n0 : MINRD);
  if (T <= alloc) {
    const uint64_t next_alloc = (alloc - T) ? alloc - T : MINRD;
    return (a_off0 + alloc <= arena_cap && ((a_off0 + T + 15) & ~15ull) + next_alloc <= arena_cap) ? 1 : 0;
  }
  // (credit_due: a status report falls into this drain -- Recv posts it read by read (pair.cc:276-284), and the first
  //  read ends inside a record: left to the general tiers)
  if (leftover0 == 0 || max_slices < 2 || credit_due) return 0;
  const uint64_t L = leftover0, rest = T - L, alloc2 = rem1 > MINRD ? rem1 : MINRD;
  if (rest > alloc2) return 0;
  const uint64_t off1 = (a_off0 + L + 15) & ~15ull, off2 = (off1 + rest + 15) & ~15ull;
  const uint64_t alloc3 = (alloc2 - rest) ? alloc2 - rest : MINRD;
  return (off1 + alloc2 <= arena_cap && off2 + alloc3 <= arena_cap) ? 2 : 0;
}

struct chain_walker {
  const uint8_t* ring;
  uint64_t cap;
  uint64_t pos;     // ring offset of the first unverified record
  uint64_t e0;      // encoded size of the record at pos when its header is already known
  uint64_t h2, h1;  // encoded sizes of the two records before pos (0 = unknown)
  bool dry;         // pos holds no complete record
  uint64_t room;    // bytes from pos the sender has reported as landed (grdma_wire_report); 2^62 = no limit
};

// One probe round of the wave tier.  Lane j loads the tag words at the offset the
// chain reaches after j records if the last two sizes keep alternating; every
// lane checks its own link and the footer in front of it; two ballots give the
// verified prefix.  Stores the payload sizes in chain[0..v) and returns v.
__device__ __forceinline__ uint32_t chain_round(chain_walker* w, uint64_t* chain, int lane) {
  const uint64_t cap = w->cap, mask = cap - 1;
  // history with the already-known first record folded in
  const uint64_t H2 = w->e0 ? w->h1 : w->h2;
  const uint64_t H1 = w->e0 ? w->e0 : w->h1;
  const uint64_t A = H2 ? H2 : H1, B = H1;  // predicted sizes alternate A, B, A, ...
  auto rel_of = [&](uint64_t j) -> uint64_t {
    uint64_t base = 0, k = j;
    if (w->e0) {
      if (j == 0) return 0;
      base = w->e0;
      k = j - 1;
    }
    return base + (k >> 1) * (A + B) + ((k & 1) ? A : 0);
  };
  const bool have_pattern = (A != 0);
  const uint64_t rel = have_pattern ? rel_of(lane) : 0;
  const uint64_t rel_next = have_pattern ? rel_of(lane + 1) : 0;
  // The sender never lets the ring hold more than cap - 8 bytes (W() keeps 24
  // free before a write), so a record can only exist where it ends by cap - 8,
  // and the footer in front of lane j only matters if record j-1 ends by then.
  const bool probe_hdr = (lane == 0) || (have_pattern && rel_next <= cap - 8);
  const bool probe_prev = lane > 0 && have_pattern && rel <= cap - 8;
  const uint64_t my_pos = (w->pos + rel) & mask;
  uint64_t hdr = 0, prev = 0;
  if (probe_hdr) hdr = ld_tag(w->ring + my_pos);
  if (probe_prev) prev = ld_tag(w->ring + ((my_pos + cap - 8) & mask));  // footer of record j-1
  const uint64_t enc = 16 + round_up8(hdr);
  // (a record the sender has not reported as complete is not there yet: its tags may be in place while a
  // tile of its payload is still in flight)
  const bool valid = probe_hdr && hdr != 0 && hdr <= cap - GRDMA_RESERVED && rel + enc <= w->room;
  const bool link_ok = valid && have_pattern && lane < 63 && enc == rel_next - rel;
  const uint64_t m_link = __ballot(link_ok);
  const uint64_t m_foot = __ballot(probe_prev && prev == GRDMA_FOOTER) >> 1;  // bit j: footer of j
  const uint64_t m_fprobed = __ballot(probe_prev) >> 1;
  const uint64_t m_hprobed = __ballot(probe_hdr);
  const uint64_t good = m_link & m_foot;
  const uint32_t v = (good == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~good);
  if ((uint32_t)lane < v) chain[lane] = hdr;
  // state for the next round: lane v is the first unverified record
  const uint64_t m_valid = __ballot(valid);
  const uint64_t rel_v = __shfl(rel, v < 64 ? v : 63, 64);
  const uint64_t enc_v = __shfl(enc, v < 64 ? v : 63, 64);
  const uint64_t enc_l1 = __shfl(enc, v >= 1 ? v - 1 : 0, 64);
  const uint64_t enc_l2 = __shfl(enc, v >= 2 ? v - 2 : 0, 64);
  if (v >= 2) {
    w->h2 = enc_l2;
    w->h1 = enc_l1;
  } else if (v == 1) {
    w->h2 = w->h1;
    w->h1 = enc_l1;
  }
  w->pos = (w->pos + (v < 64 ? rel_v : rel_v + enc_v)) & mask;
  w->room -= (v < 64 ? rel_v : rel_v + enc_v);
  w->e0 = 0;
  if (v < 64) {
    const bool v_hprobed = (m_hprobed >> v) & 1;
    const bool v_valid = (m_valid >> v) & 1;
    const bool v_link = (m_link >> v) & 1;
    const bool v_fprobed = (m_fprobed >> v) & 1;
    if (!v_hprobed) {
      // not looked at (prediction ran past the ring): probe it as lane 0 next round
    } else if (!v_valid) {
      w->dry = true;                 // no (or torn) header: nothing more is ready
    } else if (!v_link || !v_fprobed) {
      w->e0 = enc_v;                 // header known, exact footer probe next round
    } else {
      w->dry = true;                 // size as predicted but the footer has not landed
    }
  }
  return v;
}

// Transition of the endpoint-read state over one record of n bytes:
// s = bytes of space left in an open 256-byte read (0 = between reads).
__device__ __forceinline__ uint32_t read_space_after(uint64_t n, uint32_t s) {
  if (s == 0) return n >= MINRD ? 0 : (uint32_t)(MINRD - n);
  if (n < s) return s - (uint32_t)n;
  if (n == s) return 0;
  const uint64_t r = n - s;
  return r >= MINRD ? 0 : (uint32_t)(MINRD - r);
}

// What one whole record does to the read sequence, given the incoming state.
// (Plain scalars only: arrays indexed at run time would live in scratch memory.)
struct rec_plan {
  uint64_t c1, c2;    // bytes of the (at most) two Recv steps
  uint64_t sl0, sl1;  // lengths of the slices completed by this record, in order (0 = none)
  uint32_t sl_cnt;
};

__device__ __forceinline__ rec_plan replay_record(uint64_t n, uint32_t s_in) {
  rec_plan r;
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

// The same in 32-bit arithmetic, for the bulk tier (ring <= 2 GiB there): half the
// VALU work of the 64-bit form on the pass that is instruction-issue bound.
struct rec_plan32 {
  uint32_t c1, c2, sl0, sl1, sl_cnt;
};
__device__ __forceinline__ rec_plan32 replay_record32(uint32_t n, uint32_t s_in) {
  rec_plan32 r;
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
__device__ __forceinline__ uint32_t al16_32(uint32_t v) { return (v + 15u) & ~15u; }
__device__ __forceinline__ uint32_t tiles_of32(uint32_t len, uint32_t ts) { return (len + (1u << ts) - 1u) >> ts; }

// The ring pieces of one record's steps: step 1 = pieces 0,1; step 2 = pieces 2,3
// (the second piece of a step exists only when the step crosses the ring end).
struct rec_pieces {
  uint64_t off[4], len[4];
};
__device__ __forceinline__ void split_step(uint64_t pay, uint64_t off, uint64_t len, uint64_t cap,
                                           uint64_t* o0, uint64_t* l0, uint64_t* o1, uint64_t* l1) {
  const uint64_t p0 = (pay + off) & (cap - 1);
  const uint64_t first = len < cap - p0 ? len : cap - p0;
  *o0 = p0;
  *l0 = first;
  *o1 = 0;
  *l1 = len - first;
}

__device__ __forceinline__ uint64_t al16(uint64_t v) { return (v + 15) & ~15ull; }
__device__ __forceinline__ uint32_t tiles_of(uint64_t len, uint32_t ts) { return (uint32_t)((len + (1ull << ts) - 1) >> ts); }

// Everything the tiers share, kept in LDS between the phases of the kernel.
struct rx_state {
  uint64_t head, mh, remain, irs, leftover;
  uint64_t nslices, nsegs, ntiles, bytes, consumed_total, records;
  uint64_t a_off, would_block, credit, credit_head;
  uint64_t hist_count;
  uint32_t stop;
  uint32_t bulk_tries;    // bulk attempts left in this call
  uint32_t bulk_blocked;  // last bulk attempt verified nothing: let the wave tier move first
  uint32_t period;        // record-size period of this connection (0 = unknown)
  uint32_t period_searched;  // the full period search already ran in this call
  uint64_t period_retry_at;  // hist_count before which no new search is made
  uint32_t period_strikes;   // consecutive drains whose first bulk pass mispredicted
  uint32_t bulk_first;       // no bulk pass has run in this call yet
  uint32_t period_backoff;   // log2 of the cool-down after a retired / missing period
  uint32_t took;
  // Hand-over from a bulk pass that stopped short of its prediction to the wave tier: the
  // records it verified but did not consume (they do not end between two reads) wait in
  // s_chain, and what it saw at the first unverified position is not probed a second time.
  uint32_t hand_on, hand_n, hand_v, hand_dry;
  uint64_t hand_pos;
};

// SCRATCH: the big arrays (rx_lds_general, 64 KB) live in global memory instead of LDS -- the general planner as the
// rarely-taken fallback inside a launch whose other workgroups cannot keep their occupancy beside a 64 KB LDS
// allocation per workgroup (round 4's fused round, retired: no kernel instantiates SCRATCH = true now).  Same code, same results; slower.
template <bool SCRATCH = false>
__device__ __forceinline__ void rx_plan_body(const grdma_rx_op& op_in, rx_lds* scratch = nullptr) {
  // (a private copy: fields read through the reference would be re-fetched from memory
  // after every store the compiler cannot prove unrelated)
  const grdma_rx_op op = op_in;
  // (the stamps of the launch-chain paths are always taken -- the result block's dbg words, hundreds of microseconds
  //  of work behind them; the latency path's only when asked for: prof_time)
  const bool prof = !op.inline_apply || (op.inline_apply & GRDMA_OP_PROFILE) != 0;
  const uint64_t t_begin = prof_time(prof);
  const unsigned tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned wave = tid >> 6;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;
  uint8_t* ring = c->ring;
  const uint64_t cap = c->cap, mask = cap - 1;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);  // tile size of this connection's plans
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  // How far the sender's completed writes reach (grdma_wire_report), read once: what lands while this
  // drain runs belongs to the next one.  room_at(pos) = bytes from ring offset pos that may be walked.
  const bool limited = op.limit_ptr != nullptr || c->wire_limit != 0;
  const uint64_t lim_tail = !limited ? 0
                            : (op.limit_ptr != nullptr
                                   ? __hip_atomic_load(op.limit_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                   : __hip_atomic_load(&c->wire_recv.wire_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  auto room_at = [&](uint64_t pos) -> uint64_t { return limited ? ((lim_tail - pos) & mask) : (1ull << 62); };

  __shared__ rx_state S;
  __shared__ uint64_t s_chain[CHAIN_CAP];
  // (the big arrays share their allocation with the steady-state body, which has finished when this one runs)
  rx_lds* lds;
  if constexpr (SCRATCH) lds = scratch; else lds = rx_lds_get();
  auto& s_hist = lds->g.hist;
  // padded like the send plan's arrays: contiguous 16-record runs per thread
#define RXP(i) ((i) + ((i) >> 4))
  auto& s_penc = lds->g.penc;
  auto& s_xenc = lds->g.xenc;
  auto& s_n = lds->g.n;
  auto& s_sin = lds->g.sin;
  static_assert(BULK_MAX == RXG_BULK && RXP(BULK_MAX) == RXG_PAD(RXG_BULK), "rx_lds_general mirrors these arrays");
  __shared__ uint64_t s_wave[PLAN_THREADS / 64];
  __shared__ uint64_t s_wbytes[PLAN_THREADS / 64], s_wn[PLAN_THREADS / 64];
  __shared__ uint32_t s_wpk[PLAN_THREADS / 64], s_wtiles[PLAN_THREADS / 64];
  __shared__ unsigned int s_key, s_fail, s_clean;
  __shared__ uint64_t s_dbg[16];

  grdma_slice_out* out_slices = op.slices;
  uint64_t max_slices = GRDMA_MAX_SLICES;
  uint64_t a_off0 = 0;
  if (op.append) {  // streaming job: keep filling the caller's buffer / slice table
    const uint64_t s_idx = op.append == 2 ? 0 : c->rx_slice_idx;
    a_off0 = op.append == 2 ? 0 : c->rx_arena_off;
    out_slices = op.slices + s_idx;
    const uint64_t room = op.slices_cap > s_idx ? op.slices_cap - s_idx : 0;
    if (room < max_slices) max_slices = room;
  }
  if (op.max_reads < max_slices) max_slices = op.max_reads;
  const uint64_t mh0 = c->moving_head;
  const uint64_t hist_count0 = c->rx_hist_count;
  // Everything else of the connection block this body is going to need -- the pointers its commit stores go through,
  // the counters it advances, the period bookkeeping -- loaded HERE, in the round trip of the loads above: a load
  // behind the first store to the block waits for it (the block is reached through a generic pointer), and the
  // latency path paid a round trip each for them (profiles/r04_rtt_probe.txt: 1.1 us "counters loaded" + the pointer
  // loads between the commit's stores).  Nobody but this body writes these fields while it runs.
  grdma_hostline* const pre_line = c->line;
  grdma_status_report* const pre_ps = c->peer_status;
  grdma_hostline* const pre_pl = c->peer_line;
  uint32_t* const pre_hist = c->rx_hist;
  const uint64_t pre_total_read = c->total_read, pre_credit_msgs = c->credit_msgs;
  const uint64_t pre_rx_records = c->rx_records, pre_rx_rounds = c->rx_rounds;
  const uint32_t pre_h1 = c->rx_h1, pre_h2 = c->rx_h2, pre_period = c->rx_period, pre_pad3 = c->pad3;
  const uint64_t pre_retry_at = c->rx_period_retry_at;

  // ===================================================================== express drain
  // The unary small-message case (latency mode): nothing is half-read, at most EXPRESS_MAX
  // tiny records are ready and they fit the read that is open (or the 256-byte read a fresh
  // rdma_continue_read allocates).  One wavefront then does the whole drain in three
  // memory round trips -- connection state, one probe round, the payload bytes -- with no
  // plan in memory, no LDS tables and no history load: lane i owns record i for the
  // arithmetic, every lane moves eight bytes of the delivered slice.  Anything else
  // (a record cut by the read, more records than lanes looked at, a pattern the probe round
  // could not follow) falls through to the general tiers below, which produce the same
  // result for these cases too.
  __shared__ unsigned int s_express;
  constexpr uint32_t EXPRESS_MAX = 8, EXPRESS_BYTES = 512;
  if (tid == 0) s_express = 0;
  __syncthreads();
  if (op.inline_apply && op.raw_cap == 0 && !op.append && connected && wave == 0 && c->remain == 0 &&
      max_slices >= 1) {
    const uint64_t head0 = c->head, leftover0 = c->leftover_cap, irs0 = c->internal_read_size;
    const uint64_t te_a = prof_time(prof) + (head0 & 0);  // (state loaded)
    chain_walker w = {ring, cap, head0, 0, pre_h2, pre_h1, false, room_at(head0)};
    const uint32_t v = chain_round(&w, s_chain, lane);
    const bool all_seen = w.dry && v <= EXPRESS_MAX;
    const uint64_t te_b = prof_time(prof) + (v & 0);  // (records known)
    const uint32_t n = ((uint32_t)lane < v && all_seen) ? (uint32_t)s_chain[lane] : 0;
    const uint32_t enc = ((uint32_t)lane < v && all_seen) ? 16u + (uint32_t)round_up8(n) : 0;
    const uint32_t i_n = wave_incl_scan_u32(n), i_enc = wave_incl_scan_u32(enc);
    const uint32_t T = __shfl(i_n, 63, 64), E = __shfl(i_enc, 63, 64);
    const uint32_t n0 = __shfl(n, 0, 64);
    // rdma_continue_read, rdma_bp_posix.cc:306-317: the open read, or a new one of max(256, readable)
    const uint64_t alloc = leftover0 ? leftover0 : (n0 > MINRD ? n0 : MINRD);
    // (1: one read takes the message; 2: the open read completes and a fresh one takes the rest -- two slices)
    // (rem1: what is left of the record in which the open read's last byte falls -- the next read is sized by it)
    uint32_t rem1 = 0;
    {
      const uint32_t xn = i_n - n;
      const bool mine = n != 0 && leftover0 >= xn && leftover0 < (uint64_t)xn + n;
      const uint64_t who = __ballot(mine);
      if (who) rem1 = __shfl(xn + n - (uint32_t)leftover0, __builtin_ctzll(who), 64);
    }
    const int fits = all_seen ? express_fits(leftover0, n0, T, max_slices, a_off0, op.arena_cap, rem1, irs0 + E >= cap / 2) : 0;
    const bool split = fits == 2;
    const uint32_t L0 = split ? (uint32_t)leftover0 : T;                                   // bytes of the first slice
    const uint32_t off1 = split ? (uint32_t)(((a_off0 + L0 + 15) & ~15ull) - a_off0) : 0;  // the second slice, from dst
    static_assert(EXPRESS_BYTES == 512, "express_fits knows the limit");
    if (fits != 0) {
      // payload: output byte b of the slice lives in record r(b) at offset b - x_n(r)
      const uint32_t x_n = i_n - n, x_enc = i_enc - enc;
      uint8_t* dst = op.arena + a_off0;
      uint32_t src_off[8];  // ring offset of output byte 8 * lane + q, relative to head0
#pragma unroll
      for (int q = 0; q < 8; q++) src_off[q] = 0;
#pragma unroll
      for (uint32_t r = 0; r < EXPRESS_MAX; r++) {
        const uint32_t rx = __shfl(x_n, (int)r, 64), rn = __shfl(n, (int)r, 64), re = __shfl(x_enc, (int)r, 64);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t b = (uint32_t)lane * 8 + q;
          if (b >= rx && b < rx + rn) src_off[q] = re + 8 + (b - rx);
        }
      }
      // all byte loads first (global address space: no LDS counter involved), then the stores
      auto* gring = (const __attribute__((address_space(1))) uint8_t*)(uint64_t)ring;
      uint8_t bytes[8];
      if (op.inline_apply & 8u) {
        // (a watcher's drain, csrc k_watch: the bytes were written by another workgroup, process or device since this
        //  CU last looked at these lines, with no kernel boundary in between -- loads that go past the caches)
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const uint32_t b = (uint32_t)lane * 8 + q;
          const uint8_t x = __hip_atomic_load(&gring[(head0 + src_off[q]) & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          bytes[q] = b < T ? x : (uint8_t)0;
        }
      } else {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t b = (uint32_t)lane * 8 + q;
        bytes[q] = b < T ? gring[(head0 + src_off[q]) & mask] : (uint8_t)0;
      }
      }
      const uint64_t te_c = prof_time(prof) + (bytes[0] & 0);  // (payload loaded)
      // one 8-byte store per lane (the slice buffer is 16-byte aligned and `alloc` bytes long;
      // the bytes behind the slice end inside the last word are written as zero)
      uint64_t word = 0;   // output bytes 8 * lane .. 8 * lane + 7 (zero behind T)
#pragma unroll
      for (int q = 0; q < 8; q++) word |= (uint64_t)bytes[q] << (8 * q);
      if (split) {
        // The second slice starts at a 16-byte boundary of its own (off1): its word k is output bytes L0 + 8 k ..., which
        // lie in the registers of lanes (L0 >> 3) + k and the one behind it.  Both slices leave as 8-byte words, the
        // bytes behind a slice's end inside its last word as zeros (arena padding) -- byte stores were eighty posted
        // writes over PCIe when the arena is pinned host memory (profiles/r06_rtt_notes.txt; rxw_fast does the same).
        const uint32_t nw0 = (L0 + 7u) >> 3, nw1 = (T - L0 + 7u) >> 3, j0 = L0 >> 3, sh = 8u * (L0 & 7u);
        const uint32_t ja = (j0 + (uint32_t)lane) & 63u, jb = (j0 + (uint32_t)lane + 1u) & 63u;
        const uint64_t lo = __shfl(word, (int)ja, 64);
        uint64_t hi = __shfl(word, (int)jb, 64);
        if (j0 + (uint32_t)lane + 1u > 63u) hi = 0;
        const uint64_t w1 = (j0 + (uint32_t)lane > 63u) ? 0ull : (sh ? ((lo >> sh) | (hi << (64u - sh))) : lo);
        if ((uint32_t)lane < nw0) {
          uint64_t w0 = word;
          if ((uint32_t)lane == nw0 - 1 && (L0 & 7u) != 0) w0 &= (1ull << sh) - 1ull;
          *reinterpret_cast<uint64_t*>(dst + (uint32_t)lane * 8) = w0;
        }
        if ((uint32_t)lane < nw1) *reinterpret_cast<uint64_t*>(dst + off1 + (uint32_t)lane * 8) = w1;
      } else if ((uint32_t)lane * 8 < T) {
        *reinterpret_cast<uint64_t*>(dst + (uint32_t)lane * 8) = word;
      }
      // clear what was consumed: records are 8-byte granular, [head0, head0 + E) with wrap
      for (uint32_t o = (uint32_t)lane * 8; o < E; o += 64 * 8)
        *reinterpret_cast<uint64_t*>(ring + ((head0 + o) & mask)) = 0;
      // history ring and the credit rule of Recv (pair.cc:276-284), record by record
      uint32_t* gh = pre_hist;
      if ((uint32_t)lane < v) gh[(hist_count0 + lane) % GRDMA_RX_HIST] = enc;
      uint64_t irs = irs0, credit = 0, credit_head = 0;
      for (uint32_t r = 0; r < v; r++) {
        irs += __shfl(enc, (int)r, 64);
        if (irs >= cap / 2) {
          credit_head = (head0 + __shfl(i_enc, (int)r, 64)) & mask;
          credit++;
          irs = 0;
        }
      }
      if (lane == 0) {
        const uint64_t nh = (head0 + E) & mask;
        S.head = nh; S.mh = v ? nh : mh0; S.remain = 0; S.irs = irs;
        S.nsegs = S.ntiles = 0;
        S.consumed_total = E; S.records = v; S.bytes = T;
        S.credit = credit; S.credit_head = credit_head;
        S.hist_count = hist_count0 + v;
        if (v && split) {
          // the open read completes with L0 bytes; a fresh read of max(256, readable) takes the rest; a third finds nothing
          const uint64_t rest = T - L0, alloc2 = rem1 > MINRD ? rem1 : MINRD, rest2 = alloc2 - rest;
          out_slices[0].off = a_off0;
          out_slices[0].len = L0;
          out_slices[1].off = a_off0 + off1;
          out_slices[1].len = rest;
          S.nslices = 2;
          S.a_off = (a_off0 + off1 + rest + 15) & ~15ull;
          S.would_block = max_slices >= 3 ? 1 : 0;
          S.leftover = max_slices >= 3 ? (rest2 ? rest2 : MINRD) : rest2;
        } else if (v) {
          // one completed read of T bytes; a second read finds nothing and keeps its buffer
          out_slices[0].off = a_off0;
          out_slices[0].len = T;
          S.nslices = 1;
          S.a_off = (a_off0 + T + 15) & ~15ull;
          const uint64_t rest = alloc - T;
          S.would_block = max_slices >= 2 ? 1 : 0;
          S.leftover = max_slices >= 2 ? (rest ? rest : MINRD) : rest;
        } else {
          // nothing ready: notify_on_read, the slice stays allocated (rdma_bp_posix.cc:241-243)
          S.nslices = 0; S.a_off = a_off0; S.would_block = 1; S.leftover = alloc;
        }
        S.stop = 1;
        S.bulk_tries = 0; S.bulk_blocked = 0; S.took = 0; S.hand_on = 0;
        S.period = pre_period; S.period_searched = 0; S.period_retry_at = pre_retry_at;
        S.period_strikes = pre_pad3 & 0xFFFFu; S.period_backoff = pre_pad3 >> 16; S.bulk_first = 1;
        for (int q = 0; q < 16; q++) s_dbg[q] = 0;
        s_dbg[0] = 1;
        // sizes of the two newest records, for the next call's probe round
        const uint32_t e_last = v >= 1 ? (uint32_t)s_chain[v - 1] : 0, e_prev = v >= 2 ? (uint32_t)s_chain[v - 2] : 0;
        if (v >= 2) { c->rx_h1 = 16u + (uint32_t)round_up8(e_last); c->rx_h2 = 16u + (uint32_t)round_up8(e_prev); }
        else if (v == 1) { c->rx_h2 = pre_h1; c->rx_h1 = 16u + (uint32_t)round_up8(e_last); }
        s_express = 1;
        atomicAdd(&g_express_drains, 1ull);
      }
      const uint64_t te_d = prof_time(prof);  // (stores issued)
      GRDMA_WAIT_VMEM();
      if (lane == 0 && prof) {
        const uint64_t te_e = __builtin_amdgcn_s_memtime();
        g_rx_express_ticks[0] += te_a - t_begin;
        g_rx_express_ticks[1] += te_b - te_a;
        g_rx_express_ticks[2] += te_c - te_b;
        g_rx_express_ticks[3] += te_d - te_c;
        g_rx_express_ticks[4] += te_e - te_d;
        g_rx_express_ticks[8] += 1;
        s_dbg[15] = te_e;
      }
    }
  }
  __syncthreads();
  const bool express = s_express != 0;
  if ((op.inline_apply & 8u) && !express) {
    // a watcher's drain that the general tiers take: their loads of ring bytes are plain ones, and this CU has not
    // seen a kernel boundary since it last read those lines
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }

  if (!express) {
    // (all loads first: `c->rx_hist` is a generic pointer, so a store to LDS between two
    // of them would serialise the round trips)
    constexpr int NH = GRDMA_RX_HIST / PLAN_THREADS;
    const uint32_t* gh = c->rx_hist;
    uint32_t hv[NH];
#pragma unroll
    for (int r = 0; r < NH; r++) hv[r] = gh[tid + r * PLAN_THREADS];
#pragma unroll
    for (int r = 0; r < NH; r++) s_hist[tid + r * PLAN_THREADS] = hv[r];
  }
  if (tid == 0 && !express) {
    S.head = c->head; S.mh = c->moving_head; S.remain = c->remain;
    S.irs = c->internal_read_size; S.leftover = c->leftover_cap;
    S.nslices = S.nsegs = S.ntiles = S.bytes = S.consumed_total = S.records = 0;
    S.a_off = a_off0; S.would_block = 0; S.credit = 0; S.credit_head = 0;
    S.hist_count = hist_count0;
    S.stop = connected ? 0 : 1;
    S.bulk_tries = (op.raw_cap == 0 && cap <= (1ull << 31) && max_slices >= 512) ? 6 : 0;
    S.bulk_blocked = 0;
    S.took = 0;
    S.hand_on = 0;
    S.period = c->rx_period;
    S.period_searched = 0;
    S.period_retry_at = c->rx_period_retry_at;
    S.period_strikes = c->pad3 & 0xFFFFu;
    S.period_backoff = c->pad3 >> 16;
    S.bulk_first = 1;
    for (int q = 0; q < 16; q++) s_dbg[q] = 0;
  }
  __syncthreads();
  if (tid == 0) s_dbg[14] = __builtin_amdgcn_s_memtime() - t_begin;  // prologue

  // ===================================================================== bulk tier
  auto bulk = [&]() -> uint32_t {
    const uint64_t H = S.hist_count < GRDMA_RX_HIST ? S.hist_count : GRDMA_RX_HIST;
    const uint64_t hbase = S.hist_count - H;  // absolute index of the oldest entry kept
    auto hist_at = [&](uint64_t i) -> uint32_t {  // i in [0, H): oldest .. newest
      return s_hist[(hbase + i) % GRDMA_RX_HIST];
    };
    // ---- period detection -------------------------------------------------------------------
    // The newest TAIL entries are set aside: a Send that stops on the staging budget
    // splits one record in two pieces (the last record of one drain, the first of
    // the next), which would break an otherwise exact period.  The period is found
    // on the older entries and the tail is then replayed against it.
    const uint64_t TAIL = 3;
    if (H < TAIL + 6) return 0xFFFFFFFFu;
    const uint64_t Hp = H - TAIL;
    if (tid == 0) {
      s_key = 0xFFFFFFFFu;
      s_fail = 0xFFFFFFFFu;
      s_clean = 0;
    }
    __syncthreads();
    // The period found earlier is remembered in the connection and only re-validated
    // (256 threads compare up to WIN recent entries in one step); a full search runs
    // when it stops matching, at most once per call.
    const uint32_t WIN = 64;
    const uint64_t tb0 = __builtin_amdgcn_s_memtime();
    const uint32_t hp1 = (uint32_t)((hbase + Hp - 1) % GRDMA_RX_HIST);  // newest clean entry
    auto entry_back = [&](uint32_t back) -> uint32_t {   // back = 0: newest clean entry
      return s_hist[(hp1 + GRDMA_RX_HIST - back) % GRDMA_RX_HIST];
    };
    // Evidence for a period P: the newest min(L, P) clean records equal the ones P
    // earlier (a whole period of them always straddles the message boundary, which
    // is what separates the true period from the frame-header / payload alternation
    // inside one message), with at least 3P/4 records to compare.
    uint64_t P = S.period;
    if (P != 0 && P + (3 * P) / 4 <= Hp) {
      const uint32_t L = (uint32_t)(Hp - P);
      const uint32_t W = L < P ? L : (uint32_t)P;
      bool bad = false;
      for (uint32_t jj = tid; jj < W; jj += PLAN_THREADS)
        bad |= entry_back(jj) != entry_back(jj + (uint32_t)P);
      if (bad) atomicMin(&s_key, 0u);
      __syncthreads();
      if (s_key == 0u) P = 0;
      __syncthreads();
      if (tid == 0) s_key = 0xFFFFFFFFu;
      __syncthreads();
    } else {
      P = 0;
    }
    if (P == 0) {
      // a failed search is not repeated until a fresh window of records has been seen
      if (S.period_searched || S.hist_count < S.period_retry_at) return 0xFFFFFFFFu;
      // full search: the largest period up to 512 records with that evidence
      for (uint32_t Pc = tid + 1; Pc <= 512 && Pc + (3 * Pc) / 4 <= Hp; Pc += PLAN_THREADS) {
        const uint32_t L = (uint32_t)Hp - Pc;
        const uint32_t W = L < Pc ? L : Pc;
        uint32_t score = 0;
        while (score < W && entry_back(score) == entry_back(score + Pc)) score++;
        if (score == W) atomicMin(&s_key, 0xFFFFu - Pc);
      }
      __syncthreads();
      const unsigned key = s_key;
      __syncthreads();
      P = key == 0xFFFFFFFFu ? 0xFFFFFFFFull : (uint64_t)(0xFFFFu - key);
      if (tid == 0) {
        S.period_searched = 1;
        S.period = P == 0xFFFFFFFFull ? 0 : (uint32_t)P;
        if (P == 0xFFFFFFFFull) {
          if (S.period_backoff < 10) S.period_backoff++;
          S.period_retry_at = S.hist_count + ((uint64_t)GRDMA_RX_HIST << S.period_backoff) / 4;
        }
      }
      // no period: leave the drain to the wave tier (64 probes per round trip)
      if (P == 0xFFFFFFFFull) return 0xFFFFFFFFu;
    }
    {
      // A Send that stops on the staging budget (cap / 2) splits a record and shifts
      // the phase; when fewer than three periods fit between two such splits the
      // history never holds enough clean records to keep the period validated, and
      // the wave tier is the better walker.
      uint64_t part = 0;
      for (uint64_t i = tid; i < P; i += PLAN_THREADS) part += entry_back((uint32_t)i);
      uint64_t period_bytes;
      block_excl_scan(part, s_wave, &period_bytes);
      if (3 * period_bytes > cap / 2) {
        if (tid == 0) {
          S.period = 0;
          S.period_retry_at = S.hist_count + 4 * GRDMA_RX_HIST;
        }
        return 0xFFFFFFFFu;
      }
    }
    if (tid == 0) s_dbg[8] += __builtin_amdgcn_s_memtime() - tb0;
    auto pattern = [&](uint64_t i) -> uint32_t {  // slot i after the clean prefix
      return hist_at(Hp - P + (i % P));
    };
    // replay the tail: whole slots advance q, pieces of a split slot accumulate
    uint64_t q = 0, acc = 0, pieces = 0;
    bool tail_ok = true;
    for (uint64_t a = 0; a < TAIL; a++) {
      const uint64_t x = hist_at(Hp + a);
      if (acc == 0 && x == pattern(q)) { q++; continue; }
      acc += x;
      pieces++;
      const uint64_t whole = (uint64_t)pattern(q) + 16 * (pieces - 1);
      if (acc == whole) { q++; acc = 0; pieces = 0; }
      else if (acc > whole) tail_ok = false;
    }
    uint64_t rem = 0;  // encoded size of the piece that completes a split slot
    if (acc > 0) {
      rem = (uint64_t)pattern(q) + 16 * pieces - acc;
      if (rem < 24 || (rem & 7)) tail_ok = false;
      q++;
    }
    if (!tail_ok) return 0xFFFFFFFFu;
    // ---- predicted sizes and offsets for up to BULK_MAX records ----------------------
    const uint32_t per = BULK_MAX / PLAN_THREADS;  // 16 contiguous records per thread
    {
      uint64_t chunk = 0;
      // record i takes pattern slot q + i (one earlier when a split slot is being completed
      // by `rem`); the slot index modulo P is carried along instead of divided out per record
      const uint32_t P32 = (uint32_t)P, r1 = rem ? 1u : 0u;
      const uint32_t i0 = tid * per;
      uint32_t jm = (uint32_t)((q + (i0 > r1 ? i0 - r1 : 0)) % P);
      const uint32_t hoff = (uint32_t)((hbase + Hp - P) % GRDMA_RX_HIST);
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = i0 + k;
        uint32_t e;
        if (r1 && i == 0) {
          e = (uint32_t)rem;
        } else {
          e = s_hist[(hoff + jm) % GRDMA_RX_HIST];
          if (++jm == P32) jm = 0;
        }
        s_penc[RXP(i)] = e;
        chunk += e;
      }
      uint64_t total;
      uint64_t x = block_excl_scan(chunk, s_wave, &total);
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = tid * per + k;
        s_xenc[RXP(i)] = x > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)x;
        x += s_penc[RXP(i)];
      }
      if (tid == PLAN_THREADS - 1) s_xenc[RXP(BULK_MAX)] = x > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)x;
    }
    __syncthreads();
    const uint64_t tb1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_dbg[15] += tb1 - tb0;
    // ---- probe every predicted header / footer pair at once ---------------------------
    const uint64_t head = S.head;
    // room: slices <= 2 per record, segments <= 2 per record + 1 wrap, arena by offsets
    uint64_t vmax = BULK_MAX;
    {
      const uint64_t r_sl = (max_slices - S.nslices) / 2;
      const uint64_t r_sg = (GRDMA_MAX_SEGS - 524 - S.nsegs) / 2;
      if (r_sl < vmax) vmax = r_sl;
      if (r_sg < vmax) vmax = r_sg;
    }
    const uint64_t arena_room = op.arena_cap > S.a_off + 1024 ? op.arena_cap - S.a_off - 1024 : 0;
    const uint64_t wire_room = room_at(head);
    {
      // One 16-byte load per record: the footer of record i and the header of record i + 1
      // are neighbouring words of the ring, so thread i fetches both at once and checks the
      // footer of its own record and the header of the next one (two 8-byte loads per record
      // were twice the requests from this one CU, and the probe is bound by how many of them
      // it can keep in flight).  All 16 loads of a thread are issued before the first is
      // looked at; sc1 as in ld_tag (served by L2 / memory, not by this CU's L1).
      constexpr int NP = BULK_MAX / PLAN_THREADS;
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      // (readfirstlane returns int: widen through uint32_t, or a set bit 31 smears into the high word)
      const uint32_t ring_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ring);
      const uint32_t ring_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ring >> 32));
      const uint64_t ring_u = ((uint64_t)ring_hi << 32) | (uint64_t)ring_lo;
      const uint32_t cap_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cap);  // <= 2 GiB in a bulk pass
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ring_u, 0, cap_u, 0x00020000);
      u32x4 pairs[NP];
      bool want[NP];
      // the header of record 0, and the first word of the ring for the one record whose
      // footer is the last word of the ring (the same line for every lane)
      // (through the buffer as well: a FLAT load would also count on LGKM_CNT, and the LDS
      // reads between the loads below would wait for it)
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 hf = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(head & mask & ~7ull), 0, 16);
      const u32x2 wz = __builtin_amdgcn_raw_buffer_load_b64(rs, 0u, 0, 16);
#pragma unroll
      for (int r = 0; r < NP; r++) {
        const uint32_t i = tid + r * PLAN_THREADS;
        const uint64_t x = s_xenc[RXP(i)], e = s_penc[RXP(i)];
        want[r] = i < vmax && x + e <= cap - 8 && x + e <= wire_room && x + e + 32ull * (i + 1) <= arena_room;
        // (unconditional: a masked offset is always inside the ring, and straight-line
        // loads are all issued before the first wait; a footer in the last word of the
        // ring is read as the second half of the pair one word earlier)
        const uint32_t f = (uint32_t)((head + x + e - 8) & mask & ~7ull);
        pairs[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, f <= cap_u - 16u ? f : cap_u - 16u, 0, 16 /* sc1 */);
      }
      uint32_t first_bad = 0xFFFFFFFFu;
      const uint64_t hdr_first = ((uint64_t)hf.y << 32) | hf.x, word_zero = ((uint64_t)wz.y << 32) | wz.x;
      if (tid == 0) {
        const bool ok = hdr_first != 0 && hdr_first <= cap - GRDMA_RESERVED && 16 + round_up8(hdr_first) == s_penc[RXP(0)];
        s_n[RXP(0)] = (uint32_t)hdr_first;
        if (!ok) first_bad = 0;
      }
#pragma unroll
      for (int r = 0; r < NP; r++) {
        const uint32_t i = tid + r * PLAN_THREADS;
        const uint64_t x = s_xenc[RXP(i)], e = s_penc[RXP(i)];
        const uint32_t f = (uint32_t)((head + x + e - 8) & mask & ~7ull);
        const bool last_word = f > cap_u - 16u;
        const uint64_t lo = ((uint64_t)pairs[r].y << 32) | pairs[r].x, hi = ((uint64_t)pairs[r].w << 32) | pairs[r].z;
        const uint64_t foot = last_word ? hi : lo;
        const uint64_t next = last_word ? word_zero : hi;  // header of record i + 1
        if (!(want[r] && foot == GRDMA_FOOTER) && i < first_bad) first_bad = i;
        if (i + 1 < BULK_MAX) {
          const bool ok = next != 0 && next <= cap - GRDMA_RESERVED && 16 + round_up8(next) == s_penc[RXP(i + 1)];
          s_n[RXP(i + 1)] = (uint32_t)next;
          if (!ok && i + 1 < first_bad) first_bad = i + 1;
        }
      }
      // one LDS atomic per wave
      const uint64_t bm = __ballot(first_bad != 0xFFFFFFFFu);
      if (bm != 0) {
        uint32_t m = first_bad;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const uint32_t o = __shfl_xor(m, d, 64);
          m = o < m ? o : m;
        }
        if (lane == 0) atomicMin(&s_fail, m);
      }
    }
    __syncthreads();
    const uint32_t V = s_fail == 0xFFFFFFFFu ? BULK_MAX : s_fail;
    const uint64_t tb2 = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_dbg[9] += tb2 - tb1;
    if (tid == 0) { s_dbg[6] = P; s_dbg[7] = V; }
    if (V < BULK_MAX && tid == 0) {
      // The verified run ended before the prediction did: either the data ends here or
      // the pattern changed.  Another bulk pass would probe thousands of predicted
      // positions to learn the same thing, so the wave tier finishes this call.
      S.bulk_tries = 1;
      // Was the first unverified record a misprediction (a complete record of another
      // size) or simply the end of what has arrived?  Three mispredicted drain starts
      // in a row retire the remembered period.
      const uint64_t hv = s_n[RXP(V)];
      const bool mispredicted = hv != 0 && hv <= cap - GRDMA_RESERVED &&
                                16 + round_up8(hv) != s_penc[RXP(V)];
      if (S.bulk_first) {
        if (V < 16 && mispredicted) {
          if (++S.period_strikes >= 3) {
            S.period = 0;
            S.period_strikes = 0;
            if (S.period_backoff < 10) S.period_backoff++;  // up to 1024 x HIST records
            S.period_retry_at = S.hist_count + ((uint64_t)GRDMA_RX_HIST << S.period_backoff);
          }
        } else if (V >= 16) {
          S.period_strikes = 0;
          if (V >= 256) S.period_backoff = 0;
        }
      }
      S.bulk_first = 0;
    }
    // Records [first, V) passed the probe (header, predicted size, footer) but stay unconsumed;
    // position V is where the chain continues.  A zero header there (low word: a value with
    // only high bits set is no header either) means nothing more had arrived when this call
    // looked -- the wave tier takes that as its would-block instead of loading it again.
    auto hand_over = [&](uint32_t first) {
      const uint32_t k = V - first;
      if (V >= BULK_MAX || k > 64) return;  // (uniform)
      if (tid < k) s_chain[tid] = s_n[RXP(first + tid)];
      if (tid == 0) {
        S.hand_on = 1;
        S.hand_n = k;
        S.hand_v = V;
        S.hand_dry = s_n[RXP(V)] == 0 ? 1u : 0u;
        S.hand_pos = (head + s_xenc[RXP(V)]) & mask;
      }
    };
    if (V == 0) {
      hand_over(0);
      return 0;
    }
    // ---- pass 0: incoming read state of every record, last clean record ----------------
    const uint32_t per0 = (V + PLAN_THREADS - 1) / PLAN_THREADS;
    const uint32_t k0 = tid * per0;
    if (k0 < V) {
      // look back to the nearest record that resets the state
      uint32_t j = k0;
      while (j > 0 && s_n[RXP(j - 1)] < 2 * MINRD - 1) j--;
      uint32_t s = 0;
      for (; j < k0; j++) s = read_space_after(s_n[RXP(j)], s);
      uint32_t last_clean = 0;
      for (uint32_t k = k0; k < k0 + per0 && k < V; k++) {
        s_sin[RXP(k)] = (uint16_t)s;
        s = read_space_after(s_n[RXP(k)], s);
        if (s == 0) last_clean = k + 1;
      }
      if (last_clean) atomicMax(&s_clean, last_clean);
    }
    __syncthreads();
    const uint32_t cnt = s_clean;  // records [0, cnt) are processed; the state ends clean
    const uint64_t tb3 = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_dbg[10] += tb3 - tb2;
    if (cnt == 0) {
      hand_over(0);
      return 0;
    }
    // ---- pass 1: totals per wave --------------------------------------------------------
    // The cnt records are dealt to the four waves in contiguous quarters and, inside a
    // wave, to lanes round-robin: neighbouring lanes then write neighbouring plan and
    // slice-table entries in pass 2 (coalesced 32-byte / 16-byte stores).
    constexpr uint32_t NW = PLAN_THREADS / 64;
    const uint32_t wchunk = (((cnt + NW - 1) / NW) + 63u) & ~63u;
    const uint32_t wbeg = wave * wchunk < cnt ? wave * wchunk : cnt;
    const uint32_t wend = wbeg + wchunk < cnt ? wbeg + wchunk : cnt;
    // (32-bit arithmetic: the ring is at most 2 GiB in a bulk pass.  At most one record of
    // the pass straddles the ring end; only that lane takes the 4-piece path.)
    const uint32_t cap32 = (uint32_t)cap, mask32 = (uint32_t)mask, head32 = (uint32_t)head;
    {
      uint64_t t_bytes = 0, t_n = 0;
      uint32_t t_pk = 0, t_tiles = 0;  // t_pk: slices | segments << 16 (<= 2 and 4 per record)
      for (uint32_t k = wbeg + lane; k < wend; k += 64) {
        const uint32_t n = s_n[RXP(k)];
        const rec_plan32 rp = replay_record32(n, s_sin[RXP(k)]);
        const uint32_t pay = (head32 + s_xenc[RXP(k)] + 8u) & mask32;
        // (the two steps of a record are contiguous in the ring AND in the arena -- step 1 fills
        // the open slice exactly, step 2 starts the next one right behind it -- so they travel
        // as ONE segment unless the record crosses the ring end)
        uint32_t sg = 1u;
        uint32_t tl = tiles_of32(rp.c1 + rp.c2, ts);
        if (pay + n > cap32 || pay + n < pay) {  // crosses the ring end: split the step(s) it cuts
          uint64_t o0, l0, o1, l1, o2, l2, o3, l3;
          split_step(pay, 0, rp.c1, cap, &o0, &l0, &o1, &l1);
          split_step(pay, rp.c1, rp.c2, cap, &o2, &l2, &o3, &l3);
          sg = (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u) + (l3 ? 1u : 0u);
          tl = tiles_of(l0, ts) + tiles_of(l1, ts) + tiles_of(l2, ts) + tiles_of(l3, ts);
        }
        t_bytes += al16_32(rp.sl0) + al16_32(rp.sl1);
        t_pk += rp.sl_cnt + (sg << 16);
        t_tiles += tl;
        t_n += n;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        t_bytes += __shfl_xor(t_bytes, d, 64);
        t_n += __shfl_xor(t_n, d, 64);
        t_pk += __shfl_xor(t_pk, d, 64);
        t_tiles += __shfl_xor(t_tiles, d, 64);
      }
      if (lane == 0) {
        s_wbytes[wave] = t_bytes;
        s_wn[wave] = t_n;
        s_wpk[wave] = t_pk;
        s_wtiles[wave] = t_tiles;
      }
    }
    __syncthreads();
    uint64_t tot_bytes = 0, tot_sl = 0, tot_sg = 0, tot_tiles = 0, tot_n = 0;
    uint64_t c_bytes = 0, c_sl = 0, c_sg = 0, c_tiles = 0;  // running prefix of this wave
#pragma unroll
    for (uint32_t w = 0; w < NW; w++) {
      const uint64_t b = s_wbytes[w], sl = s_wpk[w] & 0xFFFFu, sg = s_wpk[w] >> 16, tl = s_wtiles[w];
      if (w < wave) { c_bytes += b; c_sl += sl; c_sg += sg; c_tiles += tl; }
      tot_bytes += b; tot_sl += sl; tot_sg += sg; tot_tiles += tl; tot_n += s_wn[w];
    }
    const uint64_t tb4 = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_dbg[11] += tb4 - tb3;
    // ---- pass 2: segments, slices, tag clearing -------------------------------------------
    const uint64_t nsegs0 = S.nsegs, ntiles0 = S.ntiles, nsl0 = S.nslices, a0 = S.a_off;
    for (uint32_t base = wbeg; base < wend; base += 64) {
      const uint32_t k = base + lane;
      const bool act = k < wend;
      const uint32_t n = act ? s_n[RXP(k)] : 0;
      const uint32_t s_in = act ? s_sin[RXP(k)] : 0;
      rec_plan32 rp = replay_record32(n, s_in);
      if (!act) { rp.c1 = rp.c2 = 0; rp.sl_cnt = 0; rp.sl0 = rp.sl1 = 0; }
      const uint32_t pos = (head32 + (act ? s_xenc[RXP(k)] : 0)) & mask32, pay = (pos + 8u) & mask32;
      // pieces of the record in the ring: step 1 = pieces 0,1; step 2 = pieces 2,3 (the
      // second piece of a step exists only for the one record that crosses the ring end)
      uint32_t o0 = pay, l0 = rp.c1, o1 = 0, l1 = 0, o2 = pay + rp.c1, l2 = rp.c2, o3 = 0, l3 = 0;
      if (act && (pay + n > cap32 || pay + n < pay)) {
        uint64_t a0, b0, a1, b1, a2, b2, a3, b3;
        split_step(pay, 0, rp.c1, cap, &a0, &b0, &a1, &b1);
        split_step(pay, rp.c1, rp.c2, cap, &a2, &b2, &a3, &b3);
        o0 = (uint32_t)a0; l0 = (uint32_t)b0; o1 = (uint32_t)a1; l1 = (uint32_t)b1;
        o2 = (uint32_t)a2; l2 = (uint32_t)b2; o3 = (uint32_t)a3; l3 = (uint32_t)b3;
      }
      if (!act) l0 = 0;
      else if (l1 == 0 && l3 == 0) {  // no ring wrap: one segment for both steps (see pass 1)
        l0 += l2;
        l2 = 0;
      }
      const uint32_t my_sg = (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u) + (l3 ? 1u : 0u);
      const uint32_t my_pk = rp.sl_cnt | (my_sg << 16);
      const uint32_t my_tiles = tiles_of32(l0, ts) + tiles_of32(l1, ts) + tiles_of32(l2, ts) + tiles_of32(l3, ts);
      // (the ring holds < 2^31 bytes in a bulk pass, so 32-bit scans of one step's bytes are exact)
      const uint32_t my_bytes = al16_32(rp.sl0) + al16_32(rp.sl1);
      const uint32_t i_pk = wave_incl_scan_u32(my_pk);
      const uint32_t i_tiles = wave_incl_scan_u32(my_tiles);
      const uint32_t i_bytes = wave_incl_scan_u32(my_bytes);
      if (act) {
        uint64_t x_sl = c_sl + (i_pk & 0xFFFFu) - rp.sl_cnt;
        uint64_t x_sg = c_sg + (i_pk >> 16) - my_sg;
        uint32_t x_tiles = (uint32_t)(ntiles0 + c_tiles) + i_tiles - my_tiles;
        const uint64_t A = a0 + c_bytes + i_bytes - my_bytes;  // start of the open / next slice
        const uint32_t filled = s_in ? MINRD - s_in : 0;
        // the steps of one record are contiguous in the arena: step 1 fills the
        // open 256-byte slice exactly, step 2 starts the next slice right behind it
        uint64_t dst = (uint64_t)op.arena + A + filled;
        // header, padding and footer (ring_buffer.cc:146,173-180) are cleared by the
        // scatter waves of the record's first / last piece: GRDMA_SEG_TAG_* in grdma_dev.h
        const int last_piece = l3 ? 3 : (l2 ? 2 : (l1 ? 1 : 0));
        auto emit = [&](uint32_t off, uint32_t len, int piece) {
          if (len == 0) return;
          const uint64_t fl = GRDMA_SEG_ZERO_SRC | (piece == 0 ? GRDMA_SEG_TAG_HDR : 0) |
                              (piece == last_piece ? GRDMA_SEG_TAG_FTR : 0);
          plan->segs[nsegs0 + x_sg] = {dst, (uint64_t)(ring + off), (uint64_t)len, fl};
          plan->tile_prefix[nsegs0 + x_sg] = x_tiles;
          x_sg++;
          x_tiles += tiles_of32(len, ts);
          dst += len;
        };
        emit(o0, l0, 0);
        emit(o1, l1, 1);
        emit(o2, l2, 2);
        emit(o3, l3, 3);
        uint64_t sof = A;
        if (rp.sl0) {
          out_slices[nsl0 + x_sl].off = sof;
          out_slices[nsl0 + x_sl].len = rp.sl0;
          x_sl++;
          sof += al16_32(rp.sl0);
        }
        if (rp.sl1) {
          out_slices[nsl0 + x_sl].off = sof;
          out_slices[nsl0 + x_sl].len = rp.sl1;
        }
      }
      const uint32_t w_pk = __shfl(i_pk, 63, 64);
      c_sl += w_pk & 0xFFFFu;
      c_sg += w_pk >> 16;
      c_tiles += __shfl(i_tiles, 63, 64);
      c_bytes += __shfl(i_bytes, 63, 64);
    }
    const uint64_t tb5 = __builtin_amdgcn_s_memtime();
    if (tid == 0) s_dbg[12] += tb5 - tb4;
    // history: the processed records become the newest entries
    {
      const uint64_t hc = S.hist_count;
      const uint32_t first = cnt > GRDMA_RX_HIST ? cnt - GRDMA_RX_HIST : 0;
      __syncthreads();  // everyone is done reading s_hist through s_penc
      for (uint32_t k = first + tid; k < cnt; k += PLAN_THREADS)
        s_hist[(hc + k) % GRDMA_RX_HIST] = s_penc[RXP(k)];
    }
    // ---- credit accounting over the Recv steps (pair.cc:276-284), thread 0 ----------------
    if (tid == 0) {
      const uint64_t T = cap / 2;
      const uint64_t Ctot = s_xenc[RXP(cnt - 1)] + s_penc[RXP(cnt - 1)];
      uint64_t base = 0, thr = T - S.irs;
      bool crossed = false;
      uint64_t credit = S.credit, credit_head = S.credit_head;
      while (Ctot >= thr) {
        // first record whose running consumption (after its last step) reaches thr
        uint32_t lo = 0, hi = cnt - 1;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if ((uint64_t)s_xenc[RXP(mid)] + s_penc[RXP(mid)] >= thr) hi = mid; else lo = mid + 1;
        }
        const uint64_t n = s_n[RXP(lo)], e = s_penc[RXP(lo)];
        const rec_plan rp = replay_record(n, s_sin[RXP(lo)]);
        const uint64_t C2 = (uint64_t)s_xenc[RXP(lo)] + e;
        const uint64_t cons2 = rp.c2 ? rp.c2 + (round_up8(n) - n + 8) : 0;
        const uint64_t C1 = C2 - cons2;
        const uint64_t pos = (head + s_xenc[RXP(lo)]) & mask;
        if (rp.c2 && C1 >= thr) {  // crossed after the first step of a two-step record
          credit_head = (pos + 8 + rp.c1) & mask;
          base = C1;
        } else {
          credit_head = (pos + e) & mask;
          base = C2;
        }
        credit++;
        crossed = true;
        thr = base + T;
      }
      S.irs = crossed ? Ctot - base : S.irs + Ctot;
      S.credit = credit;
      S.credit_head = credit_head;
      S.head = (head + Ctot) & mask;
      S.mh = S.head;
      S.consumed_total += Ctot;
      S.bytes += tot_n;
      S.records += cnt;
      S.nslices += tot_sl;
      S.nsegs += tot_sg;
      S.ntiles += tot_tiles;
      S.a_off += tot_bytes;
      S.hist_count += cnt;
    }
    hand_over(cnt);
    return cnt;
  };

  // ============================================================== wave + scalar tiers
  // Runs on wave 0 with the state in registers; returns when the call is finished
  // (S.stop) or when the state is clean and the bulk tier should try again.
  auto wave_tier = [&]() {
    uint64_t head = S.head, mh = S.mh, remain = S.remain, irs = S.irs, leftover = S.leftover;
    uint64_t nslices = S.nslices, nsegs = S.nsegs, ntiles = S.ntiles, bytes = S.bytes;
    uint64_t consumed_total = S.consumed_total, records = S.records, a_off = S.a_off;
    uint64_t would_block = S.would_block, credit = S.credit, credit_head = S.credit_head;
    uint64_t hist_count = S.hist_count;
    uint32_t stop = 0, progressed = 0;
    const bool may_bulk = S.bulk_tries > 0;
    bool blocked = S.bulk_blocked != 0;
    uint64_t n_rounds = 0, n_fast = 0, n_scalar = 0;

    chain_walker w = {ring, cap, head, 0, 0, 0, false, room_at(head)};
    if (hist_count >= 1) w.h1 = s_hist[(hist_count - 1) % GRDMA_RX_HIST];
    if (hist_count >= 2) w.h2 = s_hist[(hist_count - 2) % GRDMA_RX_HIST];
    uint32_t chain_n = 0, chain_i = 0;
    if (S.hand_on) {  // what the last bulk pass verified beyond the records it took
      chain_n = S.hand_n;
      w.pos = S.hand_pos;
      w.room = room_at(w.pos);
      w.dry = S.hand_dry != 0;
      const uint32_t V = S.hand_v;  // sizes of the two records in front of w.pos
      if (V >= 2) {
        w.h1 = s_penc[RXP(V - 1)];
        w.h2 = s_penc[RXP(V - 2)];
      } else if (V == 1) {
        w.h2 = w.h1;
        w.h1 = s_penc[RXP(0)];
      }
    }

    auto hist_push = [&](uint64_t enc) {
      if (lane == 0) s_hist[hist_count % GRDMA_RX_HIST] = (uint32_t)enc;
      hist_count++;
    };
    auto refill = [&]() {
      while (chain_i == chain_n && !w.dry) {
        chain_n = chain_round(&w, s_chain, lane);
        chain_i = 0;
        n_rounds++;
      }
    };
    // keep at least 64 verified records queued while the ring has more
    auto top_up = [&]() {
      while (chain_n - chain_i < 64 && !w.dry) {
        const uint32_t k = chain_n - chain_i;
        uint64_t keep = 0;
        GRDMA_WAVE_CONVERGE();  // (no lane still reads the queue where it is about to be moved)
        if ((uint32_t)lane < k) keep = s_chain[chain_i + lane];
        GRDMA_WAVE_CONVERGE();  // (every lane holds its entry before the entries are overwritten)
        if ((uint32_t)lane < k) s_chain[lane] = keep;
        chain_i = 0;
        chain_n = k;
        chain_n += chain_round(&w, s_chain + k, lane);
        n_rounds++;
      }
    };
    auto next_ready = [&]() -> uint64_t {
      refill();
      return chain_i < chain_n ? s_chain[chain_i] : 0;
    };

    // PairPollable::Recv -> RingBufferPollable::Read(dst, capacity)
    // (pair.cc:264-286, ring_buffer.cc:122-191); returns the bytes copied.
    auto recv_step = [&](uint64_t dst, uint64_t capacity) -> uint64_t {
      uint64_t avail = remain;
      if (avail == 0) avail = next_ready();
      const uint64_t cpy = avail < capacity ? avail : capacity;
      if (cpy == 0) return 0;
      const uint64_t prev_mh = mh;
      if (remain == 0) {  // open the record, ring_buffer.cc:133-146
        if (lane == 0) *reinterpret_cast<uint64_t*>(ring + head) = 0;  // clear header
        mh = (head + 8) & mask;
        head = (head + 16 + round_up8(avail)) & mask;
        records++;
        chain_i++;
        hist_push(16 + round_up8(avail));
        progressed = 1;
      }
      // payload bytes [mh, mh+cpy) -> dst, at most two pieces at the wrap; the
      // copying wave clears them behind itself (ring_buffer.cc:160,164)
      const uint64_t l1 = cpy < cap - mh ? cpy : cap - mh;
      if (lane == 0) {
        plan->segs[nsegs] = {dst, (uint64_t)(ring + mh), l1, GRDMA_SEG_ZERO_SRC};
        plan->tile_prefix[nsegs] = (uint32_t)ntiles;
      }
      ntiles += tiles_of(l1, ts);
      nsegs++;
      if (cpy > l1) {
        if (lane == 0) {
          plan->segs[nsegs] = {dst + l1, (uint64_t)ring, cpy - l1, GRDMA_SEG_ZERO_SRC};
          plan->tile_prefix[nsegs] = (uint32_t)ntiles;
        }
        ntiles += tiles_of(cpy - l1, ts);
        nsegs++;
      }
      mh = (mh + cpy) & mask;
      remain = avail - cpy;
      if (remain == 0) {  // finish the record, ring_buffer.cc:169-182
        const uint64_t pad_end = round_up8(mh);
        if (lane == 0) {
          for (uint64_t q = mh; q < pad_end; q++) ring[q & mask] = 0;  // clear padded space
          *reinterpret_cast<uint64_t*>(ring + (pad_end & mask)) = 0;    // clear footer
        }
        mh = pad_end & mask;
        mh = (mh + 8) & mask;
      }
      const uint64_t consumed = (mh + cap - prev_mh) & mask;
      consumed_total += consumed;
      // credit return every cap/2 consumed bytes, pair.cc:276-284
      irs += consumed;
      if (irs >= cap / 2) {
        credit_head = mh;
        credit++;
        irs = 0;
      }
      return cpy;
    };

    // 64 whole records per step, one lane per record (see the file header)
    auto fast_chunk = [&]() -> uint32_t {
      if (cap > (1ull << 31)) return 0;  // 32-bit DPP scans below
      top_up();
      uint32_t k = chain_n - chain_i;
      if (k == 0) return 0;
      if (k > 64) k = 64;
      if (nslices + 128 > max_slices || nsegs + 256 + 520 > GRDMA_MAX_SEGS) return 0;
      const bool act0 = (uint32_t)lane < k;
      const uint64_t n = act0 ? s_chain[chain_i + lane] : 0;
      const uint64_t resets = __ballot(act0 && n >= 2 * MINRD - 1);
      const uint64_t below = resets & ((1ull << lane) - 1ull);
      const uint32_t from = below ? (64 - __builtin_clzll(below)) : 0;
      uint32_t s_in = 0;
      for (uint32_t i = from; i < (uint32_t)lane && act0; i++)
        s_in = read_space_after(s_chain[chain_i + i], s_in);
      const uint32_t s_out = read_space_after(n, s_in);
      const uint64_t clean = __ballot(act0 && s_out == 0);
      if (clean == 0) return 0;
      const uint32_t cnt = 64 - __builtin_clzll(clean);
      const bool act = (uint32_t)lane < cnt;
      const uint32_t enc = act ? (uint32_t)(16 + round_up8(n)) : 0;
      rec_plan rp = replay_record(act ? n : 0, act ? s_in : 0);
      if (!act) { rp.c1 = rp.c2 = 0; rp.sl_cnt = 0; rp.sl0 = rp.sl1 = 0; }
      const uint32_t done_bytes = (uint32_t)(al16(rp.sl0) + al16(rp.sl1));
      const uint32_t i_enc = wave_incl_scan_u32(enc);
      const uint32_t i_bytes = wave_incl_scan_u32(done_bytes);
      const uint32_t i_n = wave_incl_scan_u32(act ? (uint32_t)n : 0);
      const uint64_t tot_n = __shfl(i_n, 63, 64);
      if (a_off + tot_n + 32ull * cnt + 512 > op.arena_cap) return 0;
      const uint64_t x_enc = i_enc - enc, x_bytes = i_bytes - done_bytes;
      const uint64_t pos = (head + x_enc) & mask;
      const uint64_t pay = (pos + 8) & mask;
      const uint64_t A = a_off + x_bytes;
      const uint64_t filled = s_in ? MINRD - s_in : 0;
      uint64_t o0, l0, o1, l1, o2, l2, o3, l3;
      split_step(pay, 0, rp.c1, cap, &o0, &l0, &o1, &l1);
      split_step(pay, rp.c1, rp.c2, cap, &o2, &l2, &o3, &l3);
      const uint32_t sg_cnt = (l0 ? 1u : 0u) + (l1 ? 1u : 0u) + (l2 ? 1u : 0u) + (l3 ? 1u : 0u);
      const uint32_t my_tiles = tiles_of(l0, ts) + tiles_of(l1, ts) + tiles_of(l2, ts) + tiles_of(l3, ts);
      const uint32_t packed = rp.sl_cnt | (sg_cnt << 16);
      const uint32_t i_packed = wave_incl_scan_u32(packed);
      const uint32_t i_tiles = wave_incl_scan_u32(my_tiles);
      const uint64_t x_slices = (i_packed & 0xFFFFu) - rp.sl_cnt;
      uint64_t x_segs = (i_packed >> 16) - sg_cnt;
      uint64_t x_tiles = i_tiles - my_tiles;
      if (act) {
        uint64_t dst = (uint64_t)op.arena + A + filled;  // steps are contiguous in the arena
        const int last_piece = l3 ? 3 : (l2 ? 2 : (l1 ? 1 : 0));
        auto emit = [&](uint64_t off, uint64_t len, int piece) {
          if (len == 0) return;
          const uint64_t fl = GRDMA_SEG_ZERO_SRC | (piece == 0 ? GRDMA_SEG_TAG_HDR : 0) |
                              (piece == last_piece ? GRDMA_SEG_TAG_FTR : 0);
          plan->segs[nsegs + x_segs] = {dst, (uint64_t)(ring + off), len, fl};
          plan->tile_prefix[nsegs + x_segs] = (uint32_t)(ntiles + x_tiles);
          x_segs++;
          x_tiles += tiles_of(len, ts);
          dst += len;
        };
        emit(o0, l0, 0);
        emit(o1, l1, 1);
        emit(o2, l2, 2);
        emit(o3, l3, 3);
        uint64_t so = A;
        uint64_t xs = nslices + x_slices;
        if (rp.sl0) {
          out_slices[xs].off = so;
          out_slices[xs].len = rp.sl0;
          xs++;
          so += al16(rp.sl0);
        }
        if (rp.sl1) {
          out_slices[xs].off = so;
          out_slices[xs].len = rp.sl1;
        }
        s_hist[(hist_count + lane) % GRDMA_RX_HIST] = enc;
      }
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
        credit++;
        crossed = true;
        thr = base + cap / 2;
      }
      irs = crossed ? Ctot - base : irs + Ctot;
      const uint32_t t_packed = __shfl(i_packed, 63, 64);
      head = (head + Ctot) & mask;
      mh = head;
      consumed_total += Ctot;
      bytes += tot_n;
      records += cnt;
      nslices += t_packed & 0xFFFFu;
      nsegs += t_packed >> 16;
      ntiles += __shfl(i_tiles, 63, 64);
      a_off += __shfl(i_bytes, 63, 64);
      chain_i += cnt;
      hist_count += cnt;
      progressed = 1;
      return cnt;
    };

    if (op.raw_cap > 0) {
      // grdma_pair_recv(): exactly one Recv(buf, capacity)
      uint64_t n = recv_step((uint64_t)op.arena, op.raw_cap);
      if (lane == 0) {
        out_slices[0].off = 0;
        out_slices[0].len = n;
      }
      nslices = n ? 1 : 0;
      bytes = n;
      a_off = n;
      stop = 1;
    } else {
      for (;;) {
        if (!(nslices < max_slices && nsegs + 520 <= GRDMA_MAX_SEGS)) { stop = 1; break; }
        const bool clean = remain == 0 && leftover == 0;
        // hand back to the bulk tier once something moved since its last miss
        if (clean && may_bulk && (!blocked || progressed)) break;
        if (clean && fast_chunk() > 0) { n_fast++; continue; }
        n_scalar++;
        // rdma_continue_read, rdma_bp_posix.cc:306-317
        uint64_t readable = remain;
        if (readable == 0) readable = next_ready();
        const uint64_t alloc = leftover ? leftover : (readable > MINRD ? readable : MINRD);
        if (a_off + alloc > op.arena_cap) { stop = 1; break; }  // receive arena exhausted
        uint64_t total = 0;
        // rdma_do_read loop, rdma_bp_posix.cc:195-277
        while (total < alloc) {
          uint64_t n = recv_step((uint64_t)(op.arena + a_off + total), alloc - total);
          if (n == 0) break;
          total += n;
        }
        if (total == 0) {  // nothing ready: notify_on_read, the slice stays allocated
          leftover = alloc;
          would_block = 1;
          stop = 1;
          break;
        }
        leftover = alloc - total;  // grpc_slice_buffer_trim_end -> last_read_buffer
        if (lane == 0) {
          out_slices[nslices].off = a_off;
          out_slices[nslices].len = total;
        }
        nslices++;
        bytes += total;
        a_off = al16(a_off + total);
      }
    }
    GRDMA_WAVE_CONVERGE();  // (every lane has read S at the top before lane 0 replaces it)
    if (lane == 0) {
      S.head = head; S.mh = mh; S.remain = remain; S.irs = irs; S.leftover = leftover;
      S.nslices = nslices; S.nsegs = nsegs; S.ntiles = ntiles; S.bytes = bytes;
      S.consumed_total = consumed_total; S.records = records; S.a_off = a_off;
      S.would_block = would_block; S.credit = credit; S.credit_head = credit_head;
      S.hist_count = hist_count;
      S.stop = stop;
      S.hand_on = 0;
      if (progressed) S.bulk_blocked = 0;
      s_dbg[0] += n_rounds; s_dbg[1] += n_fast; s_dbg[2] += n_scalar;
    }
  };

  // ==================================================================== driver loop
  for (;;) {
    if (S.stop) break;  // uniform: S is only written between barriers
    const bool clean = S.remain == 0 && S.leftover == 0;
    uint32_t took = 0;
    if (S.bulk_tries > 0 && !(S.nslices + 512 <= max_slices && S.nsegs + 1024 <= GRDMA_MAX_SEGS)) {
      __syncthreads();
      if (tid == 0) S.bulk_tries = 0;  // no room for a bulk pass: the wave tier finishes
      __syncthreads();
    }
    if (clean && S.bulk_tries > 0 && !S.bulk_blocked) {
      took = bulk();
      __syncthreads();
      if (tid == 0) {
        if (took == 0xFFFFFFFFu) S.bulk_tries = 0; else S.bulk_tries--;
        if (took == 0) S.bulk_blocked = 1;
        if (took != 0xFFFFFFFFu) s_dbg[3] += took;
      }
      if (took == 0xFFFFFFFFu) took = 0;
      __syncthreads();
      if (took) continue;
    }
    // Every wave has evaluated the loop head (S.stop, S.remain, S.bulk_*) before wave 0 goes on to replace S
    // at the end of the wave tier: without this barrier a wave that is late by the duration of the wave tier
    // would read the NEXT state, take the other branch and miss the barrier below.  (Found with the host
    // emulation under CPU load, tests/cc/wave_emu.h: there a wave is an OS thread and can be late by that much.)
    __syncthreads();
    if (wave == 0) wave_tier();
    __syncthreads();
  }

  const uint64_t t_loop_end = __builtin_amdgcn_s_memtime();
  // history back to the connection
  __syncthreads();
  if (!express) {
    for (unsigned i = tid; i < GRDMA_RX_HIST; i += PLAN_THREADS) c->rx_hist[i] = s_hist[i];
    if (tid == 0 && S.hist_count >= 1) {
      c->rx_h1 = s_hist[(S.hist_count - 1) % GRDMA_RX_HIST];
      c->rx_h2 = S.hist_count >= 2 ? s_hist[(S.hist_count - 2) % GRDMA_RX_HIST] : 0;
    }
  }
  if (op.inline_apply && !express) {
    // small-message path: this workgroup also scatters the payload, clears it
    // behind itself and (below, thread 0) posts the credit -- one launch per drain
    if (tid == 0) {
      plan->nsegs = (uint32_t)S.nsegs;
      plan->ntiles = (uint32_t)S.ntiles;
      plan->tile_bytes = 1u << ts;
      plan->tile_prefix[S.nsegs] = (uint32_t)S.ntiles;
      plan->tag_base = (uint64_t)ring;
      plan->tag_mask = mask;
    }
    __syncthreads();
    run_plan<1024, true>(plan, wave, PLAN_THREADS / 64, lane);
    GRDMA_WAIT_VMEM();
    __syncthreads();
  }
  // (no early return: the resident engine calls this body inside a loop with
  // barriers, so every thread must fall out of the function together)
  if (tid == 0) {
    c->rx_hist_count = S.hist_count;
    c->rx_period = S.period;
    c->rx_period_retry_at = S.period_retry_at;
    c->pad3 = (S.period_strikes & 0xFFFFu) | (S.period_backoff << 16);

    const uint64_t head = S.head, mh = S.mh, nsegs = S.nsegs, nslices = S.nslices;
    plan->nsegs = (uint32_t)nsegs;
    plan->ntiles = (uint32_t)S.ntiles;
    plan->tile_bytes = 1u << ts;
    plan->tile_prefix[nsegs] = (uint32_t)S.ntiles;
    plan->bytes = S.bytes;
    plan->tag_base = (uint64_t)ring;
    plan->tag_mask = mask;

    // (counters: all loads before the first store to the connection, one round trip)
    const uint64_t o_total_read = pre_total_read, o_credit_msgs = pre_credit_msgs;
    const uint64_t o_rx_records = pre_rx_records, o_rx_rounds = pre_rx_rounds;
    const uint64_t o_slice_idx = op.append == 1 ? c->rx_slice_idx : 0;
    const uint64_t te_f = prof_time(prof) + ((o_total_read + o_rx_rounds) & 0);  // (counters loaded)
    // The connection block's own fields: nobody but the next drain of this connection -- this workgroup again, or a
    // kernel behind a boundary -- reads them, so on the latency path (inline_apply) they are stored BEHIND the sequence
    // word the host is waiting for; the launch-chain paths keep them in front of the result block.
    auto commit_conn = [&] {
      c->head = head;
      c->moving_head = mh;
      c->remain = S.remain;
      c->internal_read_size = S.irs;
      c->leftover_cap = S.leftover;
      c->total_read = o_total_read + S.bytes;
      c->credit_msgs = o_credit_msgs + S.credit;
      c->rx_records = o_rx_records + S.records;
      if (nslices) c->rx_rounds = o_rx_rounds + 1;
      if (op.append) {
        c->rx_arena_off = S.a_off;
        c->rx_slice_idx = o_slice_idx + nslices;
      }
      op.plan->blocks_done = 0;
      // updateStatus() (pair.cc:624-641) must not overtake the copy-out and the
      // zero-fill of the bytes it grants: the 16-byte report is posted by the last
      // workgroup of k_rx_apply.
      if (S.credit) c->status_send.remote_head = S.credit_head;
    };
    if (!op.inline_apply) commit_conn();
    if (pre_line != nullptr) {  // what HasMessage() on the host compares with the sender's arrival report
      pre_line->rx_head = head;
      pre_line->rx_remain = S.remain;
    }
    res->credit_head = S.credit_head;
    res->nslices = nslices;
    res->bytes = S.bytes;
    res->consumed = S.consumed_total;
    res->records = S.records;
    res->would_block = S.would_block;
    res->credit_sent = S.credit;
    res->head = head;
    res->moving_head = mh;
    res->remain = S.remain;
    res->arena_used = S.a_off;
    // (profiling stamps: not in latency mode, where the result block lives in host memory and
    // every word of it is a PCIe write in front of the release below)
    if (!op.inline_apply) {
    res->dbg[0] = t_begin;
    res->dbg[1] = __builtin_amdgcn_s_memtime();
    res->dbg[2] = s_dbg[0];
    res->dbg[3] = s_dbg[1];
    res->dbg[4] = s_dbg[2];
    res->dbg[5] = s_dbg[3];
    for (int q = 6; q < 13; q++) res->dbg[q] = s_dbg[q];
    res->dbg[13] = t_loop_end - t_begin;
    res->dbg[14] = s_dbg[14];
    res->dbg[15] = s_dbg[15];
    }
    // consumed ring bytes are always the contiguous range [mh0, mh)
    res->zero_off[0] = res->zero_off[1] = res->zero_len[0] = res->zero_len[1] = 0;
    if (S.consumed_total > 0) {
      if (mh > mh0) {
        res->zero_off[0] = mh0;
        res->zero_len[0] = mh - mh0;
      } else {
        res->zero_off[0] = mh0;
        res->zero_len[0] = cap - mh0;
        res->zero_off[1] = 0;
        res->zero_len[1] = mh;
      }
    }
    if (op.inline_apply) {
      // (relaxed stores: the single system-scope release on `seq` below publishes them;
      // every release is an L2 write-back, and this is the latency path)
      if (S.credit) {
        // (a watcher's drain: the sender runs on another CU, possibly behind another L2 -- the zero-fill of the bytes
        //  this report grants must have left this L2 before the sender can overwrite them.  Once per ring / 2 bytes.)
        if (op.inline_apply & 8u) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        grdma_status_report* ps = pre_ps;
        if (ps != nullptr)
          __hip_atomic_store(&ps->remote_head, S.credit_head, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        if (pre_pl != nullptr)
          __hip_atomic_store(&pre_pl->remote_head, S.credit_head, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(&res->commit_seq, op.seq_next ? op.seq_next : res->commit_seq + 1, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint64_t te_g = prof_time(prof);  // (commit stores issued)
    __hip_atomic_store(&res->seq, op.seq_next ? op.seq_next : res->seq + 1, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    if (op.inline_apply) commit_conn();
    if (express && prof) {
      const uint64_t te_h = __builtin_amdgcn_s_memtime();
      g_rx_express_ticks[5] += te_f - s_dbg[15];
      g_rx_express_ticks[6] += te_g - te_f;
      g_rx_express_ticks[7] += te_h - te_g;
    }
  }
}

// one wave per SIMD: the whole 512-register file is available, no spills
__global__ __launch_bounds__(PLAN_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_rx_plan(const grdma_rx_op* ops) {
  rx_plan_body(ops[blockIdx.x]);
}

// The receive plan of a streaming job's round: the straight-line steady-state body first (grdma_rx_fast.h); what it
// declines -- nothing has been written then -- goes through the general planner in the same launch.
// Launched in the steady-state body's shape (1024 threads); when that declines, waves 4-15 leave (a barrier only
// counts the waves that are still there) and waves 0-3 run the general planner, a 256-thread body -- under this
// kernel's 128-register budget, i.e. with spills: a job whose drains keep being declined is switched to the plain
// k_rx_plan by the host (grdma_stream_job_run).
__global__ __launch_bounds__(RXF_THREADS) RXF_KERNEL_ATTR
void k_rx_plan_job(const grdma_rx_op* ops) {
  if (rxf_body(ops[blockIdx.x])) return;  // (uniform)
  if (threadIdx.x >= PLAN_THREADS) return;
  rx_plan_body(ops[blockIdx.x]);
  if (threadIdx.x == 0) ops[blockIdx.x].result->dbg[9] = 0;  // (not rxf_body's stamps)
}

// Out-of-line copies for the resident engine: with both bodies inlined into its
// command loop the structurizer merges the loop tails and parks lane 0 behind
// the other lanes' next barrier (a deadlock); real calls keep the loop simple.
__device__ __attribute__((noinline)) void tx_plan_call(const grdma_tx_op* op) { tx_plan_body(*op); }
// the one-wave small Send by itself (the latency engine's doorbell wave: engine_body).  INLINE (round 6, second half): it has no
// barrier and runs in one wave, so the loop-tail problem above does not arise -- and as a real call it cost every unary
// Send the call and the callee's register saves in scratch memory, 0.23 us per hop (RTT 18.6 -> 18.1 us p50, three
// alternations of the two builds on one box).
__device__ __forceinline__ void tx_small_direct(const grdma_tx_op* op, uint64_t nslices, int lane) {
  tx_small_wave(*op, 0, op->byte_idx, nslices, lane, g_txs_len);
}
__device__ __attribute__((noinline)) void rx_plan_call(const grdma_rx_op* op) { rx_plan_body(*op); }

#include "grdma_job_kernels.inc"
#include "grdma_engine_kernels.inc"

}  // namespace

// groups watcher workgroups beside the command workgroup: one launch on the device (s), two under the emulator (s, ws)
extern "C" __attribute__((visibility("hidden"))) hipError_t grdma_launch_engine(grdma_engine_mbox* mb, grdma_watch_ctl* wc, uint64_t epoch,
                                                                                uint32_t groups, uint32_t flags, hipStream_t s, hipStream_t ws) {
#ifdef GRDMA_WAVE_EMU
  hipLaunchKernelGGL(k_engine, dim3(1), dim3(PLAN_THREADS), 0, s, mb, wc, epoch, flags);
  hipLaunchKernelGGL(k_watch, dim3(groups), dim3(PLAN_THREADS), 0, ws, mb, wc, epoch, flags);
#else
  (void)ws;
  (void)&k_watch;
  hipLaunchKernelGGL(k_engine, dim3(1 + groups), dim3(PLAN_THREADS), 0, s, mb, wc, epoch, flags);
#endif
  return hipGetLastError();
}

// profiling aid: phase ticks of the small-send wave inside the latency engine
// {slice loads, pricing, copies issued, copies acknowledged, bookkeeping stores, release, count}
extern "C" int grdma_tx_small_ticks(uint64_t out[8]) {
  unsigned long long v[8];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tx_small_ticks), sizeof(v)) != hipSuccess) return -1;
  for (int i = 0; i < 8; i++) out[i] = v[i];
  return 0;
}

// profiling aid: 10 ns ticks summed over the watchers' drains -- {-, arrival report published -> found by the watcher,
// found -> drain done, drains, found -> plan body entered, found -> bytes loaded, found -> stores issued, found -> sequence
// word stored (the last three: the single-wave path), -, command taken off the mailbox -> found}
extern "C" int grdma_watch_ticks(uint64_t out[12]) {
  unsigned long long v[12];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_watch_ticks), sizeof(v)) != hipSuccess) return -1;
  for (int i = 0; i < 12; i++) out[i] = v[i];
  return 0;
}
extern "C" int grdma_rx_express_ticks(uint64_t out[9]) {
  unsigned long long v[9];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_rx_express_ticks), sizeof(v)) != hipSuccess) return -1;
  for (int i = 0; i < 9; i++) out[i] = v[i];
  return 0;
}
extern "C" uint64_t grdma_watch_fast_drains(void) {   /* drains the watchers' single-wave path took (rxw_fast) */
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_watch_fast_drains), sizeof(v)) != hipSuccess) return 0;
  return (uint64_t)v;
}
extern "C" uint64_t grdma_express_drains(void) {
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_express_drains), sizeof(v)) != hipSuccess) return 0;
  return (uint64_t)v;
}

extern "C" __attribute__((visibility("hidden"))) const void* grdma_kernel_fn_rx_plan(void) { return reinterpret_cast<const void*>(&k_rx_plan); }
extern "C" __attribute__((visibility("hidden"))) const void* grdma_kernel_fn_rx_plan_job(void) { return reinterpret_cast<const void*>(&k_rx_plan_job); }
extern "C" __attribute__((visibility("hidden"))) uint32_t grdma_rx_plan_job_threads(void) { return RXF_THREADS; }
extern "C" __attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_plan_job(const grdma_rx_op* d_ops, uint32_t nops, hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_plan_job, dim3(nops), dim3(RXF_THREADS), 0, s, d_ops);
  return hipGetLastError();
}
// diagnostics: drains rxf_body took, and the ones it left to the general planner by reason (g_rx_fast_drains)
extern "C" int grdma_rx_fast_drains(uint64_t out[6]) {
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_rx_fast_drains), sizeof(v)) != hipSuccess) return -1;
  for (int i = 0; i < 6; i++) out[i] = v[i];
  return 0;
}
extern "C" int grdma_rx_table_cache_stats(uint64_t out[2]) {
  unsigned long long v[2] = {0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_rx_tab_stats), sizeof(v)) != hipSuccess) return -1;
  out[0] = v[0];
  out[1] = v[1];
  return 0;
}
extern "C" int grdma_rx_verdict_counts(uint64_t out[2]) {
  unsigned long long v[2] = {0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_rx_verdicts), sizeof(v)) != hipSuccess) return -1;
  out[0] = v[0];
  out[1] = v[1];
  return 0;
}
extern "C" __attribute__((visibility("hidden"))) const void* grdma_kernel_fn_plan_pair_mw(void) { return reinterpret_cast<const void*>(&k_plan_pair_mw); }
extern "C" __attribute__((visibility("hidden"))) uint32_t grdma_rx_multi_groups(void) { return RXM_G; }
extern "C" __attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_plan_mw(const grdma_rx_op* d_ops, uint32_t nops, hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_plan_mw, dim3(nops, RXM_G), dim3(PLAN_THREADS), 0, s, d_ops);
  return hipGetLastError();
}
extern "C" __attribute__((visibility("hidden"))) uint32_t grdma_tx_multi_groups(void) { return TXM_G; }
extern "C" __attribute__((visibility("hidden"))) uint32_t grdma_tx_multi_max_sends(void) { return TXM_MAX_SENDS_FOLDED; }
extern "C" __attribute__((visibility("hidden"))) uint32_t grdma_tx_multi_seq_sends(void) { return TXM_MAX_SENDS; }
extern "C" int grdma_debug_set_promise_wait(uint32_t v) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_promise_wait_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1;
}
extern "C" uint64_t grdma_wire_wait_runouts(void) {
  unsigned long long v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_wire_wait_runout), sizeof(v)) != hipSuccess) return ~0ull;
  return v;
}
extern "C" int grdma_tx_promise_counts(uint64_t out[4]) {
  unsigned long long v[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tx_promise), sizeof(v)) != hipSuccess) return -1;
  for (int i = 0; i < 4; i++) out[i] = v[i];
  return 0;
}
// (this translation unit's copy of the index body's counters: the pair kernel's Sends)
extern "C" __attribute__((visibility("hidden"))) int grdma_tx_fast_sends_pair(uint64_t out[2]) {
  unsigned long long v[2] = {0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tx_fast_sends), sizeof(v)) != hipSuccess) return -1;
  out[0] = v[0];
  out[1] = v[1];
  return 0;
}
extern "C" __attribute__((visibility("hidden"))) const void* grdma_kernel_fn_plan_pair(void) { return reinterpret_cast<const void*>(&k_plan_pair); }

extern "C" __attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_plan(const grdma_rx_op* d_ops, uint32_t nops,
                                           hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_plan, dim3(nops), dim3(PLAN_THREADS), 0, s, d_ops);
  return hipGetLastError();
}
