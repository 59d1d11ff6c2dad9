// Body of the send plan (PairPollable::Send arithmetic + rdma_flush cursor), shared by
// the k_tx_plan kernel and the persistent latency engine.  One 256-thread workgroup.
#ifndef GRDMA_TX_BODY_H
#define GRDMA_TX_BODY_H
#include "grdma_devfn.h"
#include "grdma_ops.h"

namespace {

// The sender's side of grdma_wire_report / grdma_hostline: called by ONE thread once every byte of a Send
// has landed in the peer ring (all waves of the caller have waited for their stores and met at a barrier,
// or the wire kernel in front of k_tx_commit has completed).  tail = remote_tail_ after the Send.
__device__ __forceinline__ void tx_publish(grdma_conn* c, uint64_t tail, uint64_t partial, uint64_t seq) {
  uint64_t* pw = c->peer_wire;
  if (pw != nullptr) __hip_atomic_store(pw, tail, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  grdma_hostline* pl = c->peer_line;
  if (pl != nullptr) __hip_atomic_store(&pl->wire_tail, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  grdma_hostline* ln = c->line;
  if (ln != nullptr) {
    __hip_atomic_store(&ln->remote_tail, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&ln->partial_write, partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (seq) __hip_atomic_store(&ln->tx_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ----------------------------------------------------------------------------
// Small sends (<= 64 slices, <= 64 KiB): the whole of PairPollable::Send on ONE
// wavefront, lane i = slice i.  Same arithmetic as the block-wide plan below
// (pay_i = min(len_i, W(S - st_i), W(free0 - st_i)), first short record ends the
// send), with the prefix sum on the DPP network and the first short record from a
// single __ballot; no LDS, no barriers.  The wave then writes the tags, moves
// the payload and performs the wire write itself.
// ----------------------------------------------------------------------------
// profiling aid: ticks spent in the phases of tx_small_wave (per translation unit; the latency
// engine's copy is read by grdma_tx_small_ticks)
static __device__ unsigned long long g_tx_small_ticks[8];
// profiling aid (grdma_watch_ticks): [0] the 100 MHz clock when the last small send published its arrival report;
// the watcher that finds it adds [1] += found - published, [2] += drain done - found, [3] += 1, [4] += fire -> body entered,
// [5] += bytes loaded - found, [6] += stores issued - found, [7] += sequence word stored - found; [8] the clock when the
// command workgroup took the last command off the mailbox, [9] += found - that
static __device__ unsigned long long g_watch_ticks[12];

__device__ __forceinline__ void tx_small_wave(const grdma_tx_op& op, uint64_t start,
                                              uint64_t byte_idx, uint64_t avail, int lane, uint64_t* lds = nullptr /* [160] */) {
  const bool prof = (op.inline_copy & GRDMA_OP_PROFILE) != 0;
  const uint64_t tk0 = prof_time(prof);
  grdma_conn* c = op.conn;
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap, tail0 = c->remote_tail;
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  const grdma_sge* sl = op.slices + start;
  const bool direct = c->wire_direct != 0;
  uint8_t* const dbase = direct ? c->peer_ring : c->staging;
  // (what the commit below needs of the connection block, in the round trip of the loads above: behind the first
  //  store to the block every load of it is a round trip of its own -- the arrival report's three pointers were)
  uint64_t* const pre_pw = c->peer_wire;
  grdma_hostline* const pre_pl = c->peer_line;
  grdma_hostline* const pre_ln = c->line;
  const uint64_t pre_written = c->total_written, pre_records = c->tx_records, pre_rounds = c->tx_rounds;
  const uint64_t pre_partial = c->partial_write;
  const bool peer_limited = c->peer_limited != 0 && pre_pw != nullptr;

  uint64_t len = 0;
  const uint8_t* src = nullptr;
  if ((uint64_t)lane < avail) {
    len = sl[lane].len;
    src = sl[lane].ptr;
    if (lane == 0) {
      len = sat_sub(len, byte_idx);
      src += byte_idx;
    }
  }
  const uint64_t tk1 = prof_time(prof) + (len & 0);  // (after the slice loads)
  // total offered (pair.cc:660-663)
  uint64_t offered = len;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) offered += __shfl_xor(offered, d, 64);
  if (op.use_cursor == 1) offered = c->tx_remaining;

  uint64_t m = avail < c->max_sge ? avail : c->max_sge;
  if (!connected) m = 0;
  const bool in_m = (uint64_t)lane < m;
  const uint32_t enc = in_m ? (uint32_t)enc_size(len < (cap << 1) ? len : (cap << 1)) : 0;
  const uint32_t incl = wave_incl_scan_u32(enc);
  const uint64_t st = incl - enc;
  const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
  const uint64_t free0 = cap - occupied0;
  uint64_t pay = len;
  {
    const uint64_t a = writable_of(sat_sub(S, st)), b = writable_of(sat_sub(free0, st));
    if (a < pay) pay = a;
    if (b < pay) pay = b;
  }
  const uint64_t shorts = __ballot(in_m && (pay < len || len == 0));
  const uint64_t nrec = shorts ? (uint64_t)__builtin_ctzll(shorts) : m;
  const uint64_t short_pay = shorts ? __shfl(pay, (int)nrec, 64) : 0;
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  const uint64_t my_pay = (uint64_t)lane < nrec ? len : ((uint64_t)lane == nrec ? short_pay : 0);
  // Σ enc over the whole records, plus the short one if any
  const uint64_t whole = nrec ? (uint64_t)__shfl(incl, (int)nrec - 1, 64) : 0;
  const uint64_t staged = whole + (short_pay > 0 ? enc_size(short_pay) : 0);
  uint64_t sent = my_pay;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sent += __shfl_xor(sent, d, 64);

  const uint64_t tk2 = prof_time(prof) + (sent & 0);  // (after pricing)
  const uint64_t seg1 = staged < cap - tail0 ? staged : cap - tail0;
  uint8_t* const peer_ring = c->peer_ring;
  bool sys_stores = false;   // every byte of this Send went into the peer ring past the caches
  if (nrec_total >= 1 && nrec_total <= 4 && __ballot(my_pay > 256) == 0 && peer_ring != nullptr && lds != nullptr) {
    // ---- unary-sized Sends: at most four records of at most 256 bytes ------------------------
    // The encoded bytes of the Send -- [length][payload, zero padded][footer] per record, back to back: what the <= 2
    // RDMA WRITEs of GetWriteRequests carry (ring_buffer.cc:261-330) -- are put together in LDS and leave as 8-byte
    // words, lane l the words l, l + 64, l + 128: into staging, and into the peer ring PAST the caches (the peer's
    // watcher workgroup reads them from another CU, possibly behind another L2 -- and then the arrival report needs
    // no write-back of this L2 in front of it).  A handful of wide stores instead of the thirty byte stores per record
    // this branch began with (2.5 us of a single wave's issue, profiles/r05_rtt_notes.txt).  For a reader that finds
    // records by their tags, the footers go out behind everything else (ring_buffer.cc:67-97).
    const uint32_t nwords = (uint32_t)(staged >> 3);   // <= 4 * (16 + 256) / 8 = 136
    uint8_t* const s8 = reinterpret_cast<uint8_t*>(lds);
    for (uint32_t w = (uint32_t)lane; w < nwords; w += 64) lds[w] = 0;
    GRDMA_WAVE_CONVERGE();
    if (my_pay > 0) {  // lane i owns the tags of record i
      lds[st >> 3] = my_pay;
      lds[(st + 8 + round_up8(my_pay)) >> 3] = GRDMA_FOOTER;
    }
    uint32_t fw[4];   // the footers' words
#pragma unroll
    for (int r = 0; r < 4; r++) {
      fw[r] = 0xFFFFFFFFu;
      if ((uint64_t)r < nrec_total) {
        const uint32_t p = (uint32_t)__shfl(my_pay, r, 64), so = (uint32_t)__shfl((uint32_t)st, r, 64);
        const uint8_t* sp = reinterpret_cast<const uint8_t*>(__shfl((uint64_t)src, r, 64));
        fw[r] = (so + 8 + ((p + 7u) & ~7u)) >> 3;
        for (uint32_t j = (uint32_t)lane; j < p; j += 64) s8[so + 8 + j] = sp[j];
      }
    }
    GRDMA_WAVE_CONVERGE();
    for (uint32_t w = (uint32_t)lane; w < nwords; w += 64) {
      const uint64_t val = lds[w];
      const bool is_f = w == fw[0] || w == fw[1] || w == fw[2] || w == fw[3];
      if (!direct) *reinterpret_cast<uint64_t*>(c->staging + 8ull * w) = val;
      if (peer_limited || !is_f)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(peer_ring + ((tail0 + 8ull * w) & mask)), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!peer_limited) {
      GRDMA_WAIT_VMEM();
      GRDMA_WAVE_CONVERGE();
      if (my_pay > 0)
        __hip_atomic_store(reinterpret_cast<uint64_t*>(peer_ring + ((tail0 + st + 8 + round_up8(my_pay)) & mask)), GRDMA_FOOTER, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    sys_stores = true;
  } else {
  // tags (AppendHeader / AppendFooter, ring_buffer.h:84-99) and zero padding
  uint64_t pay_off = 0;
  if (my_pay > 0) {
    const uint64_t hdr_off = direct ? ((tail0 + st) & mask) : st;
    pay_off = direct ? ((hdr_off + 8) & mask) : st + 8;
    const uint64_t foot_off = direct ? ((hdr_off + 8 + round_up8(my_pay)) & mask) : st + 8 + round_up8(my_pay);
    *reinterpret_cast<uint64_t*>(dbase + hdr_off) = my_pay;
    *reinterpret_cast<uint64_t*>(dbase + foot_off) = GRDMA_FOOTER;
    for (uint64_t q = my_pay; q < round_up8(my_pay); q++)
      dbase[direct ? ((pay_off + q) & mask) : pay_off + q] = 0;
  }
  // payload: the wave walks the records (K1)
  for (uint64_t r = 0; r < nrec_total; r++) {
    const uint64_t p = __shfl(my_pay, (int)r, 64);
    const uint64_t po = __shfl(pay_off, (int)r, 64);
    const uint8_t* sp = reinterpret_cast<const uint8_t*>(__shfl((uint64_t)src, (int)r, 64));
    uint64_t done = 0;
    while (done < p) {
      const uint64_t at = direct ? ((po + done) & mask) : po + done;
      uint64_t n = p - done;
      if (n > GRDMA_TILE_BYTES) n = GRDMA_TILE_BYTES;
      if (direct && at + n > cap) n = cap - at;  // ring end: continue at offset 0
      wave_copy_tile(dbase + at, sp + done, n, lane);
      done += n;
    }
  }
  // wire: the <= 2 RDMA WRITEs of GetWriteRequests (ring_buffer.cc:261-330)
  if (!direct && staged > 0 && c->peer_ring != nullptr) {
    GRDMA_WAIT_VMEM();  // staging complete before it is read back
    GRDMA_WAVE_CONVERGE();  // (by every lane: a lane's wire tile holds bytes its neighbours staged)
    for (uint64_t done = 0; done < staged;) {
      uint64_t n = staged - done;
      if (n > GRDMA_TILE_BYTES) n = GRDMA_TILE_BYTES;
      uint8_t* dst;
      if (done < seg1) {
        if (n > seg1 - done) n = seg1 - done;
        dst = c->peer_ring + tail0 + done;
      } else {
        dst = c->peer_ring + (done - seg1);
      }
      wave_copy_tile(dst, c->staging + done, n, lane);
      done += n;
    }
  }
  }
  const uint64_t tk3 = prof_time(prof);  // (copies issued)
  GRDMA_WAIT_VMEM();
  const uint64_t tk4 = prof_time(prof);  // (copies acknowledged)
  if (lane == 0) {
    // The arrival report FIRST (the wave has waited for every copy above): the peer's watcher workgroup (k_watch) polls
    // this word, and everything below -- plan headers, result block, counters -- is this end's own bookkeeping.  The
    // release carries the ring bytes out of this L2 (the watcher may sit behind another one).
    const uint64_t nt = (tail0 + staged) & mask;
    if (prof) __hip_atomic_store(&g_watch_ticks[0], (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (pre_pw != nullptr) {
      // (bytes stored past the caches and acknowledged are in memory: no write-back of this L2 in front of the report)
      if (sys_stores) __hip_atomic_store(pre_pw, nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else __hip_atomic_store(pre_pw, nt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (pre_pl != nullptr) __hip_atomic_store(&pre_pl->wire_tail, nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (lane == 0) {
    grdma_plan* plan = op.plan;
    plan->nsegs = 0;
    plan->ntiles = 0;
    plan->tile_bytes = 1u << GRDMA_PLAN_TILE_SHIFT(cap);
    plan->tile_prefix[0] = 0;
    plan->bytes = sent;
    if (op.wire_plan != nullptr) {
      op.wire_plan->nsegs = 0;
      op.wire_plan->ntiles = 0;
      op.wire_plan->tile_bytes = 1u << GRDMA_PLAN_TILE_SHIFT(cap);
      op.wire_plan->tile_prefix[0] = 0;
      op.wire_plan->bytes = direct ? 0 : staged;
    }
    const uint64_t new_tail = (tail0 + staged) & mask;
    grdma_tx_result* r = op.result;
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    if (staged > 0) {
      r->wr_off[0] = tail0;
      r->wr_len[0] = seg1;
      r->wr_count = 1;
      if (tail0 + staged >= cap) {
        r->wr_off[1] = 0;
        r->wr_len[1] = staged - seg1;
        r->wr_count = 2;
      }
    }
    uint64_t idx = start + nrec, bidx = 0;
    if (short_pay > 0) bidx = (nrec == 0 ? byte_idx : 0) + short_pay;
    else if (nrec == 0) bidx = byte_idx;
    // (counters: loads before the first store to the connection, one round trip)
    const uint64_t o_written = pre_written, o_records = pre_records, o_rounds = pre_rounds;
    c->remote_tail = new_tail;
    if (connected) c->partial_write = sent < offered ? 1 : 0;  // pair.cc:709 (Send returns at :652-655 when not connected)
    c->total_written = o_written + sent;
    c->tx_records = o_records + nrec_total;
    c->tx_last_records = (uint32_t)nrec_total;
    if (nrec_total) c->tx_rounds = o_rounds + 1;
    if (op.use_cursor) {
      c->tx_slice_idx = idx;
      c->tx_byte_idx = bidx;
      c->tx_remaining = offered - sent;
    }
    r->sent = sent;
    r->records = nrec_total;
    r->staged = staged;
    r->partial = connected ? (sent < offered ? 1 : 0) : pre_partial;
    r->new_remote_tail = new_tail;
    if (op.tail_out != nullptr) *op.tail_out = new_tail;
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = (idx >= op.nslices) ? 1 : 0;
    const uint64_t tk5 = prof_time(prof);  // (bookkeeping stores issued)
    // (the wave has waited for every copy above: the arrival report may go out)
    {  // the rest of tx_publish (the arrival report went out above), with the pointers loaded in front
      const uint64_t partial = connected ? (sent < offered ? 1 : 0) : pre_partial;
      if (pre_ln != nullptr) {
        __hip_atomic_store(&pre_ln->remote_tail, new_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pre_ln->partial_write, partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // the peer reads the ring in a later command / kernel; the host needs the result
    // block (pinned memory): a system-scope release on the sequence word covers it
    __hip_atomic_store(&r->seq, op.seq_next ? op.seq_next : r->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (prof) {
      const uint64_t tk6 = __builtin_amdgcn_s_memtime();
      g_tx_small_ticks[0] += tk1 - tk0;
      g_tx_small_ticks[1] += tk2 - tk1;
      g_tx_small_ticks[2] += tk3 - tk2;
      g_tx_small_ticks[3] += tk4 - tk3;
      g_tx_small_ticks[4] += tk5 - tk4;
      g_tx_small_ticks[5] += tk6 - tk5;
      g_tx_small_ticks[6] += 1;
    }
  }
}

// LDS index padding of the send plan's arrays (see tx_plan_body).  The first of them is declared here, outside the
// function: a workgroup that never plans a Send -- a watcher of the latency engine, k_watch -- keeps its own tables in
// these 34 KB (one kernel holds both bodies, and the compiler adds their LDS up).
#define TXP(i) ((i) + ((i) >> 4))
static __shared__ uint64_t g_txs_len[TXP(GRDMA_TX_MAX_RECORDS) + 1];

// ----------------------------------------------------------------------------
// k_tx_plan: PairPollable::Send arithmetic + rdma_flush cursor, one block per op
// ----------------------------------------------------------------------------
// All records of a Send are priced at once: enc_i = 16 + round_up8(len_i) is
// prefix-summed across the block (st_i), every record tests its own budget
// pay_i = min(len_i, W(S - st_i), W(free0 - st_i)) under the assumption that all
// earlier records went out whole, and an LDS atomic-min finds the first record
// that comes up short -- which is exactly where the reference's sequential loop
// stops (pair.cc:671-707).  Global loads/stores are
// striped over the block (record i -> thread i % 256) so they coalesce.
__device__ __forceinline__ void tx_plan_body(const grdma_tx_op& op_in) {
  const grdma_tx_op op = op_in;  // private copy: no re-fetch through the reference after stores
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint64_t s_wave[PLAN_THREADS / 64];
  // LDS index padding: threads walk contiguous runs of up to 16 records, a 128-byte
  // stride that would put all 64 lanes on the same banks; one extra slot per 16 breaks it
  auto& s_len = g_txs_len;                                         // len_i, later pay_i
  __shared__ uint64_t s_excl[TXP(GRDMA_TX_MAX_RECORDS + 1) + 1];  // st_i
  __shared__ unsigned int s_first_short;
  __shared__ unsigned int s_wrap_rec;
  const unsigned tid = threadIdx.x;
  uint64_t tdbg[8];
  tdbg[0] = __builtin_amdgcn_s_memtime();

  const uint64_t cap = c->cap, mask = cap - 1;
  const uint64_t S = c->staging_cap;
  const uint64_t tail0 = c->remote_tail;
  // get_remote_head(), pair.h:229-233
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  uint64_t start = op.use_cursor == 1 ? c->tx_slice_idx : 0;
  const uint64_t byte_idx =
      op.use_cursor == 1 ? c->tx_byte_idx : (op.use_cursor ? 0 : op.byte_idx);
  if (start > op.nslices) start = op.nslices;
  const uint64_t avail = op.nslices - start;
  const grdma_sge* sl = op.slices + start;

  // latency path: small sends run on one wavefront without LDS or barriers
  if (op.inline_copy && avail <= 64 && cap <= (1ull << 31)) {
    uint64_t small_bytes = 0;
    if (tid < avail) small_bytes = sl[tid].len;
    // (uniform decision: every thread sums the same <= 64 lengths through wave 0's lanes)
    __shared__ uint64_t s_small_total;
    if (tid < 64) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) small_bytes += __shfl_xor(small_bytes, d, 64);
      if (tid == 0) s_small_total = small_bytes;
    }
    __syncthreads();
    if (s_small_total <= (64ull << 10)) {
      if (tid < 64) tx_small_wave(op, start, byte_idx, avail, (int)tid, s_len);
      return;
    }
  }
  if (tid == 0) {
    s_first_short = 0xFFFFFFFFu;
    s_wrap_rec = 0xFFFFFFFFu;
  }

  // total bytes offered (pair.cc:660-663).  A streaming job keeps the running
  // remainder in the connection instead of re-summing the whole list per round.
  uint64_t offered;
  if (op.use_cursor == 1) {
    offered = c->tx_remaining;
    __syncthreads();
  } else {
    uint64_t part = 0;
    for (uint64_t i = tid; i < avail; i += PLAN_THREADS) part += sl[i].len;
    block_excl_scan(part, s_wave, &offered);
    offered = sat_sub(offered, byte_idx);
  }

  uint64_t m = avail;
  if (m > c->max_sge) m = c->max_sge;
  if (m > GRDMA_TX_MAX_RECORDS - 1) m = GRDMA_TX_MAX_RECORDS - 1;
  if (!connected) m = 0;
  // Pricing window.  A Send takes the records that fit the staging / credit budget, usually
  // far fewer than the slices offered (a 4 MiB ring takes ~150 of 4095).  The first
  // attempt prices only a window sized from what the previous Send took; if the budget
  // is not exhausted inside the window, the whole list is priced.  Same result either
  // way: records behind the first short one never matter.
  const uint64_t m_full = m;
  {
    const uint64_t guess = (uint64_t)c->tx_last_records + (c->tx_last_records >> 2) + 64;
    if (guess < m) m = guess;
  }

  tdbg[1] = __builtin_amdgcn_s_memtime();
  uint64_t t_loaded = 0, per = 0, room0 = 0, free0 = 0;
  constexpr int NPT = GRDMA_TX_MAX_RECORDS / PLAN_THREADS;
  uint64_t r_ptr[NPT], r_len[NPT];
  for (;;) {
  // slice table, striped: thread `tid` owns records tid, tid + 256, ...; all of its
  // 16-byte {ptr, len} loads are in flight together and stay in registers for the
  // segment pass below (no second trip to the table)
#pragma unroll
  for (int r = 0; r < NPT; r++) r_ptr[r] = r_len[r] = 0;
  if (m) {
    // (clamped, unconditional, and no LDS store in between: the table pointer is a
    // generic pointer, so every store to LDS would otherwise fence the loads behind it)
#pragma unroll
    for (int r = 0; r < NPT; r++) {
      const uint64_t i = tid + (uint64_t)r * PLAN_THREADS;
      const grdma_sge g = sl[i < m ? i : m - 1];
      r_ptr[r] = reinterpret_cast<uint64_t>(g.ptr);
      r_len[r] = g.len;
    }
#pragma unroll
    for (int r = 0; r < NPT; r++) {
      const uint64_t i = tid + (uint64_t)r * PLAN_THREADS;
      if (i < m) s_len[TXP(i)] = i == 0 ? sat_sub(r_len[r], byte_idx) : r_len[r];
    }
  }
  __syncthreads();
  t_loaded = __builtin_amdgcn_s_memtime();

  // st_i: each thread scans a contiguous run of `per` records out of LDS, and tests the
  // budget of each record in the same sweep (pay_i = min(len_i, W(S - st_i), W(free - st_i)):
  // the record is short exactly when len_i > W(min(S, free) - st_i), W being monotone)
  per = (m + PLAN_THREADS - 1) / PLAN_THREADS;
  const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
  free0 = cap - occupied0;
  room0 = S < free0 ? S : free0;
  {
    uint64_t chunk = 0;
    for (uint64_t k = 0; k < per; k++) {
      const uint64_t i = tid * per + k;
      if (i < m) {
        const uint64_t l = s_len[TXP(i)];
        // clamp so that sums cannot overflow; anything above 2*cap cannot fit anyway
        chunk += enc_size(l < (cap << 1) ? l : (cap << 1));
      }
    }
    uint64_t total_enc;
    uint64_t st = block_excl_scan(chunk, s_wave, &total_enc);
    unsigned int my_short = 0xFFFFFFFFu;
    for (uint64_t k = 0; k < per; k++) {
      const uint64_t i = tid * per + k;
      if (i < m) {
        s_excl[TXP(i)] = st;
        const uint64_t l = s_len[TXP(i)];
        // a zero payload ends the send exactly like the reference's `break`
        if (my_short == 0xFFFFFFFFu && (l == 0 || l > writable_of(sat_sub(room0, st))))
          my_short = (unsigned int)i;
        st += enc_size(l < (cap << 1) ? l : (cap << 1));
      }
    }
    if (tid == PLAN_THREADS - 1 || (tid * per < m && (tid + 1) * per >= m)) s_excl[TXP(m)] = st;
    if (m == 0 && tid == 0) s_excl[TXP(0)] = 0;
    // one LDS atomic per wave: runs are in thread order, so the lowest short lane of a
    // wave holds the wave's lowest short record
    const uint64_t bm = __ballot(my_short != 0xFFFFFFFFu);
    if (bm != 0 && (tid & 63) == (unsigned)__builtin_ctzll(bm)) atomicMin(&s_first_short, my_short);
  }
  tdbg[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (s_first_short != 0xFFFFFFFFu || m == m_full) break;  // uniform
  m = m_full;  // the window did not reach the end of the budget
  }
  const uint64_t fs = s_first_short;
  const uint64_t nrec = (fs != 0xFFFFFFFFu) ? fs : m;  // records [0, nrec) go out whole
  uint64_t short_pay = 0;
  if (fs != 0xFFFFFFFFu) {
    const uint64_t st = s_excl[TXP(fs)];
    const uint64_t a = writable_of(sat_sub(S, st));
    const uint64_t b = writable_of(sat_sub(free0, st));
    short_pay = s_len[TXP(fs)];
    if (a < short_pay) short_pay = a;
    if (b < short_pay) short_pay = b;
  }
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  // Σ enc over the whole records, plus the short one if any
  const uint64_t staged = s_excl[TXP(nrec)] + (short_pay > 0 ? enc_size(short_pay) : 0);
  __syncthreads();
  if (tid == 0 && short_pay > 0) s_len[TXP(nrec)] = short_pay;  // s_len[TXP(i)] is pay_i from here on
  __syncthreads();

  // destination of record i: staging + st_i, or the peer ring itself at
  // (tail0 + st_i) & mask when the wire is direct.
  const bool direct = c->wire_direct != 0;
  // (a pipelined job alternates between two staging buffers so that the plan of the next
  // Send can be laid out while the wire still reads the previous one)
  uint8_t* const staging = op.staging_alt ? op.staging_alt : c->staging;
  uint8_t* const dbase = direct ? c->peer_ring : staging;
  if (direct) {
    for (uint64_t i = tid; i < nrec_total; i += PLAN_THREADS) {
      const uint64_t pstart = (tail0 + s_excl[TXP(i)] + 8) & mask;
      if (pstart + s_len[TXP(i)] > cap) atomicMin(&s_wrap_rec, (unsigned int)i);
    }
  }
  __syncthreads();
  const uint64_t wrap_rec = s_wrap_rec;

  tdbg[3] = __builtin_amdgcn_s_memtime();
  // segments, striped.  AppendHeader / AppendFooter (ring_buffer.h:84-99) and the
  // deterministic zero padding (the reference leaves stale staging bytes there) are
  // written by the gather waves that move the first / last tile of each record: see
  // GRDMA_SEG_TAG_* in grdma_dev.h
  uint64_t sent_part = 0;
#pragma unroll
  for (int r = 0; r < NPT; r++) {
    const uint64_t i = tid + (uint64_t)r * PLAN_THREADS;
    if (i >= nrec_total) break;
    const uint64_t p = s_len[TXP(i)];
    const uint64_t st = s_excl[TXP(i)];
    sent_part += p;
    const uint64_t hdr_off = direct ? ((tail0 + st) & mask) : st;
    const uint64_t pay_off = direct ? ((hdr_off + 8) & mask) : st + 8;
    const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (p << GRDMA_SEG_TAG_LEN_SHIFT);
    const uint8_t* src = reinterpret_cast<const uint8_t*>(r_ptr[r]) + (i == 0 ? byte_idx : 0);
    const uint64_t seg = i + (i > wrap_rec ? 1 : 0);
    if (i == wrap_rec) {
      const uint64_t l1 = cap - pay_off;
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, l1, tagw | GRDMA_SEG_TAG_HDR};
      plan->segs[seg + 1] = {(uint64_t)dbase, (uint64_t)(src + l1), p - l1, tagw | GRDMA_SEG_TAG_FTR};
    } else {
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, p,
                         tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
    }
  }
  uint64_t sent;
  block_excl_scan(sent_part, s_wave, &sent);

  tdbg[4] = __builtin_amdgcn_s_memtime();
  // tile prefix per segment (contiguous runs again, out of LDS)
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;
  uint64_t ntiles;
  {
    const uint64_t per2 = (nrec_total + PLAN_THREADS - 1) / PLAN_THREADS;
    auto tiles_of = [&](uint64_t i, uint64_t* t1) -> uint64_t {
      const uint64_t p = s_len[TXP(i)];
      if (i == wrap_rec) {
        const uint64_t pay_off = (tail0 + s_excl[TXP(i)] + 16) & mask;  // (hdr_off + 8) & mask
        const uint64_t l1 = cap - ((tail0 + s_excl[TXP(i)] + 8) & mask);
        (void)pay_off;
        *t1 = (l1 + TB - 1) >> ts;
        return *t1 + ((p - l1 + TB - 1) >> ts);
      }
      *t1 = (p + TB - 1) >> ts;
      return *t1;
    };
    uint64_t chunk = 0, t1;
    for (uint64_t k = 0; k < per2; k++) {
      const uint64_t i = tid * per2 + k;
      if (i < nrec_total) chunk += tiles_of(i, &t1);
    }
    uint64_t x = block_excl_scan(chunk, s_wave, &ntiles);
    for (uint64_t k = 0; k < per2; k++) {
      const uint64_t i = tid * per2 + k;
      if (i < nrec_total) {
        const uint64_t seg = i + (i > wrap_rec ? 1 : 0);
        const uint64_t t = tiles_of(i, &t1);
        plan->tile_prefix[seg] = (uint32_t)x;
        if (i == wrap_rec) plan->tile_prefix[seg + 1] = (uint32_t)(x + t1);
        x += t;
      }
    }
  }
  const uint64_t nsegs = nrec_total + ((wrap_rec != 0xFFFFFFFFu) ? 1 : 0);

  tdbg[5] = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    plan->nsegs = (uint32_t)nsegs;
    plan->ntiles = (uint32_t)ntiles;
    plan->tile_bytes = (uint32_t)TB;
    plan->tile_prefix[nsegs] = (uint32_t)ntiles;
    plan->bytes = sent;
    plan->tag_base = (uint64_t)dbase;
    plan->tag_mask = direct ? mask : ~0ull;
    const uint64_t new_tail = (tail0 + staged) & mask;
    // the ≤2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330
    uint64_t seg1 = staged < cap - tail0 ? staged : cap - tail0;
    grdma_tx_result* r = op.result;
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    if (staged > 0) {
      r->wr_off[0] = tail0;
      r->wr_len[0] = seg1;
      r->wr_count = 1;
      if (tail0 + staged >= cap) {  // a record reached (or crossed) the ring end
        r->wr_off[1] = 0;
        r->wr_len[1] = staged - seg1;
        r->wr_count = 2;
      }
    }
    grdma_plan* wp = op.wire_plan;
    if (wp != nullptr) {
      uint32_t ns = 0, nt = 0;
      if (!direct && staged > 0) {
        wp->segs[0] = {(uint64_t)(c->peer_ring + tail0), (uint64_t)staging, seg1, 0};
        wp->tile_prefix[0] = 0;
        nt = (uint32_t)((seg1 + TB - 1) >> ts);
        ns = 1;
        if (staged > seg1) {
          wp->segs[1] = {(uint64_t)c->peer_ring, (uint64_t)(staging + seg1), staged - seg1, 0};
          wp->tile_prefix[1] = nt;
          nt += (uint32_t)((staged - seg1 + TB - 1) >> ts);
          ns = 2;
        }
      }
      wp->nsegs = ns;
      wp->ntiles = nt;
      wp->tile_bytes = (uint32_t)TB;
      wp->tile_prefix[ns] = nt;
      wp->bytes = direct ? 0 : staged;
    }
    // rdma_flush cursor walk, rdma_bp_posix.cc:480-493
    uint64_t idx = start + nrec;
    uint64_t bidx = 0;
    if (short_pay > 0) bidx = (nrec == 0 ? byte_idx : 0) + short_pay;
    else if (nrec == 0) bidx = byte_idx;
    // (counters: loads before the first store to the connection, one round trip)
    const uint64_t o_written = c->total_written, o_records = c->tx_records, o_rounds = c->tx_rounds;
    c->remote_tail = new_tail;
    if (connected) c->partial_write = sent < offered ? 1 : 0;  // pair.cc:709 (Send returns at :652-655 when not connected)
    c->total_written = o_written + sent;
    c->tx_records = o_records + nrec_total;
    c->tx_last_records = (uint32_t)nrec_total;
    if (nrec_total) c->tx_rounds = o_rounds + 1;
    if (op.use_cursor) {
      c->tx_slice_idx = idx;
      c->tx_byte_idx = bidx;
      c->tx_remaining = offered - sent;
    }
    r->sent = sent;
    r->records = nrec_total;
    r->staged = staged;
    r->partial = connected ? (sent < offered ? 1 : 0) : c->partial_write;
    r->new_remote_tail = new_tail;
    if (op.tail_out != nullptr) *op.tail_out = new_tail;
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = (idx >= op.nslices) ? 1 : 0;
    tdbg[6] = __builtin_amdgcn_s_memtime();
    if (!op.inline_copy) {  // profiling stamps; not in latency mode (PCIe writes in front of the release)
      for (int q = 0; q < 7; q++) r->dbg[q] = tdbg[q];
      r->dbg[8] = t_loaded;
      r->dbg[7] = m;
    }
  }
  if (op.inline_copy) {
    // small-message path: the planning workgroup moves the bytes itself (its
    // own plan stores are visible to its waves after the barrier)
    __syncthreads();
    run_plan<1024>(plan, tid >> 6, PLAN_THREADS / 64, tid & 63);
    if (op.wire_plan != nullptr && !direct) {
      GRDMA_WAIT_VMEM();
      __syncthreads();
      run_plan<1024>(op.wire_plan, tid >> 6, PLAN_THREADS / 64, tid & 63);
    }
    GRDMA_WAIT_VMEM();
    __syncthreads();
  }
  if (tid == 0) {
    grdma_tx_result* r = op.result;
    // one launch did it all (gather and wire included, every wave has waited for its stores): the
    // arrival report goes out here; otherwise k_tx_commit follows the wire kernel
    if (op.inline_copy) tx_publish(c, c->remote_tail, c->partial_write, 0);
    const uint64_t nxt = op.seq_next ? op.seq_next : r->seq + 1;
    __hip_atomic_store(&r->seq, nxt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}


// ----------------------------------------------------------------------------
// A BURST of Sends of one connection on ONE wavefront (k_tx_plan_seq, max_sge <= 64 -- the
// reference's default is 30): lane i = record i of the current Send.  The connection state
// (tail, cursor, remaining bytes, counters) lives in registers across the burst and is written back
// once; per Send: one coalesced load of <= 64 slice descriptors, the same pricing as the block-wide
// plan (pay_i = min(len_i, W(S - st_i), W(free0 - st_i)), the first short record ends the Send),
// one segment + tile-prefix entry per lane, the <= 2 wire requests, the result block.  No LDS, no
// barriers, no wait between Sends except for the descriptors.
// ops[k * stride] is Send k.  Preconditions checked by the caller: cap <= 2^30, max_sge <= 64,
// every op has use_cursor == 1 -- or, with `reset`, the first one has use_cursor == 2: the cursor starts at the
// head of the slice table and the bytes offered are the table's total, which this wave sums itself (a table of a
// few thousand entries at most: the caller checks).
// ----------------------------------------------------------------------------
__device__ __forceinline__ void tx_burst_wave(const grdma_tx_op* ops, uint32_t stride, uint32_t burst, int lane,
                                              bool reset = false) {
  const grdma_tx_op op0 = ops[0];
  grdma_conn* c = op0.conn;
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap;
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  const bool direct = c->wire_direct != 0;
  const uint64_t max_sge = c->max_sge;
  uint8_t* const peer_ring = c->peer_ring;
  uint8_t* const conn_staging = c->staging;
  const grdma_sge* const sl = op0.slices;
  const uint64_t nslices = op0.nslices;
  // running state
  uint64_t tail = c->remote_tail, idx = c->tx_slice_idx, bidx = c->tx_byte_idx, remaining = c->tx_remaining;
  uint64_t total_written = c->total_written, records = c->tx_records, rounds = c->tx_rounds;
  uint32_t last_records = c->tx_last_records, partial = (uint32_t)c->partial_write;
  if (reset) {  // (pair.cc:660-663: what a Send is offered is the sum over its whole slice list)
    uint64_t part = 0;
    for (uint64_t i = (uint64_t)lane; i < nslices; i += 64) part += sl[i].len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    idx = 0;
    bidx = 0;
    remaining = part;
  }
  if (idx > nslices) idx = nslices;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;

  bool dry = false;  // a Send accepted nothing: nothing changes until the peer reads, the rest accept nothing either
  for (uint32_t k = 0; k < burst; k++) {
    const grdma_tx_op op = ops[(size_t)k * stride];
    grdma_plan* const plan = op.plan;
    if (dry) {
      // the same outcome as pricing it: empty plans, nothing sent, state untouched
      // (partial_write was set by the Send that came up empty, pair.cc:709)
      if (lane == 0) {
        plan->nsegs = 0;
        plan->ntiles = 0;
        plan->tile_bytes = (uint32_t)TB;
        plan->tile_prefix[0] = 0;
        plan->bytes = 0;
        if (op.wire_plan != nullptr) {
          op.wire_plan->nsegs = 0;
          op.wire_plan->ntiles = 0;
          op.wire_plan->tile_bytes = (uint32_t)TB;
          op.wire_plan->tile_prefix[0] = 0;
          op.wire_plan->bytes = 0;
        }
        grdma_tx_result* r = op.result;
        r->wr_count = 0;
        r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
        r->sent = 0;
        r->records = 0;
        r->staged = 0;
        r->partial = partial;
        r->new_remote_tail = tail;
        if (op.tail_out != nullptr) *op.tail_out = tail;
        r->slice_idx = idx;
        r->byte_idx = bidx;
        r->done = idx >= nslices ? 1 : 0;
      }
      last_records = 0;
      continue;
    }
    const uint64_t avail = nslices - idx;
    uint64_t m = avail < max_sge ? avail : max_sge;
    if (!connected) m = 0;
    const bool in_m = (uint64_t)lane < m;
    // (unconditional load from a clamped index)
    const uint64_t li = idx + (uint64_t)lane < nslices ? idx + (uint64_t)lane : (nslices ? nslices - 1 : 0);
    grdma_sge g = {nullptr, 0};
    if (nslices) g = sl[li];
    uint64_t len = in_m ? g.len : 0;
    const uint8_t* src = g.ptr;
    if (lane == 0 && in_m) {
      len = sat_sub(len, bidx);
      src += bidx;
    }
    const uint64_t offered = remaining;
    const uint32_t enc = in_m ? (uint32_t)enc_size(len < (cap << 1) ? len : (cap << 1)) : 0;
    const uint32_t incl = wave_incl_scan_u32(enc);
    const uint64_t st = incl - enc;
    const uint64_t free0 = cap - ((tail + cap - rhead) & mask);
    uint64_t pay = len;
    {
      const uint64_t a = writable_of(sat_sub(S, st)), b = writable_of(sat_sub(free0, st));
      if (a < pay) pay = a;
      if (b < pay) pay = b;
    }
    const uint64_t shorts = __ballot(in_m && (pay < len || len == 0));
    const uint64_t nrec = shorts ? (uint64_t)__builtin_ctzll(shorts) : m;
    const uint64_t short_pay = shorts ? __shfl(pay, (int)nrec, 64) : 0;
    const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
    const uint64_t my_pay = (uint64_t)lane < nrec ? len : ((uint64_t)lane == nrec ? short_pay : 0);
    const uint64_t whole = nrec ? (uint64_t)__shfl(incl, (int)nrec - 1, 64) : 0;
    const uint64_t staged = whole + (short_pay > 0 ? enc_size(short_pay) : 0);
    uint64_t sent = my_pay;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sent += __shfl_xor(sent, d, 64);

    // segments (see the block-wide plan: tags ride on the segments, GRDMA_SEG_TAG_*)
    uint8_t* const staging = op.staging_alt ? op.staging_alt : conn_staging;
    uint8_t* const dbase = direct ? peer_ring : staging;
    const bool mine = (uint64_t)lane < nrec_total;
    const uint64_t hdr_off = direct ? ((tail + st) & mask) : st;
    const uint64_t pay_off = direct ? ((hdr_off + 8) & mask) : st + 8;
    const uint64_t wraps = __ballot(mine && direct && pay_off + my_pay > cap);
    const uint32_t wrap_rec = wraps ? (uint32_t)__builtin_ctzll(wraps) : 0xFFFFFFFFu;
    const uint64_t l1 = cap - pay_off;  // (only meaningful for the wrapping record)
    uint32_t t1 = (uint32_t)((my_pay + TB - 1) >> ts), tiles = t1;
    if ((uint32_t)lane == wrap_rec) {
      t1 = (uint32_t)((l1 + TB - 1) >> ts);
      tiles = t1 + (uint32_t)((my_pay - l1 + TB - 1) >> ts);
    }
    if (!mine) tiles = 0;
    const uint32_t tincl = wave_incl_scan_u32(tiles);
    const uint32_t tx0 = tincl - tiles;
    const uint32_t ntiles = (uint32_t)__shfl(tincl, 63, 64);
    if (mine) {
      const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (my_pay << GRDMA_SEG_TAG_LEN_SHIFT);
      const uint64_t seg = (uint64_t)lane + ((uint32_t)lane > wrap_rec ? 1 : 0);
      if ((uint32_t)lane == wrap_rec) {
        plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, l1, tagw | GRDMA_SEG_TAG_HDR};
        plan->segs[seg + 1] = {(uint64_t)dbase, (uint64_t)(src + l1), my_pay - l1, tagw | GRDMA_SEG_TAG_FTR};
        plan->tile_prefix[seg] = tx0;
        plan->tile_prefix[seg + 1] = tx0 + t1;
      } else {
        plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, my_pay, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
        plan->tile_prefix[seg] = tx0;
      }
    }
    const uint64_t new_tail = (tail + staged) & mask;
    uint64_t nidx = idx + nrec, nbidx = 0;
    if (short_pay > 0) nbidx = (nrec == 0 ? bidx : 0) + short_pay;
    else if (nrec == 0) nbidx = bidx;
    const uint32_t npartial = connected ? (sent < offered ? 1u : 0u) : partial;
    if (lane == 0) {
      const uint32_t nsegs = (uint32_t)nrec_total + (wrap_rec != 0xFFFFFFFFu ? 1u : 0u);
      plan->nsegs = nsegs;
      plan->ntiles = ntiles;
      plan->tile_bytes = (uint32_t)TB;
      plan->tile_prefix[nsegs] = ntiles;
      plan->bytes = sent;
      plan->tag_base = (uint64_t)dbase;
      plan->tag_mask = direct ? mask : ~0ull;
      // the <= 2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330
      const uint64_t seg1 = staged < cap - tail ? staged : cap - tail;
      grdma_tx_result* r = op.result;
      r->wr_count = 0;
      r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
      if (staged > 0) {
        r->wr_off[0] = tail;
        r->wr_len[0] = seg1;
        r->wr_count = 1;
        if (tail + staged >= cap) {
          r->wr_off[1] = 0;
          r->wr_len[1] = staged - seg1;
          r->wr_count = 2;
        }
      }
      grdma_plan* wp = op.wire_plan;
      if (wp != nullptr) {
        uint32_t ns = 0, nt = 0;
        if (!direct && staged > 0) {
          wp->segs[0] = {(uint64_t)(peer_ring + tail), (uint64_t)staging, seg1, 0};
          wp->tile_prefix[0] = 0;
          nt = (uint32_t)((seg1 + TB - 1) >> ts);
          ns = 1;
          if (staged > seg1) {
            wp->segs[1] = {(uint64_t)peer_ring, (uint64_t)(staging + seg1), staged - seg1, 0};
            wp->tile_prefix[1] = nt;
            nt += (uint32_t)((staged - seg1 + TB - 1) >> ts);
            ns = 2;
          }
        }
        wp->nsegs = ns;
        wp->ntiles = nt;
        wp->tile_bytes = (uint32_t)TB;
        wp->tile_prefix[ns] = nt;
        wp->bytes = direct ? 0 : staged;
      }
      r->sent = sent;
      r->records = nrec_total;
      r->staged = staged;
      r->partial = npartial;
      r->new_remote_tail = new_tail;
      if (op.tail_out != nullptr) *op.tail_out = new_tail;
      r->slice_idx = nidx;
      r->byte_idx = nbidx;
      r->done = nidx >= nslices ? 1 : 0;
    }
    // the state the next Send starts from (PairPollable members + the rdma_flush cursor)
    tail = new_tail;
    idx = nidx;
    bidx = nbidx;
    remaining = offered - sent;
    partial = npartial;
    total_written += sent;
    records += nrec_total;
    last_records = (uint32_t)nrec_total;
    if (nrec_total) rounds++;
    else dry = true;
  }
  if (lane == 0) {
    c->remote_tail = tail;
    c->partial_write = partial;
    c->total_written = total_written;
    c->tx_records = records;
    c->tx_last_records = last_records;
    c->tx_rounds = rounds;
    c->tx_slice_idx = idx;
    c->tx_byte_idx = bidx;
    c->tx_remaining = remaining;
  }
}

// A burst that must not run: it was queued behind a write whose last Send turned out to be partial (grdma_tx_op::
// use_cursor == 3, grdma_endpoint_write_queue).  Empty plans, results that say so (done = 2), and NOTHING of the
// connection's state is touched -- the cursor still belongs to the write in front, which the host continues first.
__device__ __forceinline__ void tx_burst_skip(const grdma_tx_op* ops, uint32_t stride, uint32_t burst, int lane) {
  if (lane != 0) return;
  const grdma_conn* c = ops[0].conn;
  const uint64_t tail = c->remote_tail;
  const uint32_t TB = 1u << GRDMA_PLAN_TILE_SHIFT(c->cap);
  for (uint32_t k = 0; k < burst; k++) {
    const grdma_tx_op op = ops[(size_t)k * stride];
    for (grdma_plan* pl : {op.plan, op.wire_plan}) {
      if (pl == nullptr) continue;
      pl->nsegs = 0;
      pl->ntiles = 0;
      pl->tile_bytes = TB;
      pl->tile_prefix[0] = 0;
      pl->bytes = 0;
    }
    grdma_tx_result* r = op.result;
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    r->sent = 0;
    r->records = 0;
    r->staged = 0;
    r->partial = (uint32_t)c->partial_write;
    r->new_remote_tail = tail;
    if (op.tail_out != nullptr) *op.tail_out = tail;
    r->slice_idx = 0;
    r->byte_idx = 0;
    r->done = 2;
  }
}

}  // namespace
#endif  // GRDMA_TX_BODY_H
