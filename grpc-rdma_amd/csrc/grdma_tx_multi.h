// txm_body: the send plan of a streaming job's Send priced from the index of the slice buffer (grdma_tx_fast.h) by
// SEVERAL small workgroups, nothing exchanged between them (included by grdma_rx_plan.hip; k_plan_pair_mw runs it
// next to the drain plan of the round before, grdma_rx_multi.h has the reason for the shape: one wave per SIMD, one
// record per thread, sixteen CUs' memory pipes for the plan stores).
//
// Same contract as txf_body -- PairPollable::Send (pair.cc:645-734): how many records go out whole, the short record
// behind them (CalculateWritableSize, ring_buffer.h:185-189), the <= 2 work requests of GetWriteRequests
// (ring_buffer.cc:261-330), the rdma_flush cursor (rdma_bp_posix.cc:480-493) -- and the same segments, tile prefix,
// wire plan, state and result block.  txf_body COUNTS the whole records over the whole Send (four per thread); here
// every workgroup finds the count for itself from the index: "record i goes out whole" is monotone in i, so the count
// is a search -- 256 samples of the encoded prefix in one round trip, then the <= 16 entries between two samples in a
// second one.  The record of a direct-wire Send whose payload crosses the ring end comes out of the same two round
// trips ("starts in front of the ring end" is monotone too).  Then every thread writes the segment of its own record,
// and the workgroup dispatched last, once all have counted in (plan->mw_arrive), writes totals, wire plan, state and result.
//
// Every workgroup reads the credit word (status_recv.remote_head) for itself: inside a job's chain nothing posts a
// credit while the planner pair runs (the scatter that does is the launch before or the launch behind), which is the
// only place this body is used -- txf_body's "ONE read for the whole Send" holds by construction.
#ifndef GRDMA_TX_MULTI_H
#define GRDMA_TX_MULTI_H
#include "grdma_tx_fast.h"

namespace {

// profiling aid (grdma_tx_promise_counts): Sends priced with a promised credit / with none in the drain's result /
// with one that was not between the posted head and the tail (an older block); waits that ran out
__device__ unsigned long long g_tx_promise[4] = {0, 0, 0, 0};
// test knob (grdma_debug_set_promise_wait): 0 = every Send workgroup polls up to 2^16 times; v > 0 = the Send workgroups
// with an odd index poll v - 1 times only (v = 1: their wait runs out before the first look)
__device__ uint32_t g_promise_wait_dbg = 0;

#define TXM_THREADS 256u
#define TXM_WAVES (TXM_THREADS / 64u)
#define TXM_G 16u   // workgroups per Send: 16 x 256 records
#define TXM_MAX_SENDS 2u  // consecutive Sends priced ONE AFTER THE OTHER (grdma_txf_ctl::sends <= this)
#define TXM_MAX_SENDS_FOLDED 64u  // more Sends per plan than that: priced as one cut of the index (see txm_body)
static_assert(TXM_G * TXM_THREADS >= GRDMA_TX_MAX_RECORDS, "one pass covers a Send");

// number of threads of the workgroup whose flag is set (every thread gets the count)
__device__ __forceinline__ uint32_t txm_count(bool flag, uint32_t* s_cnt) {
  const uint64_t b = __ballot(flag);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(b);
  __syncthreads();
  uint32_t n = 0;
#pragma unroll
  for (int w = 0; w < (int)TXM_WAVES; w++) n += s_cnt[w];
  __syncthreads();
  return n;
}

// (round 6) The index window: entries [start, start + m] of the slice table's three prefix arrays, copied into LDS by one
// round of loads before a pricing looks at the credit.  A pricing used to be a chain of dependent round trips to the
// index -- the first search level, the second, the short record, the totals, the last Send of a folded round: 4-5 of
// them, 6-7 us behind the promised credit at the reference's default knobs -- and every one of them is a look-up in
// these 40 KB now.  m >= TXM_WIN (a job with more records per Send than that): the look-ups go to memory as before.
#define TXM_WIN 2048u
struct txm_win {
  uint64_t enc[TXM_WIN], len[TXM_WIN];
  uint32_t tile[TXM_WIN];
};

// What a launch's drain promises its Send (the hand-over word of the plan, csrc/grdma_devfn.h)
struct txm_promise {
  bool kept;
  uint64_t credit_sent, credit_head;
};

// What one Send of the round was priced at (uniform over the workgroup, and the same in every workgroup).
struct txm_send {
  uint64_t tail0, start, byte_idx, offered;  // the state in front of it: remote_tail_, the rdma_flush cursor, what the write still holds
  uint64_t m, nrec, short_pay, st_short;     // slices offered, records that go out whole, the short record behind them
  uint64_t base_e, base_t, D, Dt;            // the index relative to the Send's first slice (the first one may be cut)
  uint32_t wrap_rec, wrap_extra;             // direct wire: the record whose payload crosses the ring end
  uint64_t nrec_total, sent, staged, ntiles, nsegs, new_tail, idx, bidx;
  bool performed;
  // a unit that folds several Sends (price(..., ns > 1)): how many of them carried records, and the LAST Send performed
  // -- its remote_tail_ in front, what it staged / sent / was offered, its records (an empty one when the ring was full)
  uint64_t sends_nonempty, last_tail0, last_staged, last_sent, last_offered, last_records;
  bool declined;
};

// Returns 0: not the last workgroup of this round to arrive; 1: the last one, the round is planned; 2: the last one,
// and the body declined (in every workgroup alike: nothing written) -- the caller runs the general planner.
//
// ctl->sends > 1 (a streaming job's grdma_stream_job_set_sends): the plan holds that many CONSECUTIVE Sends -- what
// rdma_flush does while the ring has room (rdma_bp_posix.cc:470-524: Send, advance the cursor, Send again) -- priced one
// after the other from the same index: Send k + 1 starts at the tail, cursor and free space Send k leaves (no credit
// arrives in between: this launch posts none), is skipped when Send k took the last slice, and its records follow
// Send k's in the gather plan, in the staging buffer (which then holds `sends` x staging_cap bytes) and in the ring,
// so the wire plan is still at most two pieces.  Every workgroup prices every Send (two round trips each); the
// records of all of them are then spread over the threads of all workgroups.  State and counters advance Send by
// Send (tx_rounds counts Sends); the result block is the LAST performed Send's.
//
// `promised` (a job's "promised credit" mode, k_plan_pair_mw): the result block of the drain planned by the receive
// workgroups of the SAME launch, committed before this body starts (the caller waited).  The credit that drain's
// scatter is going to post is then known -- credit_sent / credit_head are computed by the plan, the scatter only
// publishes them once the bytes are free -- and the Send is priced with it: nothing of this Send touches the ring
// before that scatter has run (the gather fills the staging buffer; the wire kernel is the launch behind the
// scatter's), so the round sees the credit the sequential schedule would give it, a round earlier than the paired one.
// (round 6) The promise is fetched through `wait_promise` -- a callable that returns what the drain of this launch is
// going to post once it is committed, or kept = false (no promise in this launch, or given up) -- and it is CALLED from inside the first pricing, behind
// everything that pricing loads without knowing the credit: the connection's state, the index entry of the cursor, the
// 256 samples of the first search level, the encoded sizes of the folded Sends, the payload prefix of the cursor.  Those
// round trips used to begin when the wait was over (the planner launch at the reference's default knobs: 24.5 us of
// kernel time, the drain plan's 14.7 followed by the Send's 6.6, profiles/r06_plan_phases.txt); now they are in flight
// while the Send's workgroups wait, and what is left behind the wait is the second search level, the short record and
// the totals.
template <class WaitFn>
__device__ __forceinline__ int txm_body(const grdma_tx_op& op_in, const grdma_txf_ctl* ctl, const uint32_t wg, const uint32_t nwg,
                                        WaitFn wait_promise, txm_win* const W = nullptr) {
  const grdma_tx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint32_t s_cnt[TXM_WAVES];

  // ---- state; what this body takes (as txf_body)
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap;
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  const uint64_t n = ctl->n;
  const uint64_t* const enc_pre = ctl->enc_pre;
  const uint64_t* const len_pre = ctl->len_pre;
  const uint32_t* const tile_pre = ctl->tile_pre;
  const bool direct = c->wire_direct != 0;
  // sends <= TXM_MAX_SENDS: that many pricings in a row.  More (the reference's default knobs: max_sge 30, a round is
  // dozens of Sends): ONE pricing that folds them -- every Send but the last takes exactly max_sge whole records and the
  // staging budget of a single Send (staging_cap) binds nowhere (checked; declined otherwise), so "record i goes out
  // whole" is the cumulative space rule st_n(i) + 8 <= free and the round is one cut of the index; what the Sends do
  // one by one (tx_rounds, the last Send's result, partial_write_) follows from the cut in closed form.
  const uint32_t sends_cfg = ctl->sends > 1 ? (ctl->sends < TXM_MAX_SENDS_FOLDED ? ctl->sends : TXM_MAX_SENDS_FOLDED) : 1u;
  const uint32_t FOLD = sends_cfg > TXM_MAX_SENDS ? sends_cfg : 1u;
  const uint32_t NS = FOLD > 1 ? 1u : sends_cfg;
  const bool ok = ctl->valid != 0 && ctl->slices == op.slices && n == op.nslices && op.use_cursor != 0 && !op.inline_copy &&
                  connected && (direct || op.wire_plan != nullptr) && cap <= (1ull << 31) &&
                  ctl->tile_shift == GRDMA_PLAN_TILE_SHIFT(cap) &&
                  (uint64_t)NS * GRDMA_TX_MAX_RECORDS + NS <= GRDMA_MAX_SEGS && (sends_cfg == 1 || direct || op.staging_alt != nullptr);
  // (that the workgroups cover the records offered is checked per pricing: m <= nwg x 256)
  // get_remote_head(), pair.h:229-233
  uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint64_t tail_at_start = c->remote_tail;
  bool promise_pending = true;
  auto take_promise = [&]() {  // (uniform: every thread of every workgroup of the Send comes away with the same head)
    if (!promise_pending) return;
    promise_pending = false;
    // (once per body, here: every load of the connection's state this workgroup makes has been requested -- they have
    //  returned, in every wave, before it counts itself in; the committing workgroup waits for all counts, see below)
    GRDMA_WAIT_LOADS();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&plan->mw_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const txm_promise promised = wait_promise();
    if (!promised.kept) return;
    const bool first = wg == 0 && tid == 0;
    if (promised.credit_sent != 0) {
      // (a head between the one posted so far and remote_tail_: a drain that found nothing leaves an older result block)
      const uint64_t ph = promised.credit_head;
      if (((ph - rhead) & mask) <= ((tail_at_start - rhead) & mask)) {
        rhead = ph;
        if (first) atomicAdd(&g_tx_promise[0], 1ull);
      } else if (first) {
        atomicAdd(&g_tx_promise[2], 1ull);
      }
    } else if (first) {
      atomicAdd(&g_tx_promise[1], 1ull);
    }
  };
  const uint32_t max_sge = c->max_sge;
  // (what the committing workgroup adds to, and the entries of "my" record if it belongs to the first Send: requested now)
  const uint64_t o_written = c->total_written, o_records = c->tx_records, o_rounds = c->tx_rounds;
  const uint64_t o_seq = op.result->seq;
  const uint64_t start0 = op.use_cursor == 1 ? c->tx_slice_idx : 0, bidx0 = op.use_cursor == 1 ? c->tx_byte_idx : 0;
  const uint64_t gi = (uint64_t)wg * TXM_THREADS + tid;
  const bool pre_mine = ok && start0 + gi < n;
  grdma_sge g_pre = {};
  uint64_t e_pre = 0;
  uint32_t tp_pre = 0;
  if (pre_mine) {
    g_pre = op.slices[start0 + gi];
    e_pre = enc_pre[start0 + gi];
    tp_pre = tile_pre[start0 + gi];
  }
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;
  uint8_t* const staging = op.staging_alt ? op.staging_alt : c->staging;

  // ---- one Send priced from the state in front of it (every thread gets the same answer)
  auto price = [&](uint64_t tail0, uint64_t start, uint64_t byte_idx, uint64_t offered, const uint32_t ns, const uint64_t m_cap) -> txm_send {
    txm_send q;
    q.tail0 = tail0; q.byte_idx = byte_idx; q.offered = offered;
    q.declined = false;
    q.nrec = q.short_pay = q.st_short = q.base_e = q.base_t = q.D = q.Dt = q.m = 0;
    q.wrap_rec = 0xFFFFFFFFu; q.wrap_extra = 0;
    q.performed = true;
    if (start > n) start = n;
    q.start = start;
    const uint64_t avail = n - start;
    uint64_t msge = max_sge;  // slices of one Send
    if (msge > GRDMA_TX_MAX_RECORDS - 1) msge = GRDMA_TX_MAX_RECORDS - 1;
    uint64_t m = avail;
    if (m > (uint64_t)ns * msge) m = (uint64_t)ns * msge;
    if (m > m_cap) q.declined = true;  // (more records than the workgroups of this launch cover: uniform)
    if (q.declined) m = 0;
    q.m = m;
    // ---- the index window (or, m >= TXM_WIN, loads): everything this pricing reads of the index WITHOUT knowing the
    //      credit is requested here -- for the first pricing of a launch with a promise, while the drain's workgroups
    //      are still at work
    const bool win = W != nullptr && m != 0 && m < TXM_WIN;  // (uniform)
    if (W != nullptr) __syncthreads();  // (the pricing in front of this one has read the window)
    if (win) {
      constexpr uint32_t NR = TXM_WIN / TXM_THREADS;
      uint64_t ve[NR], vl[NR];
      uint32_t vt[NR];
#pragma unroll
      for (uint32_t r = 0; r < NR; r++) {
        ve[r] = vl[r] = 0;
        vt[r] = 0;
        if ((uint64_t)r * TXM_THREADS <= m) {  // (uniform: a short table is one or two rounds of loads, not eight)
          const uint64_t k = tid + r * TXM_THREADS, kk = k <= m ? k : m;
          ve[r] = enc_pre[start + kk];
          vl[r] = len_pre[start + kk];
          vt[r] = tile_pre[start + kk];
        }
      }
#pragma unroll
      for (uint32_t r = 0; r < NR; r++) {
        const uint64_t k = tid + r * TXM_THREADS;
        if (k <= m) {
          W->enc[k] = ve[r];
          W->len[k] = vl[r];
          W->tile[k] = vt[r];
        }
      }
      __syncthreads();
    }
    auto E = [&](uint64_t k) -> uint64_t { return win ? W->enc[k] : enc_pre[start + k]; };
    auto Lp = [&](uint64_t k) -> uint64_t { return win ? W->len[k] : len_pre[start + k]; };
    auto Tp = [&](uint64_t k) -> uint32_t { return win ? W->tile[k] : tile_pre[start + k]; };
    const uint64_t step = (m + TXM_THREADS - 1) / TXM_THREADS;  // <= 16
    const uint64_t k_s = ((uint64_t)tid + 1) * step - 1;
    const bool in_s = m != 0 && k_s < m;
    const uint64_t kk_s = in_s ? k_s : 0;
    const uint64_t e0_s = m ? E(kk_s) : 0, e1_s = m ? E(kk_s + 1) : 0;
    const uint64_t a_f = (uint64_t)tid * msge, b_f = a_f + msge < m ? a_f + msge : m;
    const bool in_f0 = ns > 1 && a_f < m;
    const uint64_t ea_f = in_f0 ? E(a_f) : 0, eb_f = in_f0 ? E(b_f) : 0;
    const uint64_t lp_start = m ? Lp(0) : 0;
    // the first slice may have been sent in part: its record is shorter than the index says
    if (m) {
      const uint64_t len0 = Lp(1) - lp_start, l0 = sat_sub(len0, byte_idx);
      q.base_e = E(0);
      q.base_t = Tp(0);
      q.D = enc_size(l0) - (E(1) - q.base_e);              // (<= 0 as a signed number; modular arithmetic)
      q.Dt = ((l0 + TB - 1) >> ts) - (Tp(1) - q.base_t);
    }
    const uint64_t base_e = q.base_e, D = q.D;
    // st_i(k): staging offset of record k of this Send; st_n(k): where record k ends
    auto st_i_of = [&](uint64_t k, uint64_t e_k) -> uint64_t { return k == 0 ? 0 : e_k - base_e + D; };
    // ---- records that start in front of the ring end (direct wire): a monotone predicate the credit has no part in --
    //      both levels of its search
    uint64_t nfront;
    {
      const bool front = in_s && tail0 + st_i_of(kk_s, e0_s) + 8 < cap;
      const uint64_t lo_f = (uint64_t)txm_count(front, s_cnt) * step;
      const uint64_t kf = lo_f + tid;
      const bool in_f = tid < step && kf < m;
      const uint64_t e0f = in_f ? E(kf) : 0;
      const bool front2 = in_f && tail0 + st_i_of(kf, e0f) + 8 < cap;
      nfront = lo_f + txm_count(front2, s_cnt);
    }
    // ---- the credit: the head posted so far, or -- first pricing of a launch with a promise -- the head the drain of
    //      this launch is going to post (the wait is here)
    take_promise();
    const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
    const uint64_t free0 = cap - occupied0;
    // (several Sends folded: the free space is what binds; that no single Send exceeds staging_cap is checked below)
    const uint64_t room0 = (ns == 1 && S < free0) ? S : free0;
    // ---- whole records (st_n + 8 <= room0): a monotone predicate, a two-level search over the index
    uint64_t nrec;
    {
      const bool whole = in_s && (e1_s - base_e + D) + 8 <= room0;
      const uint64_t lo_w = (uint64_t)txm_count(whole, s_cnt) * step;
      const uint64_t kw = lo_w + tid;
      const bool in_w = tid < step && kw < m;
      const uint64_t e1w = in_w ? E(kw + 1) : 0;
      const bool whole2 = in_w && (e1w - base_e + D) + 8 <= room0;
      nrec = lo_w + txm_count(whole2, s_cnt);
    }
    q.nrec = nrec;
    // ---- (folded) no Send of the round may be bound by its own staging budget: Send t is the slices
    //      [t msge, (t + 1) msge) -- thread t looks at its encoded size, one round trip
    uint64_t st_send0 = 0;  // st of the first record of the Send the short record belongs to
    if (ns > 1) {
      const uint64_t a = a_f;
      const bool in = in_f0;
      const uint64_t ea = ea_f, eb = eb_f;
      const uint64_t sz = in ? (eb - ea + (a == 0 ? D : 0)) : 0;
      const bool over = in && sz + 8 > S;
      if (txm_count(over, s_cnt) != 0 || ns > TXM_THREADS) q.declined = true;
      const uint64_t j0 = (nrec / msge) * msge;
      st_send0 = (m && j0 < m) ? st_i_of(j0, E(j0)) : 0;
    }
    // ---- the short record behind the whole ones, the record that may cross the ring end: uniform loads, one round trip
    {
      const uint64_t wi = nfront ? nfront - 1 : 0;  // the last record that starts in front of the ring end
      const bool has_short = nrec < m;
      const uint64_t len_s = has_short ? Lp(nrec + 1) - Lp(nrec) : 0, e_s = m ? E(nrec < m ? nrec : m) : 0;
      const uint64_t len_w = (direct && nfront) ? Lp(wi + 1) - Lp(wi) : 0, e_w = (direct && nfront) ? E(wi) : 0;
      q.st_short = m ? st_i_of(nrec, e_s) : 0;  // st(nrec): where the short record starts, or st(m) = the end of the Send
      if (has_short) {
        // (pay = min(len, W(S - st), W(free0 - st)): it did not fit whole)
        uint64_t p = nrec == 0 ? sat_sub(len_s, byte_idx) : len_s;
        const uint64_t a = writable_of(sat_sub(S, q.st_short - st_send0)), b = writable_of(sat_sub(free0, q.st_short));
        if (a < p) p = a;
        if (b < p) p = b;
        q.short_pay = p;
      }
      const uint64_t nrec_total = nrec + (q.short_pay > 0 ? 1 : 0);
      if (direct && nfront && wi < nrec_total) {
        const uint64_t p = wi == nrec ? q.short_pay : (wi == 0 ? sat_sub(len_w, byte_idx) : len_w);
        const uint64_t pay_off = (tail0 + st_i_of(wi, e_w) + 8) & mask;
        if (pay_off + p > cap) {
          const uint64_t l1 = cap - pay_off;
          q.wrap_rec = (uint32_t)wi;
          q.wrap_extra = (uint32_t)(((l1 + TB - 1) >> ts) + ((p - l1 + TB - 1) >> ts) - ((p + TB - 1) >> ts));
        }
      }
    }
    // ---- totals behind the whole records (two uniform loads), the rdma_flush cursor walk (rdma_bp_posix.cc:480-493)
    q.nrec_total = nrec + (q.short_pay > 0 ? 1 : 0);
    q.sent = q.short_pay;
    q.ntiles = (q.short_pay + TB - 1) >> ts;
    if (nrec) {
      q.sent += Lp(nrec) - lp_start - byte_idx;
      q.ntiles += Tp(nrec) - q.base_t + q.Dt;
    }
    q.ntiles += q.wrap_extra;
    q.staged = (nrec || q.short_pay) ? q.st_short + (q.short_pay > 0 ? enc_size(q.short_pay) : 0) : 0;
    q.nsegs = q.nrec_total + (q.wrap_rec != 0xFFFFFFFFu ? 1 : 0);
    q.new_tail = (tail0 + q.staged) & mask;
    q.idx = start + nrec;
    q.bidx = 0;
    if (q.short_pay > 0) q.bidx = (nrec == 0 ? byte_idx : 0) + q.short_pay;
    else if (nrec == 0) q.bidx = byte_idx;
    // ---- the Sends one by one: how many carried records, and the last one performed
    q.sends_nonempty = q.nrec_total ? 1 : 0;
    q.last_tail0 = tail0; q.last_staged = q.staged; q.last_sent = q.sent; q.last_offered = offered; q.last_records = q.nrec_total;
    if (ns > 1 && q.nrec_total) {
      const uint64_t s_last = (q.nrec_total - 1) / msge, j0 = s_last * msge;  // the last Send that carried records
      q.sends_nonempty = s_last + 1;
      const uint64_t st0 = j0 ? st_i_of(j0, E(j0)) : 0;
      const uint64_t sent0 = j0 ? Lp(j0) - lp_start - byte_idx : 0;
      if (q.idx < n && q.sends_nonempty < ns) {  // data is left and so is a Send: it finds the ring full and sends nothing
        q.last_tail0 = q.new_tail; q.last_staged = 0; q.last_sent = 0; q.last_offered = offered - q.sent; q.last_records = 0;
      } else {
        q.last_tail0 = (tail0 + st0) & mask; q.last_staged = q.staged - st0; q.last_sent = q.sent - sent0;
        q.last_offered = offered - sent0; q.last_records = q.nrec_total - j0;
      }
    }
    return q;
  };

  txm_send Q[TXM_MAX_SENDS];
  uint64_t rec0[TXM_MAX_SENDS + 1], seg0[TXM_MAX_SENDS + 1], tile0[TXM_MAX_SENDS + 1], stg0[TXM_MAX_SENDS + 1];
  rec0[0] = seg0[0] = tile0[0] = stg0[0] = 0;
  uint32_t performed = 0;
#pragma unroll
  for (uint32_t k = 0; k < TXM_MAX_SENDS; k++) {
    Q[k].performed = false;
    Q[k].nrec_total = Q[k].nsegs = Q[k].ntiles = Q[k].staged = Q[k].m = 0;
    if (ok && k < NS) {  // (uniform)
      if (k == 0) {
        const uint64_t remaining = c->tx_remaining;
        Q[0] = price(tail_at_start, start0, bidx0,
                     op.use_cursor == 1 ? remaining : len_pre[n], FOLD, (uint64_t)nwg * TXM_THREADS / NS);
        performed = 1;
      } else if (Q[k - 1].performed && Q[k - 1].idx < n) {  // the write still holds data: rdma_flush sends again
        Q[k] = price(Q[k - 1].new_tail, Q[k - 1].idx, Q[k - 1].bidx, Q[k - 1].offered - Q[k - 1].sent, 1u, (uint64_t)nwg * TXM_THREADS / NS);
        performed = k + 1;
      }
    }
    rec0[k + 1] = rec0[k] + Q[k].nrec_total;
    seg0[k + 1] = seg0[k] + Q[k].nsegs;
    tile0[k + 1] = tile0[k] + Q[k].ntiles;
    stg0[k + 1] = stg0[k] + Q[k].staged;
  }
  take_promise();  // (a body that priced nothing -- not ok -- still counts itself out of the hand-over: uniform)
  bool declined = false;  // (a pricing met what it does not take: in every workgroup alike)
#pragma unroll
  for (uint32_t k = 0; k < TXM_MAX_SENDS; k++) declined |= Q[k].performed && Q[k].declined;
  const uint64_t nrec_all = declined ? 0 : rec0[TXM_MAX_SENDS];
  const bool table = op.sizes_out != nullptr && nrec_all <= GRDMA_HINT_MAX_RECORDS;  // (the size table holds two Sends' worth)
  const uint64_t t_priced = __builtin_amdgcn_s_memtime();

  // ---- my record: its segment and tile-prefix entry (AppendHeader / AppendFooter ride on the segment), as txf_body
  if (ok && gi < nrec_all) {
    // (which Send the record belongs to: selected field by field, the Sends stay in registers)
    txm_send q = Q[0];
    uint64_t rec_k = 0, seg_k = 0, tile_k = 0, stg_k = 0;
#pragma unroll
    for (uint32_t kk = 1; kk < TXM_MAX_SENDS; kk++)
      if (gi >= rec0[kk]) {
        q = Q[kk];
        rec_k = rec0[kk]; seg_k = seg0[kk]; tile_k = tile0[kk]; stg_k = stg0[kk];
      }
    const uint64_t i = gi - rec_k;
    const bool mine_pre = pre_mine && q.start + i == start0 + gi;  // (a record of the first Send)
    const grdma_sge g = mine_pre ? g_pre : op.slices[q.start + i];
    const uint64_t e0 = mine_pre ? e_pre : enc_pre[q.start + i];
    const uint32_t tp0 = mine_pre ? tp_pre : tile_pre[q.start + i];
    const uint64_t st_i = i == 0 ? 0 : e0 - q.base_e + q.D;
    const uint64_t p = i == q.nrec ? q.short_pay : (i == 0 ? sat_sub(g.len, q.byte_idx) : g.len);
    const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (p << GRDMA_SEG_TAG_LEN_SHIFT);
    if (table) op.sizes_out->n[gi] = (uint32_t)p;  // (for the drain of the same round: grdma_size_hint)
    const uint8_t* src = g.ptr + (i == 0 ? q.byte_idx : 0);
    const uint32_t t0 = (uint32_t)tile_k + (i == 0 ? 0u : (uint32_t)(tp0 - q.base_t + q.Dt));
    if (!direct) {
      plan->segs[seg_k + i] = {(uint64_t)(staging + stg_k + st_i + 8), (uint64_t)src, p, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
      plan->tile_prefix[seg_k + i] = t0;
    } else {
      uint8_t* const ring = c->peer_ring;
      const uint64_t pay_off = (q.tail0 + st_i + 8) & mask;
      const uint64_t seg = seg_k + i + ((uint32_t)i > q.wrap_rec ? 1 : 0);
      const uint32_t tx0 = t0 + ((uint32_t)i > q.wrap_rec ? q.wrap_extra : 0u);
      if ((uint32_t)i == q.wrap_rec) {
        const uint64_t l1 = cap - pay_off;
        plan->segs[seg] = {(uint64_t)(ring + pay_off), (uint64_t)src, l1, tagw | GRDMA_SEG_TAG_HDR};
        plan->segs[seg + 1] = {(uint64_t)ring, (uint64_t)(src + l1), p - l1, tagw | GRDMA_SEG_TAG_FTR};
        plan->tile_prefix[seg] = tx0;
        plan->tile_prefix[seg + 1] = tx0 + (uint32_t)((l1 + TB - 1) >> ts);
      } else {
        plan->segs[seg] = {(uint64_t)(ring + pay_off), (uint64_t)src, p, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
        plan->tile_prefix[seg] = tx0;
      }
    }
  }

  // ---- the LAST workgroup of the Send writes totals, wire plan, state and result -- or hands the Send to the general
  //      planner; the verdict is the same in every workgroup.  What it writes does not depend on the others' entries
  //      (everything is at the memory side when the launch ends), but the connection's state must not change under a
  //      workgroup that has yet to read it: every workgroup has counted itself in once its state loads had returned
  //      (take_promise), and the last one -- dispatched behind the others -- waits for all the counts.  (Round 6.
  //      Through round 5 the workgroups counted in HERE, behind their entries: a device-scope round trip at the tail of
  //      every planner launch; the count now travels while the Send waits for its promise and is priced.)
  if (wg != nwg - 1) return 0;
  if (tid == 0) {
    bool all = false;
    for (uint32_t spins = 0; spins < (1u << 22) && !all; spins++) {
      all = __hip_atomic_load(&plan->mw_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nwg;
      if (!all) __builtin_amdgcn_s_sleep(1);
    }
    if (!all) atomicAdd(&g_tx_promise[3], 1ull);  // (a wait that runs out: counted with the promise's, a test asserts zero)
    __hip_atomic_store(&plan->mw_arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!ok || declined) {  // (uniform, and the same in every workgroup)
    if (tid == 0) {
      atomicAdd(&g_tx_fast_sends[1], 1ull);
      op.result->dbg[11]++;  // (dbg[10] / dbg[11]: Sends of this result block priced from the index / declined)
      if (op.sizes_out != nullptr) op.sizes_out->count = 0;  // (the general planner leaves no size table)
    }
    return 2;
  }
  if (tid == 0) {
    txm_send L = Q[0];  // the last Send performed: the state it leaves, its result
#pragma unroll
    for (uint32_t kk = 1; kk < TXM_MAX_SENDS; kk++)
      if (Q[kk].performed) L = Q[kk];
    const uint64_t tail0 = Q[0].tail0;
    const uint64_t nsegs = seg0[TXM_MAX_SENDS], ntiles = tile0[TXM_MAX_SENDS], staged_all = stg0[TXM_MAX_SENDS];
    uint64_t sent_all = 0, rounds = 0;
#pragma unroll
    for (uint32_t k = 0; k < TXM_MAX_SENDS; k++) {
      sent_all += Q[k].performed ? Q[k].sent : 0;
      rounds += Q[k].performed ? Q[k].sends_nonempty : 0;
    }
    plan->nsegs = (uint32_t)nsegs;
    plan->ntiles = (uint32_t)ntiles;
    plan->tile_bytes = (uint32_t)TB;
    plan->tile_prefix[nsegs] = (uint32_t)ntiles;
    plan->bytes = sent_all;
    plan->tag_base = direct ? (uint64_t)c->peer_ring : (uint64_t)staging;
    plan->tag_mask = direct ? mask : ~0ull;
    // the <= 2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330 (of the last Send performed)
    grdma_tx_result* r = op.result;
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    if (L.last_staged > 0) {
      const uint64_t s1 = L.last_staged < cap - L.last_tail0 ? L.last_staged : cap - L.last_tail0;
      r->wr_off[0] = L.last_tail0;
      r->wr_len[0] = s1;
      r->wr_count = 1;
      if (L.last_tail0 + L.last_staged >= cap) {  // a record reached (or crossed) the ring end
        r->wr_off[1] = 0;
        r->wr_len[1] = L.last_staged - s1;
        r->wr_count = 2;
      }
    }
    // the loop-back wire: the Sends are contiguous in the staging buffer and in the ring -- two pieces at most
    grdma_plan* wp = op.wire_plan;
    if (wp != nullptr) {
      uint32_t ns = 0, nt = 0;
      if (staged_all > 0 && !direct) {
        const uint64_t seg1 = staged_all < cap - tail0 ? staged_all : cap - tail0;
        wp->segs[0] = {(uint64_t)(c->peer_ring + tail0), (uint64_t)staging, seg1, 0};
        wp->tile_prefix[0] = 0;
        nt = (uint32_t)((seg1 + TB - 1) >> ts);
        ns = 1;
        if (staged_all > seg1) {
          wp->segs[1] = {(uint64_t)c->peer_ring, (uint64_t)(staging + seg1), staged_all - seg1, 0};
          wp->tile_prefix[1] = nt;
          nt += (uint32_t)((staged_all - seg1 + TB - 1) >> ts);
          ns = 2;
        }
      }
      wp->nsegs = ns;
      wp->ntiles = nt;
      wp->tile_bytes = (uint32_t)TB;
      wp->tile_prefix[ns] = nt;
      wp->bytes = direct ? 0 : staged_all;
    }
    c->remote_tail = L.new_tail;
    c->partial_write = L.last_sent < L.last_offered ? 1 : 0;  // pair.cc:709
    c->total_written = o_written + sent_all;
    c->tx_records = o_records + nrec_all;
    c->tx_last_records = (uint32_t)L.last_records;
    if (rounds) c->tx_rounds = o_rounds + rounds;
    c->tx_slice_idx = L.idx;
    c->tx_byte_idx = L.bidx;
    c->tx_remaining = L.offered - L.sent;
    r->sent = L.last_sent;
    r->records = L.last_records;
    r->staged = L.last_staged;
    r->partial = L.last_sent < L.last_offered ? 1 : 0;
    r->new_remote_tail = L.new_tail;
    if (op.tail_out != nullptr) *op.tail_out = L.new_tail;
    if (op.sizes_out != nullptr) {
      op.sizes_out->start_off = tail0;
      op.sizes_out->count = table ? (uint32_t)nrec_all : 0u;
    }
    r->slice_idx = L.idx;
    r->byte_idx = L.bidx;
    r->done = (L.idx >= op.nslices) ? 1 : 0;
    r->dbg[0] = t_begin;
    r->dbg[1] = t_priced;
    r->dbg[6] = __builtin_amdgcn_s_memtime();
    r->dbg[7] = Q[0].m + (TXM_MAX_SENDS > 1 ? Q[TXM_MAX_SENDS - 1].m : 0);
    r->dbg[9] = 0xFA57;  // this Send was priced from the index
    r->dbg[10] += FOLD > 1 ? (L.sends_nonempty ? L.sends_nonempty : 1) : performed;
    atomicAdd(&g_tx_fast_sends[0], (unsigned long long)(FOLD > 1 ? (L.sends_nonempty ? L.sends_nonempty : 1) : performed));
    const uint64_t nxt = op.seq_next ? op.seq_next : o_seq + 1;
    __hip_atomic_store(&r->seq, nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return 1;
}

// (no promise in this launch / the caller has the drain's result block already)
__device__ __forceinline__ int txm_body(const grdma_tx_op& op_in, const grdma_txf_ctl* ctl, const uint32_t wg, const uint32_t nwg,
                                        const grdma_rx_result* promised = nullptr) {
  return txm_body(op_in, ctl, wg, nwg, [promised]() -> txm_promise {
    if (promised == nullptr) return txm_promise{false, 0, 0};
    return txm_promise{true, xwg_ld64<true>(&promised->credit_sent), xwg_ld64<true>(&promised->credit_head)};
  }, (txm_win*)nullptr);
}

}  // namespace
#endif  // GRDMA_TX_MULTI_H
