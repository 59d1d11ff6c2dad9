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
// and the last workgroup to arrive (plan->mw_arrive) writes totals, wire plan, state and result.
//
// Every workgroup reads the credit word (status_recv.remote_head) for itself: inside a job's chain nothing posts a
// credit while the planner pair runs (the scatter that does is the launch before or the launch behind), which is the
// only place this body is used -- txf_body's "ONE read for the whole Send" holds by construction.
#ifndef GRDMA_TX_MULTI_H
#define GRDMA_TX_MULTI_H
#include "grdma_tx_fast.h"

namespace {

#define TXM_THREADS 256u
#define TXM_WAVES (TXM_THREADS / 64u)
#define TXM_G 16u   // workgroups per Send: 16 x 256 records
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

// Returns 0: not the last workgroup of this Send to arrive; 1: the last one, the Send is planned; 2: the last one,
// and the body declined (in every workgroup alike: nothing written) -- the caller runs the general planner.
__device__ __forceinline__ int txm_body(const grdma_tx_op& op_in, const grdma_txf_ctl* ctl, const uint32_t wg, const uint32_t nwg) {
  const grdma_tx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint32_t s_cnt[TXM_WAVES];
  __shared__ uint32_t s_last;

  // ---- state; what this body takes (as txf_body)
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap, tail0 = c->remote_tail;
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  const uint64_t n = ctl->n;
  const uint64_t* const enc_pre = ctl->enc_pre;
  const uint64_t* const len_pre = ctl->len_pre;
  const uint32_t* const tile_pre = ctl->tile_pre;
  const bool direct = c->wire_direct != 0;
  const bool ok = ctl->valid != 0 && ctl->slices == op.slices && n == op.nslices && op.use_cursor != 0 && !op.inline_copy &&
                  connected && (direct || op.wire_plan != nullptr) && cap <= (1ull << 31) &&
                  ctl->tile_shift == GRDMA_PLAN_TILE_SHIFT(cap) && nwg * TXM_THREADS >= GRDMA_TX_MAX_RECORDS - 1;
  // get_remote_head(), pair.h:229-233
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint64_t start = op.use_cursor == 1 ? c->tx_slice_idx : 0;
  const uint64_t byte_idx = op.use_cursor == 1 ? c->tx_byte_idx : 0;
  const uint64_t remaining = c->tx_remaining;
  uint32_t max_sge = c->max_sge;

  uint64_t nrec = 0, short_pay = 0, st_short = 0, base_e = 0, base_t = 0, D = 0, Dt = 0, m = 0;
  uint32_t wrap_rec = 0xFFFFFFFFu, wrap_extra = 0;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;
  uint64_t free0 = 0, room0 = 0;
  const grdma_sge* sl = op.slices;
  uint8_t* const staging = op.staging_alt ? op.staging_alt : c->staging;
  if (ok) {  // (uniform)
    if (start > n) start = n;
    const uint64_t avail = n - start;
    m = avail;
    if (m > max_sge) m = max_sge;
    if (m > GRDMA_TX_MAX_RECORDS - 1) m = GRDMA_TX_MAX_RECORDS - 1;
    sl = op.slices + start;
    const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
    free0 = cap - occupied0;
    room0 = S < free0 ? S : free0;
    // the first slice may have been sent in part: its record is shorter than the index says
    if (m) {
      const uint64_t len0 = sl[0].len, l0 = sat_sub(len0, byte_idx);
      base_e = enc_pre[start];
      base_t = tile_pre[start];
      D = enc_size(l0) - (enc_pre[start + 1] - base_e);              // (<= 0 as a signed number; modular arithmetic)
      Dt = ((l0 + TB - 1) >> ts) - (tile_pre[start + 1] - base_t);
    }
    // st_i(k): staging offset of record k of this Send; st_n(k): where record k ends
    auto st_i_of = [&](uint64_t k, uint64_t e_k) -> uint64_t { return k == 0 ? 0 : e_k - base_e + D; };
    // ---- whole records (st_n + 8 <= room0) and records that start in front of the ring end (direct wire): two
    //      monotone predicates, one two-level search over the index
    const uint64_t step = (m + TXM_THREADS - 1) / TXM_THREADS;  // <= 16
    uint32_t c_whole, c_front;
    {
      const uint64_t k = ((uint64_t)tid + 1) * step - 1;
      const bool in = m != 0 && k < m;
      const uint64_t kk = in ? k : 0;
      const uint64_t e0 = m ? enc_pre[start + kk] : 0, e1 = m ? enc_pre[start + kk + 1] : 0;
      const bool whole = in && (e1 - base_e + D) + 8 <= room0;
      const bool front = in && tail0 + st_i_of(kk, e0) + 8 < cap;
      c_whole = txm_count(whole, s_cnt);
      c_front = txm_count(front, s_cnt);
    }
    uint64_t nfront;
    {
      const uint64_t lo_w = (uint64_t)c_whole * step, lo_f = (uint64_t)c_front * step;
      const uint64_t kw = lo_w + tid, kf = lo_f + tid;
      const bool in_w = tid < step && kw < m, in_f = tid < step && kf < m;
      const uint64_t e1w = in_w ? enc_pre[start + kw + 1] : 0;
      const uint64_t e0f = in_f ? enc_pre[start + kf] : 0;
      const bool whole = in_w && (e1w - base_e + D) + 8 <= room0;
      const bool front = in_f && tail0 + st_i_of(kf, e0f) + 8 < cap;
      nrec = lo_w + txm_count(whole, s_cnt);
      nfront = lo_f + txm_count(front, s_cnt);
    }
    // ---- the short record behind the whole ones, the record that may cross the ring end: uniform loads, one round trip
    {
      const uint64_t wi = nfront ? nfront - 1 : 0;  // the last record that starts in front of the ring end
      const bool has_short = nrec < m;
      const uint64_t len_s = has_short ? sl[nrec].len : 0, e_s = m ? enc_pre[start + (nrec < m ? nrec : m)] : 0;
      const uint64_t len_w = (direct && nfront) ? sl[wi].len : 0, e_w = (direct && nfront) ? enc_pre[start + wi] : 0;
      st_short = m ? st_i_of(nrec, e_s) : 0;  // st(nrec): where the short record starts, or st(m) = the end of the Send
      if (has_short) {
        // (pay = min(len, W(S - st), W(free0 - st)): it did not fit whole)
        uint64_t p = nrec == 0 ? sat_sub(len_s, byte_idx) : len_s;
        const uint64_t a = writable_of(sat_sub(S, st_short)), b = writable_of(sat_sub(free0, st_short));
        if (a < p) p = a;
        if (b < p) p = b;
        short_pay = p;
      }
      const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
      if (direct && nfront && wi < nrec_total) {
        const uint64_t p = wi == nrec ? short_pay : (wi == 0 ? sat_sub(len_w, byte_idx) : len_w);
        const uint64_t pay_off = (tail0 + st_i_of(wi, e_w) + 8) & mask;
        if (pay_off + p > cap) {
          const uint64_t l1 = cap - pay_off;
          wrap_rec = (uint32_t)wi;
          wrap_extra = (uint32_t)(((l1 + TB - 1) >> ts) + ((p - l1 + TB - 1) >> ts) - ((p + TB - 1) >> ts));
        }
      }
    }
  }
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  const uint64_t t_priced = __builtin_amdgcn_s_memtime();

  // ---- my record: its segment and tile-prefix entry (AppendHeader / AppendFooter ride on the segment), as txf_body
  const uint64_t i = (uint64_t)wg * TXM_THREADS + tid;
  if (ok && i < nrec_total) {
    const grdma_sge g = sl[i];
    const uint64_t e0 = enc_pre[start + i];
    const uint32_t tp0 = tile_pre[start + i];
    const uint64_t st_i = i == 0 ? 0 : e0 - base_e + D;
    const uint64_t p = i == nrec ? short_pay : (i == 0 ? sat_sub(g.len, byte_idx) : g.len);
    const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (p << GRDMA_SEG_TAG_LEN_SHIFT);
    if (op.sizes_out != nullptr) op.sizes_out->n[i] = (uint32_t)p;  // (for the drain of the same round: grdma_size_hint)
    const uint8_t* src = g.ptr + (i == 0 ? byte_idx : 0);
    const uint32_t t0 = i == 0 ? 0u : (uint32_t)(tp0 - base_t + Dt);
    if (!direct) {
      plan->segs[i] = {(uint64_t)(staging + st_i + 8), (uint64_t)src, p, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
      plan->tile_prefix[i] = t0;
    } else {
      uint8_t* const ring = c->peer_ring;
      const uint64_t pay_off = (tail0 + st_i + 8) & mask;
      const uint64_t seg = i + ((uint32_t)i > wrap_rec ? 1 : 0);
      const uint32_t tx0 = t0 + ((uint32_t)i > wrap_rec ? wrap_extra : 0u);
      if ((uint32_t)i == wrap_rec) {
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

  // ---- arrival: the last workgroup writes totals, wire plan, state, result -- or hands the Send to the general planner
  __syncthreads();
  if (tid == 0) {
    const uint32_t prev = __hip_atomic_fetch_add(&plan->mw_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = prev == nwg - 1;
    s_last = last ? 1u : 0u;
    if (last) __hip_atomic_store(&plan->mw_arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return 0;
  if (!ok) {  // (uniform, and the same in every workgroup)
    if (tid == 0) {
      atomicAdd(&g_tx_fast_sends[1], 1ull);
      op.result->dbg[11]++;  // (dbg[10] / dbg[11]: Sends of this result block priced from the index / declined)
      if (op.sizes_out != nullptr) op.sizes_out->count = 0;  // (the general planner leaves no size table)
    }
    return 2;
  }
  if (tid == 0) {
    const uint64_t offered = op.use_cursor == 1 ? remaining : len_pre[n];
    const uint64_t o_written = c->total_written, o_records = c->tx_records, o_rounds = c->tx_rounds;
    const uint64_t o_seq = op.result->seq, lp_start = m ? len_pre[start] : 0;
    // totals behind the whole records
    uint64_t sent = short_pay, ntiles = (short_pay + TB - 1) >> ts;
    if (nrec) {
      sent += len_pre[start + nrec] - lp_start - byte_idx;
      ntiles += tile_pre[start + nrec] - base_t + Dt;
    }
    const uint64_t st_last = st_short;  // st(nrec)
    const uint64_t staged = (nrec || short_pay) ? st_last + (short_pay > 0 ? enc_size(short_pay) : 0) : 0;
    const uint64_t nsegs = nrec_total + (wrap_rec != 0xFFFFFFFFu ? 1 : 0);
    ntiles += wrap_extra;
    plan->nsegs = (uint32_t)nsegs;
    plan->ntiles = (uint32_t)ntiles;
    plan->tile_bytes = (uint32_t)TB;
    plan->tile_prefix[nsegs] = (uint32_t)ntiles;
    plan->bytes = sent;
    plan->tag_base = direct ? (uint64_t)c->peer_ring : (uint64_t)staging;
    plan->tag_mask = direct ? mask : ~0ull;
    const uint64_t new_tail = (tail0 + staged) & mask;
    // the <= 2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330
    const uint64_t seg1 = staged < cap - tail0 ? staged : cap - tail0;
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
      if (staged > 0 && !direct) {
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
    const uint64_t idx = start + nrec;
    uint64_t bidx = 0;
    if (short_pay > 0) bidx = (nrec == 0 ? byte_idx : 0) + short_pay;
    else if (nrec == 0) bidx = byte_idx;
    c->remote_tail = new_tail;
    c->partial_write = sent < offered ? 1 : 0;  // pair.cc:709
    c->total_written = o_written + sent;
    c->tx_records = o_records + nrec_total;
    c->tx_last_records = (uint32_t)nrec_total;
    if (nrec_total) c->tx_rounds = o_rounds + 1;
    c->tx_slice_idx = idx;
    c->tx_byte_idx = bidx;
    c->tx_remaining = offered - sent;
    r->sent = sent;
    r->records = nrec_total;
    r->staged = staged;
    r->partial = sent < offered ? 1 : 0;
    r->new_remote_tail = new_tail;
    if (op.tail_out != nullptr) *op.tail_out = new_tail;
    if (op.sizes_out != nullptr) {
      op.sizes_out->start_off = tail0;
      op.sizes_out->count = (uint32_t)nrec_total;
    }
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = (idx >= op.nslices) ? 1 : 0;
    r->dbg[0] = t_begin;
    r->dbg[1] = t_priced;
    r->dbg[6] = __builtin_amdgcn_s_memtime();
    r->dbg[7] = m;
    r->dbg[9] = 0xFA57;  // this Send was priced from the index
    r->dbg[10]++;
    atomicAdd(&g_tx_fast_sends[0], 1ull);
    const uint64_t nxt = op.seq_next ? op.seq_next : o_seq + 1;
    __hip_atomic_store(&r->seq, nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return 1;
}

}  // namespace
#endif  // GRDMA_TX_MULTI_H
