// txf_body: the send plan of a streaming job's Send priced from the INDEX of the slice buffer (k_tx_index,
// grdma_tx_fast.hip has the story); included by grdma_kernels.hip, where k_tx_plan_job runs it in front of the
// general planner.
#ifndef GRDMA_TX_FAST_H
#define GRDMA_TX_FAST_H
#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"

namespace {

#define TXB_THREADS 1024
#define TXB_PER 4
#define TXB_KERNEL_ATTR
#define TXB_WAVES (TXB_THREADS / 64)
static_assert(TXB_THREADS * TXB_PER >= GRDMA_TX_MAX_RECORDS, "one pass covers a Send");

// diagnostics: Sends taken by txf_body / left to the general planner
__device__ unsigned long long g_tx_fast_sends[2] = {0, 0};

// ---------------------------------------------------------------------------------------------------------
// txf_body: one Send over the index.  1024 threads, four records per thread, striped (k_tx_plan_job launches this shape;
// when the body declines, waves 4-15 leave and waves 0-3 run tx_plan_body).  Returns true when it
// planned the Send, false -- nothing written -- when the general planner has to.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool txf_body(const grdma_tx_op& op_in, const grdma_txf_ctl* ctl) {
  const grdma_tx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint32_t s_cnt[TXB_WAVES];
  __shared__ uint64_t s_rhead;

  // ---- state; what this kernel takes (nothing is stored before the decision)
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap, tail0 = c->remote_tail;
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  const uint64_t n = ctl->n;
  const uint64_t* const enc_pre = ctl->enc_pre;
  const uint64_t* const len_pre = ctl->len_pre;
  const uint32_t* const tile_pre = ctl->tile_pre;
  const bool direct = c->wire_direct != 0;  // records straight into the peer ring (no staging copy, no wire plan)
  const bool ok = ctl->valid != 0 && ctl->slices == op.slices && n == op.nslices && op.use_cursor != 0 && !op.inline_copy &&
                  connected && (direct || op.wire_plan != nullptr) && cap <= (1ull << 31) &&
                  ctl->tile_shift == GRDMA_PLAN_TILE_SHIFT(cap);
  if (tid == 0)  // get_remote_head(), pair.h:229-233 -- ONE read for the whole Send (the peer's scatter may post meanwhile)
    s_rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (!ok) {  // (uniform)
    if (tid == 0) {
      atomicAdd(&g_tx_fast_sends[1], 1ull);
      op.result->dbg[11]++;  // (dbg[10] / dbg[11]: Sends of this result block priced from the index / declined)
    }
    return false;
  }
  const uint64_t rhead = s_rhead;
  uint64_t start = op.use_cursor == 1 ? c->tx_slice_idx : 0;
  const uint64_t byte_idx = op.use_cursor == 1 ? c->tx_byte_idx : 0;
  if (start > n) start = n;
  const uint64_t avail = n - start;
  const uint64_t offered = op.use_cursor == 1 ? c->tx_remaining : len_pre[n];
  uint64_t m = avail;
  if (m > c->max_sge) m = c->max_sge;
  if (m > GRDMA_TX_MAX_RECORDS - 1) m = GRDMA_TX_MAX_RECORDS - 1;
  const grdma_sge* const sl = op.slices + start;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;
  const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
  const uint64_t free0 = cap - occupied0;
  const uint64_t room0 = S < free0 ? S : free0;
  // the first slice may have been sent in part: its record is shorter than the index says
  uint64_t base_e = 0, base_t = 0, D = 0, Dt = 0;
  if (m) {
    const uint64_t len0 = sl[0].len, l0 = sat_sub(len0, byte_idx);
    base_e = enc_pre[start];
    base_t = tile_pre[start];
    D = enc_size(l0) - (enc_pre[start + 1] - base_e);              // (<= 0 as a signed number; modular arithmetic)
    Dt = ((l0 + TB - 1) >> ts) - (tile_pre[start + 1] - base_t);
  }
  // st(k): staging offset of record k of this Send (k >= 1), from the index
  // ---- my records: striped, all loads in flight together
  uint64_t r_ptr[TXB_PER], r_len[TXB_PER], r_e0[TXB_PER], r_e1[TXB_PER];
  uint32_t r_t0[TXB_PER], r_t1[TXB_PER];
  uint64_t r_l1[TXB_PER];  // len_pre / tile_pre BEHIND my record: what the totals need from the last whole record's owner
  // (what thread 0 adds to at the very end: fetched now, in the same round trip)
  const uint64_t o_written = c->total_written, o_records = c->tx_records, o_rounds = c->tx_rounds;
  const uint64_t o_seq = op.result->seq, lp_start = m ? len_pre[start] : 0;
#pragma unroll
  for (int r = 0; r < TXB_PER; r++) {
    const uint64_t i = tid + (uint64_t)r * TXB_THREADS;
    const uint64_t k = i < m ? i : (m ? m - 1 : 0);  // clamped, unconditional
    const grdma_sge g = m ? sl[k] : grdma_sge{nullptr, 0};
    r_ptr[r] = reinterpret_cast<uint64_t>(g.ptr);
    r_len[r] = g.len;
    r_e0[r] = m ? enc_pre[start + k] : 0;
    r_e1[r] = m ? enc_pre[start + k + 1] : 0;
    r_t0[r] = m ? tile_pre[start + k] : 0;
    r_t1[r] = m ? tile_pre[start + k + 1] : 0;
    r_l1[r] = m ? len_pre[start + k + 1] : 0;
  }
  // ---- whole records: a count
  uint32_t my_whole = 0;
  uint64_t st_i[TXB_PER], st_n[TXB_PER];
#pragma unroll
  for (int r = 0; r < TXB_PER; r++) {
    const uint64_t i = tid + (uint64_t)r * TXB_THREADS;
    st_i[r] = i == 0 ? 0 : r_e0[r] - base_e + D;
    st_n[r] = r_e1[r] - base_e + D;
    if (i < m && st_n[r] + 8 <= room0) my_whole++;
  }
  {
    uint32_t w = my_whole;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d, 64);
    if (lane == 0) s_cnt[tid >> 6] = w;
  }
  __syncthreads();
  uint64_t nrec = 0;
#pragma unroll
  for (int w = 0; w < TXB_WAVES; w++) nrec += s_cnt[w];
  // the short record behind them (pay = min(len, W(S - st), W(free0 - st)) = W(room0 - st): it did not fit whole)
  __shared__ uint64_t s_short[4];  // {short_pay, st of record nrec, len_pre and tile_pre behind the last whole record}
  if (tid == 0) s_short[0] = s_short[1] = s_short[2] = s_short[3] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < TXB_PER; r++) {
    const uint64_t i = tid + (uint64_t)r * TXB_THREADS;
    if (i == nrec && i < m) {
      uint64_t p = i == 0 ? sat_sub(r_len[r], byte_idx) : r_len[r];
      const uint64_t a = writable_of(sat_sub(S, st_i[r])), b = writable_of(sat_sub(free0, st_i[r]));
      if (a < p) p = a;
      if (b < p) p = b;
      s_short[0] = p;
      s_short[1] = st_i[r];
    }
    if (nrec == m && m != 0 && i == m - 1) s_short[1] = st_n[r];  // every record went out whole: st(m)
    if (nrec != 0 && i == nrec - 1) {
      s_short[2] = r_l1[r];
      s_short[3] = r_t1[r];
    }
  }
  __syncthreads();
  const uint64_t short_pay = s_short[0];
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  uint8_t* const staging = op.staging_alt ? op.staging_alt : c->staging;
  const uint64_t t_priced = __builtin_amdgcn_s_memtime();

  // ---- direct wire: the one record whose payload crosses the ring end becomes two segments (a Send stages at most
  //      ring / 2 bytes: one wrap at most); the records behind it sit one segment further, their tiles `extra` further
  __shared__ uint32_t s_wrap[2];  // {record that wraps (0xFFFFFFFF = none), extra tiles of the split}
  if (tid == 0) {
    s_wrap[0] = 0xFFFFFFFFu;
    s_wrap[1] = 0;
  }
  __syncthreads();
  if (direct) {
#pragma unroll
    for (int r = 0; r < TXB_PER; r++) {
      const uint64_t i = tid + (uint64_t)r * TXB_THREADS;
      if (i >= nrec_total) continue;
      const uint64_t p = i == nrec ? short_pay : (i == 0 ? sat_sub(r_len[r], byte_idx) : r_len[r]);
      const uint64_t pay_off = (tail0 + st_i[r] + 8) & mask;
      if (pay_off + p > cap) {
        const uint64_t l1 = cap - pay_off;
        s_wrap[0] = (uint32_t)i;
        s_wrap[1] = (uint32_t)(((l1 + TB - 1) >> ts) + ((p - l1 + TB - 1) >> ts) - ((p + TB - 1) >> ts));
      }
    }
    __syncthreads();
  }
  const uint32_t wrap_rec = s_wrap[0], wrap_extra = s_wrap[1];

  // ---- segments and tile prefix, one record per thread-step (AppendHeader / AppendFooter ride on the segment)
#pragma unroll
  for (int r = 0; r < TXB_PER; r++) {
    const uint64_t i = tid + (uint64_t)r * TXB_THREADS;
    if (i >= nrec_total) continue;
    const uint64_t p = i == nrec ? short_pay : (i == 0 ? sat_sub(r_len[r], byte_idx) : r_len[r]);
    const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (p << GRDMA_SEG_TAG_LEN_SHIFT);
    const uint8_t* src = reinterpret_cast<const uint8_t*>(r_ptr[r]) + (i == 0 ? byte_idx : 0);
    const uint32_t t0 = i == 0 ? 0u : (uint32_t)(r_t0[r] - base_t + Dt);
    if (!direct) {
      plan->segs[i] = {(uint64_t)(staging + st_i[r] + 8), (uint64_t)src, p, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
      plan->tile_prefix[i] = t0;
      continue;
    }
    uint8_t* const ring = c->peer_ring;
    const uint64_t pay_off = (tail0 + st_i[r] + 8) & mask;
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

  if (tid == 0) {
    // totals behind the whole records (their owner fetched the index entries with its record)
    uint64_t sent = short_pay, ntiles = (short_pay + TB - 1) >> ts;
    if (nrec) {
      sent += s_short[2] - lp_start - byte_idx;
      ntiles += s_short[3] - base_t + Dt;
    }
    const uint64_t st_last = s_short[1];  // st(nrec)
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
    // (relaxed: a streaming job's consumers are later kernels of the graph; a system-scope release would write the
    // XCD's L2 back -- the plan just laid out -- before the kernel may end)
    __hip_atomic_store(&r->seq, nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return true;
}

}  // namespace
#endif  // GRDMA_TX_FAST_H
