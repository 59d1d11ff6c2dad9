// rxh_body: the drain of a streaming job's round whose record sizes are NOT periodic, laid out by several small
// workgroups from the sizes its own Send computed (included by grdma_rx_plan.hip; k_plan_pair_mw runs it when the
// connection has no period and the round carries a size table).
//
// A receiver learns where a record ends from the record in front of it: RingBufferPollable::Read walks header by header
// (ring_buffer.cc:122-191), and a walk is a chain of dependent loads -- 4095 records are a millisecond on a GPU.  The
// steady-state bodies (grdma_rx_fast.h, grdma_rx_multi.h) get around the walk by PREDICTING every record from the
// period of the record sizes and verifying all predictions in one round of loads; traffic without a period (the
// reference's own test distribution: message sizes uniform in [1, 4 MiB), examples/cpp/test/common.h:4-31) went to the
// general planner (64 us per round, value_mixed_sizes 0.34 x value).  A streaming job has both ends of the connection
// on one device, and already hands the drain of a round the ring offset its Send of the same round ended at
// (grdma_rx_op::limit_ptr).  Here it also hands over the payload sizes of the records that Send produced
// (grdma_tx_op::sizes_out -> grdma_rx_op::sizes_in: what the sender's planner knows anyway), and the drain uses them
// exactly as the steady-state bodies use the pattern: as a PREDICTION.  Every record's header and footer in the ring is
// checked against it -- by the thread that owns the record, one round of loads -- before anything is consumed; a
// table that does not start at the reader's head, does not end at the limit or disagrees with a single header makes
// the body decline without having written a byte, and the general planner walks the chain.  What is delivered is
// what the ring holds, never what the table says.  A pair whose peer is another process has no such table: this
// body is for the device-resident job (bench.py's legs), the endpoint's drains keep walking.
//
// Shape: workgroup b of G owns records [256 b, 256 b + 256), one per thread, one wave per SIMD (grdma_rx_multi.h has
// the reasons).  What the records in front of a workgroup's own took -- slices, segments, tiles, arena bytes -- has no
// closed form here: every workgroup lays out the records in front of its own for itself (at most sixteen per thread
// in the last workgroup; the sizes are in LDS, the read state in front of a record comes from the <= 192 records
// before it), so that again nothing is exchanged between the workgroups but the arrival word.
#ifndef GRDMA_RX_HINT_H
#define GRDMA_RX_HINT_H
#include <type_traits>
#include "grdma_rx_multi.h"

namespace {

#define RXH_HALF (RXM_G * RXM_THREADS)  // records of one Send's worth: the half of the table every drain loads at once
#define RXH_MAX (2 * RXH_HALF)          // records per drain (round 6: a round of two Sends -- 8190 records of mixed sizes)
static_assert(RXH_MAX <= GRDMA_HINT_MAX_RECORDS, "the size table of a round covers a drain");

struct rx_lds_hint {
  uint32_t n[RXH_MAX];          // payload sizes
  uint32_t x[RXH_MAX + 1];      // exclusive prefix of the encoded sizes
};

// (round 6) The table's header and this thread's sixteen sizes, REQUESTED when the launch begins -- beside the connection
// state the periodic body (rxm_body) asks for, before that body has found out that the connection has no period: the
// table's address is in the op, its words do not depend on anything the drain reads first.  What used to be two more
// dependent round trips behind rxm_body's (the header, then the sizes clamped by its count: 4.7 us of this body's 14,
// profiles/r06_plan_phases.txt) is in flight with the first.  The sizes are read unclamped -- the table has
// GRDMA_TX_MAX_RECORDS entries whatever its count says -- and the entries behind the count are zeroed below as before.
struct rxh_pre {
  uint32_t nv[RXH_HALF / RXM_THREADS];
  uint32_t V;
  uint64_t h_start;
};
__device__ __forceinline__ void rxh_preload(const grdma_rx_op& op, rxh_pre& p) {
  constexpr uint32_t PER = RXH_HALF / RXM_THREADS;
  const grdma_size_hint* const hint = op.sizes_in;
  p.V = 0;
  p.h_start = ~0ull;
#pragma unroll
  for (uint32_t r = 0; r < PER; r++) p.nv[r] = 0;
  if (hint != nullptr) {  // (uniform)
    p.h_start = hint->start_off;
    p.V = hint->count;
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) p.nv[r] = hint->n[threadIdx.x * PER + r];
  }
}

template <bool WT = false, bool EWT = WT, class RingWait = ring_ready_now, class Publish = credit_unpublished>  // (see rxm_body)
__device__ __forceinline__ int rxh_body(const grdma_rx_op& op_in, const uint32_t wg, const uint32_t nwg, const rxh_pre* pre = nullptr,
                                        RingWait* ring_wait = nullptr, Publish* publish = nullptr) {
  static_assert(sizeof(rx_lds_hint) <= sizeof(rx_lds) && !WT, "the tables fit the planners' LDS (not the small one of a write-through body)");
  rx_lds_hint& H = *reinterpret_cast<rx_lds_hint*>(rx_tables<WT>());
  const grdma_rx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;
  __shared__ uint32_t s_w[4][RXM_WAVES];
  __shared__ uint32_t s_bad, s_first, s_F;

  // ---- 0. state, the table's header, preconditions
  uint8_t* const ring = c->ring;
  const uint64_t cap64 = c->cap;
  const uint64_t head64 = c->head, mh0 = c->moving_head, remain0 = c->remain, leftover0 = c->leftover_cap;
  const uint64_t irs0 = c->internal_read_size;
  const uint64_t hc = c->rx_hist_count;
  const uint32_t status = c->status;
  const uint64_t lim = op.limit_ptr ? __hip_atomic_load(op.limit_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  const uint64_t slice_idx0 = op.append == 1 ? c->rx_slice_idx : 0;
  const uint64_t a_off0 = op.append == 1 ? c->rx_arena_off : 0;
  const grdma_size_hint* const hint = op.sizes_in;
  const uint64_t h_start = pre ? pre->h_start : (hint ? hint->start_off : ~0ull);
  const uint32_t V = pre ? pre->V : (hint ? hint->count : 0);

  // A round of more than one Send's worth of records (round 6: the table holds two) is what a PERIODIC stream's rounds look
  // like too -- bench.py's headline: 8190 records, period 390 -- and those belong to rxm_body, which lays them out in a
  // third of the time; but a connection's period is found by the general planner's search over the record-size history,
  // and a drain this body takes never gets there.  So: no period yet and a search due (the planner's own back-off,
  // rx_period_retry_at) -> this body declines a big round, the general planner walks it and searches.  A stream without
  // a period costs a handful of walked drains, ever rarer; one with a period is rxm_body's from its second drain on.
  const bool search_due = V > RXH_HALF && c->rx_period == 0 && hc >= c->rx_period_retry_at;
  bool ok = !search_due && status == GRDMA_PAIR_CONNECTED && op.raw_cap == 0 && op.append != 0 && !op.inline_apply &&
            op.limit_ptr != nullptr && remain0 == 0 && leftover0 <= RXF_MINRD && cap64 <= (1ull << 31) &&
            a_off0 < (1ull << 31) && hint != nullptr && h_start == head64 && V != 0 && V <= nwg * RXM_CHUNK && V <= RXH_MAX;
  uint64_t max_slices = GRDMA_MAX_SLICES;
  {
    const uint64_t room = op.slices_cap > slice_idx0 ? op.slices_cap - slice_idx0 : 0;
    if (room < max_slices) max_slices = room;
    if (op.max_reads < max_slices) max_slices = op.max_reads;
  }
  const uint32_t cap = (uint32_t)cap64, mask = cap - 1u, head = (uint32_t)head64;
  const uint32_t Lr = ((uint32_t)lim - head) & mask;  // ring bytes between my head and the sender's tail
  const bool idle = Lr == 0;
  if (tid == 0) {
    s_bad = 0;
    s_first = RXM_NONE;
    s_F = RXM_NONE;
  }
  __syncthreads();
  uint32_t reason = (!ok || idle) ? 1u : 0u;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap64);

  // ---- 1. the sizes into LDS, their encoded prefix: thread t holds records 16 t .. 16 t + 15 of the table's first half
  //      (one Send's worth: what every drain loads, requested when the launch began) and, for a round of more than 4096
  //      records, 4096 + 16 t .. of the second (a second round of loads and a second scan, for those rounds only)
  constexpr uint32_t PER = RXH_HALF / RXM_THREADS;
  if (!reason) {
    bool bad = false;
    uint32_t base_x = 0;
    // (the half as a compile-time constant: the preloaded sizes stay in registers -- indexed through a loop variable
    //  they went to scratch memory, and a planner kernel with a scratch segment costs every launch 1-2 us)
    auto load_half = [&](auto HALF) {
      constexpr uint32_t half = decltype(HALF)::value;
      uint32_t nv[PER];
      const uint32_t i0 = half * RXH_HALF + tid * PER;
#pragma unroll
      for (uint32_t r = 0; r < PER; r++)
        nv[r] = (half == 0 && pre) ? pre->nv[r] : hint->n[i0 + r < V ? i0 + r : 0];  // (clamped: all loads in flight)
      uint32_t sum = 0;
#pragma unroll
      for (uint32_t r = 0; r < PER; r++) {
        if (i0 + r >= V) nv[r] = 0;
        else bad |= nv[r] == 0 || nv[r] > cap - (uint32_t)GRDMA_RESERVED;
        sum += i0 + r < V ? 16u + ((nv[r] + 7u) & ~7u) : 0u;
      }
      uint32_t px, d1, d2, d3, tot[4];
      rxm_scan4(sum, 0, 0, 0, s_w, &px, &d1, &d2, &d3, tot);
      px += base_x;
#pragma unroll
      for (uint32_t r = 0; r < PER; r++) {
        H.n[i0 + r] = nv[r];
        H.x[i0 + r] = px;
        px += i0 + r < V ? 16u + ((nv[r] + 7u) & ~7u) : 0u;
      }
      if (tid == RXM_THREADS - 1) H.x[(half + 1) * RXH_HALF] = px;  // (the end of this half: x[4096], x[8192])
      base_x += tot[0];
    };
    load_half(std::integral_constant<uint32_t, 0>{});
    if (V > RXH_HALF) load_half(std::integral_constant<uint32_t, 1>{});  // (uniform)
    if (bad) s_bad = 1;
    __syncthreads();
    if (s_bad || H.x[V] != Lr) reason = 2;  // the table does not end where the sender's tail is
  }
  const uint64_t t_pattern = __builtin_amdgcn_s_memtime();
  if (ring_wait != nullptr) (*ring_wait)();

  // ---- 2. one round trip: header and footer of my record
  const uint32_t i_mine = wg * RXM_CHUNK + tid;
  const bool have = !reason && i_mine < V;
  uint32_t xe = 0, n_mine = 0;
  if (!reason) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t ring_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ring);
    const uint32_t ring_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ring >> 32));
    const uint64_t ring_u = ((uint64_t)ring_hi << 32) | (uint64_t)ring_lo;
    const uint32_t cap_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ring_u, 0, cap_u, 0x00020000);
    uint32_t ee = 0;
    if (have) {
      xe = H.x[i_mine];
      n_mine = H.n[i_mine];
      ee = 16u + ((n_mine + 7u) & ~7u);
    }
    const uint32_t o_h = have ? ((head + xe) & mask & ~7u) : 0u;
    const uint32_t o_f = have ? ((head + xe + ee - 8u) & mask & ~7u) : 0u;
    const u32x2 hw = __builtin_amdgcn_raw_buffer_load_b64(rs, o_h, 0, 16 /* sc1 */);
    const u32x2 fw = __builtin_amdgcn_raw_buffer_load_b64(rs, o_f, 0, 16);
    if (have) {
      const uint64_t h = ((uint64_t)hw.y << 32) | hw.x, f = ((uint64_t)fw.y << 32) | fw.x;
      if (h != (uint64_t)n_mine || f != GRDMA_FOOTER) s_bad = 1;  // the ring does not hold what the table says
    }
    // the first record that leaves no read open behind it (F), among the first RXM_PFX records
    if (tid < RXM_PFX && tid < V && H.n[tid] >= RXM_RESET) atomicMin(&s_F, tid);
    __syncthreads();
    if (s_bad) reason = 3;
  }
  const uint64_t t_probe = __builtin_amdgcn_s_memtime();

  // ---- 3. read state in front of a record: back to the nearest record that leaves no read open, or to the drain's start
  const uint32_t s0 = (uint32_t)leftover0;
  const bool odd_open = s0 != 0 && s0 != RXF_MINRD;
  auto state_of = [&](uint32_t i, bool* too_far) -> uint32_t {
    uint32_t j = i, steps = 0;
    while (j > 0 && H.n[j - 1] < RXM_RESET && steps <= RXF_LOOKBACK) {
      j--;
      steps++;
    }
    if (j > 0 && H.n[j - 1] < RXM_RESET) *too_far = true;  // a long run of small records: not this body's case
    uint32_t s = j == 0 ? s0 : 0;
    for (; j < i; j++) s = rxf_space_after(H.n[j], s);
    return s;
  };
  // the first record that completes a slice closes the read that was open when the drain began: it lies at or in
  // front of the first record that leaves no read open (F) -- among the first RXM_PFX records, or the body declines
  uint32_t first_done = RXM_NONE;
  if (!reason) {
    const uint32_t F = s_F;
    if (F == RXM_NONE && V > RXM_PFX) reason = 4;
    else {
      const uint32_t NF = F == RXM_NONE ? V : F + 1;
      if (tid < NF) {
        bool far = false;
        const uint32_t s = state_of(tid, &far);
        if (rxf_replay(H.n[tid], s).sl_cnt != 0) atomicMin(&s_first, tid);
      }
      __syncthreads();
      first_done = s_first;
    }
  }
  auto layout_of = [&](uint32_t i, bool* too_far) -> rxf_layout {
    const uint32_t s_in = state_of(i, too_far);
    const bool in_first = odd_open && i <= first_done;
    return rxf_lay(H.n[i], s_in, (head + H.x[i] + 8u) & mask, cap, in_first ? s0 : RXF_MINRD, odd_open && i == first_done, ts);
  };

  // ---- 4. what the records in front of my workgroup's took (every workgroup for itself), then my own record's place
  uint32_t b_pk = 0, b_tl = 0, b_by = 0;     // before my workgroup
  uint32_t x_pk = 0, x_tl = 0, x_by = 0;     // before my record, inside my workgroup
  uint32_t tot_pk = 0, tot_tl = 0, tot_by = 0, tot_n = 0, s_end = 0;
  rxf_layout Lm = {};
  if (!reason) {
    const uint32_t base = wg * RXM_CHUNK;
    uint32_t a_pk = 0, a_tl = 0, a_by = 0, a_n = 0;   // records in front of the workgroup, strided over the threads
    uint32_t z_pk = 0, z_tl = 0, z_by = 0, z_n = 0;   // records behind the workgroup (the drain's totals)
    bool far = false;
    for (uint32_t i = tid; i < V; i += RXM_THREADS) {
      if (i >= base && i < base + RXM_CHUNK) continue;
      const rxf_layout L = layout_of(i, &far);
      const uint32_t pk = L.sl_cnt | (L.nsg << 16);
      if (i < base) { a_pk += pk; a_tl += L.ntl; a_by += L.bytes; a_n += H.n[i]; }
      else { z_pk += pk; z_tl += L.ntl; z_by += L.bytes; z_n += H.n[i]; }
    }
    uint32_t m_pk = 0, m_tl = 0, m_by = 0;
    if (have) {
      Lm = layout_of(i_mine, &far);
      m_pk = Lm.sl_cnt | (Lm.nsg << 16);
      m_tl = Lm.ntl;
      m_by = Lm.bytes;
    }
    if (far) s_bad = 1;
    uint32_t d0, d1, d2, d3, ta[4], tz[4], tm[4];
    rxm_scan4(a_pk, a_tl, a_by, a_n, s_w, &d0, &d1, &d2, &d3, ta);
    rxm_scan4(z_pk, z_tl, z_by, z_n, s_w, &d0, &d1, &d2, &d3, tz);
    rxm_scan4(m_pk, m_tl, m_by, have ? n_mine : 0u, s_w, &x_pk, &x_tl, &x_by, &d3, tm);
    b_pk = ta[0]; b_tl = ta[1]; b_by = ta[2];
    tot_pk = ta[0] + tm[0] + tz[0];
    tot_tl = ta[1] + tm[1] + tz[1];
    tot_by = ta[2] + tm[2] + tz[2];
    tot_n = ta[3] + tm[3] + tz[3];
    if (s_bad) reason = 4;
    if (!reason) {
      bool far2 = false;
      s_end = rxf_space_after(H.n[V - 1], state_of(V - 1, &far2));
    }
  }
  const uint64_t t_state = __builtin_amdgcn_s_memtime();
  const uint32_t tot_sl = tot_pk & 0xFFFFu, tot_sg = tot_pk >> 16;
  // the would-block at the end (rdma_do_read, rdma_bp_posix.cc:195-277), as rxf_body
  const uint32_t cap_open_end = (odd_open && first_done == RXM_NONE) ? s0 : RXF_MINRD;
  const uint32_t short_len = s_end ? cap_open_end - s_end : 0;
  const uint32_t nsl_final = tot_sl + (short_len ? 1u : 0u);
  const uint32_t leftover_final = s_end ? s_end : RXF_MINRD;
  const uint64_t a_end = a_off0 + tot_by + rxf_al16(short_len);
  if (!reason && !(nsl_final + 2 <= max_slices && tot_sg + 8 <= GRDMA_MAX_SEGS && a_end + leftover_final + 16 <= op.arena_cap &&
                   a_end < (1ull << 32)))
    reason = 5;
  const uint64_t t_scan = __builtin_amdgcn_s_memtime();
  // ---- my verdict is final: counted in (grdma_rx_multi.h: drain_count_in), behind the loads of everything the commit overwrites
  const bool committer = wg == nwg - 1;
  GRDMA_WAIT_LOADS();
  __syncthreads();
  if (tid == 0) drain_count_in(plan, reason);

  // ---- 5. my record: segments, tile prefix, slices (entries beyond any committed count if the drain is declined)
  grdma_slice_out* const out_slices = op.slices + slice_idx0;
  if (!reason && have) {
    const uint32_t pk = b_pk + x_pk;
    uint32_t sl = pk & 0xFFFFu, sg = pk >> 16, tl = b_tl + x_tl;
    const uint64_t A = a_off0 + b_by + x_by;
    int last_piece = 0;
#pragma unroll
    for (int k = 1; k < 4; k++)
      if (Lm.len[k]) last_piece = k;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (Lm.len[k] == 0) continue;
      const uint64_t fl = GRDMA_SEG_ZERO_SRC | (k == 0 ? GRDMA_SEG_TAG_HDR : 0) | (k == last_piece ? GRDMA_SEG_TAG_FTR : 0);
      xwg_put_seg<EWT>(&plan->segs[sg], (uint64_t)op.arena + A + Lm.dst_rel[k], (uint64_t)(ring + Lm.off[k]), (uint64_t)Lm.len[k], fl);
      xwg_st32<EWT>(&plan->tile_prefix[sg], tl);
      sg++;
      tl += rxf_tiles(Lm.len[k], ts);
    }
    uint64_t sof = A;
    if (Lm.sl0) {
      xwg_st64<EWT>(&out_slices[sl].off, sof);
      xwg_st64<EWT>(&out_slices[sl].len, (uint64_t)Lm.sl0);
      sl++;
      sof += rxf_al16(Lm.sl0);
    }
    if (Lm.sl1) {
      xwg_st64<EWT>(&out_slices[sl].off, sof);
      xwg_st64<EWT>(&out_slices[sl].len, (uint64_t)Lm.sl1);
    }
  }
  // ---- (round 6) the counters the commit adds to and the credit of the drain, by thread 0 of EVERY workgroup while its
  //      entries are on their way to the memory side (see rxm_body)
  uint64_t o_total_read = 0, o_credit_msgs = 0, o_rx_records = 0, o_rx_rounds = 0, o_seq = 0;
  uint32_t o_h1 = 0;
  grdma_hostline* line = nullptr;
  uint64_t base = 0, credit = 0, credit_head = 0;
  bool crossed = false;
  if (tid == 0 && !reason && committer) {
    o_total_read = c->total_read; o_credit_msgs = c->credit_msgs;
    o_rx_records = c->rx_records; o_rx_rounds = c->rx_rounds;
    o_h1 = c->rx_h1;
    o_seq = res->seq;
    line = c->line;
    const uint64_t T = cap64 / 2, Ctot = Lr;
    uint64_t thr = T - irs0;
    while (Ctot >= thr) {
      uint32_t lo = 0, hi = V - 1;  // first record whose running consumption (after its last step) reaches thr
      const uint32_t t32 = (uint32_t)thr;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (H.x[mid + 1] >= t32) hi = mid; else lo = mid + 1;
      }
      const uint32_t n = H.n[lo];
      bool far = false;
      const rxf_rec rp = rxf_replay(n, state_of(lo, &far));
      const uint64_t C2 = H.x[lo + 1];
      const uint64_t e = 16u + ((n + 7u) & ~7u);
      const uint64_t cons2 = rp.c2 ? rp.c2 + (((n + 7u) & ~7u) - n + 8u) : 0;
      const uint64_t C1 = C2 - cons2;
      const uint64_t pos = (head64 + C2 - e) & (cap64 - 1);
      if (rp.c2 && C1 >= thr) {  // crossed after the first step of a two-step record
        credit_head = (pos + 8 + rp.c1) & (cap64 - 1);
        base = C1;
      } else {
        credit_head = (pos + e) & (cap64 - 1);
        base = C2;
      }
      credit++;
      crossed = true;
      thr = base + T;
    }
  }
  const uint64_t t_emit = __builtin_amdgcn_s_memtime();

  // ---- 6. the committer goes on once every verdict is in; the others leave behind their entries' acknowledgement
  if (!committer) {
    if (EWT) GRDMA_WAIT_VMEM();
    __syncthreads();
    if (tid == 0) drain_leave(plan);
    return 0;
  }
  const uint32_t s_any = drain_verdicts(plan, nwg);
  if (s_any) {  // (uniform)
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[reason ? reason : 3u], 1ull);
      atomicAdd(&g_rx_verdicts[s_any < nwg ? 0 : 1], 1ull);
      if (!idle) res->pad0++;
    }
    // (the general planner rewrites entries: everybody's are at the memory side first, mine included)
    if (EWT) GRDMA_WAIT_VMEM();
    if (tid == 0) drain_close(plan, nwg);
    __syncthreads();
    return 2;
  }
  // (the drain is committed as far as its Send is concerned: the promise leaves before the bookkeeping)
  if (tid == 0 && publish != nullptr) (*publish)(credit, credit_head);

  // history: the encoded sizes of this drain's records become the newest entries (what the period detector of the
  // general planner looks at when the traffic changes)
  constexpr int NH = GRDMA_RX_HIST / RXM_THREADS;
#pragma unroll
  for (int r = 0; r < NH; r++) {
    const uint32_t back = tid + r * RXM_THREADS;
    if (back < V) {
      const uint32_t i = V - 1 - back;
      c->rx_hist[(uint32_t)((hc + i) % GRDMA_RX_HIST)] = 16u + ((H.n[i] + 7u) & ~7u);
    }
  }
  // ---- 7. credit (pair.cc:276-284), state, result: thread 0
  if (tid == 0) {
    const uint64_t Ctot = Lr;
    const uint64_t irs = crossed ? Ctot - base : irs0 + Ctot;
    const uint64_t nh = (head64 + Lr) & (cap64 - 1);
    if (short_len) {
      out_slices[tot_sl].off = a_off0 + tot_by;
      out_slices[tot_sl].len = short_len;
    }
    xwg_st32<WT>(&plan->nsegs, tot_sg);
    xwg_st32<WT>(&plan->ntiles, tot_tl);
    xwg_st32<WT>(&plan->tile_bytes, 1u << ts);
    xwg_st32<WT>(&plan->tile_prefix[tot_sg], tot_tl);
    plan->bytes = tot_n;
    xwg_st64<WT>(&plan->tag_base, (uint64_t)ring);
    xwg_st64<WT>(&plan->tag_mask, cap64 - 1);
    xwg_st32<WT>(&plan->blocks_done, 0u);
    c->head = nh;
    c->moving_head = nh;
    c->remain = 0;
    if (line != nullptr) {
      line->rx_head = nh;
      line->rx_remain = 0;
    }
    c->internal_read_size = irs;
    c->leftover_cap = leftover_final;
    c->total_read = o_total_read + tot_n;
    c->credit_msgs = o_credit_msgs + credit;
    c->rx_records = o_rx_records + V;
    if (nsl_final) c->rx_rounds = o_rx_rounds + 1;
    c->rx_arena_off = a_end;
    c->rx_slice_idx = slice_idx0 + nsl_final;
    c->rx_hist_count = hc + V;
    c->rx_h1 = 16u + ((H.n[V - 1] + 7u) & ~7u);
    c->rx_h2 = V >= 2 ? 16u + ((H.n[V - 2] + 7u) & ~7u) : o_h1;
    if (credit) c->status_send.remote_head = credit_head;
    xwg_st64<WT>(&res->credit_head, credit_head);
    res->nslices = nsl_final;
    res->bytes = tot_n;
    res->consumed = Lr;
    res->records = V;
    res->would_block = 1;
    xwg_st64<WT>(&res->credit_sent, credit);
    res->head = nh;
    res->moving_head = nh;
    res->remain = 0;
    res->arena_used = a_end;
    res->zero_off[0] = res->zero_off[1] = res->zero_len[0] = res->zero_len[1] = 0;
    if (nh > mh0) {
      res->zero_off[0] = mh0;
      res->zero_len[0] = nh - mh0;
    } else {
      res->zero_off[0] = mh0;
      res->zero_len[0] = cap64 - mh0;
      res->zero_off[1] = 0;
      res->zero_len[1] = nh;
    }
    res->dbg[0] = t_begin;
    res->dbg[2] = t_pattern - t_begin;
    res->dbg[3] = t_probe - t_begin;
    res->dbg[4] = t_state - t_begin;
    res->dbg[5] = t_scan - t_begin;
    res->dbg[6] = t_emit - t_begin;
    res->dbg[7] = V;
    res->dbg[8] = 0;       // (no period)
    res->dbg[9] = 0xFA57;  // this stamp set comes from a steady-state body
    res->dbg[10] = nwg;
    res->dbg[11] = res->dbg[12] = res->dbg[13] = 0;
    res->pad1++;
    res->dbg[1] = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_rx_fast_drains[0], 1ull);
    __hip_atomic_store(&res->seq, op.seq_next ? op.seq_next : o_seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    drain_close(plan, nwg);
  }
  return 1;
}

}  // namespace
#endif  // GRDMA_RX_HINT_H
