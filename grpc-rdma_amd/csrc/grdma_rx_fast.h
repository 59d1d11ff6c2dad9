// rxf_body: the drain of a streaming connection in its steady state, in a straight line (the first thing
// k_rx_plan_job runs; included by grdma_rx_plan.hip).
//
// Same contract as k_rx_plan (grdma_rx_plan.hip): GetReadableSize / Read (ring_buffer.cc:67-191), Recv with
// its credit rule (pair.cc:264-286) and the endpoint-read loop (rdma_bp_posix.cc:180-326) replayed for every
// record between the reader's head and the tail its sender reported -- identical slices, segments, ring
// state, credit reports and history.  k_rx_plan gets there in tiers (wave tier until the read state is clean,
// bulk passes, wave tier for the tail, scalar reads to the would-block), each a handful of dependent memory
// round trips and barriers: ~50 us of fixed cost per drain whatever its size.  This kernel takes the one
// case a streaming job is in round after round and does it in a straight line:
//
//   1. The record sizes of the connection are periodic with the period P that k_rx_plan detected and keeps
//      validating (conn->rx_period): record i of this drain is predicted to have the size of the record P
//      earlier.  With the pattern's prefix sums in LDS the ring offset of EVERY record is one multiply-add
//      (x_i = (i / P) * S_P + pre[i % P]) -- no 4096-element scan -- and the drain's length is known from the
//      sender's arrival limit (grdma_rx_op::limit_ptr): V records must end exactly there.
//   2. ONE probe round: thread t loads footer(i) | header(i + 1) for its four records (16 bytes each, all in
//      flight together) and checks them against the prediction.  Anything unexpected -- a size that differs,
//      a footer that is missing, a limit the pattern does not hit, too many records -- and the body returns false
//      WITHOUT having written a byte; the caller then runs the general planner (rx_plan_body) in the same launch.
//   3. The endpoint-read state machine data-parallel over all records, from ANY starting state (the read left
//      open by the last drain's would-block: leftover_cap in [0, 256]) and through to the would-block at the
//      end (a read cut short is delivered as a short slice, its rest stays open), so no sequential tier runs
//      in front of or behind the parallel pass.
//   4. Segments, tile prefix, slices: one record per lane-step, coalesced stores.
//
// 1024 threads (16 wavefronts on one CU: a single wave per SIMD issues an instruction every 4-5 cycles and this body is
// a few thousand instructions per thread at 16 records per thread -- measured 2.2x SLOWER with 256 threads), four
// records per thread in contiguous runs (LDS index padded), <= 4096 records per drain.  k_rx_plan_job launches this
// shape; when the body declines, waves 4-15 leave and waves 0-3 run the general planner (a 256-thread body).
// Everything here is u32 arithmetic: the ring is at most 2 GiB on this path.
#ifndef GRDMA_RX_FAST_H
#define GRDMA_RX_FAST_H
#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"

namespace {

#define RXF_THREADS 1024
#define RXF_PER 4
#define RXF_KERNEL_ATTR
#define RXF_MAX (RXF_THREADS * RXF_PER)
#define RXF_WAVES (RXF_THREADS / 64)
#define RXF_MINRD 256u
#define RXF_PMAX 512u
#define RXF_LOOKBACK 192u

// diagnostics: [0] drains taken by rxf_body; declined: [1] preconditions (state, period unknown), [2] the pattern does not
// end at the limit / too many records, [3] the ring does not hold the predicted records, [4] long run of small records,
// [5] no room in plan / slice table / arena
__device__ unsigned long long g_rx_fast_drains[6] = {0, 0, 0, 0, 0, 0};

// LDS of the receive planners.  The steady-state body runs first and the general planner (grdma_rx_plan.hip) only
// after it has declined, so their big arrays share one allocation: that is what lets one planner launch (k_plan_pair, k_plan_pair_mw) hold both
// receive bodies AND both send bodies within the CU's 160 KB.
#define RXG_BULK 4096                       // BULK_MAX of the general planner
#define RXG_PAD(i) ((i) + ((i) >> 4))       // its index padding (RXP)
struct rx_lds_general {
  uint32_t hist[GRDMA_RX_HIST];
  uint32_t penc[RXG_PAD(RXG_BULK) + 1];
  uint32_t xenc[RXG_PAD(RXG_BULK + 1) + 1];
  uint32_t n[RXG_PAD(RXG_BULK) + 1];
  uint16_t sin[RXG_PAD(RXG_BULK) + 1];
};
struct rx_lds_fast {
  uint32_t hist[GRDMA_RX_HIST];
  uint32_t pat[RXF_PMAX], pre[RXF_PMAX + 1];
  uint32_t n[RXF_MAX + RXF_MAX / RXF_PER + 2];
  uint16_t sin[RXF_MAX + RXF_MAX / RXF_PER + 2];
};
union rx_lds {
  rx_lds_general g;
  rx_lds_fast f;
  uint32_t hint_room[2 * GRDMA_HINT_MAX_RECORDS + 2];  // (rx_lds_hint, grdma_rx_hint.h: the sizes of a round and their prefix)
};
__device__ __forceinline__ rx_lds* rx_lds_get() {
  __shared__ rx_lds L;
  return &L;
}

// (the three functions below restate read_space_after / replay_record32 of grdma_rx_plan.hip)
// s = bytes of space left in the open read (0 = between reads) after a record of n bytes
__device__ __forceinline__ uint32_t rxf_space_after(uint32_t n, uint32_t s) {
  if (s == 0) return n >= RXF_MINRD ? 0 : RXF_MINRD - n;
  if (n < s) return s - n;
  if (n == s) return 0;
  const uint32_t r = n - s;
  return r >= RXF_MINRD ? 0 : RXF_MINRD - r;
}

struct rxf_rec {
  uint32_t c1, c2;    // bytes of the (at most) two Recv steps
  uint32_t sl0, sl1;  // slices completed by this record (0 = none); sl0 == 256 stands for "the open read"
  uint32_t sl_cnt;
};
__device__ __forceinline__ rxf_rec rxf_replay(uint32_t n, uint32_t s_in) {
  rxf_rec r;
  r.c1 = n;
  r.c2 = 0;
  r.sl0 = r.sl1 = 0;
  if (s_in == 0) {
    if (n >= RXF_MINRD) r.sl0 = n;
  } else if (n <= s_in) {
    if (n == s_in) r.sl0 = RXF_MINRD;
  } else {
    r.c1 = s_in;
    r.c2 = n - s_in;
    r.sl0 = RXF_MINRD;
    if (r.c2 >= RXF_MINRD) r.sl1 = r.c2;
  }
  r.sl_cnt = (r.sl0 ? 1u : 0u) + (r.sl1 ? 1u : 0u);
  return r;
}
__device__ __forceinline__ uint32_t rxf_al16(uint32_t v) { return (v + 15u) & ~15u; }
__device__ __forceinline__ uint32_t rxf_tiles(uint32_t len, uint32_t ts) { return (len + (1u << ts) - 1u) >> ts; }

// Layout of one record: ring pieces (<= 4: two steps, each cut once at the ring end), where they go in the
// arena, what it completes.  `odd` = this record completes the read that was open when the drain began and
// that read's capacity s0 is not 256 (a read cut short by the previous drain's would-block keeps its rest as
// the next read's capacity, rdma_bp_posix.cc:283-287): the slice it completes is s0 long and the next slice
// starts at the next 16-byte boundary, so the two steps are NOT contiguous in the arena.
struct rxf_layout {
  uint32_t off[4], len[4];
  uint32_t dst_rel[4];  // arena offset of each piece relative to A (the start of the open / next slice)
  uint32_t nsg, ntl;
  uint32_t sl0, sl1, sl_cnt, bytes;  // slices completed, arena bytes they take (16-byte granular)
};
__device__ __forceinline__ rxf_layout rxf_lay(uint32_t n, uint32_t s_in, uint32_t pay, uint32_t cap, uint32_t cap_open,
                                              bool odd, uint32_t ts) {
  rxf_layout L;
  const rxf_rec rp = rxf_replay(n, s_in);
  L.sl0 = (rp.sl0 == RXF_MINRD && s_in != 0) ? cap_open : rp.sl0;
  L.sl1 = rp.sl1;
  L.sl_cnt = rp.sl_cnt;
  L.bytes = rxf_al16(L.sl0) + rxf_al16(L.sl1);
  const uint32_t filled = s_in ? cap_open - s_in : 0;
  const uint32_t d2 = (odd && rp.c2) ? rxf_al16(cap_open) : filled + rp.c1;  // where step 2 lands
  const uint32_t mask = cap - 1;
  // step 1 = pieces 0, 1; step 2 = pieces 2, 3
  const uint32_t p0 = pay, l_a = rp.c1 < cap - p0 ? rp.c1 : cap - p0;
  const uint32_t p2 = (pay + rp.c1) & mask, l_c = rp.c2 < cap - p2 ? rp.c2 : cap - p2;
  L.off[0] = p0;  L.len[0] = l_a;          L.dst_rel[0] = filled;
  L.off[1] = 0;   L.len[1] = rp.c1 - l_a;  L.dst_rel[1] = filled + l_a;
  L.off[2] = p2;  L.len[2] = l_c;          L.dst_rel[2] = d2;
  L.off[3] = 0;   L.len[3] = rp.c2 - l_c;  L.dst_rel[3] = d2 + l_c;
  // the two steps are one run in the ring and in the arena unless the ring end or the odd slice cuts them
  if (pay + n <= cap && L.len[2] != 0 && d2 == filled + rp.c1) {  // (pay + n <= cap: no piece is cut, step 2 follows step 1)
    L.len[0] += L.len[2];
    L.len[2] = 0;
  }
  L.nsg = 0;
  L.ntl = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    L.nsg += L.len[k] ? 1u : 0u;
    L.ntl += rxf_tiles(L.len[k], ts);
  }
  return L;
}

// exclusive block scans of three u32 values at once (1024 threads); totals in tot[3]
__device__ __forceinline__ void rxf_scan3(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t (*s_w)[RXF_WAVES],
                                          uint32_t* x0, uint32_t* x1, uint32_t* x2, uint32_t tot[3]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1), i2 = wave_incl_scan_u32(v2);
  if (lane == 63) {
    s_w[0][wave] = i0;
    s_w[1][wave] = i1;
    s_w[2][wave] = i2;
  }
  __syncthreads();
  uint32_t b0 = 0, b1 = 0, b2 = 0, t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
  for (int w = 0; w < RXF_WAVES; w++) {
    const uint32_t a = s_w[0][w], b = s_w[1][w], c = s_w[2][w];
    if (w < wave) { b0 += a; b1 += b; b2 += c; }
    t0 += a; t1 += b; t2 += c;
  }
  __syncthreads();
  *x0 = b0 + i0 - v0;
  *x1 = b1 + i1 - v1;
  *x2 = b2 + i2 - v2;
  tot[0] = t0; tot[1] = t1; tot[2] = t2;
}

// LDS index padding: a thread walks a contiguous run of RXF_PER records; one extra slot per run makes the
// lanes' stride odd (5 words), which spreads them over the banks.
#define RXFP(i) ((i) + ((i) / RXF_PER))

// Returns true when it took the drain (everything is written), false when the general planner has to
// (nothing is written).  Every thread of the workgroup returns the same value.
__device__ __forceinline__ bool rxf_body(const grdma_rx_op& op_in) {
  const grdma_rx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;

  rx_lds* const lds = rx_lds_get();
  auto& s_hist = lds->f.hist;
  auto& s_pat = lds->f.pat;
  auto& s_pre = lds->f.pre;
  auto& s_n = lds->f.n;
  auto& s_sin = lds->f.sin;
  static_assert(RXFP(RXF_MAX) + 2 <= sizeof(lds->f.n) / sizeof(uint32_t), "padded record arrays fit");
  __shared__ uint32_t s_w[3][RXF_WAVES];
  __shared__ uint32_t s_bad, s_vj, s_first, s_send, s_totn;

  // ---- 0. state, preconditions (every thread reads the same words; nothing is stored before the probe passed)
  uint8_t* const ring = c->ring;
  const uint64_t cap64 = c->cap;
  const uint64_t head64 = c->head, mh0 = c->moving_head, remain0 = c->remain, leftover0 = c->leftover_cap;
  const uint64_t irs0 = c->internal_read_size;
  const uint64_t hc = c->rx_hist_count;
  const uint32_t P = c->rx_period;
  const uint32_t status = c->status;
  const uint32_t* const gh = c->rx_hist;
  const uint64_t lim = op.limit_ptr ? __hip_atomic_load(op.limit_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  const uint64_t slice_idx0 = op.append == 1 ? c->rx_slice_idx : 0;
  const uint64_t a_off0 = op.append == 1 ? c->rx_arena_off : 0;
  // (what thread 0 adds to at the very end: fetched now, in the same round trip as the state above)
  const uint64_t o_total_read = c->total_read, o_credit_msgs = c->credit_msgs;
  const uint64_t o_rx_records = c->rx_records, o_rx_rounds = c->rx_rounds;
  const uint32_t o_h1 = c->rx_h1;
  const uint64_t o_seq = res->seq;
  grdma_hostline* const line = c->line;
  constexpr int NH = GRDMA_RX_HIST / RXF_THREADS;
  uint32_t hv[NH];
#pragma unroll
  for (int r = 0; r < NH; r++) hv[r] = gh[tid + r * RXF_THREADS];

  bool ok = status == GRDMA_PAIR_CONNECTED && op.raw_cap == 0 && op.append != 0 && !op.inline_apply &&
            op.limit_ptr != nullptr && remain0 == 0 && leftover0 <= RXF_MINRD && P != 0 && P <= RXF_PMAX && hc >= P &&
            cap64 <= (1ull << 31) && a_off0 < (1ull << 31);
  uint64_t max_slices = GRDMA_MAX_SLICES;
  {
    const uint64_t room = op.slices_cap > slice_idx0 ? op.slices_cap - slice_idx0 : 0;
    if (room < max_slices) max_slices = room;
    if (op.max_reads < max_slices) max_slices = op.max_reads;
  }
  const uint32_t cap = (uint32_t)cap64, mask = cap - 1u, head = (uint32_t)head64;
  const uint32_t Lr = ((uint32_t)lim - head) & mask;  // ring bytes between my head and the sender's tail
  const bool idle = Lr == 0;  // nothing has arrived: the general planner records the would-block
  if (tid == 0) {
    s_bad = 0;
    s_vj = 0xFFFFFFFFu;
    s_first = 0xFFFFFFFFu;
    s_send = 0;
    s_totn = 0;
  }
#pragma unroll
  for (int r = 0; r < NH; r++) s_hist[tid + r * RXF_THREADS] = hv[r];
  __syncthreads();
  if (!ok || idle) {  // (uniform)
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[1], 1ull);
      if (!idle) res->pad0++;  // (pad1 / pad0: drains of this result block taken / declined with data waiting)
    }
    return false;
  }

  // ---- 1. the pattern: the newest P record sizes, their prefix sums, and where the limit falls in it
  //         (thread t holds pattern entries 2t and 2t + 1)
  static_assert(RXF_PMAX <= 2 * RXF_THREADS, "two pattern entries per thread");
  const uint32_t j0 = 2 * tid, j1 = j0 + 1;
  const uint32_t pv0 = j0 < P ? s_hist[(uint32_t)((hc - P + j0) % GRDMA_RX_HIST)] : 0;
  const uint32_t pv1 = j1 < P ? s_hist[(uint32_t)((hc - P + j1) % GRDMA_RX_HIST)] : 0;
  uint32_t px, dummy1, dummy2, ptot[3];
  rxf_scan3(pv0 + pv1, 0, 0, s_w, &px, &dummy1, &dummy2, ptot);
  const uint32_t SP = ptot[0];
  if (j0 < P) {
    s_pat[j0] = pv0;
    s_pre[j0] = px;
  }
  if (j1 < P) {
    s_pat[j1] = pv1;
    s_pre[j1] = px + pv0;
  }
  if (tid == 0) s_pre[P] = SP;
  const uint32_t q_full = SP ? Lr / SP : 0, rem = SP ? Lr - q_full * SP : 0;
  if (SP != 0) {  // (pre[] is strictly increasing: at most one match)
    if (j0 < P && px == rem) s_vj = j0;
    if (j1 < P && px + pv0 == rem) s_vj = j1;
  }
  __syncthreads();
  const uint32_t vj = s_vj;
  const uint64_t V64 = (uint64_t)q_full * P + vj;
  if (SP == 0 || vj == 0xFFFFFFFFu || V64 == 0 || V64 > RXF_MAX) {  // (uniform)
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[2], 1ull);
      res->pad0++;
    }
    return false;
  }
  const uint32_t V = (uint32_t)V64;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap64);
  const uint64_t t_pattern = __builtin_amdgcn_s_memtime();

  // ---- 2. one probe round: footer of record i and header of record i + 1 are neighbouring words
  const uint32_t i0 = tid * RXF_PER;
  uint32_t xe[RXF_PER], ee[RXF_PER];  // exclusive encoded prefix and encoded size of my records
  uint32_t e_after;                    // size the pattern predicts for the record behind my last one
  {
    uint32_t qi = i0 / P, ri = i0 - qi * P;
#pragma unroll
    for (int r = 0; r < RXF_PER; r++) {
      xe[r] = qi * SP + s_pre[ri];
      ee[r] = s_pat[ri];
      if (++ri == P) {
        ri = 0;
        qi++;
      }
    }
    e_after = s_pat[ri];
  }
  {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t ring_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ring);
    const uint32_t ring_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ring >> 32));
    const uint64_t ring_u = ((uint64_t)ring_hi << 32) | (uint64_t)ring_lo;
    const uint32_t cap_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ring_u, 0, cap_u, 0x00020000);
    u32x4 pairs[RXF_PER];
    const u32x2 hf = __builtin_amdgcn_raw_buffer_load_b64(rs, head & ~7u, 0, 16);  // header of record 0
    const u32x2 wz = __builtin_amdgcn_raw_buffer_load_b64(rs, 0u, 0, 16);          // first word of the ring
#pragma unroll
    for (int r = 0; r < RXF_PER; r++) {
      const uint32_t f = (head + xe[r] + ee[r] - 8u) & mask & ~7u;
      pairs[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, f <= cap_u - 16u ? f : cap_u - 16u, 0, 16 /* sc1 */);
    }
    bool bad = false;
    if (tid == 0) {
      const uint64_t h0 = ((uint64_t)hf.y << 32) | hf.x;
      bad |= !(h0 != 0 && h0 <= cap64 - GRDMA_RESERVED && 16u + (uint32_t)round_up8(h0) == ee[0]);
      s_n[RXFP(0)] = (uint32_t)h0;
    }
#pragma unroll
    for (int r = 0; r < RXF_PER; r++) {
      const uint32_t i = i0 + r;
      if (i < V) {
        const uint32_t f = (head + xe[r] + ee[r] - 8u) & mask & ~7u;
        const bool last_word = f > cap_u - 16u;
        const uint64_t lo = ((uint64_t)pairs[r].y << 32) | pairs[r].x, hi = ((uint64_t)pairs[r].w << 32) | pairs[r].z;
        const uint64_t foot = last_word ? hi : lo;
        const uint64_t next = last_word ? (((uint64_t)wz.y << 32) | wz.x) : hi;
        const uint32_t e_next = r + 1 < RXF_PER ? ee[r + 1 < RXF_PER ? r + 1 : 0] : e_after;
        bad |= foot != GRDMA_FOOTER;
        if (i + 1 < V) {
          bad |= !(next != 0 && next <= cap64 - GRDMA_RESERVED && 16u + (uint32_t)round_up8(next) == e_next);
          s_n[RXFP(i + 1)] = (uint32_t)next;
        }
      }
    }
    if (bad) s_bad = 1;
  }
  __syncthreads();
  if (s_bad) {  // (uniform) the ring does not hold what the pattern says: the general planner takes this drain
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[3], 1ull);
      res->pad0++;
    }
    return false;
  }
  const uint64_t t_probe = __builtin_amdgcn_s_memtime();

  // ---- 3. incoming read state of every record (look back to the nearest record that resets it)
  const uint32_t s0 = (uint32_t)leftover0;
  if (i0 < V) {
    uint32_t j = i0, steps = 0;
    while (j > 0 && s_n[RXFP(j - 1)] < 2 * RXF_MINRD && steps < RXF_LOOKBACK) {
      j--;
      steps++;
    }
    if (j > 0 && s_n[RXFP(j - 1)] < 2 * RXF_MINRD) s_bad = 1;  // a long run of small records: not this body's case
    const bool from_start = j == 0;  // my chain began at the drain's start: it may hold the slice that closes the open read
    uint32_t s = j == 0 ? s0 : 0;
    for (; j < i0; j++) s = rxf_space_after(s_n[RXFP(j)], s);
    for (uint32_t r = 0; r < RXF_PER; r++) {
      const uint32_t i = i0 + r;
      if (i < V) {
        s_sin[RXFP(i)] = (uint16_t)s;
        const uint32_t n = s_n[RXFP(i)];
        // the first record that completes a slice: it closes the read that was open at the start (if any)
        if (from_start && rxf_replay(n, s).sl_cnt != 0) atomicMin(&s_first, i);
        s = rxf_space_after(n, s);
        if (i == V - 1) s_send = s;
      }
    }
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[4], 1ull);
      res->pad0++;
    }
    return false;
  }
  const uint32_t first_done = s_first;  // 0xFFFFFFFF: no slice completes in this drain
  const uint32_t s_end = s_send;
  const bool odd_open = s0 != 0 && s0 != RXF_MINRD;  // the open read's capacity is not a fresh read's 256
  const uint64_t t_state = __builtin_amdgcn_s_memtime();

  // ---- 4. counts, prefix sums, room
  uint32_t my_pk = 0, my_tl = 0, my_by = 0, my_n = 0;
#pragma unroll
  for (int r = 0; r < RXF_PER; r++) {
    const uint32_t i = i0 + r;
    if (i < V) {
      const uint32_t n = s_n[RXFP(i)], s_in = s_sin[RXFP(i)];
      const bool in_first = odd_open && i <= first_done;
      const rxf_layout L = rxf_lay(n, s_in, (head + xe[r] + 8u) & mask, cap, in_first ? s0 : RXF_MINRD,
                                   odd_open && i == first_done, ts);
      my_pk += L.sl_cnt | (L.nsg << 16);
      my_tl += L.ntl;
      my_by += L.bytes;
      my_n += n;
    }
  }
  uint32_t x_pk, x_tl, x_by, tot[3];
  rxf_scan3(my_pk, my_tl, my_by, s_w, &x_pk, &x_tl, &x_by, tot);
  // (payload total: a plain reduction riding on the scan's barriers would need a fourth lane array; one more
  // wave reduction + LDS atomic is cheaper than a second scan)
  {
    uint32_t wn = my_n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wn += __shfl_xor(wn, d, 64);
    if ((tid & 63) == 0 && wn) atomicAdd(&s_totn, wn);
  }
  const uint32_t tot_sl = tot[0] & 0xFFFFu, tot_sg = tot[0] >> 16, tot_tl = tot[1], tot_by = tot[2];
  // the would-block at the end (rdma_do_read, rdma_bp_posix.cc:195-277): a read with bytes in it is handed up
  // as it is -- a short slice -- and its rest stays open; a clean state allocates a fresh 256-byte read
  const uint32_t cap_open_end = (odd_open && first_done == 0xFFFFFFFFu) ? s0 : RXF_MINRD;
  const uint32_t short_len = s_end ? cap_open_end - s_end : 0;
  const uint32_t nsl_final = tot_sl + (short_len ? 1u : 0u);
  const uint32_t leftover_final = s_end ? s_end : RXF_MINRD;
  const uint64_t a_end = a_off0 + tot_by + rxf_al16(short_len);
  if (!(nsl_final + 2 <= max_slices && tot_sg + 8 <= GRDMA_MAX_SEGS && a_end + leftover_final + 16 <= op.arena_cap &&
        a_end < (1ull << 32))) {  // (uniform)
    if (tid == 0) {
      atomicAdd(&g_rx_fast_drains[5], 1ull);
      res->pad0++;
    }
    return false;
  }
  const uint64_t t_scan = __builtin_amdgcn_s_memtime();

  // ---- 5. segments, tile prefix, slices: the drain is committed from here on
  grdma_slice_out* const out_slices = op.slices + slice_idx0;
  {
    uint32_t sl = x_pk & 0xFFFFu, sg = x_pk >> 16, tl = x_tl, by = x_by;
#pragma unroll
    for (int r = 0; r < RXF_PER; r++) {
      const uint32_t i = i0 + r;
      if (i < V) {
        const uint32_t n = s_n[RXFP(i)], s_in = s_sin[RXFP(i)];
        const bool in_first = odd_open && i <= first_done;
        const rxf_layout L = rxf_lay(n, s_in, (head + xe[r] + 8u) & mask, cap, in_first ? s0 : RXF_MINRD,
                                     odd_open && i == first_done, ts);
        const uint64_t A = a_off0 + by;  // start of the open slice, or of the slice this record begins
        int last_piece = 0;
#pragma unroll
        for (int k = 1; k < 4; k++)
          if (L.len[k]) last_piece = k;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (L.len[k] == 0) continue;
          const uint64_t fl = GRDMA_SEG_ZERO_SRC | (k == 0 ? GRDMA_SEG_TAG_HDR : 0) | (k == last_piece ? GRDMA_SEG_TAG_FTR : 0);
          plan->segs[sg] = {(uint64_t)op.arena + A + L.dst_rel[k], (uint64_t)(ring + L.off[k]), (uint64_t)L.len[k], fl};
          plan->tile_prefix[sg] = tl;
          sg++;
          tl += rxf_tiles(L.len[k], ts);
        }
        uint64_t sof = A;
        if (L.sl0) {
          out_slices[sl].off = sof;
          out_slices[sl].len = L.sl0;
          sl++;
          sof += rxf_al16(L.sl0);
        }
        if (L.sl1) {
          out_slices[sl].off = sof;
          out_slices[sl].len = L.sl1;
          sl++;
        }
        by += L.bytes;
      }
    }
  }
  // history: the records of this drain become the newest entries (the pattern simply continues)
  {
    const uint32_t first = V > GRDMA_RX_HIST ? V - GRDMA_RX_HIST : 0;
#pragma unroll
    for (int r = 0; r < RXF_PER; r++) {
      const uint32_t i = i0 + r;
      if (i >= first && i < V) c->rx_hist[(uint32_t)((hc + i) % GRDMA_RX_HIST)] = ee[r];
    }
  }
  const uint64_t t_emit = __builtin_amdgcn_s_memtime();

  // ---- 6. credit (pair.cc:276-284), state, result: thread 0
  __syncthreads();  // (s_totn is complete)
  if (tid == 0) {
    const uint32_t tot_n = s_totn;
    auto enc_end = [&](uint32_t i) -> uint64_t {  // ring bytes consumed once record i is finished
      const uint32_t qi = i / P, ri = i - qi * P;
      return (uint64_t)qi * SP + s_pre[ri] + s_pat[ri];
    };
    const uint64_t T = cap64 / 2, Ctot = Lr;
    uint64_t base = 0, thr = T - irs0, credit = 0, credit_head = 0;
    bool crossed = false;
    while (Ctot >= thr) {
      uint32_t lo = 0, hi = V - 1;  // first record whose running consumption (after its last step) reaches thr
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (enc_end(mid) >= thr) hi = mid; else lo = mid + 1;
      }
      const uint32_t n = s_n[RXFP(lo)];
      const rxf_rec rp = rxf_replay(n, s_sin[RXFP(lo)]);
      const uint64_t C2 = enc_end(lo);
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
    const uint64_t irs = crossed ? Ctot - base : irs0 + Ctot;
    const uint64_t nh = (head64 + Lr) & (cap64 - 1);
    if (short_len) {
      out_slices[tot_sl].off = a_off0 + tot_by;
      out_slices[tot_sl].len = short_len;
    }
    plan->nsegs = tot_sg;
    plan->ntiles = tot_tl;
    plan->tile_bytes = 1u << ts;
    plan->tile_prefix[tot_sg] = tot_tl;
    plan->bytes = tot_n;
    plan->tag_base = (uint64_t)ring;
    plan->tag_mask = cap64 - 1;
    plan->blocks_done = 0;
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
    {
      const uint32_t ql = (V - 1) / P, rl = (V - 1) - ql * P;
      c->rx_h1 = s_pat[rl];
      c->rx_h2 = V >= 2 ? s_pat[rl ? rl - 1 : P - 1] : o_h1;
    }
    if (credit) c->status_send.remote_head = credit_head;
    res->credit_head = credit_head;
    res->nslices = nsl_final;
    res->bytes = tot_n;
    res->consumed = Lr;
    res->records = V;
    res->would_block = 1;
    res->credit_sent = credit;
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
    res->dbg[8] = P;
    res->dbg[9] = 0xFA57;  // this stamp set comes from rxf_body
    res->pad1++;
    res->dbg[1] = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_rx_fast_drains[0], 1ull);
    // (relaxed: the consumers of a streaming job's drain are later kernels of the graph; a release at system scope
    // here would write the XCD's L2 back -- the plan just laid out -- before the kernel may end)
    __hip_atomic_store(&res->seq, op.seq_next ? op.seq_next : o_seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return true;
}

}  // namespace
#endif  // GRDMA_RX_FAST_H
