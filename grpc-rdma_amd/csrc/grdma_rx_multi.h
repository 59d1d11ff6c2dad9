// rxm_body: the steady-state drain of a streaming connection (grdma_rx_fast.h) laid out by SEVERAL workgroups
// at once, with no word exchanged between them (included by grdma_rx_plan.hip; k_plan_pair_mw / k_rx_plan_mw run it).
//
// Why: rxf_body is one workgroup, and what bounds it is not the dependent round trips (a chained load costs
// ~0.2 us behind the launch before, tools/trip_probe.hip) but what ONE CU can push through its memory pipe and
// issue: the probe of 4095 records 8 KB apart takes 6.9 us from one workgroup and 2.2 us from four, 393 KB of
// segments 5.4 us against 1.3 us, and four records per thread are four times the instruction stream.  Same
// contract as rxf_body -- GetReadableSize / Read (ring_buffer.cc:67-191), Recv with its credit rule
// (pair.cc:264-286), the endpoint-read loop (rdma_bp_posix.cc:180-326) replayed for every record between the
// reader's head and the tail its sender reported; identical slices, segments, ring state, credit reports and
// history -- but workgroup b of G owns records [b * 256, (b + 1) * 256), ONE record per thread, and a workgroup is
// FOUR wavefronts, one per SIMD: a SIMD issues one wave-instruction every four cycles, so the sixteen waves of a
// 1024-thread planner on one CU take 16 x (instructions per thread) cycles whatever the memory system does (measured:
// tools/trip_probe.hip, straight-line arithmetic at 64 / 256 / 1024 threads) -- a steady-state body of ~3000
// instructions per thread is 20 us that way and 5 us as one wave per SIMD.
//
// What a record needs from the records in front of it -- how many slices, segments, tiles and arena bytes they
// took -- is a block scan in rxf_body.  Here it is a closed form, so that no workgroup waits for another:
//
//   * The record sizes are periodic (period P, the precondition of the steady-state body).  The payload sizes of
//     the pattern are the headers of the first min(P, V) records of this drain (every workgroup loads them in
//     the same round trip as its own records' header and footer; every record's owner checks its header against
//     the pattern, so a drain whose sizes are not periodic in the PAYLOAD size is declined as a whole).
//   * The endpoint-read state in front of a record (space left in the open read) depends on the records back to
//     the nearest one that resets it (n >= 512: whatever was open, such a record closes it and leaves nothing
//     open).  Let F be the first such record of the drain (F <= RXF_LOOKBACK, else this body declines -- as
//     rxf_body does for long runs of small records).  Records 0..F -- the PREFIX REGION -- depend on the read the
//     last drain left open; every workgroup lays them out for itself (<= 193 records, a scan).  Behind F the
//     state is a function of the position in the pattern alone, so "what records F+1 .. i-1 took" is
//     cyc(i) - cyc(F + 1) with cyc(k) = (k / P) * (sum over one period) + (prefix inside the period): two table
//     look-ups in LDS.  The tables are laid out for a ring without an end; the one record of a drain whose
//     payload crosses the ring end is found by every workgroup from the same closed form (the record whose span
//     contains offset `cap`), laid out both ways, and the difference added behind it.
//
// Every workgroup reaches the same verdict on everything that is not its own probe (same inputs, same
// arithmetic); the probes are combined by the arrival counter: a workgroup adds 1 (and 65536 if it has to decline)
// to plan->mw_arrive, the LAST one to arrive sees the sum, and either commits -- credit (pair.cc:276-284), state,
// result, history: thread 0 and one thread per history entry -- or, if any workgroup declined, returns 2: its
// caller then runs the general planner in the same launch (what the others wrote are plan / slice-table entries
// beyond any count that was committed; the general planner rewrites them).  Nothing but that counter is shared.
#ifndef GRDMA_RX_MULTI_H
#define GRDMA_RX_MULTI_H
#include "grdma_rx_fast.h"

namespace {

#define RXM_THREADS 256u               // one wave per SIMD
#define RXM_WAVES (RXM_THREADS / 64u)
#define RXM_G 16u                      // workgroups per drain (k_plan_pair_mw's grid): 16 x 256 records
#define RXM_CHUNK RXM_THREADS          // records per workgroup = threads
#define RXM_PFX (RXF_LOOKBACK + 1u)    // most records the prefix region may hold
#define RXM_RESET (2u * RXF_MINRD)     // a record of this many bytes or more leaves no read open (rxf_space_after)
#define RXM_NONE 0xFFFFFFFFu

// diagnostics: committed drains whose read-state tables came out of the connection's table cache [0] / were computed and
// written back [1] (grdma_rx_table_cache_stats)
__device__ unsigned long long g_rx_tab_stats[2] = {0, 0};
// diagnostics: drains of the multi-workgroup bodies that went to the general planner with a MIXED verdict -- some
// workgroups' probes passed (their plan entries were written, and are rewritten by that planner in the same launch),
// some declined -- [0], and with every workgroup declining [1] (grdma_rx_verdict_counts)
__device__ unsigned long long g_rx_verdicts[2] = {0, 0};

struct rx_lds_multi {
  uint32_t hist[GRDMA_RX_HIST];
  uint32_t pat[RXF_PMAX], pre[RXF_PMAX + 1];   // encoded sizes of the pattern and their exclusive prefix
  uint32_t npat[RXF_PMAX];                      // payload sizes of the pattern (RXM_NONE: not part of this drain)
  uint16_t sss[RXF_PMAX];                       // steady read state in front of pattern position j
  uint32_t qpk[RXF_PMAX + 1], qtl[RXF_PMAX + 1], qby[RXF_PMAX + 1];  // exclusive prefix over the steady pattern of
                                                // (slices | segments << 16), tiles, arena bytes; [P] = one period
  uint32_t qn[RXF_PMAX + 1];                    // exclusive prefix of the payload sizes
  uint16_t as[RXM_PFX + 1];                     // read state in front of prefix-region record t
  uint32_t apk[RXM_PFX + 1], atl[RXM_PFX + 1], aby[RXM_PFX + 1];     // exclusive prefix over the prefix region
};

// exclusive block scans of four u32 values at once (RXM_THREADS threads); totals in tot[4]
__device__ __forceinline__ void rxm_scan4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t (*s_w)[RXM_WAVES],
                                          uint32_t* x0, uint32_t* x1, uint32_t* x2, uint32_t* x3, uint32_t tot[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1), i2 = wave_incl_scan_u32(v2), i3 = wave_incl_scan_u32(v3);
  if (lane == 63) {
    s_w[0][wave] = i0;
    s_w[1][wave] = i1;
    s_w[2][wave] = i2;
    s_w[3][wave] = i3;
  }
  __syncthreads();
  uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
  for (int w = 0; w < (int)RXM_WAVES; w++) {
    const uint32_t a = s_w[0][w], b = s_w[1][w], c = s_w[2][w], d = s_w[3][w];
    if (w < wave) { b0 += a; b1 += b; b2 += c; b3 += d; }
    t0 += a; t1 += b; t2 += c; t3 += d;
  }
  __syncthreads();
  *x0 = b0 + i0 - v0;
  *x1 = b1 + i1 - v1;
  *x2 = b2 + i2 - v2;
  *x3 = b3 + i3 - v3;
  tot[0] = t0; tot[1] = t1; tot[2] = t2; tot[3] = t3;
}

// k / P for k * P < 2^32 from inv = ceil(2^32 / P) (one real division per launch instead of ~20: a u32 division is
// ~40 instructions, and this body's time is its instruction count)
// (P == 1: the reciprocal does not fit 32 bits; the quotient is k)
__device__ __forceinline__ uint32_t rxm_div(uint32_t k, uint32_t P, uint32_t inv) { return P > 1 ? (uint32_t)(((uint64_t)k * inv) >> 32) : k; }

// what the records [0, k) of an endless steady pattern took: k = q * P + r
__device__ __forceinline__ void rxm_cyc(const rx_lds_multi& M, uint32_t P, uint32_t inv, uint32_t k, uint32_t* pk, uint32_t* tl, uint32_t* by) {
  const uint32_t q = rxm_div(k, P, inv), r = k - q * P;
  *pk = q * M.qpk[P] + M.qpk[r];
  *tl = q * M.qtl[P] + M.qtl[r];
  *by = q * M.qby[P] + M.qby[r];
}

// The tables of the multi-workgroup bodies: the receive planners' shared LDS (64 KB, the general planner's arrays).
// (SMALL: an allocation of their own size, for a body that shares its launch with copy workgroups which must keep
//  their occupancy -- round 4's fused round, retired; no kernel instantiates it now.)
#define RXM_SMALL_LDS_BYTES 33024
template <bool SMALL>
__device__ __forceinline__ void* rx_tables() {
  if (SMALL) {
    __shared__ __attribute__((aligned(16))) uint8_t s_small[RXM_SMALL_LDS_BYTES];
    return s_small;
  }
  return rx_lds_get();
}

// Wire inside the planner pair's launch (k_plan_pair_mw; a job of few links with small rings): the wire's
// workgroups -- the lowest blockIdx.y, dispatched first -- move the round into the peer ring, write their L2 back and
// count in at the WIRE plan's arrival word; the drain's workgroups do everything that does not look at the ring, then
// wait here for all of them.  Every drain workgroup of the launch calls this exactly once (all threads); the last one
// through zeroes the words for the next launch.  A wait that runs out is counted (g_wire_wait_runout: a test reads it)
// and the drain goes on -- it delivers what the ring holds, as it always does.
__device__ unsigned long long g_wire_wait_runout = 0;
__device__ __forceinline__ void wire_arrive(grdma_plan* wp) {  // (thread 0 of a wire workgroup, behind a barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_fetch_add(&wp->mw_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wire_wait(grdma_plan* wp, uint32_t wire_groups, uint32_t participants) {
  if (threadIdx.x == 0) {
    bool seen = false;
    for (uint32_t spins = 0; spins < (1u << 22) && !seen; spins++) {
      seen = __hip_atomic_load(&wp->mw_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wire_groups;
      if (!seen) __builtin_amdgcn_s_sleep(2);
    }
    if (!seen) atomicAdd(&g_wire_wait_runout, 1ull);
    const uint32_t prev = __hip_atomic_fetch_add(&wp->promise_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == participants) {
      __hip_atomic_store(&wp->promise_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&wp->mw_arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}
// (what a drain body calls before it looks at the ring for the first time)
struct ring_ready_now {
  __device__ __forceinline__ void operator()() {}
};
struct ring_behind_wire {
  grdma_plan* wp;
  uint32_t wire_groups, participants;
  bool done;
  __device__ __forceinline__ void operator()() {
    if (wp != nullptr && !done) wire_wait(wp, wire_groups, participants);  // (uniform)
    done = true;
  }
};
// (what the committing thread of a drain body calls with the credit the drain posts, as soon as it is known -- before
//  the body's bookkeeping: the promised credit of a launch, csrc/grdma_devfn.h promise_keep)
struct credit_unpublished {
  __device__ __forceinline__ void operator()(uint64_t, uint64_t) {}
};
struct credit_promised {
  grdma_plan* plan;
  uint32_t participants;
  bool done;
  __device__ __forceinline__ void operator()(uint64_t credit_sent, uint64_t credit_head) {
    if (plan != nullptr && !done) (void)promise_keep(plan, participants, credit_sent, credit_head);
    done = true;
  }
};

// ---- (round 6) Who commits a drain, and when.  Through the first half of the round every workgroup counted in behind its
// entries -- write-through stores acknowledged first, because the general planner may rewrite them in this launch -- and the
// LAST TO ARRIVE committed: an acknowledgement round trip and a device-scope round trip at the tail of every drain plan.
// What the count carries is the verdict, and that is final before a workgroup emits anything: it counts in THERE (its
// loads of the connection block and of the table slot have returned by then -- what the commit will overwrite), and the
// workgroup dispatched LAST commits once all verdicts are in, which they usually are when it has emitted its own
// entries.  The acknowledgement moved off the path: a workgroup that is not the committer counts OUT behind it
// (grdma_plan::mw_done); the committer waits for those counts before it lets the general planner rewrite entries (a
// declined drain), or else at the very end, before it clears both words for the next launch.  (Under the emulator, which
// runs the workgroups of a launch in index order, the others are through when the last one starts.)
__device__ __forceinline__ void drain_count_in(grdma_plan* plan, uint32_t reason) {  // (thread 0, behind a barrier)
  __hip_atomic_fetch_add(&plan->mw_arrive, 1u + (reason ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_leave(grdma_plan* plan) {  // (thread 0 of a workgroup that does not commit)
  __hip_atomic_fetch_add(&plan->mw_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the committer, all threads: every verdict is in; returns how many workgroups declined
__device__ __forceinline__ uint32_t drain_verdicts(grdma_plan* plan, uint32_t nwg) {
  __shared__ uint32_t s_declined;
  if (threadIdx.x == 0) {
    uint32_t v = 0;
    for (uint32_t spins = 0; spins < (1u << 22); spins++) {
      v = __hip_atomic_load(&plan->mw_arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v & 0xFFFFu) >= nwg) break;
      __builtin_amdgcn_s_sleep(1);
    }
    // (a wait that runs out -- never seen -- declines the drain: the general planner walks what the ring holds)
    s_declined = (v & 0xFFFFu) >= nwg ? (v >> 16) : nwg;
  }
  __syncthreads();
  return s_declined;
}
// the committer, thread 0: the other workgroups' entries are at the memory side; the words are clear for the next launch
__device__ __forceinline__ void drain_close(grdma_plan* plan, uint32_t nwg) {
  for (uint32_t spins = 0; spins < (1u << 22); spins++) {
    if (__hip_atomic_load(&plan->mw_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nwg - 1) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(&plan->mw_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&plan->mw_arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns 0: not the committing workgroup of this drain (nothing more to do); 1: the committing one, the drain is
// committed; 2: the committing one, and a workgroup declined -- the caller runs the general planner; 3: every workgroup alike
// found the connection without a usable period and the round carries a size table -- the caller runs rxh_body (every
// thread of the workgroup returns the same value).
// WT: every plan and credit word is stored write-through and acknowledged before a workgroup leaves (grdma_devfn.h:
// xwg_*), for a plan consumed inside the SAME launch; unused since round 4's fused round was retired -- every kernel
// passes false, the plan's consumers are later launches.
// EWT: the entries a workgroup emits for its own records (segments, tile prefix, slices) are stored write-through
// and acknowledged before it arrives, whatever WT says about the rest.  The kernels whose last workgroup runs the general
// planner IN THE SAME LAUNCH when some workgroup declined (k_plan_pair_mw, k_rx_plan_mw) need it: that planner rewrites
// the same slots from index 0, and entries left dirty in another XCD's L2 by a workgroup whose own probe passed would
// be written back over them, or not, in no defined order when the kernel ends.
// ring_wait (may be null): called once by every thread, before the body looks at the ring for the first time -- the
// wire of the round may share the launch (ring_behind_wire); a body that returns 3 has not called it.
// publish (may be null): called by the committing thread with the credit of the drain (credit_promised).
template <bool WT = false, bool EWT = WT, class RingWait = ring_ready_now, class Publish = credit_unpublished>
__device__ __forceinline__ int rxm_body(const grdma_rx_op& op_in, const uint32_t wg, const uint32_t nwg, RingWait* ring_wait = nullptr,
                                        Publish* publish = nullptr) {
  static_assert(sizeof(rx_lds_multi) <= sizeof(rx_lds) && sizeof(rx_lds_multi) <= RXM_SMALL_LDS_BYTES, "the tables fit their LDS");
  rx_lds_multi& M = *reinterpret_cast<rx_lds_multi*>(rx_tables<WT>());
  const grdma_rx_op op = op_in;
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const uint32_t tid = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;
  __shared__ uint32_t s_w[4][RXM_WAVES];
  __shared__ uint32_t s_bad, s_vj, s_first, s_F, s_miss, s_lb;

  // ---- 0. state, preconditions (as rxf_body; nothing of the connection is stored before the last workgroup commits)
  uint8_t* const ring = c->ring;
  const uint64_t cap64 = c->cap;
  const uint64_t head64 = c->head, mh0 = c->moving_head, remain0 = c->remain, leftover0 = c->leftover_cap;
  const uint64_t irs0 = c->internal_read_size;
  const uint64_t hc = c->rx_hist_count;
  const uint32_t P = c->rx_period;
  const uint32_t status = c->status;
  const uint32_t* const gh = c->rx_hist;
  // how far the sender's completed writes reach: the tail a streaming job's Send of the same round computed, or -- an
  // endpoint's drain -- the connection's arrival report (grdma_wire_report), read once
  const bool limited = op.limit_ptr != nullptr || c->wire_limit != 0;
  const uint64_t lim = op.limit_ptr ? __hip_atomic_load(op.limit_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : __hip_atomic_load(&c->wire_recv.wire_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint64_t slice_idx0 = op.append == 1 ? c->rx_slice_idx : 0;
  const uint64_t a_off0 = op.append == 1 ? c->rx_arena_off : 0;
  constexpr int NH = GRDMA_RX_HIST / RXM_THREADS;
  uint32_t hv[NH];
#pragma unroll
  for (int r = 0; r < NH; r++) hv[r] = gh[tid + r * RXM_THREADS];

  bool ok = status == GRDMA_PAIR_CONNECTED && op.raw_cap == 0 && !op.inline_apply &&
            limited && remain0 == 0 && leftover0 <= RXF_MINRD && P != 0 && P <= RXF_PMAX && hc >= P &&
            cap64 <= (1ull << 31) && a_off0 < (1ull << 31);
  uint64_t max_slices = GRDMA_MAX_SLICES;
  {
    const uint64_t room = op.slices_cap > slice_idx0 ? op.slices_cap - slice_idx0 : 0;
    if (room < max_slices) max_slices = room;
    if (op.max_reads < max_slices) max_slices = op.max_reads;
  }
  const uint32_t cap = (uint32_t)cap64, mask = cap - 1u, head = (uint32_t)head64;
  const uint32_t Lr = ((uint32_t)lim - head) & mask;  // ring bytes between my head and the sender's tail
  const bool idle = Lr == 0;
  // (k / P as a multiplication: exact for k * P < 2^32; k <= 4096 + RXM_PFX here, P <= 512)
  const uint32_t invP = P > 1 ? (uint32_t)(0xFFFFFFFFu / P) + 1u : 0u;
  auto divP = [&](uint32_t k) -> uint32_t { return rxm_div(k, P, invP); };
  if (tid == 0) {
    s_bad = 0;
    s_vj = RXM_NONE;
    s_first = RXM_NONE;
    s_F = RXM_NONE;
    s_miss = 0;
    s_lb = 0;
  }
  // ---- table cache, part 1: the tables of step 4 are a function of the pattern's payload sizes, the period and the tile
  // size alone -- what every workgroup of every drain of a periodic stream recomputed (3.9 us of a drain plan's 13-15,
  // profiles/r05_plan_phases.txt).  They are kept behind the connection's history, eight slots chosen by a hash of the
  // pattern as the drain sees it (a drain begins wherever the last one ended), each slot with the payload sizes it was derived from as
  // its key.  The slot's words are requested in step 2; step 4 compares the key with the sizes the probe has read from
  // the ring and takes the tables, or computes them as before.  The LAST workgroup of a drain to arrive writes the slot
  // (every other one has long read it; the next reader is the next launch).
  uint32_t c_hdr[4] = {0, 0, 0, 0}, c_key[2] = {0, 0}, c_sss[2] = {0, 0}, c_qpk[2] = {0, 0}, c_qtl[2] = {0, 0}, c_qby[2] = {0, 0},
           c_qn[2] = {0, 0}, c_tot[4] = {0, 0, 0, 0};
  uint32_t* c_tab = nullptr;
#pragma unroll
  for (int r = 0; r < NH; r++) M.hist[tid + r * RXM_THREADS] = hv[r];
  __syncthreads();
  // `reason` (uniform over the workgroup once set): 0 = going on; the slots of g_rx_fast_drains otherwise
  uint32_t reason = (!ok || idle) ? 1u : 0u;

  // ---- 1. the pattern: the newest P encoded sizes, their prefix sums, where the limit falls in it (as rxf_body)
  uint32_t SP = 0, V = 0;
  if (!reason) {
    const uint32_t j0 = 2 * tid, j1 = j0 + 1;
    const uint32_t pv0 = j0 < P ? M.hist[(uint32_t)((hc - P + j0) % GRDMA_RX_HIST)] : 0;
    const uint32_t pv1 = j1 < P ? M.hist[(uint32_t)((hc - P + j1) % GRDMA_RX_HIST)] : 0;
    uint32_t px, d1, d2, d3, ptot[4];
    rxm_scan4(pv0 + pv1, 0, 0, 0, s_w, &px, &d1, &d2, &d3, ptot);
    SP = ptot[0];
    if (j0 < P) {
      M.pat[j0] = pv0;
      M.pre[j0] = px;
    }
    if (j1 < P) {
      M.pat[j1] = pv1;
      M.pre[j1] = px + pv0;
    }
    if (tid == 0) M.pre[P] = SP;
    const uint32_t q_full = SP ? Lr / SP : 0, rem = SP ? Lr - q_full * SP : 0;
    if (SP != 0) {  // (pre[] is strictly increasing: at most one match)
      if (j0 < P && px == rem) s_vj = j0;
      if (j1 < P && px + pv0 == rem) s_vj = j1;
    }
    __syncthreads();
    const uint32_t vj = s_vj;
    const uint64_t V64 = (uint64_t)q_full * P + vj;
    if (SP == 0 || vj == RXM_NONE || V64 == 0 || V64 > (uint64_t)nwg * RXM_CHUNK) reason = 2;
    else V = (uint32_t)V64;
  }
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap64);
  const uint64_t t_pattern = __builtin_amdgcn_s_memtime();
  // No period, or a pattern that does not end at the sender's tail: known to EVERY workgroup alike before the ring has
  // been looked at and before anyone has arrived -- the caller may try the body that predicts from the Send's own size
  // table instead (grdma_rx_hint.h); nothing has been written, nothing counted.
  if (reason && !idle && op.sizes_in != nullptr) return 3;
  if (ring_wait != nullptr) (*ring_wait)();

  // ---- 2. one round trip: header and footer of my record, header of pattern record `tid`
  const uint32_t i_mine = wg * RXM_CHUNK + tid;
  const bool have = !reason && i_mine < V;
  const uint32_t PV = reason ? 0u : (V < P ? V : P);  // pattern positions this drain holds a record for
  uint32_t xe = 0, ee = 0, ri_mine = 0, n_mine = 0;
  if (!reason) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t ring_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)ring);
    const uint32_t ring_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)ring >> 32));
    const uint64_t ring_u = ((uint64_t)ring_hi << 32) | (uint64_t)ring_lo;
    const uint32_t cap_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)cap);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ring_u, 0, cap_u, 0x00020000);
    if (have) {
      const uint32_t qi = divP(i_mine);
      ri_mine = i_mine - qi * P;
      xe = qi * SP + M.pre[ri_mine];
      ee = M.pat[ri_mine];
    }
    // (clamped, unconditional loads: every lane's four words are in flight together; pattern positions 2 tid, 2 tid + 1)
    const uint32_t pj0 = 2 * tid, pj1 = pj0 + 1;
    const uint32_t o_h = have ? ((head + xe) & mask & ~7u) : 0u;
    const uint32_t o_f = have ? ((head + xe + ee - 8u) & mask & ~7u) : 0u;
    const uint32_t o_p0 = pj0 < PV ? ((head + M.pre[pj0]) & mask & ~7u) : 0u;
    const uint32_t o_p1 = pj1 < PV ? ((head + M.pre[pj1]) & mask & ~7u) : 0u;
    const u32x2 hw = __builtin_amdgcn_raw_buffer_load_b64(rs, o_h, 0, 16 /* sc1 */);
    const u32x2 fw = __builtin_amdgcn_raw_buffer_load_b64(rs, o_f, 0, 16);
    const u32x2 pw0 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_p0, 0, 16);
    const u32x2 pw1 = __builtin_amdgcn_raw_buffer_load_b64(rs, o_p1, 0, 16);
    // (table cache: the slot's words, requested BEHIND the probe's loads -- loads return in order, the probe does not
    //  wait for them -- and in flight under the probe's round trip and step 3)
    {
      // which slot: a hash of eight samples of the pattern AS THIS DRAIN SEES IT (its encoded sizes, rotated to where the
      // drain begins).  Two drains with the same key have the same samples -- a stream of identical messages has as many
      // distinct rotations as its true period allows, however large the detected period P is (a multiple of it) --, and
      // drains whose keys differ but share the samples only cost each other a recomputation.
      uint32_t c_h = P;
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) c_h = c_h * 0x9E3779B1u + M.pat[(k * P) >> 3];
      c_tab = const_cast<uint32_t*>(gh) + GRDMA_RX_HIST + ((c_h * 0x85EBCA6Bu) >> 29) * GRDMA_RX_TAB_WORDS;
      static_assert(GRDMA_RX_TAB_SLOTS == 8 && GRDMA_RX_TAB_STRIDE >= RXF_PMAX + 1 && 2 * RXM_THREADS >= RXF_PMAX, "slot index and layout");
      const uint32_t* const a = c_tab + 4;
#pragma unroll
      for (int k = 0; k < 4; k++) c_hdr[k] = c_tab[k];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const uint32_t pj = 2 * tid + r;  // (< 512: inside every array)
        c_key[r] = a[pj];
        c_sss[r] = a[GRDMA_RX_TAB_STRIDE + pj];
        c_qpk[r] = a[2 * GRDMA_RX_TAB_STRIDE + pj];
        c_qtl[r] = a[3 * GRDMA_RX_TAB_STRIDE + pj];
        c_qby[r] = a[4 * GRDMA_RX_TAB_STRIDE + pj];
        c_qn[r] = a[5 * GRDMA_RX_TAB_STRIDE + pj];
      }
      if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) c_tot[k] = a[(2 + k) * GRDMA_RX_TAB_STRIDE + RXF_PMAX];  // one period's totals
      }
    }
    bool bad = false;
    if (have) {
      const uint64_t h = ((uint64_t)hw.y << 32) | hw.x, f = ((uint64_t)fw.y << 32) | fw.x;
      bad |= !(h != 0 && h <= cap64 - GRDMA_RESERVED && 16u + (uint32_t)round_up8(h) == ee);
      bad |= f != GRDMA_FOOTER;
      n_mine = (uint32_t)h;
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t pj = r ? pj1 : pj0;
      if (pj < P) {
        uint32_t np = RXM_NONE;
        if (pj < PV) {
          const uint64_t h = r ? (((uint64_t)pw1.y << 32) | pw1.x) : (((uint64_t)pw0.y << 32) | pw0.x);
          const uint32_t e = M.pat[pj];
          // (a header that does not fit its slot makes the drain's owner of that record decline; keep the value harmless)
          np = (h != 0 && h <= cap64 - GRDMA_RESERVED && 16u + (uint32_t)round_up8(h) == e) ? (uint32_t)h : e - 16u;
          if (np != (uint32_t)h) bad = true;
        }
        M.npat[pj] = np;
      }
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (have && n_mine != M.npat[ri_mine]) s_bad = 1;  // periodic in the encoded size but not in the payload size
    // the first record that leaves no read open behind it (F), among the first RXM_PFX records
    if (tid < RXM_PFX && tid < V && M.npat[tid - divP(tid) * P] >= RXM_RESET) atomicMin(&s_F, tid);
    __syncthreads();
    if (s_bad) reason = 3;
  }
  const uint64_t t_probe = __builtin_amdgcn_s_memtime();

  // ---- 3. the prefix region: records 0 .. NF - 1 from the read the last drain left open
  const uint32_t s0 = (uint32_t)leftover0;
  const bool odd_open = s0 != 0 && s0 != RXF_MINRD;  // the open read's capacity is not a fresh read's 256
  uint32_t NF = 0, first_done = RXM_NONE;
  uint32_t PS_pk = 0, PS_tl = 0, PS_by = 0;
  if (!reason) {
    const uint32_t F = s_F;
    if (F != RXM_NONE) NF = F + 1;
    else if (V <= RXM_PFX) NF = V;   // no such record in a short drain: all of it is prefix region
    else reason = 4;                  // a long run of small records: not this body's case
  }
  if (!reason) {
    uint32_t s_in = 0, n_t = 0;
    const bool mine = tid < NF;
    if (mine) {
      uint32_t s = s0, rj = 0;
      for (uint32_t j = 0; j < tid; j++) {
        s = rxf_space_after(M.npat[rj], s);
        if (++rj == P) rj = 0;
      }
      s_in = s;
      n_t = M.npat[rj];
      M.as[tid] = (uint16_t)s;
      if (rxf_replay(n_t, s).sl_cnt != 0) atomicMin(&s_first, tid);
    }
    __syncthreads();
    first_done = s_first;  // RXM_NONE: no slice completes in the prefix region (then it is the whole drain)
    uint32_t my_pk = 0, my_tl = 0, my_by = 0;
    if (mine) {
      const uint32_t q = divP(tid), r = tid - q * P;
      const uint32_t x = q * SP + M.pre[r];
      const bool in_first = odd_open && tid <= first_done;
      const rxf_layout L = rxf_lay(n_t, s_in, (head + x + 8u) & mask, cap, in_first ? s0 : RXF_MINRD,
                                   odd_open && tid == first_done, ts);
      my_pk = L.sl_cnt | (L.nsg << 16);
      my_tl = L.ntl;
      my_by = L.bytes;
    }
    uint32_t x_pk, x_tl, x_by, x_d, tot[4];
    rxm_scan4(my_pk, my_tl, my_by, 0, s_w, &x_pk, &x_tl, &x_by, &x_d, tot);
    if (mine) {
      M.apk[tid] = x_pk;
      M.atl[tid] = x_tl;
      M.aby[tid] = x_by;
    }
    PS_pk = tot[0];
    PS_tl = tot[1];
    PS_by = tot[2];
  }

  // ---- 4. the steady pattern: read state in front of every position, what every position takes, prefix sums
  // (table cache, part 2: taken from the slot when its key is this drain's pattern)
  bool c_fresh = false;  // this workgroup computed the tables itself (it writes the slot if it is the committing one)
  bool c_hit = false;
  if (!reason) {
    bool mine_ok = c_hdr[3] == GRDMA_RX_TAB_MAGIC && c_hdr[0] == P && c_hdr[1] == ts;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t pj = 2 * tid + r;
      if (pj < P && c_key[r] != M.npat[pj]) mine_ok = false;
    }
    if (!mine_ok) s_miss = 1;
    __syncthreads();
    c_hit = s_miss == 0;
  }
  if (!reason && c_hit) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t pj = 2 * tid + r;
      if (pj < P) {
        M.sss[pj] = (uint16_t)c_sss[r];
        M.qpk[pj] = c_qpk[r];
        M.qtl[pj] = c_qtl[r];
        M.qby[pj] = c_qby[r];
        M.qn[pj] = c_qn[r];
      }
    }
    if (tid == 0) {
      M.qpk[P] = c_tot[0];
      M.qtl[P] = c_tot[1];
      M.qby[P] = c_tot[2];
      M.qn[P] = c_tot[3];
      if (c_hdr[2] != 0 && V > NF) s_bad = 1;  // (a position beyond the look-back: as the computation below finds it)
    }
    __syncthreads();
    if (s_bad) reason = 4;
  } else if (!reason) {
    c_fresh = true;
    // (thread t: positions 2 t and 2 t + 1)
    uint32_t v_pk[2] = {0, 0}, v_tl[2] = {0, 0}, v_by[2] = {0, 0}, v_n[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t pj = 2 * tid + r;
      if (pj >= P) continue;
      // back to the nearest position that resets the state (cyclic; positions this drain has no record for count
      // as resets: they are only reached from positions whose values are never used)
      uint32_t k = pj, steps = 0;
      for (;;) {
        const uint32_t kp = k == 0 ? P - 1 : k - 1;
        if (M.npat[kp] >= RXM_RESET) break;
        k = kp;
        if (++steps > RXF_LOOKBACK) break;
      }
      if (steps > RXF_LOOKBACK) s_lb = 1;            // (kept with the tables: whether they may be used depends on the drain)
      if (steps > RXF_LOOKBACK && V > NF) s_bad = 1;  // (a drain that is all prefix region does not use these tables)
      uint32_t s = 0;
      for (uint32_t j = 0; j < steps && steps <= RXF_LOOKBACK; j++) {
        s = rxf_space_after(M.npat[k], s);
        if (++k == P) k = 0;
      }
      M.sss[pj] = (uint16_t)s;
      const uint32_t n = M.npat[pj];
      if (n != RXM_NONE) {
        const rxf_layout L = rxf_lay(n, s, 0u, 0x80000000u, RXF_MINRD, false, ts);  // a ring without an end
        v_pk[r] = L.sl_cnt | (L.nsg << 16);
        v_tl[r] = L.ntl;
        v_by[r] = L.bytes;
        v_n[r] = n;
      }
    }
    uint32_t x_pk, x_tl, x_by, x_n, tot[4];
    rxm_scan4(v_pk[0] + v_pk[1], v_tl[0] + v_tl[1], v_by[0] + v_by[1], v_n[0] + v_n[1], s_w, &x_pk, &x_tl, &x_by, &x_n, tot);
    if (2 * tid < P) {
      M.qpk[2 * tid] = x_pk;
      M.qtl[2 * tid] = x_tl;
      M.qby[2 * tid] = x_by;
      M.qn[2 * tid] = x_n;
    }
    if (2 * tid + 1 < P) {
      M.qpk[2 * tid + 1] = x_pk + v_pk[0];
      M.qtl[2 * tid + 1] = x_tl + v_tl[0];
      M.qby[2 * tid + 1] = x_by + v_by[0];
      M.qn[2 * tid + 1] = x_n + v_n[0];
    }
    if (tid == 0) {
      M.qpk[P] = tot[0];
      M.qtl[P] = tot[1];
      M.qby[P] = tot[2];
      M.qn[P] = tot[3];
    }
    __syncthreads();
    if (s_bad) reason = 4;
  }
  const uint64_t t_state = __builtin_amdgcn_s_memtime();

  // ---- 5. the record whose payload crosses the ring end (behind the prefix region), totals, room
  // read state in front of record i, its payload size
  auto state_of = [&](uint32_t i, uint32_t* n_out) -> uint32_t {
    const uint32_t r = i - divP(i) * P;
    *n_out = M.npat[r];
    return i < NF ? (uint32_t)M.as[i] : (uint32_t)M.sss[r];
  };
  uint32_t w_rec = RXM_NONE, d_sg = 0, d_tl = 0;
  // what records [0, i) took, NF <= i <= V
  auto before = [&](uint32_t i, uint32_t* pk, uint32_t* tl, uint32_t* by) {
    uint32_t a_pk, a_tl, a_by, b_pk, b_tl, b_by;
    rxm_cyc(M, P, invP, i, &a_pk, &a_tl, &a_by);
    rxm_cyc(M, P, invP, NF, &b_pk, &b_tl, &b_by);
    const bool past = w_rec != RXM_NONE && i > w_rec;
    *pk = PS_pk + (a_pk - b_pk) + (past ? d_sg << 16 : 0u);
    *tl = PS_tl + (a_tl - b_tl) + (past ? d_tl : 0u);
    *by = PS_by + (a_by - b_by);
  };
  uint32_t tot_pk = 0, tot_tl = 0, tot_by = 0, s_end = 0;
  if (!reason) {
    const uint32_t u = cap - head;  // ring bytes from my head to the ring end
    if (u < Lr) {
      const uint32_t q = u / SP, r = u - q * SP;
      uint32_t lo = 0, hi = P;  // the last position whose prefix is <= r
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (M.pre[mid] <= r) lo = mid; else hi = mid;
      }
      const uint32_t w = q * P + lo;
      if (w >= NF && w < V) {
        const uint32_t n = M.npat[lo], s = M.sss[lo];
        const rxf_layout La = rxf_lay(n, s, (head + q * SP + M.pre[lo] + 8u) & mask, cap, RXF_MINRD, false, ts);
        // (what this position takes in a ring without an end: the difference of two table entries)
        w_rec = w;
        d_sg = La.nsg - ((M.qpk[lo + 1] - M.qpk[lo]) >> 16);
        d_tl = La.ntl - (M.qtl[lo + 1] - M.qtl[lo]);
      }
    }
    if (V > NF) before(V, &tot_pk, &tot_tl, &tot_by);
    else { tot_pk = PS_pk; tot_tl = PS_tl; tot_by = PS_by; }
    uint32_t n_last;
    const uint32_t s_last_in = state_of(V - 1, &n_last);
    s_end = rxf_space_after(n_last, s_last_in);
  }
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
  // ---- my verdict is final: counted in (drain_count_in, above), behind the loads of everything the commit overwrites
  const bool committer = wg == nwg - 1;
  GRDMA_WAIT_LOADS();
  __syncthreads();
  if (tid == 0) drain_count_in(plan, reason);

  // ---- 6. my record: segments, tile prefix, slices (entries beyond any committed count if the drain is declined)
  grdma_slice_out* const out_slices = op.slices + slice_idx0;
  if (!reason && have) {
    uint32_t n, pk, tl, by;
    const uint32_t s_in = state_of(i_mine, &n);
    if (i_mine < NF) {
      pk = M.apk[i_mine];
      tl = M.atl[i_mine];
      by = M.aby[i_mine];
    } else {
      before(i_mine, &pk, &tl, &by);
    }
    uint32_t sl = pk & 0xFFFFu, sg = pk >> 16;
    const bool in_first = odd_open && i_mine <= first_done;
    const rxf_layout L = rxf_lay(n, s_in, (head + xe + 8u) & mask, cap, in_first ? s0 : RXF_MINRD,
                                 odd_open && i_mine == first_done, ts);
    const uint64_t A = a_off0 + by;  // start of the open slice, or of the slice this record begins
    int last_piece = 0;
#pragma unroll
    for (int k = 1; k < 4; k++)
      if (L.len[k]) last_piece = k;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (L.len[k] == 0) continue;
      const uint64_t fl = GRDMA_SEG_ZERO_SRC | (k == 0 ? GRDMA_SEG_TAG_HDR : 0) | (k == last_piece ? GRDMA_SEG_TAG_FTR : 0);
      xwg_put_seg<EWT>(&plan->segs[sg], (uint64_t)op.arena + A + L.dst_rel[k], (uint64_t)(ring + L.off[k]), (uint64_t)L.len[k], fl);
      xwg_st32<EWT>(&plan->tile_prefix[sg], tl);
      sg++;
      tl += rxf_tiles(L.len[k], ts);
    }
    uint64_t sof = A;
    if (L.sl0) {
      xwg_st64<EWT>(&out_slices[sl].off, sof);
      xwg_st64<EWT>(&out_slices[sl].len, (uint64_t)L.sl0);
      sl++;
      sof += rxf_al16(L.sl0);
    }
    if (L.sl1) {
      xwg_st64<EWT>(&out_slices[sl].off, sof);
      xwg_st64<EWT>(&out_slices[sl].len, (uint64_t)L.sl1);
    }
  }
  // ---- (round 6) what the commit needs and no probe has a part in -- the counters it adds to, and the credit of the
  //      drain (pair.cc:276-284: a function of the pattern, the read-state tables and internal_read_size) -- by thread 0
  //      of EVERY workgroup, here, while its entries are on their way to the memory side: through round 5 the one
  //      workgroup that commits fetched the counters and walked the credit loop behind the arrival, 2 us at the tail of
  //      every planner launch (profiles/r06_plan_phases.txt: "commit loads", "credit").  Nothing of this is written
  //      before the last workgroup commits; the counters only ever change there.
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
    auto enc_end = [&](uint32_t i) -> uint64_t {  // ring bytes consumed once record i is finished
      const uint32_t qi = divP(i), ri = i - qi * P;
      return (uint64_t)qi * SP + M.pre[ri] + M.pat[ri];
    };
    const uint64_t T = cap64 / 2, Ctot = Lr;
    uint64_t thr = T - irs0;
    while (Ctot >= thr) {
      // first record whose running consumption (after its last step) reaches thr: enc_end(i) = E(i + 1) with
      // E(k) = (k / P) SP + pre[k mod P] (pre[P] = SP), so the period comes from one division and the position from
      // a search over the pattern's prefix in LDS
      uint32_t lo;
      {
        const uint32_t t32 = (uint32_t)thr;  // (thr <= Ctot < 2^31 here)
        const uint32_t qk = t32 / SP, r = t32 - qk * SP;
        uint32_t a = 0, b = P;  // the smallest j in [1, P] with pre[j] >= r (r > 0), or j = 0 (r == 0)
        if (r == 0) b = 0;
        while (b - a > 1) {
          const uint32_t mid = (a + b) >> 1;
          if (M.pre[mid] >= r) b = mid; else a = mid;
        }
        const uint32_t k = qk * P + b;  // smallest k with E(k) >= thr
        lo = k ? k - 1 : 0;
        if (lo > V - 1) lo = V - 1;
      }
      uint32_t n;
      const uint32_t s_in = state_of(lo, &n);
      const rxf_rec rp = rxf_replay(n, s_in);
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
  }
  const uint64_t t_emit = __builtin_amdgcn_s_memtime();

  // ---- 7. the committer goes on once every verdict is in; the others leave behind their entries' acknowledgement
  if (!committer) {
    if (EWT) GRDMA_WAIT_VMEM();
    __syncthreads();
    if (tid == 0) drain_leave(plan);
    return 0;
  }
  const uint32_t s_any = drain_verdicts(plan, nwg);
  // (table cache, part 3: the slot is written by the committing workgroup, from the tables it computed itself)
  if (c_fresh) {
    uint32_t* const a = c_tab + 4;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint32_t pj = 2 * tid + r;
      if (pj < P) {
        a[pj] = M.npat[pj];
        a[GRDMA_RX_TAB_STRIDE + pj] = (uint32_t)M.sss[pj];
        a[2 * GRDMA_RX_TAB_STRIDE + pj] = M.qpk[pj];
        a[3 * GRDMA_RX_TAB_STRIDE + pj] = M.qtl[pj];
        a[4 * GRDMA_RX_TAB_STRIDE + pj] = M.qby[pj];
        a[5 * GRDMA_RX_TAB_STRIDE + pj] = M.qn[pj];
      }
    }
    if (tid == 0) {
      a[2 * GRDMA_RX_TAB_STRIDE + RXF_PMAX] = M.qpk[P];
      a[3 * GRDMA_RX_TAB_STRIDE + RXF_PMAX] = M.qtl[P];
      a[4 * GRDMA_RX_TAB_STRIDE + RXF_PMAX] = M.qby[P];
      a[5 * GRDMA_RX_TAB_STRIDE + RXF_PMAX] = M.qn[P];
      c_tab[0] = P;
      c_tab[1] = ts;
      c_tab[2] = s_lb;
      c_tab[3] = GRDMA_RX_TAB_MAGIC;
    }
  }
  const uint64_t t_arrived = __builtin_amdgcn_s_memtime();
  if (s_any) {  // (uniform)
    if (tid == 0) {
      // (the slot of this workgroup's own reason if it has one, the probe's otherwise: another workgroup's records)
      atomicAdd(&g_rx_fast_drains[reason ? reason : 3u], 1ull);
      atomicAdd(&g_rx_verdicts[s_any < nwg ? 0 : 1], 1ull);
      if (!idle) res->pad0++;  // (pad1 / pad0: drains of this result block taken / declined with data waiting)
    }
    // (the general planner rewrites entries: everybody's are at the memory side first, mine included)
    if (EWT) GRDMA_WAIT_VMEM();
    if (tid == 0) drain_close(plan, nwg);
    __syncthreads();
    return 2;
  }
  // (the drain is committed as far as its Send is concerned: the promise leaves before the bookkeeping)
  if (tid == 0 && publish != nullptr) (*publish)(credit, credit_head);

  if (tid == 0) atomicAdd(&g_rx_tab_stats[c_hit ? 0 : 1], 1ull);
  // history: the records of this drain become the newest entries (the pattern simply continues)
#pragma unroll
  for (int r = 0; r < NH; r++) {
    const uint32_t back = tid + r * RXM_THREADS;  // the newest GRDMA_RX_HIST records
    if (back < V) {
      const uint32_t i = V - 1 - back;
      c->rx_hist[(uint32_t)((hc + i) % GRDMA_RX_HIST)] = M.pat[i - divP(i) * P];
    }
  }
  // ---- 8. credit (pair.cc:276-284), state, result: thread 0 (as rxf_body, per-record values from the tables)
  if (tid == 0) {
    const uint64_t t_loaded = __builtin_amdgcn_s_memtime() + (o_seq & 0);
    const uint32_t tot_n = divP(V) * M.qn[P] + M.qn[V - divP(V) * P];  // payload bytes of the drain
    const uint64_t Ctot = Lr;
    const uint64_t t_credit = __builtin_amdgcn_s_memtime();
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
    if (op.append) {  // (a streaming job's cursors)
      c->rx_arena_off = a_end;
      c->rx_slice_idx = slice_idx0 + nsl_final;
    }
    c->rx_hist_count = hc + V;
    {
      const uint32_t rl = (V - 1) - divP(V - 1) * P;
      c->rx_h1 = M.pat[rl];
      c->rx_h2 = V >= 2 ? M.pat[rl ? rl - 1 : P - 1] : o_h1;
    }
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
    res->dbg[8] = P;
    res->dbg[9] = 0xFA57;  // this stamp set comes from a steady-state body
    res->dbg[10] = nwg;
    res->dbg[11] = t_arrived - t_begin;
    res->dbg[12] = t_loaded - t_begin;
    res->dbg[13] = t_credit - t_begin;
    res->pad1++;
    res->dbg[1] = __builtin_amdgcn_s_memtime();
    atomicAdd(&g_rx_fast_drains[0], 1ull);
    __hip_atomic_store(&res->seq, op.seq_next ? op.seq_next : o_seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    drain_close(plan, nwg);
  }
  return 1;
}

}  // namespace
#endif  // GRDMA_RX_MULTI_H
