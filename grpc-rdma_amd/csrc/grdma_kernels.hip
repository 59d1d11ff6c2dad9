// CDNA4 (gfx950) kernels of the ring-buffer endpoint data plane.
//
// The reference moves every byte with CPU memcpy/memset inside
// PairPollable::Send (src/core/lib/ibverbs/pair.cc:645-734) and
// RingBufferPollable::Read (src/core/lib/ibverbs/ring_buffer.cc:122-191).  Here
// the same protocol is split the way the hardware wants it:
//
//   k_tx_plan   one workgroup per connection: the head/tail credit arithmetic of
//               Send() as a block-wide prefix scan (all records priced at once,
//               first short record found with an LDS atomic-min), record
//               header/footer tags, the rdma_flush slice cursor, the ≤2 wire
//               work requests.  Emits a list of byte-copy segments.
//   k_copy      the only kernel that touches payload: every wave takes 4 KiB
//               tiles of the segment list; 16-byte destination-aligned stores,
//               source realigned in registers (two aligned 16-byte loads + a
//               funnel shift), so arbitrary grpc_slice alignment costs no
//               extra HBM transactions.  Used for slice gather (TX), the
//               loop-back wire, and ring->slice scatter (RX).
//   k_rx_plan   one wave per connection: walks the record chain from head_
//               (header tag + footer tag = message-ready test of
//               GetReadableSize, ring_buffer.cc:67-97) 64 speculative probes per
//               memory round trip, replays the endpoint_read loop of
//               rdma_bp_posix.cc:180-326 to decide slice boundaries, does the
//               credit accounting of Recv() (pair.cc:264-286), clears the tags.
//   k_rx_apply  K4: scatters the payload to the slices, clears it behind itself
//               (reader zero-fill invariant, ring_buffer.cc:146,160,164,173-180);
//               the last workgroup posts the 16-byte status report.
//   k_poll      K3 batched: one lane per connection, 64 connections per wave,
//               __ballot() of the ready set (HasMessage / GetReadableSize).
//
// No MFMA anywhere: this is HBM-bound byte shuffling.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"
#include "grdma_tx_body.h"
#include "grdma_tx_fast.h"

#ifndef GRDMA_COPY_CONTIG
#define GRDMA_COPY_CONTIG true
#endif
#ifndef GRDMA_APPLY_CONTIG
#define GRDMA_APPLY_CONTIG true
#endif

namespace {

// ----------------------------------------------------------------------------
// k_tx_plan: see grdma_tx_body.h
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(PLAN_THREADS) void k_tx_plan(const grdma_tx_op* ops) {
  tx_plan_body(ops[blockIdx.x]);
}

// The send plan of a streaming job's round: priced from the index of the slice buffer first (grdma_tx_fast.h); what
// that body declines -- nothing has been written then -- goes through the general planner in the same launch.
// (launched in the index body's shape, 1024 threads; the general planner is a 256-thread body: waves 4-15 leave)
__global__ __launch_bounds__(TXB_THREADS) TXB_KERNEL_ATTR void k_tx_plan_job(const grdma_tx_op* ops, const grdma_txf_ctl* ctls) {
  if (txf_body(ops[blockIdx.x], &ctls[blockIdx.x])) return;  // (uniform)
  if (threadIdx.x >= PLAN_THREADS) return;
  tx_plan_body(ops[blockIdx.x]);
  if (threadIdx.x == 0) ops[blockIdx.x].result->dbg[9] = 0;  // (not priced from the index)
}

// k_tx_plan_seq: gridDim.y Sends of the SAME connection back to back in one launch (a sender that
// runs ahead of its reader: rdma_flush retried before the peer has read).  ops[k * gridDim.x + link]
// is Send k of connection `link`; every Send has its own plans, staging buffer and result block and
// continues from the cursor and ring tail the one before left in the connection block.  Between two
// Sends the workgroup makes its own stores visible to itself (agent-scope fence around a barrier:
// the next body reads the connection block the last one wrote).
__device__ __attribute__((noinline)) void tx_plan_seq_call(const grdma_tx_op* op) { tx_plan_body(*op); }
__global__ __launch_bounds__(PLAN_THREADS) void k_tx_plan_seq(const grdma_tx_op* ops) {
  if (blockIdx.y != 0) return;  // (the grid's y extent only carries the burst length)
  const uint32_t n = gridDim.x, burst = gridDim.y;
  const grdma_tx_op* mine = ops + blockIdx.x;
  const grdma_conn* c = mine[0].conn;
  uint32_t k0 = 0;
  if (c->max_sge <= 64 && c->cap <= (1ull << 30)) {
    // the Sends are small: one wavefront prices the whole burst (tx_burst_wave).
    // A first Send that resets the cursor offers the whole slice table: the wave sums a table of up to 4096 entries
    // itself (a write of a few hundred slices then costs one short wave instead of the block-wide planner);
    // anything else that is not a continuation takes the block-wide plan.
    const uint32_t uc0 = mine[0].use_cursor;
    const bool reset = (uc0 == 2 || uc0 == 3) && mine[0].nslices <= 4096;
    if (mine[0].use_cursor != 1 && !reset) {
      tx_plan_seq_call(&mine[0]);
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
      __syncthreads();
      k0 = 1;
    }
    if (reset) {
      // The ops and the slice table of an asynchronous endpoint's write live in pinned HOST memory: the wave below
      // would fetch them Send by Send, two dependent PCIe round trips each.  The whole workgroup pulls them into LDS
      // in ONE round trip instead (every load in flight together), and the wave prices the burst from there.
      constexpr uint32_t kOps = 64, kSl = 1024, kWords = sizeof(grdma_tx_op) / 8;
      static_assert(sizeof(grdma_tx_op) % 8 == 0 && offsetof(grdma_tx_op, slices) == 8, "op layout");
      __shared__ __attribute__((aligned(16))) uint64_t s_ops[kOps * kWords];
      __shared__ __attribute__((aligned(16))) grdma_sge s_sl[kSl];
      const uint64_t ns = mine[0].nslices;
      // a burst queued behind another write runs only if that write's last Send took everything it was offered
      __shared__ uint32_t s_gate;
      if (threadIdx.x == 0)
        s_gate = uc0 == 3 ? (reinterpret_cast<const grdma_tx_result*>(mine[0].byte_idx)->done == 1 ? 1u : 0u) : 1u;
      if (burst <= kOps && ns <= kSl) {
        const grdma_sge* const gsl = mine[0].slices;
        for (uint32_t w = threadIdx.x; w < burst * kWords; w += PLAN_THREADS) {
          const uint32_t k = w / kWords, f = w - k * kWords;
          s_ops[w] = reinterpret_cast<const uint64_t*>(&mine[(size_t)k * n])[f];
        }
        for (uint64_t i = threadIdx.x; i < ns; i += PLAN_THREADS) s_sl[i] = gsl[i];
        __syncthreads();
        // (every Send of a write walks the same table)
        if (threadIdx.x < burst) s_ops[threadIdx.x * kWords + 1] = (uint64_t)(uintptr_t)&s_sl[0];
        __syncthreads();
        if (threadIdx.x < 64) {
          if (s_gate) tx_burst_wave(reinterpret_cast<const grdma_tx_op*>(s_ops), 1, burst, (int)threadIdx.x, true);
          else tx_burst_skip(reinterpret_cast<const grdma_tx_op*>(s_ops), 1, burst, (int)threadIdx.x);
        }
        return;
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        if (s_gate) tx_burst_wave(mine, n, burst, (int)threadIdx.x, true);
        else tx_burst_skip(mine, n, burst, (int)threadIdx.x);
      }
      return;
    }
    if (threadIdx.x < 64 && k0 < burst) tx_burst_wave(mine + (size_t)k0 * n, n, burst - k0, (int)threadIdx.x);
    return;
  }
  for (uint32_t k = 0; k < burst; k++) {
    tx_plan_seq_call(&mine[(size_t)k * n]);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------
// k_copy: segment-list byte mover (slice gather, wire, ring scatter); the tile
// machinery lives in grdma_devfn.h so the fused small-message paths share it
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(COPY_THREADS) void k_copy(const grdma_plan* const* plans) {
  const grdma_plan* plan = plans[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  run_plan<256, GRDMA_COPY_CONTIG>(plan, wave, nwaves, lane);
}

// ----------------------------------------------------------------------------
// k_rx_apply: K4 in one launch -- copy the payload out, clear it behind, and let
// the last workgroup post the credit (status report) once every byte is free.
// ----------------------------------------------------------------------------
// (bx of gx: this workgroup's place among the workgroups that move the plan -- the launch's own x by default)
__device__ __forceinline__ void rx_apply_body(const grdma_rx_op op, const uint32_t bx = blockIdx.x, const uint32_t gx_in = gridDim.x) {
  const int lane = threadIdx.x & 63;
  // A plan with one tile per segment is moved by waves that take segment PAIRS (run_plan): the workgroups beyond the
  // last pair have nothing to move -- the fifth round of a step is 520 segments under a grid sized for 8190, a round at
  // the reference's default knobs 260 -- and they do not count in either: hundreds of arrivals at one counter were
  // ~6 us of the launch that moves the fifth round's 12 MB (value +1.4 %, six alternations:
  // profiles/r05_copy_launch_interleave.txt, "active").
  uint32_t gx = gx_in;
  {
    const uint32_t nsegs = op.plan->nsegs, ntiles = op.plan->ntiles;
    if (ntiles == nsegs) {
      const uint32_t need = (((ntiles + 1) >> 1) + (COPY_THREADS / 64) - 1) / (COPY_THREADS / 64);  // workgroups with a pair
      const uint32_t active = need < 1 ? 1u : (need < gx ? need : gx);  // (an empty plan: one workgroup posts the commit)
      if (bx >= active) return;
      gx = active;
    }
  }
  const uint32_t wave = (bx * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gx * COPY_THREADS) >> 6;
  run_plan<256, GRDMA_APPLY_CONTIG>(op.plan, wave, nwaves, lane);
  // arrival: my stores have been issued and acknowledged (vmcnt(0)); count in,
  // the last workgroup publishes.  Consumers on this device run in later
  // kernels of the stream (a kernel boundary makes the writes visible); a ring
  // registered for a NIC is uncached memory, where acknowledged stores are
  // already at their destination -- so no L2 write-back fence is paid here.
  __shared__ unsigned int s_last;
  GRDMA_WAIT_VMEM();
  __syncthreads();
  if (threadIdx.x == 0) {
    // Relaxed on purpose.  An acq_rel arrival makes every workgroup write its L2 back (agent
    // scope spans eight XCDs with an L2 each): measured 95 us per launch instead of 30.  Who reads
    // what the other workgroups wrote?  (a) kernels later in the stream -- the kernel boundary
    // publishes it; (b) a peer in another process, after it saw the credit -- its ring is
    // allocated fine-grained (GRDMA_RING_FINE_GRAINED: stores are write-through, acknowledged =
    // at memory), and every workgroup waited for its acknowledgements (vmcnt(0)) before arriving.
    unsigned int prev = __hip_atomic_fetch_add(&op.plan->blocks_done, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == gx - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    grdma_rx_result* res = op.result;
    if (res->credit_sent) {
      grdma_status_report* ps = op.conn->peer_status;
      if (ps != nullptr)
        __hip_atomic_store(&ps->remote_head, res->credit_head, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      // the same report into the sender's host-visible state line (GetWritableSize() there is a host load)
      grdma_hostline* pl = op.conn->peer_line;
      if (pl != nullptr)
        __hip_atomic_store(&pl->remote_head, res->credit_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&res->commit_seq, res->commit_seq + 1, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(COPY_THREADS) void k_rx_apply(const grdma_rx_op* ops) { rx_apply_body(ops[blockIdx.y]); }

// k_rx_apply_gather: the scatter of round t and the GATHER of round t + 1 in one launch (gridDim.y = 2 x links: y < links
// scatters link y, y >= links moves the gather plan of link y - links).  Both are ready when the planner pair of round t
// has run -- the drain plan for the scatter, the next send plan for the gather, whose staging buffer the wire of round
// t - 1 has left -- and neither touches what the other does; two copy kernels back to back pay a kernel boundary and a
// ramp each, and the tail of the first leaves most of the chip idle.  One launch less per round.
//
// Which workgroup does what (round 5).  The grid is (gx, 2 x links) and workgroups are dispatched x first: taken as
// "y < links scatters" the whole scatter (1 byte read, 2 written per payload byte) was resident before the first
// gather workgroup (1 read, 1 written) started, and a launch was a write-heavy phase followed by a balanced one.  Now
// the 2 gx workgroups of a link alternate between the two plans in GROUPS OF EIGHT in dispatch order -- eight consecutive
// workgroups are one per XCD, so every XCD, every L2 and every moment of the launch sees both mixes: 60.2 -> 57.7 us per
// launch on one box, 59.3 -> 56.9 on another (profiles/r05_copy_launch_interleave.txt; alternating one by one, which
// gives the even XCDs the scatter and the odd ones the gather, gains a third of that; a 3 : 2 split by bytes nothing).
// Each plan's workgroups know their own index and count (rx_apply_body's arrival counter, run_plan's grid stride).
__global__ __launch_bounds__(COPY_THREADS) void k_rx_apply_gather(const grdma_rx_op* ops, const grdma_plan* const* gplans) {
  const uint32_t gx = gridDim.x, per = 2 * gx;
  const uint32_t F = blockIdx.y * gx + blockIdx.x, link = F / per, f = F - link * per;
  uint32_t role, idx, cnt;
  if (gx >= 8) {
    const uint32_t g = f >> 3;
    const uint32_t D = per >> 4, rem = per & 15u;  // whole double groups; what is left of the last one
    role = g & 1u;
    idx = ((g >> 1) << 3) + (f & 7u);
    cnt = role == 0 ? D * 8 + (rem < 8 ? rem : 8u) : D * 8 + (rem > 8 ? rem - 8 : 0u);
  } else {  // (a launch of a few workgroups: one by one, so that each plan has at least one)
    role = f & 1u;
    idx = f >> 1;
    cnt = gx;
  }
  if (role == 0) {
    rx_apply_body(ops[link], idx, cnt);
    return;
  }
  const int lane = threadIdx.x & 63;
  run_plan<256, GRDMA_COPY_CONTIG>(gplans[link], (idx * COPY_THREADS + threadIdx.x) >> 6, (cnt * COPY_THREADS) >> 6, lane);
}

// ----------------------------------------------------------------------------
// k_tx_commit: runs behind the wire kernel of a Send (a kernel boundary: every byte of the Send is in
// the peer ring) and reports the arrival -- grdma_wire_report into the peer's connection block, the
// same into the peer's host line when it lives in this process, the sender's own line (remote_tail_,
// partial_write_, and the sequence number the host waits for: the caller's slices are free).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_tx_commit(grdma_conn* const* conns, const uint64_t* seqs) {
  if (threadIdx.x != 0) return;
  grdma_conn* c = conns[blockIdx.x];
  tx_publish(c, c->remote_tail, c->partial_write, seqs ? seqs[blockIdx.x] : 0);
}

__global__ __launch_bounds__(64) void k_tx_commit1(grdma_conn* c, uint64_t seq) {  // one pair, arguments by value
  if (threadIdx.x == 0) tx_publish(c, c->remote_tail, c->partial_write, seq);
}

// k_rx_commit1: runs behind k_rx_apply of an asynchronous drain whose slices land in pinned HOST memory.  The
// scatter's workgroups write over PCIe from eight XCDs; that each of them has seen its stores acknowledged before
// it counted in does not order those posted writes before the last workgroup's flag as the HOST sees them.  A
// kernel boundary does (the end-of-kernel release the runtime's own completion signals rely on), so the word the
// host polls is written by this kernel.
__global__ __launch_bounds__(64) void k_rx_commit1(grdma_conn* c, uint64_t seq) {
  if (threadIdx.x == 0 && c->line != nullptr)
    __hip_atomic_store(&c->line->rx_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_rx_idle: an endpoint read that found nothing and never reached the device (the host saw "no message" in the pair's
// state line).  rdma_continue_read had allocated its 256-byte slice by then, and that slice stays in the incoming buffer
// for the next edge (rdma_bp_posix.cc:283-287, 306-326): the read state of the connection says so, exactly as a drain
// that ended in a would-block leaves it.
__global__ __launch_bounds__(64) void k_rx_idle(grdma_conn* c) {
  if (threadIdx.x == 0 && c->remain == 0 && c->leftover_cap == 0) c->leftover_cap = GRDMA_MIN_READ_SLICE;
}

struct ring_probe {
  uint64_t n;      // header value at `pos`
  bool ready;      // header valid and footer tag present
};

__device__ __forceinline__ ring_probe probe_record(const uint8_t* ring, uint64_t cap,
                                                   uint64_t pos) {
  // GetReadableSize, ring_buffer.cc:67-97
  ring_probe r;
  r.n = ld_tag(ring + pos);
  r.ready = false;
  if (r.n == 0 || r.n > cap - GRDMA_RESERVED) return r;  // empty / torn header
  uint64_t f = (pos + 8 + round_up8(r.n)) & (cap - 1);
  r.ready = ld_tag(ring + f) == GRDMA_FOOTER;
  return r;
}

// ----------------------------------------------------------------------------
// k_poll: batched message-ready detection (K3), one lane per connection
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_poll(grdma_conn* const* conns, uint32_t nconns,
                                             uint64_t* readable_out, uint64_t* ready_mask,
                                             uint64_t* has_msg_mask, uint64_t* trigger_mask) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  uint64_t readable = 0;
  bool has = false, trigger = false;
  if (i < nconns) {
    const grdma_conn* c = conns[i];
    if (c->remain > 0) {  // HasMessage / GetReadableSize fast path
      readable = c->remain;
      has = true;
    } else {
      ring_probe pr = probe_record(c->ring, c->cap, c->head);
      if (c->wire_limit) {  // (a record the sender has not reported as landed is not there yet)
        const uint64_t wt = __hip_atomic_load(&c->wire_recv.wire_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t room = (wt - c->head) & (c->cap - 1);
        if (pr.n != 0 && 16 + round_up8(pr.n) > room) pr.n = 0, pr.ready = false;
      }
      has = pr.n > 0;  // HasMessage, ring_buffer.cc:56-65: header only
      readable = pr.ready ? pr.n : 0;
    }
    readable_out[i] = readable;
    // what makes Poller::begin_polling kick the pair's wakeup fd, poller.cc:80-98
    uint32_t st = c->status;
    // get_status(), pair.cc:349-356: a connected pair whose peer announced its exit is half closed
    if (st == GRDMA_PAIR_CONNECTED && c->status_recv.peer_exit == 1) st = GRDMA_PAIR_HALF_CLOSED;
    trigger = st == GRDMA_PAIR_CONNECTED ? (has || c->partial_write != 0)
                                         : (st == GRDMA_PAIR_HALF_CLOSED || st == GRDMA_PAIR_ERROR);
    // refresh pass: a peer in another process cannot reach my host line, so what it wrote into my
    // connection block (arrival report, credit report, peer_exit) is copied there
    grdma_hostline* ln = c->line;
    if (ln != nullptr && (c->line_remote || !c->wire_limit)) {
      // (an ordered wire sends no arrival report: the probe above stands in for it -- a value that differs
      // from every head while a header is there, the head itself while none is)
      const uint64_t wt = c->wire_limit
                              ? __hip_atomic_load(&c->wire_recv.wire_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                              : (has ? (c->head | (1ull << 63)) : c->head);
      const uint64_t rh = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const int32_t px = __hip_atomic_load(&c->status_recv.peer_exit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ln->wire_tail, wt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ln->remote_head, rh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ln->peer_exit, px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ln->refresh_seq, ln->refresh_seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // wavefront ballots: 64 connections -> 64-bit words
  const uint64_t m_ready = __ballot(readable > 0);
  const uint64_t m_has = __ballot(has);
  const uint64_t m_trig = __ballot(trigger);
  if (threadIdx.x == 0) {
    ready_mask[blockIdx.x] = m_ready;
    has_msg_mask[blockIdx.x] = m_has;
    if (trigger_mask) trigger_mask[blockIdx.x] = m_trig;
  }
}

}  // namespace

// ------------------------------------------------------------------ launchers
extern "C" {

__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_plan(const grdma_tx_op* d_ops, uint32_t nops, hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_plan, dim3(nops), dim3(PLAN_THREADS), 0, s, d_ops);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_commit(grdma_conn* const* d_conns, const uint64_t* d_seqs, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_commit, dim3(n), dim3(64), 0, s, d_conns, d_seqs);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_commit1(grdma_conn* d_conn, uint64_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_tx_commit1, dim3(1), dim3(64), 0, s, d_conn, seq);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_idle(grdma_conn* d_conn, hipStream_t s) {
  hipLaunchKernelGGL(k_rx_idle, dim3(1), dim3(64), 0, s, d_conn);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_commit1(grdma_conn* d_conn, uint64_t seq, hipStream_t s) {
  hipLaunchKernelGGL(k_rx_commit1, dim3(1), dim3(64), 0, s, d_conn, seq);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_plan_seq(const grdma_tx_op* d_ops, uint32_t nlinks, uint32_t burst, hipStream_t s) {
  if (nlinks == 0 || burst == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_plan_seq, dim3(nlinks, burst), dim3(PLAN_THREADS), 0, s, d_ops);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_copy(const grdma_plan* const* d_plans, uint32_t nplans,
                             uint32_t blocks_per_plan, hipStream_t s) {
  if (nplans == 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy, dim3(blocks_per_plan, nplans), dim3(COPY_THREADS), 0, s, d_plans);
  return hipGetLastError();
}

__attribute__((visibility("hidden"))) uint32_t grdma_tx_plan_job_threads(void) { return TXB_THREADS; }
__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_plan_job(const grdma_tx_op* d_ops, const grdma_txf_ctl* d_ctls, uint32_t nops,
                                                                   hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_plan_job, dim3(nops), dim3(TXB_THREADS), 0, s, d_ops, d_ctls);
  return hipGetLastError();
}
// diagnostics: Sends of streaming jobs planned by txf_body [0], left to the general planner [1]
int grdma_tx_fast_sends_pair(uint64_t out[2]);  // (grdma_rx_plan.hip: the Sends planned inside the planner-pair launches)
int grdma_tx_fast_sends(uint64_t out[2]) {
  unsigned long long v[2] = {0, 0};
  uint64_t w[2] = {0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_tx_fast_sends), sizeof(v)) != hipSuccess) return -1;
  if (grdma_tx_fast_sends_pair(w) != 0) return -1;
  out[0] = v[0] + w[0];
  out[1] = v[1] + w[1];
  return 0;
}

__attribute__((visibility("hidden"))) hipError_t grdma_launch_rx_apply(const grdma_rx_op* d_ops, uint32_t nops, uint32_t blocks_per_op,
                                 hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_apply, dim3(blocks_per_op, nops), dim3(COPY_THREADS), 0, s, d_ops);
  return hipGetLastError();
}

// Kernel entry points for explicitly built HIP graphs (hipGraphAddKernelNode): the job
// graph is assembled node by node instead of being recorded from streams.
__attribute__((visibility("hidden"))) const void* grdma_kernel_fn(int which) {
  switch (which) {
    case 0: return reinterpret_cast<const void*>(&k_tx_plan);
    case 1: return reinterpret_cast<const void*>(&k_copy);
    case 3: return reinterpret_cast<const void*>(&k_rx_apply);
    case 4: return reinterpret_cast<const void*>(&k_tx_plan_seq);
    case 5: return reinterpret_cast<const void*>(&k_tx_commit);
    case 6: return reinterpret_cast<const void*>(&k_tx_plan_job);
    case 8: return reinterpret_cast<const void*>(&k_rx_apply_gather);
    default: return nullptr;
  }
}
// Workgroups of the copy kernels that are resident at once on the current device: the
// grid is capped there (the tile loop is grid-strided), so that no second, partial wave
// of workgroups trails the first.
__attribute__((visibility("hidden"))) uint32_t grdma_copy_resident_blocks(void) {
  int dev = 0, cus = 0, a = 0, b = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1024;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return 1024;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_copy, COPY_THREADS, 0) != hipSuccess) a = 4;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_rx_apply, COPY_THREADS, 0) != hipSuccess) b = 4;
  const int per_cu = a < b ? a : b;
  return (uint32_t)((per_cu > 0 ? per_cu : 1) * cus);
}

__attribute__((visibility("hidden"))) uint32_t grdma_kernel_threads(int which) { return (which == 0 || which == 2 || which == 4) ? PLAN_THREADS : COPY_THREADS; }

__attribute__((visibility("hidden"))) hipError_t grdma_launch_poll(grdma_conn* const* d_conns, uint32_t nconns, uint64_t* d_readable,
                             uint64_t* d_ready_mask, uint64_t* d_has_mask, uint64_t* d_trigger_mask,
                             hipStream_t s) {
  if (nconns == 0) return hipSuccess;
  hipLaunchKernelGGL(k_poll, dim3((nconns + 63) / 64), dim3(64), 0, s, d_conns, nconns,
                     d_readable, d_ready_mask, d_has_mask, d_trigger_mask);
  return hipGetLastError();
}

}  // extern "C"
