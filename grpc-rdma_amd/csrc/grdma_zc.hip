// PairPollable::SendZerocopy (src/core/lib/ibverbs/pair.cc:793-941) on the device.
//
// The reference posts, per slice that lies inside its registered zero-copy send buffer, four
// scatter-gather entries -- header (staged), the payload WHERE IT LIES, padding and footer
// (staged) -- and per other slice one entry of a record encoded into the staging buffer; the
// NIC then writes the concatenation to the peer ring in <= 2 RDMA WRITEs.  Here the payload of
// EVERY record goes straight from where it lies into the peer ring (k_copy over the plan this
// kernel lays out, record tags written by the copy waves: GRDMA_SEG_TAG_*), which is what the
// scatter-gather list means on a loop-back / xGMI wire; nothing is materialised in the staging
// buffer.  The ARITHMETIC is the reference's, record by record: a zero-copy record is limited by
// the receiver's credit only, consumes 16 + padding bytes of the staging budget and four of the
// max_sge entries (and needs 24 free staging bytes); any other record is priced as Send prices it.
//
// One wavefront per op.  Zero-copy sends are few, large slices (the reference's threshold is a
// message size in KiB), so the pricing loop is the reference's sequential loop as it stands: the
// slice table is pulled 64 entries at a time (one coalesced load), every lane runs the same scalar
// arithmetic on values broadcast with readlane, lane 0 stores segments and results.  The loop body
// (zc_price, csrc/grdma_zc_core.h) is plain integer code that tests/cc/zc_core_host.cc also compiles
// for the host and runs against the oracle.
#include <hip/hip_runtime.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"
#include "grdma_zc_core.h"

namespace {

__global__ __launch_bounds__(64) void k_tx_plan_zc(const grdma_zc_op* ops) {
  const grdma_zc_op op = ops[blockIdx.x];
  const int lane = (int)threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  const uint64_t cap = c->cap, mask = cap - 1, S = c->staging_cap, tail0 = c->remote_tail;
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;  // pair.cc:806-808
  const uint64_t max_sge = c->max_sge;
  uint8_t* const ring = c->peer_ring;
  const uint64_t n = op.nslices;
  const grdma_sge* const sl = op.slices;
  const uint32_t ts = GRDMA_PLAN_TILE_SHIFT(cap);
  const uint64_t TB = 1ull << ts;

  // total bytes offered, pair.cc:810-813
  uint64_t offered = 0;
  for (uint64_t i = (uint64_t)lane; i < n; i += 64) offered += sl[i].len;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) offered += __shfl_xor(offered, d, 64);
  offered = sat_sub(offered, op.byte_idx);

  zc_params P;
  P.cap = cap; P.S = S; P.tail0 = tail0; P.rhead = rhead; P.max_sge = max_sge; P.ring = (uint64_t)ring;
  P.zc_base = (uint64_t)op.zc_base; P.zc_cap = op.zc_cap; P.byte_idx = op.byte_idx; P.ts = ts;
  zc_state Z;
  zc_begin(P, Z);
  bool stop = !connected || ring == nullptr || n == 0;
  for (uint64_t base = 0; base < n && !stop; base += 64) {
    const uint64_t mine = base + (uint64_t)lane < n ? base + (uint64_t)lane : n - 1;
    const grdma_sge g = sl[mine];
    for (int j = 0; j < 64 && base + (uint64_t)j < n; j++) {
      const uint64_t ptr = __shfl((uint64_t)g.ptr, j, 64);
      const uint64_t len = __shfl(g.len, j, 64);
      const uint64_t seg_at = Z.nseg;
      const zc_record r = zc_price(P, Z, base + (uint64_t)j, ptr, len);  // (every lane: the same scalar arithmetic)
      if (r.stop) { stop = true; break; }
      if (lane == 0) {
        plan->segs[seg_at] = r.seg[0];
        plan->tile_prefix[seg_at] = r.tile0[0];
        if (r.nsegs == 2) {
          plan->segs[seg_at + 1] = r.seg[1];
          plan->tile_prefix[seg_at + 1] = r.tile0[1];
        }
      }
    }
  }
  const uint64_t rt = Z.rt, st = Z.st, nsge = Z.nsge, splits = Z.splits, written = Z.written, zc_bytes = Z.zc_bytes,
                 copy_bytes = Z.copy_bytes, zc_records = Z.zc_records, nrec = Z.nrec, nseg = Z.nseg, ntiles = Z.ntiles,
                 staged = Z.staged, idx = Z.idx, bidx = Z.bidx;

  if (lane == 0) {
    plan->nsegs = (uint32_t)nseg;
    plan->ntiles = (uint32_t)ntiles;
    plan->tile_bytes = (uint32_t)TB;
    plan->tile_prefix[nseg] = (uint32_t)ntiles;
    plan->bytes = written;
    plan->tag_base = (uint64_t)ring;
    plan->tag_mask = mask;
    grdma_tx_result* r = op.result;
    // the <= 2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    if (staged > 0) {
      const uint64_t seg1 = staged < cap - tail0 ? staged : cap - tail0;
      r->wr_off[0] = tail0;
      r->wr_len[0] = seg1;
      r->wr_count = 1;
      if (tail0 + staged >= cap) {
        r->wr_off[1] = 0;
        r->wr_len[1] = staged - seg1;
        r->wr_count = 2;
      }
    }
    const uint64_t o_written = c->total_written, o_records = c->tx_records, o_rounds = c->tx_rounds;
    if (connected) {
      c->remote_tail = rt;
      c->partial_write = written < offered ? 1 : 0;  // :908
      c->total_written = o_written + written;
      c->tx_records = o_records + nrec;
      if (nrec) c->tx_rounds = o_rounds + 1;
    }
    r->sent = written;
    r->records = nrec;
    r->staged = staged;
    r->partial = connected ? (written < offered ? 1 : 0) : c->partial_write;
    r->new_remote_tail = rt;
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = idx >= n ? 1 : 0;
    r->dbg[0] = zc_bytes;
    r->dbg[1] = copy_bytes;
    r->dbg[2] = nsge + splits;
    r->dbg[3] = zc_records;
    r->dbg[4] = st;  // staging bytes the reference would have used
    __hip_atomic_store(&r->seq, r->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_plan_zc(const grdma_zc_op* d_ops, uint32_t nops,
                                                                                  hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_plan_zc, dim3(nops), dim3(64), 0, s, d_ops);
  return hipGetLastError();
}
