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
// arithmetic on values broadcast with readlane, lane 0 stores segments and results.
#include <hip/hip_runtime.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"

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

  uint64_t rt = tail0, st = 0, nsge = 0, written = 0, zc_bytes = 0, copy_bytes = 0, zc_records = 0;
  uint64_t nrec = 0, nseg = 0, ntiles = 0, staged = 0, splits = 0;
  uint64_t idx = 0, bidx = op.byte_idx;  // cursor behind the last byte accepted
  bool stop = !connected || ring == nullptr || n == 0;
  for (uint64_t base = 0; base < n && !stop; base += 64) {
    const uint64_t mine = base + (uint64_t)lane < n ? base + (uint64_t)lane : n - 1;
    const grdma_sge g = sl[mine];
    for (int j = 0; j < 64 && base + (uint64_t)j < n; j++) {
      if (nsge >= max_sge) { stop = true; break; }                       // loop condition, pair.cc:818
      const uint8_t* ptr = reinterpret_cast<const uint8_t*>(__shfl((uint64_t)g.ptr, j, 64));
      uint64_t len = __shfl(g.len, j, 64);
      const uint64_t skip = (base + (uint64_t)j == 0) ? op.byte_idx : 0;  // :819-824
      ptr += skip;
      len = sat_sub(len, skip);
      const uint64_t recv_free = cap - ((rt + cap - rhead) & mask);       // GetFreeSize, ring_buffer.cc:99-104
      const uint64_t send_free = S - st;
      const bool in_zc = op.zc_base != nullptr && ptr >= op.zc_base && ptr + len <= op.zc_base + op.zc_cap;  // :825-826
      uint64_t pay = len;
      {
        const uint64_t b = writable_of(recv_free);
        if (b < pay) pay = b;
      }
      uint64_t pad = 0;
      if (in_zc) {
        if (pay == 0 || send_free < 3ull * GRDMA_ALIGN || nsge + 4 > max_sge) { stop = true; break; }  // :830-834
        pad = round_up8(pay) - pay;
      } else {
        const uint64_t a = writable_of(send_free);
        if (a < pay) pay = a;
        if (pay == 0) { stop = true; break; }                               // :885-887
      }
      const uint64_t enc = enc_size(pay);
      // scatter-gather entries and where the ring end falls among them (GetWriteRequests splits the
      // entry that crosses it, ring_buffer.cc:271-303).  Offsets from tail0, not wrapped.
      {
        const uint64_t o = tail0 + staged;  // < 2 * cap
        if (in_zc) {
          const uint64_t e0 = o, e1 = o + 8, e2 = e1 + pay, e3 = e2 + pad, e4 = e3 + 8;
          splits += (e0 < cap && e1 > cap) + (e1 < cap && e2 > cap) + (pad && e2 < cap && e3 > cap) + (e3 < cap && e4 > cap);
          nsge += pad ? 4 : 3;
          st += 16 + pad;
          zc_bytes += pay;
          zc_records++;
        } else {
          splits += (o < cap && o + enc > cap);
          nsge += 1;
          st += enc;
          copy_bytes += pay;
        }
      }
      // the record in the peer ring: header at rt, payload behind it (wrapping), tags by the copy waves
      const uint64_t pay_off = (rt + 8) & mask;
      const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (pay << GRDMA_SEG_TAG_LEN_SHIFT);
      if (pay_off + pay > cap) {
        const uint64_t l1 = cap - pay_off;
        if (lane == 0) {
          plan->segs[nseg] = {(uint64_t)(ring + pay_off), (uint64_t)ptr, l1, tagw | GRDMA_SEG_TAG_HDR};
          plan->tile_prefix[nseg] = (uint32_t)ntiles;
          plan->segs[nseg + 1] = {(uint64_t)ring, (uint64_t)(ptr + l1), pay - l1, tagw | GRDMA_SEG_TAG_FTR};
          plan->tile_prefix[nseg + 1] = (uint32_t)(ntiles + ((l1 + TB - 1) >> ts));
        }
        ntiles += ((l1 + TB - 1) >> ts) + ((pay - l1 + TB - 1) >> ts);
        nseg += 2;
      } else {
        if (lane == 0) {
          plan->segs[nseg] = {(uint64_t)(ring + pay_off), (uint64_t)ptr, pay, tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR};
          plan->tile_prefix[nseg] = (uint32_t)ntiles;
        }
        ntiles += (pay + TB - 1) >> ts;
        nseg += 1;
      }
      rt = (rt + enc) & mask;  // NextTail
      staged += enc;
      written += pay;
      nrec++;
      if (pay == len) { idx = base + (uint64_t)j + 1; bidx = 0; }
      else { idx = base + (uint64_t)j; bidx = skip + pay; }
    }
  }

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
