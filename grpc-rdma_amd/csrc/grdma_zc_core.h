// Record pricing of PairPollable::SendZerocopy (src/core/lib/ibverbs/pair.cc:793-941): plain integer code
// without wave intrinsics, shared by k_tx_plan_zc (csrc/grdma_zc.hip) and by the host check in
// tests/cc/zc_core_host.cc, which runs it against the CPU oracle's orc_pair_send_zerocopy.
//
// zc_price takes the next slice of the list and the running state of the call and decides, as the
// reference's loop body does, whether the loop ends here (no credit, no staging budget, no entries left)
// or a record goes out; for a record it yields the one or two gather segments that carry the payload
// from where it lies into the peer ring (the record tags ride on the segments: GRDMA_SEG_TAG_*).
#pragma once
#include <stdint.h>

#include "grdma_dev.h"

#if defined(__HIPCC__)
#define GRDMA_ZC_HD __host__ __device__ __forceinline__
#else
#define GRDMA_ZC_HD inline
#endif

struct zc_params {
  uint64_t cap;       // ring size (power of two)
  uint64_t S;         // staging buffer size (send_buffers_[kDataBuffer])
  uint64_t tail0;     // remote_tail_ when the call starts
  uint64_t rhead;     // get_remote_head()
  uint64_t max_sge;   // max_sge_num_
  uint64_t ring;      // address of the peer ring
  uint64_t zc_base;   // address of the zero-copy buffer (0 = none)
  uint64_t zc_cap;
  uint64_t byte_idx;  // bytes of the first slice already sent
  uint32_t ts;        // log2 of the plan's tile size
};

struct zc_state {
  uint64_t rt;          // remote_tail as the records are appended
  uint64_t st;          // send_buf_tail: staging bytes the reference would have used
  uint64_t nsge;        // entries of sg_list_
  uint64_t splits;      // 1 once an entry crosses the ring end (GetWriteRequests splits it in two)
  uint64_t written, zc_bytes, copy_bytes, zc_records;
  uint64_t nrec, nseg, ntiles, staged;
  uint64_t idx, bidx;   // cursor behind the last byte accepted: slice index, byte offset in it
};

struct zc_record {
  uint32_t stop;        // 1: the loop ends before this slice
  uint32_t nsegs;       // 1 or 2 (the payload wraps around the ring end)
  grdma_seg seg[2];
  uint32_t tile0[2];    // tile_prefix entries of the segments
};

GRDMA_ZC_HD uint64_t zc_round_up8(uint64_t v) { return (v + 7ull) & ~7ull; }
GRDMA_ZC_HD uint64_t zc_writable(uint64_t space) {  // CalculateWritableSize, ring_buffer.h:185-189
  return space > GRDMA_RESERVED ? ((space - GRDMA_RESERVED) & ~7ull) : 0ull;
}

GRDMA_ZC_HD void zc_begin(const zc_params& P, zc_state& s) {
  s.rt = P.tail0;
  s.st = s.nsge = s.splits = s.written = s.zc_bytes = s.copy_bytes = s.zc_records = 0;
  s.nrec = s.nseg = s.ntiles = s.staged = 0;
  s.idx = 0;
  s.bidx = P.byte_idx;
}

// slice `i` of the list: {ptr, len} as the caller passed it
GRDMA_ZC_HD zc_record zc_price(const zc_params& P, zc_state& s, uint64_t i, uint64_t ptr, uint64_t len) {
  zc_record r;
  r.stop = 1;
  r.nsegs = 0;
  r.tile0[0] = r.tile0[1] = 0;
  r.seg[0].dst = r.seg[0].src = r.seg[0].len = r.seg[0].flags = 0;
  r.seg[1] = r.seg[0];
  if (s.nsge >= P.max_sge) return r;                                   // loop condition, pair.cc:818
  const uint64_t mask = P.cap - 1;
  const uint64_t skip = i == 0 ? P.byte_idx : 0;                       // :819-824
  ptr += skip;
  len = len > skip ? len - skip : 0;
  const uint64_t recv_free = P.cap - ((s.rt + P.cap - P.rhead) & mask);  // GetFreeSize, ring_buffer.cc:99-104
  const uint64_t send_free = P.S - s.st;
  const bool in_zc = P.zc_base != 0 && ptr >= P.zc_base && ptr + len <= P.zc_base + P.zc_cap;  // :825-826
  uint64_t pay = len;
  {
    const uint64_t b = zc_writable(recv_free);
    if (b < pay) pay = b;
  }
  uint64_t pad = 0;
  if (in_zc) {
    if (pay == 0 || send_free < 3ull * GRDMA_ALIGN || s.nsge + 4 > P.max_sge) return r;  // :830-834
    pad = zc_round_up8(pay) - pay;
  } else {
    const uint64_t a = zc_writable(send_free);
    if (a < pay) pay = a;
    if (pay == 0) return r;                                             // :885-887
  }
  const uint64_t enc = 16 + zc_round_up8(pay);
  // scatter-gather entries, and whether the ring end falls strictly inside one of them (GetWriteRequests
  // splits that entry, ring_buffer.cc:271-303).  Offsets from tail0, not wrapped.
  {
    const uint64_t o = P.tail0 + s.staged, cap = P.cap;
    if (in_zc) {
      const uint64_t e0 = o, e1 = o + 8, e2 = e1 + pay, e3 = e2 + pad, e4 = e3 + 8;
      s.splits += (e0 < cap && e1 > cap) + (e1 < cap && e2 > cap) + (pad && e2 < cap && e3 > cap) + (e3 < cap && e4 > cap);
      s.nsge += pad ? 4 : 3;
      s.st += 16 + pad;
      s.zc_bytes += pay;
      s.zc_records++;
    } else {
      s.splits += (o < cap && o + enc > cap);
      s.nsge += 1;
      s.st += enc;
      s.copy_bytes += pay;
    }
  }
  // the record in the peer ring: header at rt, payload behind it (wrapping), tags by the copy waves
  const uint64_t TB = 1ull << P.ts;
  const uint64_t pay_off = (s.rt + 8) & mask;
  const uint64_t tagw = GRDMA_SEG_TAG_WRITE | (pay << GRDMA_SEG_TAG_LEN_SHIFT);
  r.stop = 0;
  if (pay_off + pay > P.cap) {
    const uint64_t l1 = P.cap - pay_off;
    r.nsegs = 2;
    r.seg[0].dst = P.ring + pay_off; r.seg[0].src = ptr; r.seg[0].len = l1; r.seg[0].flags = tagw | GRDMA_SEG_TAG_HDR;
    r.seg[1].dst = P.ring; r.seg[1].src = ptr + l1; r.seg[1].len = pay - l1; r.seg[1].flags = tagw | GRDMA_SEG_TAG_FTR;
    r.tile0[0] = (uint32_t)s.ntiles;
    r.tile0[1] = (uint32_t)(s.ntiles + ((l1 + TB - 1) >> P.ts));
    s.ntiles += ((l1 + TB - 1) >> P.ts) + ((pay - l1 + TB - 1) >> P.ts);
    s.nseg += 2;
  } else {
    r.nsegs = 1;
    r.seg[0].dst = P.ring + pay_off; r.seg[0].src = ptr; r.seg[0].len = pay;
    r.seg[0].flags = tagw | GRDMA_SEG_TAG_HDR | GRDMA_SEG_TAG_FTR;
    r.tile0[0] = (uint32_t)s.ntiles;
    s.ntiles += (pay + TB - 1) >> P.ts;
    s.nseg += 1;
  }
  s.rt = (s.rt + enc) & mask;  // NextTail
  s.staged += enc;
  s.written += pay;
  s.nrec++;
  if (pay == len) { s.idx = i + 1; s.bidx = 0; }
  else { s.idx = i; s.bidx = skip + pay; }
  return r;
}
