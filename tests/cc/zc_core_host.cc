// TEST INFRASTRUCTURE.  Host check of the zero-copy send's record pricing (grpc-rdma_amd/csrc/
// grdma_zc_core.h, the loop body of k_tx_plan_zc): the same function the kernel calls, driven here over a
// host model of the device side -- the loop of the kernel, then the gather segments executed the way the copy
// waves execute them (payload bytes, header in front of the first segment, zero padding and footer behind the
// last one, tag addresses wrapping inside the ring: plan_tile in csrc/grdma_devfn.h) -- and compared after
// every call with the CPU oracle's orc_pair_send_zerocopy: accepted bytes, work requests, scatter-gather
// entries, buffer tail, counters, remote_tail, partial_write and the ring image.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

extern "C" {
#include "../../oracle/grdma_oracle.h"
}
#include "../../grpc-rdma_amd/csrc/grdma_zc_core.h"

namespace {
typedef std::vector<uint8_t> bytes;
uint32_t g_rng;
uint32_t rnd() { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }

// every differing byte must be a padding byte: zero on the model, the next word the footer in both
bool ring_eq(const uint8_t* x, const uint8_t* y, size_t R) {
  for (size_t i = 0; i < R; i++) {
    if (x[i] == y[i]) continue;
    if (x[i] != 0) return false;
    const size_t nxt = ((i & ~(size_t)7) + 8) % R;
    for (size_t k = 0; k < 8; k++)
      if (x[nxt + k] != 0xFF || y[nxt + k] != 0xFF) return false;
  }
  return true;
}

// one gather segment as the copy waves run it (plan_tile: the tags depend on the segment only)
void run_segment(const grdma_seg& sg, uint64_t tag_base, uint64_t tm) {
  memcpy(reinterpret_cast<void*>(sg.dst), reinterpret_cast<const void*>(sg.src), sg.len);
  uint8_t* tb = reinterpret_cast<uint8_t*>(tag_base);
  if (sg.flags & GRDMA_SEG_TAG_HDR) {
    const uint64_t v = sg.flags >> GRDMA_SEG_TAG_LEN_SHIFT;
    memcpy(tb + ((sg.dst - 8 - tag_base) & tm), &v, 8);
  }
  if (sg.flags & GRDMA_SEG_TAG_FTR) {
    const uint64_t e = (sg.dst + sg.len - tag_base) & tm;
    const uint64_t pad = (0 - e) & 7;
    for (uint64_t k = 0; k < pad; k++) tb[e + k] = 0;
    const uint64_t f = GRDMA_FOOTER;
    memcpy(tb + ((e + pad) & tm), &f, 8);
  }
}
}  // namespace

// One seeded sequence of zero-copy sends, plain Sends and Recvs.  Returns 0, or a step number + 1 with a
// message in `why`.
extern "C" int zc_core_sequence(uint64_t R, int max_sge, uint64_t Z, uint32_t seed, int steps, char* why, uint64_t why_cap,
                                uint64_t* records_out, uint64_t* zc_records_out, uint64_t* wraps_out) {
  g_rng = seed;
  orc_pair oa, ob;
  if (orc_pair_init(&oa, R, max_sge) || orc_pair_init(&ob, R, max_sge) || orc_pair_enable_zerocopy(&oa, Z)) return -1;
  orc_pair_connect(&oa, &ob);
  bytes ring(R, 0), zc(Z, 0);
  uint64_t remote_tail = 0, zc_tail = 0, zc_bytes = 0, copy_bytes = 0, records = 0, zc_records = 0, wraps = 0;
  int partial = 0;
  const uint64_t sizes[] = {1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257, R / 3, R, Z / 2};
  int rc = 0;
  auto fail = [&](int step, const char* what) {
    snprintf(why, why_cap, "%s (R %llu sge %d Z %llu seed %u step %d)", what, (unsigned long long)R, max_sge,
             (unsigned long long)Z, seed, step);
    rc = step + 1;
  };
  for (int step = 0; step < steps && rc == 0; step++) {
    const uint32_t op = rnd() % 100;
    if (op < 60) {
      const int n = 1 + rnd() % 6;
      std::vector<bytes> keep;
      keep.reserve(n);
      std::vector<uint64_t> mptr(n), mlen(n);
      std::vector<orc_slice> osl(n);
      for (int i = 0; i < n; i++) {
        uint64_t len = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))];
        if (len == 0) len = 1;
        bytes data(len);
        for (auto& v : data) v = (uint8_t)rnd();
        uint64_t off = ~0ull;
        if (rnd() % 2 == 0 && len <= Z) {
          // AllocateSendBuffer on both sides (the model's allocator is the host function's rule)
          uint8_t* optr = orc_pair_allocate_send_buffer(&oa, len);
          const bool model_ok = zc_tail == 0 && zc_tail + len <= Z;
          if ((optr != nullptr) != model_ok) { fail(step, "allocator disagrees"); break; }
          if (optr) { off = zc_tail; zc_tail += len; if (off != (uint64_t)(optr - oa.zc_buf)) { fail(step, "allocator offset"); break; } }
          else if (rnd() % 2) off = rnd() % (Z - len + 1);
        }
        if (off != ~0ull) {
          memcpy(zc.data() + off, data.data(), len);
          memcpy(oa.zc_buf + off, data.data(), len);
          mptr[i] = (uint64_t)(zc.data() + off);
          osl[i].ptr = oa.zc_buf + off;
        } else {
          keep.push_back(data);
          mptr[i] = (uint64_t)keep.back().data();
          osl[i].ptr = keep.back().data();
        }
        mlen[i] = len;
        osl[i].len = len;
      }
      if (rc) break;
      const uint64_t bi = (rnd() % 10 < 3) ? rnd() % mlen[0] : 0;
      // ---- the model: the kernel's loop over zc_price, then the segments ----
      zc_params P;
      P.cap = R; P.S = R / 2; P.tail0 = remote_tail; P.rhead = oa.status_recv.remote_head; P.max_sge = (uint64_t)max_sge;
      P.ring = (uint64_t)ring.data(); P.zc_base = (uint64_t)zc.data(); P.zc_cap = Z; P.byte_idx = bi;
      P.ts = GRDMA_PLAN_TILE_SHIFT(R);
      zc_state S;
      zc_begin(P, S);
      uint64_t offered = 0;
      for (int i = 0; i < n; i++) offered += mlen[i];
      offered -= bi;
      std::vector<grdma_seg> segs;
      std::vector<uint32_t> prefix;
      for (int i = 0; i < n; i++) {
        const zc_record r = zc_price(P, S, (uint64_t)i, mptr[i], mlen[i]);
        if (r.stop) break;
        for (uint32_t k = 0; k < r.nsegs; k++) { segs.push_back(r.seg[k]); prefix.push_back(r.tile0[k]); }
        if (r.nsegs == 2) wraps++;
      }
      // the plan is well formed: tile prefix monotone, every segment at least one tile, counts agree
      if (segs.size() != S.nseg) { fail(step, "segment count"); break; }
      {
        const uint64_t TB = 1ull << P.ts;
        uint64_t t = 0;
        bool good = true;
        for (size_t k = 0; k < segs.size(); k++) {
          if (prefix[k] != t || segs[k].len == 0) good = false;
          t += (segs[k].len + TB - 1) >> P.ts;
        }
        if (!good || t != S.ntiles) { fail(step, "tile prefix"); break; }
      }
      for (const grdma_seg& sg : segs) run_segment(sg, P.ring, R - 1);
      uint64_t wr[2][2] = {{0, 0}, {0, 0}};
      int nwr = 0;
      if (S.staged > 0) {
        const uint64_t seg1 = S.staged < R - P.tail0 ? S.staged : R - P.tail0;
        wr[0][0] = P.tail0; wr[0][1] = seg1; nwr = 1;
        if (P.tail0 + S.staged >= R) { wr[1][0] = 0; wr[1][1] = S.staged - seg1; nwr = 2; }
      }
      remote_tail = S.rt;
      partial = S.written < offered;
      zc_tail = (uint32_t)(zc_tail - S.zc_bytes);
      zc_bytes += S.zc_bytes;
      copy_bytes += S.copy_bytes;
      records += S.nrec;
      zc_records += S.zc_records;
      // ---- the oracle ----
      const uint64_t so = orc_pair_send_zerocopy(&oa, osl.data(), (uint64_t)n, bi);
      if (S.written != so) { fail(step, "accepted bytes"); break; }
      if (nwr != oa.wr_count) { fail(step, "work request count"); break; }
      for (int k = 0; k < nwr; k++)
        if (wr[k][0] != oa.wr[k][0] || wr[k][1] != oa.wr[k][1]) fail(step, "work requests");
      if (rc) break;
      if (S.nsge + S.splits != oa.sge_count) { fail(step, "scatter-gather entries"); break; }
      if (S.st != oa.staging_used) { fail(step, "staging bytes"); break; }
      if (zc_tail != oa.zc_tail || zc_bytes != oa.zc_bytes || copy_bytes != oa.copy_bytes) { fail(step, "buffer tail / counters"); break; }
      if (remote_tail != oa.remote_tail || partial != oa.partial_write) { fail(step, "remote_tail / partial_write"); break; }
      // the cursor: bytes accepted = slices [0, idx) whole (minus byte_idx) + bidx bytes into slice idx
      {
        uint64_t acc = 0;
        for (uint64_t i = 0; i < S.idx; i++) acc += mlen[i];
        acc += S.bidx;
        acc -= bi;
        if (S.written > 0 && acc != S.written) { fail(step, "cursor"); break; }
      }
      if (!ring_eq(ring.data(), ob.ring.buf, R)) { fail(step, "ring image"); break; }
    } else if (op < 70) {
      // a plain Send in between: the oracle's, mirrored into the model
      const uint64_t len = sizes[rnd() % 14];
      bytes data(len);
      for (auto& v : data) v = (uint8_t)rnd();
      orc_slice os = {data.data(), len};
      orc_pair_send(&oa, &os, 1, 0);
      memcpy(ring.data(), ob.ring.buf, R);
      remote_tail = oa.remote_tail;
      partial = oa.partial_write;
    } else {
      // the reader: the oracle's Recv (zero-fill, credit), mirrored into the model
      const uint64_t caps[] = {1, 8, 64, 256, R};
      bytes dst(R);
      orc_pair_recv(&ob, dst.data(), caps[rnd() % 5]);
      memcpy(ring.data(), ob.ring.buf, R);
    }
  }
  if (records_out) *records_out = records;
  if (zc_records_out) *zc_records_out = zc_records;
  if (wraps_out) *wraps_out = wraps;
  orc_pair_destroy(&oa);
  orc_pair_destroy(&ob);
  return rc;
}
