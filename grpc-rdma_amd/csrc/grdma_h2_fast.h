// Message-boundary step of the HTTP/2 deframer (K8/K9): plain integer code, no wave intrinsics,
// shared by k_h2_deframe (csrc/grdma_h2.hip) and by the host check in tests/cc/h2_fast_host.cc,
// which runs it inside the CPU oracle's parser and compares the events with the oracle alone.
//
// The bulk step of k_h2_deframe covers the frames BETWEEN the first and the last frame of a
// message.  What is left for the byte-wise automaton in a client-streaming call is the slice in
// which a message starts -- two shapes, both decidable from the 32 staged bytes of the slice:
//
//   sending side (what grpc_chttp2_encode_data hands the endpoint, frame_data.cc:64-90 with the
//   5-byte header of chttp2_transport.cc:1502-1510 merged into the inlined header slice):
//       [FH_B 9][msg header 5][p0 payload bytes]   then   [rest of frame B]
//   receiving side (the endpoint sized its read to the first record, max(256, 9),
//   rdma_bp_posix.cc:308, so the closing frame of the previous message sits in front):
//       [FH_A 9][a bytes: the end of the previous message][FH_B 9][msg header 5][p0]   then   [rest of frame B]
//
// h2_boundary_match decides whether slice s (and, if frame B continues, slice s + 1) has this
// shape for the stream that is current; h2_boundary_event yields event k of the list the automaton
// (parsing.cc:111-250 + frame_data.cc:92-276) would have produced for the same bytes.  Anything
// else -- flags, another stream, a frame that holds the end of one message and the start of the
// next, a slice that goes on behind frame B, a message header outside the staged bytes -- is "no
// match" and goes through the automaton as before.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GRDMA_HD __host__ __device__ __forceinline__
#else
#define GRDMA_HD inline
#endif

struct h2_bstep {
  uint32_t ok;       // 1 = the slice(s) match
  uint32_t o;        // offset of frame B's header in slice s (0, or 9 + a)
  uint32_t a;        // payload bytes of the closing frame A (0 when there is none)
  uint32_t fs;       // size of frame B
  uint32_t comp;     // message header: compressed flag
  uint32_t mlen;     //                 message length
  uint32_t t0;       // bytes of frame B's payload in slice s (the 5 header bytes included)
  uint32_t p1;       // bytes of frame B in slice s + 1 (0 = frame B ends with slice s)
  uint32_t rem;      // message bytes still to come after frame B (0 = the message ended)
  uint32_t nev;      // events of the step
  uint32_t nslices;  // slices consumed: 1 or 2
};

// bytes [off, off + 8) of the staged 32 bytes (c0..c3 little endian, zero beyond the slice and
// beyond byte 31)
GRDMA_HD uint64_t h2f_word(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint32_t off) {
  const uint32_t q = off >> 3;
  const uint64_t lo = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : q == 3 ? c3 : 0;
  const uint64_t hi = q == 0 ? c1 : q == 1 ? c2 : q == 2 ? c3 : 0;
  const uint32_t bs = (off & 7u) * 8u;
  return bs ? (lo >> bs) | (hi << (64u - bs)) : lo;
}

// the 9-byte frame header at `off` (parsing.cc:111-193)
GRDMA_HD void h2f_frame_header(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint32_t off, uint32_t* fs,
                               uint32_t* type_flags, uint32_t* sid) {
  const uint64_t w = h2f_word(c0, c1, c2, c3, off);
  const uint32_t b9 = (uint32_t)(h2f_word(c0, c1, c2, c3, off + 8) & 0xFF);
  *fs = (uint32_t)(((w & 0xFF) << 16) | (((w >> 8) & 0xFF) << 8) | ((w >> 16) & 0xFF));
  *type_flags = (uint32_t)((w >> 24) & 0xFFFF);  // type | flags << 8
  *sid = (uint32_t)((((w >> 40) & 0x7F) << 24) | (((w >> 48) & 0xFF) << 16) | (((w >> 56) & 0xFF) << 8)) | b9;
}

// d_state / d_fsz / d_id: the data parser of the current stream (0 = at a message header,
// 5 = d_fsz message bytes to come).  next_len: length of slice s + 1, ~0 when there is none.
GRDMA_HD h2_bstep h2_boundary_match(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint64_t len,
                                    uint64_t next_len, int32_t d_state, uint32_t d_fsz, uint32_t d_id,
                                    uint32_t max_frame) {
  h2_bstep r;
  r.ok = 0; r.o = 0; r.a = 0; r.fs = 0; r.comp = 0; r.mlen = 0; r.t0 = 0; r.p1 = 0; r.rem = 0; r.nev = 0;
  r.nslices = 0;
  uint32_t o = 0, a = 0, fs, tf, sid;
  if (d_state == 5) {
    // closing frame A: exactly the d_fsz bytes the message still lacks, and few enough of them
    // that frame B's header and the message header are inside the staged bytes (9 + a + 14 <= 32)
    if (d_fsz == 0 || d_fsz > 9 || d_fsz > max_frame) return r;
    if (len < 9ull + d_fsz + 14ull) return r;
    h2f_frame_header(c0, c1, c2, c3, 0, &fs, &tf, &sid);
    if (tf != 0 || sid != d_id || fs != d_fsz) return r;
    a = d_fsz;
    o = 9 + a;
  } else if (d_state != 0 || len < 14) {
    return r;
  }
  h2f_frame_header(c0, c1, c2, c3, o, &fs, &tf, &sid);
  if (tf != 0 || sid != d_id || fs < 5 || fs > max_frame) return r;
  const uint64_t m = h2f_word(c0, c1, c2, c3, o + 9);
  const uint32_t comp = (uint32_t)(m & 0xFF);
  const uint32_t mlen = (uint32_t)((((m >> 8) & 0xFF) << 24) | (((m >> 16) & 0xFF) << 16) |
                                   (((m >> 24) & 0xFF) << 8) | ((m >> 32) & 0xFF));
  // the whole frame lies inside the message (a frame that ends one message and starts the next
  // is the automaton's), and the message is not empty
  if (comp > 1 || mlen == 0 || (uint64_t)mlen + 5 < fs) return r;
  const uint64_t avail = len - o - 9;  // >= 5
  if (avail > fs) return r;            // the slice goes on behind frame B
  const uint32_t t0 = (uint32_t)avail, p1 = fs - t0;
  if (p1 != 0 && next_len != (uint64_t)p1) return r;  // frame B must end exactly with slice s + 1
  r.ok = 1;
  r.o = o; r.a = a; r.fs = fs; r.comp = comp; r.mlen = mlen; r.t0 = t0; r.p1 = p1;
  r.rem = mlen - (fs - 5);
  r.nev = (o ? 4u : 0u) + 3u + (t0 > 5 ? 1u : 0u) + (p1 ? 2u : 0u) + (r.rem == 0 ? 1u : 0u);
  r.nslices = p1 ? 2u : 1u;
  return r;
}

// event k (0 <= k < B.nev) of the step that starts at slice s for stream id:
// out = {kind, a, b, c, d, slice} with the kinds of grdma_h2_event (include/grdma_amd.h)
GRDMA_HD void h2_boundary_event(const h2_bstep& B, uint32_t id, uint32_t s, uint32_t k, uint32_t out[6]) {
  enum { F = 1, P = 2, MB = 3, MY = 4, ME = 5 };  // EV_FRAME, EV_PAYLOAD, EV_MSG_BEGIN, EV_MSG_BYTES, EV_MSG_END
  int j = (int)k;
  uint32_t kind = 0, a = 0, b = 0, c = 0, d = 0, sl = s;
  if (B.o) {
    if (j == 0) { kind = F; c = id; d = B.a; }
    else if (j == 1) { kind = P; a = 9; b = B.a; c = 1; }
    else if (j == 2) { kind = MY; a = 9; b = B.a; c = id; }
    else if (j == 3) { kind = ME; c = id; }
    j -= 4;
  }
  if (j == 0) { kind = F; c = id; d = B.fs; }
  else if (j == 1) { kind = P; a = B.o + 9; b = B.t0; c = B.t0 == B.fs ? 1u : 0u; }
  else if (j == 2) { kind = MB; a = B.comp; b = B.mlen; c = id; }
  j -= 3;
  if (B.t0 > 5) {
    if (j == 0) { kind = MY; a = B.o + 14; b = B.t0 - 5; c = id; }
    j -= 1;
  }
  if (B.p1) {
    if (j == 0) { kind = P; a = 0; b = B.p1; c = 1; sl = s + 1; }
    else if (j == 1) { kind = MY; a = 0; b = B.p1; c = id; sl = s + 1; }
    j -= 2;
  }
  if (B.rem == 0 && j == 0) { kind = ME; c = id; sl = B.p1 ? s + 1 : s; }
  out[0] = kind; out[1] = a; out[2] = b; out[3] = c; out[4] = d; out[5] = sl;
}
