// HTTP/2 DATA framing (K6/K7) and deframing (K8/K9) on the device.
//
// TX  k_h2_frame     builds, in HBM, the slice list chttp2 hands to
//                    grpc_endpoint_write for a batch of gRPC messages:
//                    5-byte message header (chttp2_transport.cc:1502-1510), 9-byte
//                    DATA frame headers (grpc_chttp2_encode_data, frame_data.cc:64-90),
//                    payload sub-slices by reference, with the inlined-slice merge
//                    rule of grpc_slice_buffer_add (slice_buffer.cc:136-171) and the
//                    split rule of move_first_no_ref (slice_buffer.cc:270-313).
//                    No payload byte is copied: K1 (k_copy) gathers straight from
//                    the message buffers.
// RX  k_h2_deframe   the resumable frame-header state machine of
//                    grpc_chttp2_perform_read (parsing.cc:56-253) and the gRPC
//                    message deframer (frame_data.cc:92-276) over the slices an
//                    endpoint_read delivered.  Frame headers sit at data-dependent
//                    offsets (a linked list again); one wave stages the first 32
//                    bytes of the next 64 slices in registers so the automaton
//                    never waits on memory for a header that starts a slice -- the
//                    common case, because the ring preserves slice boundaries.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/grdma_amd.h"
#include "grdma_dev.h"
#include "grdma_devfn.h"

struct grdma_h2_msg_dev {
  const uint8_t* payload;
  uint64_t len;
  uint32_t stream_id;
  uint32_t flags;  // 1 = compressed, 2 = end_stream
};

struct grdma_h2_frame_result {
  uint64_t nslices;
  uint64_t hdr_bytes;   // bytes of the header arena used
  uint64_t wire_bytes;  // Σ slice lengths
  uint64_t overflow;
};

#define H2_INLINED 23u
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// grpc_chttp2_data_parser + the deframe fields of grpc_chttp2_transport, one
// per connection, resident in HBM between calls.
struct grdma_h2_stream_dev {
  uint32_t stream_id;
  int32_t state;        // 0..4 FH_0..FH_4, 5 FRAME, 6 ERROR
  uint32_t frame_size;
  int32_t compressed;
};
#define H2_MAX_STREAMS 16
struct grdma_h2_parser_dev {
  int32_t state;        // 0..23 client prefix, 24..32 FH_0..FH_8, 33 FRAME
  uint32_t incoming_frame_size;
  uint32_t incoming_frame_type;
  uint32_t incoming_frame_flags;
  uint32_t incoming_stream_id;
  uint32_t max_frame_size;
  int32_t cur_parser;   // 0 skip, 1 data
  int32_t nstreams;
  grdma_h2_stream_dev streams[H2_MAX_STREAMS];
  int32_t error;        // connection error (grdma_h2_error), sticky
  int32_t pad;
};

struct grdma_h2_deframe_result {
  uint64_t nevents;
  uint64_t overflow;
  uint64_t slices_done;
  int64_t error;
};

namespace {

// ------------------------------------------------------------------ TX framing
// Sequential layout of ONE message given the state of the slice buffer's back
// slice (inlined length, 0 = not inlined).  emit(kind, a, b): kind 0 = inlined
// bytes [a = source selector, b = length] appended as a new slice or merged,
// handled by the caller through the callbacks below.
struct frame_walk {
  // output cursors
  uint64_t nslices;
  uint64_t hdr_off;
  uint64_t wire;
  uint32_t back_inl;  // length of the back slice if it is inlined, else 0
};

template <bool EMIT>
__device__ __forceinline__ void add_inlined(frame_walk* w, const uint8_t* bytes, uint32_t n,
                                            bool merge, grdma_sge* out, uint8_t* hdr, uint64_t cap,
                                            uint64_t hdr_cap, uint64_t* overflow) {
  // grpc_slice_buffer_add (merge) / grpc_slice_buffer_add_indexed (no merge)
  uint32_t done = 0;
  if (merge && w->back_inl && w->back_inl < H2_INLINED) {
    const uint32_t room = H2_INLINED - w->back_inl;
    const uint32_t cp = n < room ? n : room;
    if (EMIT) {
      // the back slice's bytes end at hdr_off (slots are packed per slice start)
      grdma_sge* back = &out[w->nslices - 1];
      uint8_t* dst = const_cast<uint8_t*>(back->ptr) + back->len;
      for (uint32_t i = 0; i < cp; i++) dst[i] = bytes[i];
      back->len += cp;
    }
    w->back_inl += cp;
    done = cp;
    w->wire += cp;
    if (done == n) return;
  }
  const uint32_t rest = n - done;
  if (w->nslices >= cap || w->hdr_off + 32 > hdr_cap) {
    *overflow = 1;
    return;
  }
  if (EMIT) {
    uint8_t* dst = hdr + w->hdr_off;
    for (uint32_t i = 0; i < rest; i++) dst[i] = bytes[done + i];
    out[w->nslices].ptr = dst;
    out[w->nslices].len = rest;
  }
  w->nslices++;
  w->hdr_off += 32;  // one 32-byte slot per inlined slice (sizeof(grpc_slice))
  w->back_inl = rest;
  w->wire += rest;
}

template <bool EMIT>
__device__ __forceinline__ void add_ref(frame_walk* w, const uint8_t* ptr, uint64_t n,
                                        grdma_sge* out, uint64_t cap, uint64_t* overflow) {
  if (w->nslices >= cap) {
    *overflow = 1;
    return;
  }
  if (EMIT) {
    out[w->nslices].ptr = ptr;
    out[w->nslices].len = n;
  }
  w->nslices++;
  w->back_inl = 0;
  w->wire += n;
}

template <bool EMIT>
__device__ void walk_message(const grdma_h2_msg_dev& m, uint32_t max_frame, frame_walk* w,
                             grdma_sge* out, uint8_t* hdr, uint64_t cap, uint64_t hdr_cap,
                             uint64_t* overflow) {
  uint8_t h5[5];
  h5[0] = (m.flags & 1) ? 1 : 0;  // chttp2_transport.cc:1504-1509
  h5[1] = (uint8_t)(m.len >> 24);
  h5[2] = (uint8_t)(m.len >> 16);
  h5[3] = (uint8_t)(m.len >> 8);
  h5[4] = (uint8_t)m.len;
  uint64_t h5_left = 5, pay_left = m.len, pay_off = 0;
  uint64_t fcb = 5 + m.len;
  while (fcb > 0) {
    const uint64_t send = fcb < max_frame ? fcb : max_frame;
    const bool last = (m.flags & 2) && send == fcb;
    uint8_t fh[9];  // frame_data.cc:73-82
    fh[0] = (uint8_t)(send >> 16); fh[1] = (uint8_t)(send >> 8); fh[2] = (uint8_t)send;
    fh[3] = 0; fh[4] = last ? 1 : 0;
    fh[5] = (uint8_t)(m.stream_id >> 24); fh[6] = (uint8_t)(m.stream_id >> 16);
    fh[7] = (uint8_t)(m.stream_id >> 8); fh[8] = (uint8_t)m.stream_id;
    add_inlined<EMIT>(w, fh, 9, true, out, hdr, cap, hdr_cap, overflow);
    uint64_t n = send;
    const bool whole = (fcb == n);  // grpc_slice_buffer_move_into: every slice via add()
    if (h5_left > 0) {
      const uint64_t take = n < h5_left ? n : h5_left;
      // n >= slice_len (or the final move_into): merged add; n < slice_len: split,
      // the head goes in un-merged (add_indexed)
      const bool merge = whole || n >= h5_left;
      add_inlined<EMIT>(w, h5 + (5 - h5_left), (uint32_t)take, merge, out, hdr, cap, hdr_cap, overflow);
      h5_left -= take;
      n -= take;
    }
    if (n > 0) {
      add_ref<EMIT>(w, m.payload + pay_off, n, out, cap, overflow);
      pay_off += n;
      pay_left -= n;
    }
    fcb -= send;
  }
  (void)pay_left;
}

// One thread lays out one message.  A message that ends with an inlined slice
// (only an empty message does) lets the next header merge into it, so a thread
// first replays the run of empty messages in front of it to learn the state of
// the back slice.
__global__ __launch_bounds__(256) void k_h2_frame(const grdma_h2_msg_dev* msgs, uint64_t nmsgs,
                                                  uint32_t max_frame, grdma_sge* out,
                                                  uint64_t cap, uint8_t* hdr, uint64_t hdr_cap,
                                                  uint64_t* counts /* 3*nmsgs scratch */,
                                                  grdma_h2_frame_result* res) {
  __shared__ uint64_t s_wave[4];
  const uint64_t tid = threadIdx.x;
  uint64_t overflow = 0;
  // pass 1: sizes (serial over blocks of 256 messages; one block is launched)
  uint64_t base_sl = 0, base_hdr = 0, base_wire = 0;
  for (uint64_t m0 = 0; m0 < nmsgs; m0 += 256) {
    const uint64_t i = m0 + tid;
    frame_walk w = {0, 0, 0, 0};
    uint64_t merged_into_prev = 0;
    if (i < nmsgs) {
      // incoming back-slice state
      uint64_t j = i;
      while (j > 0 && msgs[j - 1].len == 0) j--;
      frame_walk pre = {0, 0, 0, 0};
      for (; j < i; j++) walk_message<false>(msgs[j], max_frame, &pre, nullptr, nullptr, ~0ull, ~0ull, &overflow);
      frame_walk me = {0, 0, 0, pre.back_inl};
      walk_message<false>(msgs[i], max_frame, &me, nullptr, nullptr, ~0ull, ~0ull, &overflow);
      w = me;
      merged_into_prev = pre.back_inl;
    }
    uint64_t tot_sl, tot_hdr, tot_wire;
    const uint64_t x_sl = block_excl_scan(w.nslices, s_wave, &tot_sl);
    const uint64_t x_hdr = block_excl_scan(w.hdr_off, s_wave, &tot_hdr);
    block_excl_scan(w.wire, s_wave, &tot_wire);
    if (i < nmsgs && (i == 0 || msgs[i - 1].len != 0)) {
      // pass 2: emit at the exact position.  A run of empty messages shares
      // inlined slices across message boundaries, so the first message of such
      // a run emits the whole run (sequentially, like the reference would).
      frame_walk me = {base_sl + x_sl, base_hdr + x_hdr, 0, 0};
      uint64_t k = i;
      for (;;) {
        walk_message<true>(msgs[k], max_frame, &me, out, hdr, cap, hdr_cap, &overflow);
        if (msgs[k].len != 0 || k + 1 >= nmsgs) break;
        k++;
      }
    }
    base_sl += tot_sl;
    base_hdr += tot_hdr;
    base_wire += tot_wire;
    __syncthreads();
  }
  (void)counts;
  if (overflow) atomicExch((unsigned long long*)&res->overflow, 1ull);
  if (tid == 0) {
    res->nslices = base_sl;
    res->hdr_bytes = base_hdr;
    res->wire_bytes = base_wire;
  }
}

// ---------------------------------------------------------------- RX deframing
enum { EV_FRAME = 1, EV_PAYLOAD = 2, EV_MSG_BEGIN = 3, EV_MSG_BYTES = 4, EV_MSG_END = 5 };
enum { ST_FH0 = 24, ST_FRAME = 33 };

__device__ __forceinline__ grdma_h2_stream_dev* find_stream(grdma_h2_parser_dev* p, uint32_t id) {
  for (int i = 0; i < p->nstreams; i++)
    if (p->streams[i].stream_id == id) return &p->streams[i];
  if (id == 0 || p->nstreams >= H2_MAX_STREAMS) return nullptr;
  grdma_h2_stream_dev* d = &p->streams[p->nstreams++];
  d->stream_id = id;
  d->state = 0;
  d->frame_size = 0;
  d->compressed = 0;
  return d;
}

// Register cache of slices [cbase, cbase + 64): lane i holds the descriptor and the first
// 32 bytes of slice cbase + i (one memory round trip per 64 slices).  Plain functions over a
// plain struct (no capturing lambdas): everything stays in registers.
struct h2_slice_cache {
  uint64_t cbase, c0, c1, c2, c3, c_off, c_len;
};

__device__ __forceinline__ uint64_t h2_keep(uint64_t v, uint64_t first, uint64_t n) {
  // bytes at and beyond the slice end read as zero
  if (n >= first + 8) return v;
  if (n <= first) return 0;
  return v & ((1ull << ((n - first) * 8)) - 1);
}

__device__ __forceinline__ void h2_ensure(h2_slice_cache& C, uint64_t s, const uint8_t* arena,
                                          const grdma_slice_out* slices, uint64_t nslices, int lane) {
  if (s >= C.cbase && s < C.cbase + 64) return;
  C.cbase = s;
  const uint64_t mine = s + lane;
  C.c0 = C.c1 = C.c2 = C.c3 = 0;
  C.c_off = C.c_len = 0;
  if (mine < nslices) {
    C.c_off = slices[mine].off;
    C.c_len = slices[mine].len;
    const uint8_t* p = arena + C.c_off;
    const uint64_t n = C.c_len;
    // the aligned 16-byte blocks that hold the first 32 bytes of the slice (two when the
    // slice starts on a 16-byte boundary, three otherwise); a block is only fetched when
    // it overlaps the slice, so nothing outside the blocks the slice touches is read
    const uint64_t sh = (uint64_t)p & 15;
    const u64x2* q = reinterpret_cast<const u64x2*>((uint64_t)p & ~15ull);
    const uint64_t need = (n < 32 ? n : 32) + sh;
    u64x2 v0 = {0, 0}, v1 = {0, 0}, v2 = {0, 0};
    if (need > 0) v0 = q[0];
    if (need > 16) v1 = q[1];
    if (need > 32) v2 = q[2];
    // 48-byte window w0..w5, shifted right by sh bytes
    uint64_t w0 = v0.x, w1 = v0.y, w2 = v1.x, w3 = v1.y, w4 = v2.x, w5 = v2.y;
    if (sh & 8) { w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; }
    const unsigned bs = (unsigned)(sh & 7) * 8;
    uint64_t o0 = w0, o1 = w1, o2 = w2, o3 = w3;
    if (bs) {
      o0 = (w0 >> bs) | (w1 << (64 - bs));
      o1 = (w1 >> bs) | (w2 << (64 - bs));
      o2 = (w2 >> bs) | (w3 << (64 - bs));
      o3 = (w3 >> bs) | (w4 << (64 - bs));
    }
    C.c0 = h2_keep(o0, 0, n);
    C.c1 = h2_keep(o1, 8, n);
    C.c2 = h2_keep(o2, 16, n);
    C.c3 = h2_keep(o3, 24, n);
  }
}

__device__ __forceinline__ uint32_t h2_byte_at(h2_slice_cache& C, uint64_t s, uint64_t off,
                                               const uint8_t* arena, const grdma_slice_out* slices,
                                               uint64_t nslices, int lane) {
  h2_ensure(C, s, arena, slices, nslices, lane);
  const int src = (int)(s - C.cbase);
  if (off < 32) {
    const uint64_t q = off >> 3;
    const uint64_t word = __shfl(q == 0 ? C.c0 : q == 1 ? C.c1 : q == 2 ? C.c2 : C.c3, src, 64);
    return (uint32_t)((word >> ((off & 7) * 8)) & 0xFF);
  }
  return arena[__shfl(C.c_off, src, 64) + off];
}

// bytes [off, off + 8) of slice s as a little-endian word; needs off + 8 <= 32 (inside the
// cached look-ahead).  Two shuffles instead of eight byte fetches.
__device__ __forceinline__ uint64_t h2_bytes8(h2_slice_cache& C, uint64_t s, uint64_t off,
                                              const uint8_t* arena, const grdma_slice_out* slices,
                                              uint64_t nslices, int lane) {
  h2_ensure(C, s, arena, slices, nslices, lane);
  const int src = (int)(s - C.cbase);
  const uint64_t q = off >> 3;
  const uint64_t lo = __shfl(q == 0 ? C.c0 : q == 1 ? C.c1 : q == 2 ? C.c2 : C.c3, src, 64);
  const uint64_t hi = __shfl(q == 0 ? C.c1 : q == 1 ? C.c2 : C.c3, src, 64);  // q == 3: unused
  const unsigned bs = (unsigned)(off & 7) * 8;
  return bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
}

__device__ __forceinline__ void h2_push(grdma_h2_event* ev, uint64_t ev_cap, uint64_t& nev,
                                        uint64_t& overflow, int lane, uint32_t kind, uint32_t a,
                                        uint32_t b, uint32_t c, uint32_t d, uint32_t sl) {
  if (nev >= ev_cap) {
    overflow = 1;
    return;
  }
  if (lane == 0) {
    // (global address space: a generic store would also count against the LDS counter
    // and stall the next shuffle of the slice cache)
    auto* e = (__attribute__((address_space(1))) grdma_h2_event*)(uint64_t)(ev + nev);
    e->kind = kind; e->a = a; e->b = b; e->c = c; e->d = d;
    e->slice = sl;
  }
  nev++;
}

// the data parser of the current stream (grpc_chttp2_data_parser), cached in registers
struct h2_cur_stream {
  int idx;
  uint32_t id, fsz;
  int32_t state, comp;
};

__device__ __forceinline__ void h2_flush_stream(grdma_h2_parser_dev* P, const h2_cur_stream& D, int lane) {
  if (D.idx >= 0 && lane == 0) {
    P->streams[D.idx].state = D.state;
    P->streams[D.idx].frame_size = D.fsz;
    P->streams[D.idx].compressed = D.comp;
  }
  __syncthreads();
}

// false: unknown stream and no room in the table / stream id 0
__device__ __forceinline__ bool h2_select_stream(grdma_h2_parser_dev* P, h2_cur_stream& D, uint32_t id,
                                                 int* s_idx, int lane) {
  if (D.idx >= 0 && D.id == id) return true;
  h2_flush_stream(P, D, lane);
  D.idx = -1;
  if (lane == 0) {
    grdma_h2_stream_dev* d = find_stream(P, id);
    *s_idx = d ? (int)(d - P->streams) : -1;
  }
  __syncthreads();
  const int idx = *s_idx;
  __syncthreads();
  if (idx < 0) return false;
  D.idx = idx;
  D.id = id;
  D.state = P->streams[idx].state;
  D.fsz = P->streams[idx].frame_size;
  D.comp = P->streams[idx].compressed;
  return true;
}

__global__ __launch_bounds__(64) void k_h2_deframe(grdma_h2_parser_dev* gp, const uint8_t* arena,
                                                   const grdma_slice_out* slices, uint64_t nslices,
                                                   grdma_h2_event* ev, uint64_t ev_cap,
                                                   grdma_h2_deframe_result* res) {
  const int lane = threadIdx.x;
  __shared__ grdma_h2_parser_dev P;
  __shared__ int s_idx;
  static_assert(sizeof(grdma_h2_parser_dev) % 4 == 0, "word copy");
  for (unsigned i = lane; i < sizeof(grdma_h2_parser_dev) / 4; i += 64)
    reinterpret_cast<uint32_t*>(&P)[i] = reinterpret_cast<const uint32_t*>(gp)[i];
  __syncthreads();
  uint64_t nev = 0, overflow = 0;
  static const char kPrefix[] = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n";  // internal.h:781

  // The automaton state lives in (wave-uniform) registers for the whole call; the copy in
  // LDS is only the stream table, touched when a frame changes the current stream.
  int32_t st = P.state;
  uint32_t fsz = P.incoming_frame_size, ftype = P.incoming_frame_type;
  uint32_t fflags = P.incoming_frame_flags, sid = P.incoming_stream_id;
  int32_t cur_parser = P.cur_parser;
  const uint32_t max_frame = P.max_frame_size;
  h2_cur_stream D = {-1, 0, 0, 0, 0};
  h2_slice_cache C = {~0ull, 0, 0, 0, 0, 0, 0};
#define H2_PUSH(kind, a, b, c, d, sl) h2_push(ev, ev_cap, nev, overflow, lane, kind, a, b, c, d, sl)
#define H2_BYTE(s_, off_) h2_byte_at(C, s_, off_, arena, slices, nslices, lane)

  uint64_t s = 0;
  int err = P.error;
  for (; s < nslices && !err && !overflow; s++) {
    h2_ensure(C, s, arena, slices, nslices, lane);
    const uint64_t len = __shfl(C.c_len, (int)(s - C.cbase), 64);
    uint64_t cur = 0;
    while (cur < len && !err && !overflow) {
      if (st < ST_FH0) {  // client connection preface, parsing.cc:70-109
        if (H2_BYTE(s, cur) != (uint8_t)kPrefix[st]) { err = 1; break; }
        cur++; st++;
        continue;
      }
      if (st < ST_FRAME) {
        if (st == ST_FH0 && len - cur >= 9 && cur + 9 <= 32) {
          // the whole 9-byte frame header sits in the cached look-ahead: same fields as
          // the byte-wise FH_0..FH_8 walk below (parsing.cc:111-193), taken in one step
          const uint64_t b8 = h2_bytes8(C, s, cur, arena, slices, nslices, lane);
          const uint32_t b9 = H2_BYTE(s, cur + 8);
          fsz = (uint32_t)(((b8 & 0xFF) << 16) | (((b8 >> 8) & 0xFF) << 8) | ((b8 >> 16) & 0xFF));
          ftype = (uint32_t)((b8 >> 24) & 0xFF);
          fflags = (uint32_t)((b8 >> 32) & 0xFF);
          sid = (uint32_t)((((b8 >> 40) & 0x7F) << 24) | (((b8 >> 48) & 0xFF) << 16) |
                           (((b8 >> 56) & 0xFF) << 8)) | b9;
          cur += 9;
          st = 32;
        } else {
        const uint32_t c = H2_BYTE(s, cur);
        switch (st) {
          case 24: fsz = c << 16; break;
          case 25: fsz |= c << 8; break;
          case 26: fsz |= c; break;
          case 27: ftype = c; break;
          case 28: fflags = c; break;
          case 29: sid = (c & 0x7f) << 24; break;
          case 30: sid |= c << 16; break;
          case 31: sid |= c << 8; break;
          case 32: sid |= c; break;
        }
        cur++;
        if (st < 32) { st++; continue; }
        }
        // FH_8 done: init_frame_parser (parsing.cc:255-308), DATA branch :340-397
        uint32_t status = 0;
        cur_parser = 0;
        if (ftype == 0) {
          if (h2_select_stream(&P, D, sid, &s_idx, lane)) {
            if (fflags & ~1u) status = 3;  // frame_data.cc:47-52
            else cur_parser = 1;
          }
        }
        H2_PUSH(EV_FRAME, ftype, fflags | (status << 8), sid, fsz, (uint32_t)s);
        if (fsz == 0) {
          H2_PUSH(EV_PAYLOAD, (uint32_t)cur, 0, 1, 0, (uint32_t)s);
          st = ST_FH0;
        } else if (fsz > max_frame) {
          err = 2;  // parsing.cc:195-205
        } else {
          st = ST_FRAME;
        }
        continue;
      }
      // FRAME: parsing.cc:215-250
      const uint64_t avail = len - cur;
      const uint64_t take = avail < fsz ? avail : fsz;
      const uint32_t is_last = take == fsz;
      H2_PUSH(EV_PAYLOAD, (uint32_t)cur, (uint32_t)take, is_last, 0, (uint32_t)s);
      if (cur_parser == 1 && h2_select_stream(&P, D, sid, &s_idx, lane)) {
        // grpc_deframe_unprocessed_incoming_frames, frame_data.cc:92-276
        uint64_t q = cur;
        const uint64_t end = cur + take;
        while (q < end && D.state != 6 && !overflow) {
          if (D.state == 0 && end - q >= 5 && q + 8 <= 32 &&
              (h2_bytes8(C, s, q, arena, slices, nslices, lane) & 0xFF) <= 1) {
            // the 5-byte message header in one step (frame_data.cc:112-176)
            const uint64_t b8 = h2_bytes8(C, s, q, arena, slices, nslices, lane);
            D.comp = (int32_t)(b8 & 0xFF);
            D.fsz = (uint32_t)((((b8 >> 8) & 0xFF) << 24) | (((b8 >> 16) & 0xFF) << 16) |
                               (((b8 >> 24) & 0xFF) << 8) | ((b8 >> 32) & 0xFF));
            H2_PUSH(EV_MSG_BEGIN, (uint32_t)D.comp, D.fsz, D.id, 0, (uint32_t)s);
            if (D.fsz == 0) {
              H2_PUSH(EV_MSG_END, 0, 0, D.id, 0, (uint32_t)s);
              D.state = 0;
            } else {
              D.state = 5;
            }
            q += 5;
          } else if (D.state < 5) {
            const uint32_t c = H2_BYTE(s, q);
            if (D.state == 0) {
              if (c > 1) {  // "Bad GRPC frame type", frame_data.cc:123-140: stream error
                D.state = 6;
                H2_PUSH(EV_FRAME, 0xff, 0, sid, 4, (uint32_t)s);
                break;
              }
              D.comp = (int32_t)c;
              D.state = 1;
            } else if (D.state == 1) { D.fsz = c << 24; D.state = 2; }
            else if (D.state == 2) { D.fsz |= c << 16; D.state = 3; }
            else if (D.state == 3) { D.fsz |= c << 8; D.state = 4; }
            else {
              D.fsz |= c;
              H2_PUSH(EV_MSG_BEGIN, (uint32_t)D.comp, D.fsz, D.id, 0, (uint32_t)s);
              if (D.fsz == 0) {
                H2_PUSH(EV_MSG_END, 0, 0, D.id, 0, (uint32_t)s);
                D.state = 0;
              } else {
                D.state = 5;
              }
            }
            q++;
          } else {
            const uint64_t rem = end - q;
            const uint64_t tk = rem < D.fsz ? rem : D.fsz;
            H2_PUSH(EV_MSG_BYTES, (uint32_t)q, (uint32_t)tk, D.id, 0, (uint32_t)s);
            D.fsz -= (uint32_t)tk;
            q += tk;
            if (D.fsz == 0) {
              H2_PUSH(EV_MSG_END, 0, 0, D.id, 0, (uint32_t)s);
              D.state = 0;
            }
          }
        }
      }
      fsz -= (uint32_t)take;
      cur += take;
      if (is_last) st = ST_FH0;
    }
    if (err || overflow) break;
  }
#undef H2_PUSH
#undef H2_BYTE
  h2_flush_stream(&P, D, lane);
  if (lane == 0) {
    P.state = st;
    P.incoming_frame_size = fsz;
    P.incoming_frame_type = ftype;
    P.incoming_frame_flags = fflags;
    P.incoming_stream_id = sid;
    P.cur_parser = cur_parser;
    P.error = err;
    res->nevents = nev;
    res->overflow = overflow;
    res->slices_done = s;
    res->error = err;
  }
  __syncthreads();
  for (unsigned i = lane; i < sizeof(grdma_h2_parser_dev) / 4; i += 64)
    reinterpret_cast<uint32_t*>(gp)[i] = reinterpret_cast<const uint32_t*>(&P)[i];
}

}  // namespace

// --------------------------------------------------------------------- host API
struct grdma_h2_parser {
  grdma_h2_parser_dev* d = nullptr;
};

static double g_h2_last_kernel_us = 0;

extern "C" {

const char* grdma_last_error(void);
// duration of the framing / deframing kernel of the last call (HIP events), microseconds
double grdma_h2_last_kernel_us(void) { return g_h2_last_kernel_us; }

int64_t grdma_h2_frame_messages(const grdma_h2_msg* msgs, uint64_t n, uint32_t max_frame,
                                grdma_slice* d_slices_out, uint64_t slices_cap,
                                void* d_hdr_arena, uint64_t hdr_cap, uint64_t* wire_bytes) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!msgs || !n || !d_slices_out || !d_hdr_arena || max_frame == 0 || max_frame >= (1u << 24))
    return -GRDMA_ERR_INVALID;
  std::vector<grdma_h2_msg_dev> tmp(n);
  for (uint64_t i = 0; i < n; i++) {
    tmp[i].payload = static_cast<const uint8_t*>(msgs[i].payload);
    tmp[i].len = msgs[i].len;
    tmp[i].stream_id = msgs[i].stream_id;
    tmp[i].flags = msgs[i].flags;
    if (msgs[i].len >= (1ull << 32)) return -GRDMA_ERR_INVALID;  // 32-bit message length field
  }
  grdma_h2_msg_dev* d_msgs = nullptr;
  grdma_h2_frame_result* d_res = nullptr;
  grdma_h2_frame_result h_res;
  int64_t rc = -GRDMA_ERR_HIP;
  if (hipMalloc((void**)&d_msgs, sizeof(grdma_h2_msg_dev) * n) == hipSuccess &&
      hipMalloc((void**)&d_res, sizeof(grdma_h2_frame_result)) == hipSuccess &&
      hipMemcpy(d_msgs, tmp.data(), sizeof(grdma_h2_msg_dev) * n, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemset(d_res, 0, sizeof(grdma_h2_frame_result)) == hipSuccess) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_h2_frame, dim3(1), dim3(256), 0, 0, d_msgs, n, max_frame,
                       reinterpret_cast<grdma_sge*>(d_slices_out), slices_cap,
                       static_cast<uint8_t*>(d_hdr_arena), hdr_cap, (uint64_t*)nullptr, d_res);
    hipEventRecord(e1, 0);
    const bool synced = hipDeviceSynchronize() == hipSuccess;
    float ms = 0;
    if (synced && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_h2_last_kernel_us = 1e3 * ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (synced &&
        hipMemcpy(&h_res, d_res, sizeof(h_res), hipMemcpyDeviceToHost) == hipSuccess) {
      if (h_res.overflow) rc = -GRDMA_ERR_CAPACITY;
      else {
        rc = (int64_t)h_res.nslices;
        if (wire_bytes) *wire_bytes = h_res.wire_bytes;
      }
    }
  }
  hipFree(d_msgs);
  hipFree(d_res);
  return rc;
}

grdma_h2_parser* grdma_h2_parser_create(int expect_client_prefix, uint32_t max_frame_size) {
  if (grdma_device_count() <= 0) return nullptr;
  grdma_h2_parser* p = new grdma_h2_parser();
  grdma_h2_parser_dev init;
  memset(&init, 0, sizeof(init));
  init.state = expect_client_prefix ? 0 : 24;
  init.max_frame_size = max_frame_size;  // http2_settings.cc:56 default 16384
  if (hipMalloc((void**)&p->d, sizeof(init)) != hipSuccess ||
      hipMemcpy(p->d, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
    delete p;
    return nullptr;
  }
  return p;
}

void grdma_h2_parser_destroy(grdma_h2_parser* p) {
  if (!p) return;
  hipFree(p->d);
  delete p;
}

int64_t grdma_h2_deframe(grdma_h2_parser* p, const void* d_arena, const grdma_read_slice* slices,
                         uint64_t n, grdma_h2_event* events_out, uint64_t cap, int* h2_error) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p || !d_arena || (!slices && n) || !events_out) return -GRDMA_ERR_INVALID;
  grdma_slice_out* d_sl = nullptr;
  grdma_h2_event* d_ev = nullptr;
  grdma_h2_deframe_result* d_res = nullptr;
  grdma_h2_deframe_result h_res;
  memset(&h_res, 0, sizeof(h_res));
  int64_t rc = -GRDMA_ERR_HIP;
  static_assert(sizeof(grdma_read_slice) == sizeof(grdma_slice_out), "layout");
  if (hipMalloc((void**)&d_sl, sizeof(grdma_slice_out) * (n ? n : 1)) == hipSuccess &&
      hipMalloc((void**)&d_ev, sizeof(grdma_h2_event) * (cap ? cap : 1)) == hipSuccess &&
      hipMalloc((void**)&d_res, sizeof(h_res)) == hipSuccess &&
      (n == 0 || hipMemcpy(d_sl, slices, sizeof(grdma_slice_out) * n, hipMemcpyHostToDevice) == hipSuccess)) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_h2_deframe, dim3(1), dim3(64), 0, 0, p->d,
                       static_cast<const uint8_t*>(d_arena), d_sl, n, d_ev, cap, d_res);
    hipEventRecord(e1, 0);
    const bool synced = hipDeviceSynchronize() == hipSuccess;
    float ms = 0;
    if (synced && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_h2_last_kernel_us = 1e3 * ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (synced &&
        hipMemcpy(&h_res, d_res, sizeof(h_res), hipMemcpyDeviceToHost) == hipSuccess) {
      const uint64_t m = h_res.nevents < cap ? h_res.nevents : cap;
      if (m == 0 || hipMemcpy(events_out, d_ev, sizeof(grdma_h2_event) * m, hipMemcpyDeviceToHost) == hipSuccess)
        rc = h_res.overflow ? -GRDMA_ERR_CAPACITY : (int64_t)m;
      if (h2_error) *h2_error = (int)h_res.error;
    }
  }
  hipFree(d_sl);
  hipFree(d_ev);
  hipFree(d_res);
  return rc;
}

}  // extern "C"
