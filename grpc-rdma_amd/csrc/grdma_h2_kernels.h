// Device side of the HTTP/2 DATA framing (K6/K7) and deframing (K8/K9): the structures the kernels keep
// in HBM and the kernels themselves (k_h2_frame_index, k_h2_frame_emit, k_h2_deframe and the chunked deframer, k_h2_table_ops).  Included by
// csrc/grdma_h2.hip, which holds the host API, and -- under the wave emulator of tests/cc/wave_emu.h -- by
// tests/cc/h2_emu_host.cc, which runs k_h2_deframe on the CPU against the oracle.
#ifndef GRDMA_H2_KERNELS_H
#define GRDMA_H2_KERNELS_H
#include "../../include/grdma_amd.h"
#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_h2_fast.h"

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

// One entry of the transport's stream map (grpc_chttp2_stream_map, internal.h) with what
// the deframe path reads: the per-stream grpc_chttp2_data_parser (frame_data.h),
// read_closed / write_closed (chttp2_transport.cc:2194-2244), header_frames_received.
// The map is an open-addressing table in HBM (linear probing from (id >> 1) & mask --
// stream ids of one side are consecutive odd numbers -- with backward-shift deletion, so
// it never fills with tombstones however many RPCs a connection has carried).
struct grdma_h2_stream_dev {
  uint32_t stream_id;   // 0 = empty slot
  int32_t state;        // 0..4 FH_0..FH_4, 5 FRAME, 6 ERROR
  uint32_t frame_size;
  uint8_t compressed, read_closed, write_closed, hdr_frames;
};
// The deframe fields of grpc_chttp2_transport (internal.h), one per connection, resident
// in HBM between calls.
struct grdma_h2_parser_dev {
  int32_t state;        // 0..23 client prefix, 24..32 FH_0..FH_8, 33 FRAME
  uint32_t incoming_frame_size;
  uint32_t incoming_frame_type;
  uint32_t incoming_frame_flags;
  uint32_t incoming_stream_id;
  uint32_t max_frame_size;
  int32_t cur_parser;   // 0 skip, 1 data, 2 header, 3 rst_stream, 4 a third header block without END_HEADERS
  int32_t is_server;
  int32_t is_first_frame;
  uint32_t expect_continuation;
  uint32_t header_eof, header_boundary, received_last_frame;
  uint32_t last_new_stream_id;
  uint32_t max_concurrent;
  uint32_t live_streams;
  uint32_t tab_mask;
  int32_t error;        // connection error (grdma_h2_error), sticky
  int32_t boundary_step;  // 1 = message starts go through h2_boundary_match (grdma_h2_fast.h)
  int32_t bulk_pairs;     // 1 = the bulk step gives every lane a frame (64 frames, 128 slices per step)
  int32_t ticks;          // 1 = the parsing wave samples the device clock around its phases (profiling aid)
  // what the data parser of the current stream held in front of the last slice the boundary step took (a slice in
  // which a message starts): the state a chunk of the parallel deframer is assumed to start in (h2_chunks below)
  int32_t hint_valid, hint_state;
  uint32_t hint_fsz, hint_id;
  grdma_h2_stream_dev* tab;
};

struct grdma_h2_deframe_result {
  uint64_t nevents;
  uint64_t overflow;
  uint64_t slices_done;
  int64_t error;
  uint64_t bulk_steps, bulk_frames;  // how much of the call went through the bulk step
  uint64_t t_wait, t_bulk, t_total, t_serial;  // profiling aid (s_memtime ticks): waiting for staged windows, inside bulk steps, whole parse, byte-wise path
  uint64_t boundary_steps, t_boundary;         // message starts taken by the boundary step, ticks inside it
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

// Framing in two launches.  The layout of a message is a closed form unless empty messages are around: a
// non-empty message whose predecessor is non-empty starts behind a by-reference slice (nothing to merge into), its
// first frame is one inlined slice [frame header 9 | message header 5] plus the first payload piece, every further
// frame a 9-byte inlined slice plus a payload piece -- 2 F slices, 32 F header-arena bytes, 9 F + 5 + len wire bytes
// for F = ceil((5 + len) / max_frame) frames (max_frame > 5).  A message that ends with an inlined slice (only an
// empty message does) lets the next header merge into it, so a run of empty messages and the message behind it are
// laid out sequentially by the thread of the run's first message, as the reference would.
//   k_h2_frame_index   one workgroup: per-message sizes (closed form, or the sequential walk around empty messages),
//                      block scans -> the first slice slot and header-arena offset of every message, the totals;
//   k_h2_frame_emit    one WAVE per message: lane j writes frame j, j + 64, ... (two 16-byte slice entries side by
//                      side, one 16-byte header store) -- 256 messages of 64 frames in a few microseconds where one
//                      thread per message took 150.
struct grdma_h2_msg_pos {
  uint64_t sl;    // first slice slot
  uint64_t hdr;   // first byte in the header arena
  uint32_t mode;  // 0 = laid out by the leader of its run, 1 = closed form, 2 = leader: sequential walk of the run
  uint32_t pad;
};

__device__ __forceinline__ bool h2_msg_plain(const grdma_h2_msg_dev* msgs, uint64_t i, uint32_t max_frame) {
  return max_frame > 5 && msgs[i].len != 0 && (i == 0 || msgs[i - 1].len != 0);
}

// sizes of message i and who lays it out (grdma_h2_msg_pos::mode)
__device__ __forceinline__ void h2_msg_size(const grdma_h2_msg_dev* msgs, uint64_t i, uint32_t max_frame, uint64_t* n_sl,
                                            uint64_t* n_hdr, uint64_t* n_wire, uint32_t* mode) {
  if (h2_msg_plain(msgs, i, max_frame)) {
    const uint64_t fcb = 5 + msgs[i].len;
    const uint64_t F = (fcb + max_frame - 1) / max_frame;
    *n_sl = 2 * F;
    *n_hdr = 32 * F;
    *n_wire = 9 * F + fcb;
    *mode = 1;
    return;
  }
  // incoming back-slice state: replay the run of empty messages in front
  uint64_t overflow = 0;
  uint64_t j = i;
  while (j > 0 && msgs[j - 1].len == 0) j--;
  frame_walk pre = {0, 0, 0, 0};
  for (; j < i; j++) walk_message<false>(msgs[j], max_frame, &pre, nullptr, nullptr, ~0ull, ~0ull, &overflow);
  frame_walk me = {0, 0, 0, pre.back_inl};
  walk_message<false>(msgs[i], max_frame, &me, nullptr, nullptr, ~0ull, ~0ull, &overflow);
  *n_sl = me.nslices;
  *n_hdr = me.hdr_off;
  *n_wire = me.wire;
  *mode = (i == 0 || msgs[i - 1].len != 0) ? 2u : 0u;
}

__global__ __launch_bounds__(256) void k_h2_frame_index(const grdma_h2_msg_dev* msgs, uint64_t nmsgs,
                                                        uint32_t max_frame, uint64_t cap, uint64_t hdr_cap,
                                                        grdma_h2_msg_pos* pos, grdma_h2_frame_result* res) {
  __shared__ uint64_t s_wave[4];
  const uint64_t tid = threadIdx.x;
  uint64_t base_sl = 0, base_hdr = 0, base_wire = 0;
  for (uint64_t m0 = 0; m0 < nmsgs; m0 += 256) {
    const uint64_t i = m0 + tid;
    uint64_t n_sl = 0, n_hdr = 0, n_wire = 0;
    uint32_t mode = 0;
    if (i < nmsgs) h2_msg_size(msgs, i, max_frame, &n_sl, &n_hdr, &n_wire, &mode);
    uint64_t tot_sl, tot_hdr, tot_wire;
    const uint64_t x_sl = block_excl_scan(n_sl, s_wave, &tot_sl);
    const uint64_t x_hdr = block_excl_scan(n_hdr, s_wave, &tot_hdr);
    block_excl_scan(n_wire, s_wave, &tot_wire);
    if (i < nmsgs) {
      grdma_h2_msg_pos q;
      q.sl = base_sl + x_sl;
      q.hdr = base_hdr + x_hdr;
      q.mode = mode;
      q.pad = 0;
      pos[i] = q;
    }
    base_sl += tot_sl;
    base_hdr += tot_hdr;
    base_wire += tot_wire;
  }
  if (tid == 0) {
    res->nslices = base_sl;
    res->hdr_bytes = base_hdr;
    res->wire_bytes = base_wire;
    res->overflow = (base_sl > cap || base_hdr > hdr_cap) ? 1 : 0;
  }
}

#define H2_EMIT_THREADS 256
// one wave lays out message i at position q
__device__ __forceinline__ void h2_emit_message(const grdma_h2_msg_dev* msgs, uint64_t i, uint64_t nmsgs, uint32_t max_frame,
                                                grdma_sge* out, uint64_t cap, uint8_t* hdr, uint64_t hdr_cap,
                                                const grdma_h2_msg_pos q, int lane) {
  if (q.mode == 2) {
    if (lane == 0) {
      uint64_t overflow = 0;  // (k_h2_frame_index has reported it; here it only keeps the stores inside the arrays)
      frame_walk me = {q.sl, q.hdr, 0, 0};
      for (uint64_t k = i;;) {
        walk_message<true>(msgs[k], max_frame, &me, out, hdr, cap, hdr_cap, &overflow);
        if (msgs[k].len != 0 || k + 1 >= nmsgs) break;
        k++;
      }
    }
    return;
  }
  if (q.mode != 1) return;
  const grdma_h2_msg_dev m = msgs[i];
  const uint64_t fcb = 5 + m.len;
  const uint64_t F = (fcb + max_frame - 1) / max_frame;
  for (uint64_t j = (uint64_t)lane; j < F; j += 64) {
    const uint64_t slot = q.sl + 2 * j, hoff = q.hdr + 32 * j;
    if (slot + 2 > cap || hoff + 32 > hdr_cap) break;  // (reported by k_h2_frame_index)
    const uint64_t done = j * max_frame;
    const uint64_t send = fcb - done < max_frame ? fcb - done : max_frame;
    const uint64_t last = ((m.flags & 2) && done + send == fcb) ? 1 : 0;
    // frame header, big-endian fields (frame_data.cc:73-82), then -- first frame -- the message header
    // (chttp2_transport.cc:1504-1509): byte k of the slot is byte k of the two words
    uint64_t w0 = ((send >> 16) & 0xFF) | (((send >> 8) & 0xFF) << 8) | ((send & 0xFF) << 16) | (last << 32) |
                  ((uint64_t)((m.stream_id >> 24) & 0xFF) << 40) | ((uint64_t)((m.stream_id >> 16) & 0xFF) << 48) |
                  ((uint64_t)((m.stream_id >> 8) & 0xFF) << 56);
    uint64_t w1 = (uint64_t)(m.stream_id & 0xFF);
    uint64_t hl = 9;
    if (j == 0) {
      w1 |= (uint64_t)((m.flags & 1) ? 1 : 0) << 8 | ((m.len >> 24) & 0xFF) << 16 | ((m.len >> 16) & 0xFF) << 24 |
            ((m.len >> 8) & 0xFF) << 32 | (m.len & 0xFF) << 40;
      hl = 14;
    }
    uint8_t* hp = hdr + hoff;
    u64x2 hv;
    hv.x = w0;
    hv.y = w1;
    if (((uint64_t)hp & 15) == 0) {
      *reinterpret_cast<u64x2*>(hp) = hv;  // (the slot is 32 bytes: the bytes behind the header are nobody's)
    } else {
      for (uint64_t k = 0; k < hl; k++) hp[k] = (uint8_t)((k < 8 ? w0 >> (8 * k) : w1 >> (8 * (k - 8))) & 0xFF);
    }
    u64x2 e0, e1;
    e0.x = (uint64_t)hp;
    e0.y = hl;
    e1.x = (uint64_t)(m.payload + (j == 0 ? 0 : done - 5));
    e1.y = j == 0 ? send - 5 : send;
    *reinterpret_cast<u64x2*>(&out[slot]) = e0;
    *reinterpret_cast<u64x2*>(&out[slot + 1]) = e1;
  }
}

__global__ __launch_bounds__(H2_EMIT_THREADS) void k_h2_frame_emit(const grdma_h2_msg_dev* msgs, uint64_t nmsgs,
                                                                  uint32_t max_frame, grdma_sge* out, uint64_t cap,
                                                                  uint8_t* hdr, uint64_t hdr_cap,
                                                                  const grdma_h2_msg_pos* pos) {
  const int lane = threadIdx.x & 63;
  const uint64_t i = (uint64_t)blockIdx.x * (H2_EMIT_THREADS / 64) + (threadIdx.x >> 6);
  if (i >= nmsgs) return;  // (wave-uniform)
  h2_emit_message(msgs, i, nmsgs, max_frame, out, cap, hdr, hdr_cap, pos[i], lane);
}

// Both steps in ONE launch for tables of up to H2_FRAME_ONE_MAX messages: every workgroup sums the sizes of the messages
// in front of its own four itself (closed form per message: a few hundred multiply-adds against a kernel boundary and a
// trip through memory for the positions), then its waves lay their messages out.  The workgroup of the last message
// reports the totals.
#define H2_FRAME_ONE_MAX 4096
__global__ __launch_bounds__(H2_EMIT_THREADS) void k_h2_frame_one(const grdma_h2_msg_dev* msgs, uint64_t nmsgs,
                                                                 uint32_t max_frame, grdma_sge* out, uint64_t cap,
                                                                 uint8_t* hdr, uint64_t hdr_cap, grdma_h2_frame_result* res) {
  __shared__ uint64_t s_part[H2_EMIT_THREADS / 64][3];
  __shared__ uint64_t s_mine[H2_EMIT_THREADS / 64][3];
  __shared__ uint32_t s_mode[H2_EMIT_THREADS / 64];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  constexpr uint32_t PER = H2_EMIT_THREADS / 64;
  const uint64_t i0 = (uint64_t)blockIdx.x * PER;
  // sizes of the messages in front of this workgroup's, one per thread per pass
  uint64_t a_sl = 0, a_hdr = 0, a_wire = 0;
  for (uint64_t j = threadIdx.x; j < i0; j += H2_EMIT_THREADS) {
    uint64_t n_sl, n_hdr, n_wire;
    uint32_t mode;
    h2_msg_size(msgs, j, max_frame, &n_sl, &n_hdr, &n_wire, &mode);
    a_sl += n_sl;
    a_hdr += n_hdr;
    a_wire += n_wire;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a_sl += __shfl_xor(a_sl, d, 64);
    a_hdr += __shfl_xor(a_hdr, d, 64);
    a_wire += __shfl_xor(a_wire, d, 64);
  }
  if (lane == 0) {
    s_part[wave][0] = a_sl;
    s_part[wave][1] = a_hdr;
    s_part[wave][2] = a_wire;
    // ... and of my own message
    uint64_t n_sl = 0, n_hdr = 0, n_wire = 0;
    uint32_t mode = 0;
    if (i0 + wave < nmsgs) h2_msg_size(msgs, i0 + wave, max_frame, &n_sl, &n_hdr, &n_wire, &mode);
    s_mine[wave][0] = n_sl;
    s_mine[wave][1] = n_hdr;
    s_mine[wave][2] = n_wire;
    s_mode[wave] = mode;
  }
  __syncthreads();
  uint64_t b_sl = 0, b_hdr = 0, b_wire = 0;
  for (uint32_t w = 0; w < PER; w++) {
    b_sl += s_part[w][0];
    b_hdr += s_part[w][1];
    b_wire += s_part[w][2];
  }
  for (uint32_t w = 0; w < wave; w++) {
    b_sl += s_mine[w][0];
    b_hdr += s_mine[w][1];
    b_wire += s_mine[w][2];
  }
  const uint64_t i = i0 + wave;
  if (i + 1 == nmsgs && lane == 0) {  // the last message's wave knows the totals
    const uint64_t t_sl = b_sl + s_mine[wave][0], t_hdr = b_hdr + s_mine[wave][1];
    res->nslices = t_sl;
    res->hdr_bytes = t_hdr;
    res->wire_bytes = b_wire + s_mine[wave][2];
    res->overflow = (t_sl > cap || t_hdr > hdr_cap) ? 1 : 0;
  }
  if (i >= nmsgs) return;  // (wave-uniform)
  grdma_h2_msg_pos q;
  q.sl = b_sl;
  q.hdr = b_hdr;
  q.mode = s_mode[wave];
  q.pad = 0;
  h2_emit_message(msgs, i, nmsgs, max_frame, out, cap, hdr, hdr_cap, q, lane);
}

// ---------------------------------------------------------------- RX deframing
enum { EV_FRAME = 1, EV_PAYLOAD = 2, EV_MSG_BEGIN = 3, EV_MSG_BYTES = 4, EV_MSG_END = 5,
       EV_STREAM_OPEN = 6, EV_STREAM_CLOSED = 7 };
enum { ST_FH0 = 24, ST_FRAME = 33 };
enum { PARSER_SKIP = 0, PARSER_DATA = 1, PARSER_HEADER = 2, PARSER_RST = 3, PARSER_HEADER3 = 4 };
enum { FT_DATA = 0, FT_HEADERS = 1, FT_RST_STREAM = 3, FT_SETTINGS = 4, FT_PING = 6, FT_GOAWAY = 7, FT_WINDOW_UPDATE = 8,
       FT_CONTINUATION = 9 };

// ---- stream map (single-lane code; callers broadcast the result) ----
__device__ __forceinline__ uint32_t tab_home(uint32_t id, uint32_t mask) { return (id >> 1) & mask; }

// grpc_chttp2_parsing_lookup_stream: plain lookup, never creates
__device__ int tab_find(const grdma_h2_stream_dev* tab, uint32_t mask, uint32_t id) {
  if (id == 0) return -1;
  uint32_t i = tab_home(id, mask);
  for (uint32_t n = 0; n <= mask; n++, i = (i + 1) & mask) {
    const uint32_t k = tab[i].stream_id;
    if (k == id) return (int)i;
    if (k == 0) return -1;
  }
  return -1;
}

// the caller keeps the table at most half full, so a free slot always exists
__device__ int tab_insert(grdma_h2_stream_dev* tab, uint32_t mask, uint32_t id) {
  uint32_t i = tab_home(id, mask);
  while (tab[i].stream_id != 0) i = (i + 1) & mask;
  grdma_h2_stream_dev e;
  e.stream_id = id; e.state = 0; e.frame_size = 0;
  e.compressed = e.read_closed = e.write_closed = e.hdr_frames = 0;
  tab[i] = e;
  return (int)i;
}

// backward-shift deletion: entries behind the hole move up while that keeps them
// reachable from their home slot
__device__ void tab_remove(grdma_h2_stream_dev* tab, uint32_t mask, uint32_t i) {
  uint32_t j = i;
  for (;;) {
    j = (j + 1) & mask;
    const grdma_h2_stream_dev e = tab[j];
    if (e.stream_id == 0) break;
    const uint32_t k = tab_home(e.stream_id, mask);
    // keep e where it is if its home k lies cyclically in (i, j]
    const bool stays = (i <= j) ? (i < k && k <= j) : (i < k || k <= j);
    if (stays) continue;
    tab[i] = e;
    i = j;
  }
  tab[i].stream_id = 0;
}

// Look-ahead ring in LDS.  Parsing a slice needs its descriptor and its first bytes -- two
// DEPENDENT memory round trips, far more than the few hundred cycles the parse takes.  So the
// kernel is a small pipeline: seven helper waves run ahead of the parsing wave and stage, window
// by window (64 slices, one per lane), {offset, length, first 32 bytes} of every slice in an LDS
// ring of sixteen windows; the parser finds whatever it looks at in LDS.  Hand-off per window:
// seq[slot] = window + 1 (release / acquire at workgroup scope); `consumed` (the window the parser
// is in) lets the helpers reuse slots.
#define H2_RING 16
struct h2_win_ent {
  uint64_t off, len, c0, c1, c2, c3;
};
struct h2_lds {
  h2_win_ent win[H2_RING][64];
  uint32_t seq[H2_RING];
  uint32_t consumed;
  uint32_t stop;
};
// File-scope LDS object, always named directly: an access through a generic pointer is a FLAT
// load, and a flat load's result can only be waited for with vmcnt(0) -- behind every event
// store still in flight, a memory round trip per look at the ring (measured: 3.7 us per slice
// of the byte-wise path).
__shared__ h2_lds g_h2;

__device__ __forceinline__ uint64_t h2_keep(uint64_t v, uint64_t first, uint64_t n) {
  // bytes at and beyond the slice end read as zero
  if (n >= first + 8) return v;
  if (n <= first) return 0;
  return v & ((1ull << ((n - first) * 8)) - 1);
}

// A helper wave: stages windows h, h + nh, h + 2 nh, ... of the slice list.
__device__ void h2_stage_windows(uint32_t h, uint32_t nh, const uint8_t* arena,
                                 const grdma_slice_out* slices, uint64_t nslices, int lane) {
  h2_lds* const L = &g_h2;
  for (uint64_t k = h; k * 64 < nslices; k += nh) {
    // the slot is free once the parser has left window k - H2_RING
    for (uint32_t spins = 0;; spins++) {
      const uint32_t cons = __hip_atomic_load(&L->consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t stop = __hip_atomic_load(&L->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (stop) return;
      if (k < (uint64_t)cons + H2_RING) break;
      __builtin_amdgcn_s_sleep(2);
    }
    const uint64_t mine = k * 64 + (uint64_t)lane;
    const bool have = mine < nslices;
    // (unconditional loads from clamped addresses: a load under a branch is followed by a full wait)
    const u64x2 d = *reinterpret_cast<const u64x2*>(&slices[have ? mine : nslices - 1]);
    const uint64_t off = have ? d.x : 0, n = have ? d.y : 0;
    // the aligned 16-byte blocks that hold the first 32 bytes of the slice (two when the slice
    // starts on a 16-byte boundary, three otherwise).  Nothing outside the blocks the slice
    // touches is read: a block beyond them is replaced by block 0 (an empty slice reads the
    // arena's first block).
    const uint8_t* p = arena + off;
    const uint64_t sh = n ? (uint64_t)p & 15 : 0;
    const u64x2* q = n ? reinterpret_cast<const u64x2*>((uint64_t)p & ~15ull)
                       : reinterpret_cast<const u64x2*>((uint64_t)arena & ~15ull);
    const uint64_t need = n ? (n < 32 ? n : 32) + sh : 0;
    const u64x2 v0 = q[0];
    const u64x2 v1 = q[need > 16 ? 1 : 0];
    const u64x2 v2 = q[need > 32 ? 2 : 0];
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
    h2_win_ent e;
    e.off = off;
    e.len = n;
    e.c0 = h2_keep(o0, 0, n);
    e.c1 = h2_keep(o1, 8, n);
    e.c2 = h2_keep(o2, 16, n);
    e.c3 = h2_keep(o3, 24, n);
    L->win[k % H2_RING][lane] = e;
    GRDMA_WAVE_CONVERGE();  // (every lane has stored its entry: on the GPU the store is one instruction of the wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every lane's entry before the flag
    if (lane == 0) __hip_atomic_store(&L->seq[k % H2_RING], (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// The parser's view: which windows it has seen arrive.
struct h2_view {
  uint64_t have_upto;  // windows [.., have_upto) are known to be staged (and not yet released)
  uint64_t t_wait;     // profiling aid
  bool ticks;          // sample the clock around waits
};
// the device clock, or 0 when the call does not collect phase ticks (a sample is a scalar memory operation
// the wave waits for: a dozen of them per message show in a loop that runs a few hundred instructions)
__device__ __forceinline__ uint64_t h2_clock(bool on) { return on ? __builtin_amdgcn_s_memtime() : 0; }

// make sure the window of slice s is staged (wave-uniform spin)
__device__ __forceinline__ void h2_need(h2_view& V, uint64_t s) {
  const uint64_t k = s >> 6;
  if (k < V.have_upto) return;
  const uint64_t t0 = h2_clock(V.ticks);
  for (;;) {
    const uint32_t q = __hip_atomic_load(&g_h2.seq[k % H2_RING], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (q == (uint32_t)(k + 1)) break;
    __builtin_amdgcn_s_sleep(1);
  }
  V.t_wait += h2_clock(V.ticks) - t0;
  V.have_upto = k + 1;
}
__device__ __forceinline__ const h2_win_ent* h2_ent(h2_view&, uint64_t s) {
  return &g_h2.win[(s >> 6) % H2_RING][s & 63];
}

__device__ __forceinline__ uint32_t h2_byte_at(h2_view& V, uint64_t s, uint64_t off, const uint8_t* arena) {
  h2_need(V, s);
  const h2_win_ent* e = h2_ent(V, s);
  if (off < 32) {
    const uint64_t q = off >> 3;
    const uint64_t word = q == 0 ? e->c0 : q == 1 ? e->c1 : q == 2 ? e->c2 : e->c3;
    return (uint32_t)((word >> ((off & 7) * 8)) & 0xFF);
  }
  return arena[e->off + off];
}

// bytes [off, off + 8) of slice s as a little-endian word; needs off + 8 <= 32 (inside the
// staged look-ahead)
__device__ __forceinline__ uint64_t h2_bytes8(h2_view& V, uint64_t s, uint64_t off) {
  h2_need(V, s);
  const h2_win_ent* e = h2_ent(V, s);
  const uint64_t q = off >> 3;
  const uint64_t lo = q == 0 ? e->c0 : q == 1 ? e->c1 : q == 2 ? e->c2 : e->c3;
  const uint64_t hi = q == 0 ? e->c1 : q == 1 ? e->c2 : e->c3;  // q == 3: unused
  const unsigned bs = (unsigned)(off & 7) * 8;
  return bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
}

// one event = three 8-byte stores (the array is 8-byte aligned, 24 bytes per event)
__device__ __forceinline__ void h2_store_event(grdma_h2_event* at, uint32_t kind, uint32_t a, uint32_t b, uint32_t c,
                                               uint32_t d, uint32_t sl) {
  auto* w = (__attribute__((address_space(1))) uint64_t*)(uint64_t)at;
  w[0] = (uint64_t)kind | ((uint64_t)a << 32);
  w[1] = (uint64_t)b | ((uint64_t)c << 32);
  w[2] = (uint64_t)d | ((uint64_t)sl << 32);
}
static_assert(sizeof(grdma_h2_event) == 24, "event layout");

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
    h2_store_event(ev + nev, kind, a, b, c, d, sl);
  }
  nev++;
}

// a wave-uniform value, moved to scalar registers
__device__ __forceinline__ uint32_t h2_uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t h2_uni64(uint64_t v) {
  return ((uint64_t)h2_uni32((uint32_t)(v >> 32)) << 32) | (uint64_t)h2_uni32((uint32_t)v);
}

// the map entry of the current stream, cached in (wave-uniform) registers
struct h2_cur_stream {
  int idx;
  uint32_t id, fsz;
  int32_t state, comp;
  uint32_t read_closed, write_closed, hdr_frames;
};

__device__ __forceinline__ void h2_flush_stream(grdma_h2_stream_dev* tab, const h2_cur_stream& D, int lane) {
  if (D.idx >= 0 && lane == 0) {
    grdma_h2_stream_dev e;
    e.stream_id = D.id; e.state = D.state; e.frame_size = D.fsz;
    e.compressed = (uint8_t)D.comp; e.read_closed = (uint8_t)D.read_closed;
    e.write_closed = (uint8_t)D.write_closed; e.hdr_frames = (uint8_t)D.hdr_frames;
    tab[D.idx] = e;
  }
}

// lookup (never creates): false = not in the map
__device__ __forceinline__ bool h2_select_stream(grdma_h2_stream_dev* tab, uint32_t mask, h2_cur_stream& D,
                                                 uint32_t id, int lane) {
  if (D.idx >= 0 && D.id == id) return true;
  h2_flush_stream(tab, D, lane);
  D.idx = -1;
  int idx = -1;
  if (lane == 0) idx = tab_find(tab, mask, id);
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (idx < 0) return false;
  const grdma_h2_stream_dev e = tab[idx];  // (uniform address: one load, broadcast)
  D.idx = idx;
  D.id = id;
  D.state = e.state;
  D.fsz = e.frame_size;
  D.comp = e.compressed;
  D.read_closed = e.read_closed;
  D.write_closed = e.write_closed;
  D.hdr_frames = e.hdr_frames;
  GRDMA_WAVE_CONVERGE();  // (every lane holds the entry before lane 0 may write it back)
  return true;
}

// grpc_chttp2_mark_stream_closed(t, s, close_reads, close_writes), chttp2_transport.cc:2194-2244:
// the stream leaves the map once both sides are closed.  Returns 1 if it left.
__device__ __forceinline__ uint32_t h2_mark_closed(grdma_h2_stream_dev* tab, uint32_t mask, h2_cur_stream& D,
                                                   uint32_t& live, bool close_writes, int lane) {
  // D is the stream being closed
  D.read_closed = 1;
  if (close_writes) D.write_closed = 1;
  const uint32_t gone = D.write_closed ? 1u : 0u;
  h2_flush_stream(tab, D, lane);
  if (gone) {
    if (lane == 0) tab_remove(tab, mask, (uint32_t)D.idx);
    live--;
    D.idx = -1;  // entries may have moved
  }
  return gone;
}

#define H2_DEFRAME_THREADS 512
// The deframer of one run of delivered slices: one parsing wave, seven staging waves (every thread of the
// 512-thread workgroup calls it; the staging waves return when the parser is done).
__device__ __forceinline__ void h2_deframe_body(grdma_h2_parser_dev* gp, const uint8_t* arena,
                                                const grdma_slice_out* slices, uint64_t nslices,
                                                grdma_h2_event* ev, uint64_t ev_cap,
                                                grdma_h2_deframe_result* res) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = threadIdx.x >> 6;
  if (threadIdx.x < H2_RING) g_h2.seq[threadIdx.x] = 0;
  if (threadIdx.x == 0) g_h2.consumed = g_h2.stop = 0;
  __syncthreads();  // (the only barrier: before the roles part ways)
  if (wave != 0) {
    h2_stage_windows(wave - 1, H2_DEFRAME_THREADS / 64 - 1, arena, slices, nslices, lane);
    return;
  }
  __builtin_amdgcn_s_setprio(3);  // the one serial wave of the kernel: it gets the issue slots it asks for
  const grdma_h2_parser_dev P = *gp;  // uniform loads: the whole block sits in scalar registers
  const bool ticks = P.ticks != 0;
  h2_view V = {0, 0, ticks};
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  uint64_t t_bulk = 0, t_serial = 0, t_boundary = 0;
  uint32_t consumed_pub = 0;
  uint64_t bulk_steps = 0, bulk_frames = 0, boundary_steps = 0;
  uint64_t nev = 0, overflow = 0;
  static const char kPrefix[] = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n";  // internal.h:781

  // The automaton state lives in (wave-uniform) registers for the whole call.
  int32_t st = P.state;
  uint32_t fsz = P.incoming_frame_size, ftype = P.incoming_frame_type;
  uint32_t fflags = P.incoming_frame_flags, sid = P.incoming_stream_id;
  int32_t cur_parser = P.cur_parser;
  int32_t is_first_frame = P.is_first_frame;
  uint32_t expect_cont = P.expect_continuation, header_eof = P.header_eof, header_boundary = P.header_boundary;
  uint32_t received_last = P.received_last_frame, last_new = P.last_new_stream_id, live = P.live_streams;
  const uint32_t max_frame = P.max_frame_size, mask = P.tab_mask, max_conc = P.max_concurrent;
  const bool is_server = P.is_server != 0;
  grdma_h2_stream_dev* const tab = P.tab;
  h2_cur_stream D = {-1, 0, 0, 0, 0, 0, 0, 0};
  int32_t hint_valid = P.hint_valid, hint_state = P.hint_state;
  uint32_t hint_fsz = P.hint_fsz, hint_id = P.hint_id;
#define H2_PUSH(kind, a, b, c, d, sl) h2_push(ev, ev_cap, nev, overflow, lane, kind, a, b, c, d, sl)
#define H2_BYTE(s_, off_) h2_byte_at(V, s_, off_, arena)
  // what the payload parser does with the last piece of a frame (frame_data.cc:299-305,
  // hpack_parser.cc:1746-1782, frame_rst_stream.cc:99-119)
#define H2_END_FRAME(sl)                                                                          \
  do {                                                                                            \
    if (cur_parser == PARSER_DATA) {                                                              \
      if (received_last && h2_select_stream(tab, mask, D, sid, lane)) {                           \
        const uint32_t gone = h2_mark_closed(tab, mask, D, live, false, lane);                    \
        H2_PUSH(EV_STREAM_CLOSED, gone, 0, sid, 0, (uint32_t)(sl));                               \
      }                                                                                           \
    } else if (cur_parser == PARSER_HEADER) {                                                     \
      if (header_boundary && h2_select_stream(tab, mask, D, sid, lane)) {                         \
        D.hdr_frames++;                                                                           \
        if (header_eof) {                                                                         \
          const uint32_t gone = h2_mark_closed(tab, mask, D, live, false, lane);                  \
          H2_PUSH(EV_STREAM_CLOSED, gone, 0, sid, 0, (uint32_t)(sl));                             \
        }                                                                                         \
      }                                                                                           \
    } else if (cur_parser == PARSER_RST) {                                                        \
      if (h2_select_stream(tab, mask, D, sid, lane)) {                                            \
        const uint32_t gone = h2_mark_closed(tab, mask, D, live, true, lane);                     \
        H2_PUSH(EV_STREAM_CLOSED, gone, 0, sid, 0, (uint32_t)(sl));                               \
      }                                                                                           \
    } else if (cur_parser == PARSER_HEADER3) {                                                    \
      err = 18; /* "Too many trailer frames", hpack_parser.cc:1756-1759 */                        \
    }                                                                                             \
  } while (0)

  uint64_t s = 0;
  int err = P.error;
  for (; s < nslices && !err && !overflow; s++) {
    if ((uint32_t)(s >> 6) != consumed_pub) {  // the parser left a window: its slot may be refilled
      consumed_pub = (uint32_t)(s >> 6);
      // Relaxed + a compiler barrier: LDS operations of one wave execute in order, so the reads of
      // the window left behind are done when this store lands.  (A release store would also wait
      // for every event store still in flight -- a memory round trip per window.)
      asm volatile("" ::: "memory");
      GRDMA_WAVE_CONVERGE();  // (the emulator's lanes: every lane has done its reads of the window left behind)
      if (lane == 0) __hip_atomic_store(&g_h2.consumed, consumed_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    h2_need(V, s);
    // ---- boundary step: the slice in which a message starts (grdma_h2_fast.h) -------------
    // Wave-uniform: the staged entry of slice s and the length of slice s + 1 are pulled into
    // scalar registers, the match runs on the scalar unit, and lane k stores event k.
    if (P.boundary_step && st == ST_FH0 && expect_cont == 0 && !is_first_frame && D.idx >= 0 && !D.read_closed &&
        (D.state == 0 || (D.state == 5 && D.fsz - 1u < 9u))) {
      const uint64_t tq0 = h2_clock(ticks);
      const h2_win_ent me = *h2_ent(V, s);
      uint64_t next_len = ~0ull;
      if (s + 1 < nslices) {
        h2_need(V, s + 1);
        next_len = h2_ent(V, s + 1)->len;
      }
      const uint32_t d_id = h2_uni32(D.id);
      const h2_bstep B = h2_boundary_match(h2_uni64(me.c0), h2_uni64(me.c1), h2_uni64(me.c2), h2_uni64(me.c3),
                                           h2_uni64(me.len), h2_uni64(next_len), (int32_t)h2_uni32((uint32_t)D.state),
                                           h2_uni32(D.fsz), d_id, max_frame);
      if (B.ok && nev + B.nev <= ev_cap) {
        hint_valid = 1;
        hint_state = (int32_t)h2_uni32((uint32_t)D.state);
        hint_fsz = h2_uni32(D.fsz);
        hint_id = d_id;
        if ((uint32_t)lane < B.nev) {
          uint32_t e[6];
          h2_boundary_event(B, d_id, (uint32_t)s, (uint32_t)lane, e);
          h2_store_event(ev + nev + (uint64_t)lane, e[0], e[1], e[2], e[3], e[4], e[5]);
        }
        nev += B.nev;
        D.state = B.rem ? 5 : 0;
        D.fsz = B.rem;
        D.comp = (int32_t)B.comp;
        // what the automaton's registers hold after frame B
        fsz = 0;
        ftype = FT_DATA;
        fflags = 0;
        sid = d_id;
        cur_parser = PARSER_DATA;
        received_last = 0;
        boundary_steps++;
        t_boundary += h2_clock(ticks) - tq0;
        s += (uint64_t)B.nslices - 1;  // (the loop adds the last one)
        continue;
      }
      t_boundary += h2_clock(ticks) - tq0;
    }
    // ---- bulk step: the streaming steady state, many frames at once ---------------------
    // Between a message's first and last frame every DATA frame of a stream spans exactly TWO
    // slices: on the sending side a 9-byte header slice and one payload slice
    // (frame_data.cc:64-90); on the receiving side -- where the endpoint sized its read to the
    // 9-byte record, max(256, 9) (rdma_bp_posix.cc:308) -- a 256-byte slice holding the header
    // and the first 247 payload bytes, then one slice with the rest.  In general: slice a = 9
    // header bytes + p0 >= 0 payload bytes, slice a + 1 = the remaining p1 > 0 bytes.
    // The slice cache holds descriptor + first bytes of the next 64 slices, one per lane: every
    // even lane (counted from s) checks "my slice starts such a frame, for the stream that is
    // mid-message, and the next slice ends it"; a ballot finds the verified prefix, a prefix
    // sum cuts it where the message ends, and each verified lane writes the events the byte-wise
    // automaton below would have produced for its frame.
    if (st == ST_FH0 && expect_cont == 0 && !is_first_frame && D.idx >= 0 && !D.read_closed && D.state == 5 &&
        D.fsz != 0) {
      const uint64_t tb0 = h2_clock(ticks);
      // lane i looks at slice s + i and every even lane owns a frame (the windows needed: the current
      // one and, when s is not window aligned, the next one -- staged, or beyond the end of the
      // list); with bulk_pairs lane i owns the frame in slices s + 2 i and s + 2 i + 1, 64 frames per
      // step over up to three windows (asked for in order: a window's flag says nothing about the
      // one before it, which another helper wave stages)
      const bool pairs = P.bulk_pairs != 0;
      if (pairs) h2_need(V, s + 64 < nslices ? s + 64 : nslices - 1);
      const uint64_t last_ix = s + (pairs ? 127 : 63) < nslices ? s + (pairs ? 127 : 63) : nslices - 1;
      h2_need(V, last_ix);
      const uint64_t my_raw = s + (pairs ? 2ull * (uint64_t)lane : (uint64_t)lane);
      const uint64_t my_ix = my_raw < nslices ? my_raw : nslices - 1;
      const uint64_t nx_ix = my_ix + 1 < nslices ? my_ix + 1 : nslices - 1;
      const h2_win_ent me = *h2_ent(V, my_ix);
      const bool hdr_lane = (pairs || (lane & 1) == 0) && my_raw + 1 < nslices;
      const uint64_t my_len = me.len;
      const uint64_t b8 = me.c0;
      const uint32_t fs = (uint32_t)(((b8 & 0xFF) << 16) | (((b8 >> 8) & 0xFF) << 8) | ((b8 >> 16) & 0xFF));
      const uint32_t sd = (uint32_t)((((b8 >> 40) & 0x7F) << 24) | (((b8 >> 48) & 0xFF) << 16) |
                                     (((b8 >> 56) & 0xFF) << 8) | (me.c1 & 0xFF));
      const uint64_t next_len = h2_ent(V, nx_ix)->len;
      const uint32_t p0 = (uint32_t)(my_len - 9);  // (only looked at when my_len >= 9)
      // type DATA, flags 0 (an END_STREAM frame closes the stream: left to the automaton)
      const bool ok = hdr_lane && my_len >= 9 && my_len < 9ull + fs && ((b8 >> 24) & 0xFFFF) == 0 &&
                      sd == D.id && fs <= max_frame && next_len == (uint64_t)fs - p0;
      const uint64_t bad = __ballot(hdr_lane && !ok);
      const int first_bad = bad ? __builtin_ctzll(bad) : 64;
      const bool cand = ok && lane < first_bad;
      // payload bytes / events up to and including my frame: two inclusive scans on the DPP network (six VALU
      // instructions each, no LDS; the twelve dependent bpermutes they replace were a good part of the step)
      const uint32_t cum = wave_incl_scan_u32(cand ? fs : 0u);
      const uint32_t epos = wave_incl_scan_u32(cand ? (p0 ? 5u : 3u) : 0u);
      const bool within = cand && cum <= D.fsz;  // (cum is monotone: the lanes within form a prefix)
      const uint64_t wmask = __ballot(within);
      if (wmask != 0) {
        const int last_lane = 63 - __builtin_clzll(wmask);
        // (last_lane comes from a ballot, a scalar: v_readlane instead of a trip through the LDS crossbar)
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)cum, last_lane);
        const uint32_t nevs = (uint32_t)__builtin_amdgcn_readlane((int)epos, last_lane);
        const bool ends = total == D.fsz;
        if (nev + nevs + 1 <= ev_cap) {
          if (within) {
            grdma_h2_event* e = ev + nev + epos - (p0 ? 5u : 3u);
            const uint32_t sl = (uint32_t)my_raw;
            h2_store_event(e, EV_FRAME, FT_DATA, 0, sd, fs, sl);
            if (p0) {
              h2_store_event(e + 1, EV_PAYLOAD, 9, p0, 0, 0, sl);
              h2_store_event(e + 2, EV_MSG_BYTES, 9, p0, sd, 0, sl);
              e += 2;
            }
            const uint32_t p1 = fs - p0;
            h2_store_event(e + 1, EV_PAYLOAD, 0, p1, 1, 0, sl + 1);
            h2_store_event(e + 2, EV_MSG_BYTES, 0, p1, sd, 0, sl + 1);
            if (ends && lane == last_lane) h2_store_event(e + 3, EV_MSG_END, 0, 0, sd, 0, sl + 1);
          }
          nev += nevs + (ends ? 1 : 0);
          D.fsz -= total;
          if (ends) D.state = 0;
          // what the automaton's registers hold after the last of these frames
          fsz = 0;
          ftype = FT_DATA;
          fflags = 0;
          sid = D.id;
          cur_parser = PARSER_DATA;
          received_last = 0;
          bulk_steps++;
          bulk_frames += (uint64_t)__builtin_popcountll(wmask);
          t_bulk += h2_clock(ticks) - tb0;
          s += 2ull * (uint64_t)__builtin_popcountll(wmask) - 1;  // (the loop adds the last one)
          continue;
        }
      }
    }
    const uint64_t ts0 = h2_clock(ticks);
    const uint64_t len = h2_ent(V, s)->len;
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
          const uint64_t b8 = h2_bytes8(V, s, cur);
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
        // FH_8 done: init_frame_parser (parsing.cc:255-306)
        uint32_t status = 0;
        bool opened = false, hdr_frame = false, is_cont = false;
        cur_parser = PARSER_SKIP;
        if (is_first_frame && ftype != FT_SETTINGS) { err = 8; break; }  // :256-263
        is_first_frame = 0;
        if (expect_cont != 0) {                                          // :265-283
          if (ftype != FT_CONTINUATION) { err = 5; break; }
          if (expect_cont != sid) { err = 6; break; }
          hdr_frame = is_cont = true;
        } else if (ftype == FT_DATA) {
          // init_data_frame_parser (:341-397) + grpc_chttp2_data_parser_begin_frame (frame_data.cc:43-62)
          if (h2_select_stream(tab, mask, D, sid, lane) && !D.read_closed) {
            if (fflags & ~1u) status = 3;  // frame_data.cc:47-52: stream error
            else {
              received_last = fflags & 1u;
              cur_parser = PARSER_DATA;
            }
          }
        } else if (ftype == FT_HEADERS) {
          hdr_frame = true;
        } else if (ftype == FT_CONTINUATION) {
          err = 7;                                                       // :287-289
          break;
        } else if (ftype == FT_RST_STREAM) {
          if (fsz != 4) { err = 10; break; }                             // frame_rst_stream.cc:73-79
          if (h2_select_stream(tab, mask, D, sid, lane)) cur_parser = PARSER_RST;
        } else if (ftype == FT_SETTINGS) {
          // init_settings_frame_parser (:732-757) + grpc_chttp2_settings_parser_begin_frame (frame_settings.cc:88-111)
          if (sid != 0) { err = 11; break; }
          if (fflags == 1u) {
            if (fsz != 0) { err = 12; break; }
          } else if (fflags != 0) {
            err = 13;
            break;
          } else if (fsz % 6 != 0) {
            err = 14;
            break;
          }
        } else if (ftype == FT_PING) {          // frame_ping.cc:58-64
          if ((fflags & 0xfeu) || fsz != 8) { err = 15; break; }
        } else if (ftype == FT_GOAWAY) {        // frame_goaway.cc:39-44
          if (fsz < 8) { err = 17; break; }
        } else if (ftype == FT_WINDOW_UPDATE) { // frame_window_update.cc:56-63 (before the stream is looked up)
          if (fflags || fsz != 4) { err = 16; break; }
        }  // the PAYLOADS of SETTINGS, WINDOW_UPDATE, PING, GOAWAY: control plane, skipped
        if (hdr_frame) {
          // init_header_frame_parser (parsing.cc:566-680): stream lookup / acceptance; the
          // HPACK bytes themselves are control plane and are skipped
          header_boundary = (fflags & 4u) ? 1u : 0u;
          expect_cont = header_boundary ? 0u : sid;
          if (!is_cont) header_eof = fflags & 1u;
          bool have = h2_select_stream(tab, mask, D, sid, lane);
          if (!have && !is_cont && is_server && last_new < sid && (sid & 1u)) {
            if (live >= max_conc || 2 * (live + 1) > mask + 1) { err = 9; break; }  // :623-627
            last_new = sid;                                              // :629-631 accept_stream
            h2_flush_stream(tab, D, lane);
            D.idx = -1;
            if (lane == 0) tab_insert(tab, mask, sid);
            live++;
            opened = true;
            have = h2_select_stream(tab, mask, D, sid, lane);
          }
          if (have && !D.read_closed && D.hdr_frames < 2) cur_parser = PARSER_HEADER;
          // (a third header block: the header parser in skipping mode with is_boundary = "END_HEADERS is missing"
          //  (parsing.cc:667-669, 318-327) -- the end of such a frame fails the connection, see H2_END_FRAME)
          else if (have && !D.read_closed && !header_boundary) cur_parser = PARSER_HEADER3;
        }
        H2_PUSH(EV_FRAME, ftype, fflags | (status << 8), sid, fsz, (uint32_t)s);
        if (opened) H2_PUSH(EV_STREAM_OPEN, 0, 0, sid, 0, (uint32_t)s);
        if (status == 3) {  // parsing.cc:388-391: the stream is closed for reads
          const uint32_t gone = h2_mark_closed(tab, mask, D, live, false, lane);
          H2_PUSH(EV_STREAM_CLOSED, gone, 0, sid, 0, (uint32_t)s);
        }
        if (fsz == 0) {
          H2_PUSH(EV_PAYLOAD, (uint32_t)cur, 0, 1, 0, (uint32_t)s);
          H2_END_FRAME(s);
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
      if (cur_parser == PARSER_DATA && h2_select_stream(tab, mask, D, sid, lane)) {
        // grpc_deframe_unprocessed_incoming_frames, frame_data.cc:92-276
        uint64_t q = cur;
        const uint64_t end = cur + take;
        while (q < end && D.state != 6 && !overflow) {
          if (D.state == 0 && end - q >= 5 && q + 8 <= 32 &&
              (h2_bytes8(V, s, q) & 0xFF) <= 1) {
            // the 5-byte message header in one step (frame_data.cc:112-176)
            const uint64_t b8 = h2_bytes8(V, s, q);
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
      if (is_last) {
        H2_END_FRAME(s);
        st = ST_FH0;
      }
    }
    t_serial += h2_clock(ticks) - ts0;
    if (err || overflow) break;
  }
#undef H2_PUSH
#undef H2_BYTE
#undef H2_END_FRAME
  GRDMA_WAVE_CONVERGE();  // (every lane has read the parser block before lane 0 replaces it)
  if (lane == 0) __hip_atomic_store(&g_h2.stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // helpers leave
  h2_flush_stream(tab, D, lane);
  if (lane == 0) {
    gp->state = st;
    gp->incoming_frame_size = fsz;
    gp->incoming_frame_type = ftype;
    gp->incoming_frame_flags = fflags;
    gp->incoming_stream_id = sid;
    gp->cur_parser = cur_parser;
    gp->is_first_frame = is_first_frame;
    gp->expect_continuation = expect_cont;
    gp->header_eof = header_eof;
    gp->header_boundary = header_boundary;
    gp->received_last_frame = received_last;
    gp->last_new_stream_id = last_new;
    gp->live_streams = live;
    gp->error = err;
    gp->hint_valid = hint_valid;
    gp->hint_state = hint_state;
    gp->hint_fsz = hint_fsz;
    gp->hint_id = hint_id;
    res->nevents = nev;
    res->overflow = overflow;
    res->slices_done = s;
    res->error = err;
    res->bulk_steps = bulk_steps;
    res->bulk_frames = bulk_frames;
    res->t_wait = V.t_wait;
    res->t_bulk = t_bulk;
    res->t_total = __builtin_amdgcn_s_memtime() - t_begin;
    res->t_serial = t_serial;
    res->boundary_steps = boundary_steps;
    res->t_boundary = t_boundary;
  }
}

__global__ __launch_bounds__(H2_DEFRAME_THREADS) void k_h2_deframe(grdma_h2_parser_dev* gp, const uint8_t* arena,
                                                                  const grdma_slice_out* slices, uint64_t nslices,
                                                                  grdma_h2_event* ev, uint64_t ev_cap,
                                                                  grdma_h2_deframe_result* res) {
  h2_deframe_body(gp, arena, slices, nslices, ev, ev_cap, res);
}

// ------------------------------------------------------------------------------------------------------------
// The deframer over CHUNKS of the delivered slices (round 3).  parsing.cc is a byte automaton: one wave walks
// 33 000 slices in 0.8 ms however wide the machine is.  In a streaming call, though, the parser is in the SAME
// state in front of every slice in which a message starts (frame boundary; the stream's data parser either at a
// message header -- sending side -- or a few bytes short of the message end -- receiving side): the state the
// boundary step (grdma_h2_fast.h) records as it goes (grdma_h2_parser_dev::hint_*).  So, in two launches:
//   k_h2_deframe_chunks   K workgroups.  Workgroup k finds its own two cuts -- the first slice at or behind the
//                      k-th and the (k+1)-th K-quantile of the list in which a message starts for the hinted stream
//                      (h2_boundary_match on the staged bytes; cut 0 = 0, cut K = the end) --, copies the parser
//                      block and the stream map into private ones (chunks 1.. in the hinted state), runs the
//                      ordinary deframer (h2_deframe_body) over its chunk into a private event segment, and checks
//                      that its stream map is the one it started with;
//   k_h2_merge_or_deframe   every workgroup verifies the CHAIN -- chunk k must have ended, without error, exactly
//                      where and in exactly the state chunk k + 1 was assumed to start -- and only then the events
//                      are concatenated (slice indices rebased) and the last chunk's parser block and stream map
//                      installed; when anything did not hold (no hint yet, a cut that was no frame boundary, a
//                      stream that opened or closed, an error, events that did not fit) workgroup 0 runs the ordinary
//                      deframer over the whole list: nothing of the chunks' work was visible, the result is the
//                      sequential one.
// The events, the parser state and the stream map are those of the sequential parse by construction.
// ------------------------------------------------------------------------------------------------------------
#define H2_KMAX 256
#define H2_CHUNK_MIN_SLICES 2048   // lists shorter than this go to the sequential deframer directly
#define H2_CHUNK_SLICES 128        // a chunk is at least this long (K = min(wanted, nslices / this)): one 1 MiB message
#define H2_MERGE_GRID 192
struct grdma_h2_chunks {
  uint32_t K;          // chunks of this call, 0 = no plan
  uint32_t ok;         // k_h2_merge_or_deframe: 1 = merged
  uint32_t hint_idx;   // slot of the hinted stream in the stream map
  uint32_t want;       // chunks the host asks for (<= H2_KMAX)
  uint64_t s_begin[H2_KMAX + 1];
  grdma_h2_parser_dev ref;            // the parser block the call started from
  grdma_h2_stream_dev ref_entry;      // ... and the hinted stream's entry: what chunks 1.. are assumed to start with
  grdma_h2_parser_dev gp[H2_KMAX];
  grdma_h2_deframe_result res[H2_KMAX];
  grdma_h2_stream_dev end_entry[H2_KMAX];
  uint32_t clean[H2_KMAX];            // bit 0 = both cuts found, bit 1 = every other map entry as it was
  uint64_t n_planned, n_merged;       // calls that were planned / merged since the parser was created
  grdma_h2_stream_dev* tabs;          // [H2_KMAX][slots] private stream maps
  grdma_h2_event* ev_tmp;             // ev_total events: K private segments of ev_total / K
  uint64_t ev_total;
  uint32_t slots, pad;
  uint64_t dbg[H2_KMAX + 1][8];       // s_memtime stamps of the phases (profiling aid; row H2_KMAX: the merge)
};

// the first 32 bytes of a slice as four little-endian words (zero beyond the slice): what the staging waves put
// into the look-ahead ring, for one slice
__device__ __forceinline__ void h2_first32(const uint8_t* arena, uint64_t off, uint64_t n, uint64_t c[4]) {
  const uint8_t* p = arena + off;
  const uint64_t sh = n ? (uint64_t)p & 15 : 0;
  const u64x2* q = n ? reinterpret_cast<const u64x2*>((uint64_t)p & ~15ull)
                     : reinterpret_cast<const u64x2*>((uint64_t)arena & ~15ull);
  const uint64_t need = n ? (n < 32 ? n : 32) + sh : 0;
  const u64x2 v0 = q[0];
  const u64x2 v1 = q[need > 16 ? 1 : 0];
  const u64x2 v2 = q[need > 32 ? 2 : 0];
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
  c[0] = h2_keep(o0, 0, n);
  c[1] = h2_keep(o1, 8, n);
  c[2] = h2_keep(o2, 16, n);
  c[3] = h2_keep(o3, 24, n);
}

// how many chunks a list of nslices is cut into (every workgroup and the merge compute the same number)
__device__ __forceinline__ uint32_t h2_chunk_count(uint32_t want, uint64_t nslices) {
  uint64_t K = want < H2_KMAX ? want : H2_KMAX;
  const uint64_t fit = nslices / H2_CHUNK_SLICES;
  if (K > fit) K = fit;
  return K >= 2 ? (uint32_t)K : 0u;
}

// The first slice at or behind `from` in which a message starts for the hinted stream (one wave; ~0 = none within
// `span` slices; the end of the list when the list ends first).  The match is the boundary step's own (grdma_h2_fast.h): a slice it would take from the hinted state.
__device__ __forceinline__ uint64_t h2_find_cut(const grdma_h2_parser_dev& P, const uint8_t* arena,
                                                const grdma_slice_out* slices, uint64_t nslices, uint64_t from,
                                                uint64_t span, int lane) {
  const uint64_t end = from + span < nslices ? from + span : nslices;
  // three windows of 64 slices per pass: descriptors, then first bytes -- two memory round trips for 192 slices
  for (uint64_t base = from; base < end; base += 192) {
    u64x2 d[3], dn[3];
    bool have[3];
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const uint64_t sx = base + 64u * w + (uint64_t)lane;
      have[w] = sx < end;
      d[w] = *reinterpret_cast<const u64x2*>(&slices[have[w] ? sx : nslices - 1]);
      dn[w] = *reinterpret_cast<const u64x2*>(&slices[sx + 1 < nslices ? sx + 1 : nslices - 1]);
    }
    uint64_t c[3][4];
#pragma unroll
    for (int w = 0; w < 3; w++) h2_first32(arena, d[w].x, have[w] ? d[w].y : 0, c[w]);
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const uint64_t sx = base + 64u * w + (uint64_t)lane;
      const h2_bstep B = h2_boundary_match(c[w][0], c[w][1], c[w][2], c[w][3], have[w] ? d[w].y : 0,
                                           sx + 1 < nslices ? dn[w].y : ~0ull, P.hint_state, P.hint_fsz, P.hint_id,
                                           P.max_frame_size);
      const uint64_t hit = __ballot(have[w] && B.ok);
      if (hit) return base + 64u * w + (uint64_t)__builtin_ctzll(hit);
    }
  }
  return end == nslices ? nslices : ~0ull;  // (no message starts behind `from`: the chunk in front runs to the end)
}

// Stream-map copies and comparisons of a whole workgroup, eight 16-byte entries per thread in flight (a plain loop is a
// load -> store chain: the compiler cannot know the two maps do not alias).  patch_idx: the entry that is stored with
// (state, frame_size) replaced (~0 = none); skip_idx: the entry a comparison leaves out.
template <uint32_t NT = H2_DEFRAME_THREADS>  // threads taking part (tid < NT)
__device__ __forceinline__ void h2_tab_copy(grdma_h2_stream_dev* dst, const grdma_h2_stream_dev* src, uint32_t slots,
                                            uint32_t tid, uint32_t patch_idx, int32_t p_state, uint32_t p_fsz) {
  static_assert(sizeof(grdma_h2_stream_dev) == 16, "two words per map entry");
  for (uint32_t base = 0; base < slots; base += 8 * NT) {
    u64x2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t e = base + (uint32_t)j * NT + tid;
      v[j] = *reinterpret_cast<const u64x2*>(&src[e < slots ? e : slots - 1]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t e = base + (uint32_t)j * NT + tid;
      if (e >= slots) continue;
      if (e == patch_idx) {
        grdma_h2_stream_dev t = src[e];
        t.state = p_state;
        t.frame_size = p_fsz;
        dst[e] = t;
      } else {
        *reinterpret_cast<u64x2*>(&dst[e]) = v[j];
      }
    }
  }
}
__device__ __forceinline__ bool h2_tab_differs(const grdma_h2_stream_dev* a, const grdma_h2_stream_dev* b, uint32_t slots,
                                               uint32_t tid, uint32_t skip_idx) {
  bool diff = false;
  for (uint32_t base = 0; base < slots; base += 8 * H2_DEFRAME_THREADS) {
    u64x2 x[8], y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t e = base + (uint32_t)j * H2_DEFRAME_THREADS + tid;
      const uint32_t c = e < slots ? e : slots - 1;
      x[j] = *reinterpret_cast<const u64x2*>(&a[c]);
      y[j] = *reinterpret_cast<const u64x2*>(&b[c]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t e = base + (uint32_t)j * H2_DEFRAME_THREADS + tid;
      if (e < slots && e != skip_idx) diff |= x[j].x != y[j].x || x[j].y != y[j].y;
    }
  }
  return diff;
}

__global__ __launch_bounds__(H2_DEFRAME_THREADS) void k_h2_deframe_chunks(grdma_h2_parser_dev* gp, grdma_h2_chunks* ctl,
                                                                         const uint8_t* arena, const grdma_slice_out* slices,
                                                                         uint64_t nslices) {
  const uint32_t k = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const int lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint64_t tk0 = __builtin_amdgcn_s_memtime();
  const grdma_h2_parser_dev P = *gp;  // (nothing writes the connection's block before the merge)
  const uint32_t slots = P.tab_mask + 1;
  const uint32_t K = h2_chunk_count(ctl->want, nslices);
  const bool can = K >= 2 && P.error == 0 && P.hint_valid && P.boundary_step && P.state == ST_FH0 &&
                   P.expect_continuation == 0 && !P.is_first_frame && slots == ctl->slots && nslices < (1ull << 31);
  if (!can) {  // (uniform over the grid: the merge finds K = 0 and runs the sequential deframer)
    if (k == 0 && tid == 0) {
      ctl->K = 0;
      ctl->ok = 0;
    }
    return;
  }
  if (k >= K) return;
  __shared__ uint64_t s_cut[2];
  __shared__ int s_idx;
  __shared__ uint32_t s_dirty;
  // my two cuts (wave 0: where I start, wave 1: where the next chunk starts), the hinted stream's slot (wave 2)
  if (wave < 2) {
    const uint32_t j = k + wave;
    const uint64_t span = 8 * (nslices / K) > 2048 ? 8 * (nslices / K) : 2048;  // (messages of up to ~16 MiB at 16 KiB frames)
    uint64_t cut = j == 0 ? 0 : j == K ? nslices : h2_find_cut(P, arena, slices, nslices, nslices * j / K, span, lane);
    if (lane == 0) s_cut[wave] = cut;
  } else {
    // waves 2-7 meanwhile: the private stream map (a plain copy; the hinted entry is patched below), and the hinted
    // stream's slot
    grdma_h2_stream_dev* const mine0 = ctl->tabs + (size_t)k * slots;
    h2_tab_copy<H2_DEFRAME_THREADS - 128>(mine0, P.tab, slots, tid - 128, ~0u, 0, 0);
    if (tid == 128) {
      s_idx = tab_find(P.tab, P.tab_mask, P.hint_id);
      s_dirty = 0;
    }
  }
  __syncthreads();
  const uint64_t tk1 = __builtin_amdgcn_s_memtime();
  const uint64_t s0 = s_cut[0], s1 = s_cut[1];
  const int idx = s_idx;
  const bool found = s0 != ~0ull && s1 != ~0ull && idx >= 0;  // (s0 <= s1: both are "first at or behind" a quantile)
  grdma_h2_stream_dev* const mine = ctl->tabs + (size_t)k * slots;
  if (found && k >= 1 && tid == 64) {  // chunks 1..: the hinted stream's data parser in the hinted state
    grdma_h2_stream_dev t = P.tab[idx];
    t.state = P.hint_state;
    t.frame_size = P.hint_fsz;
    mine[idx] = t;
  }
  if (tid == 0) {
    grdma_h2_parser_dev g = P;  // (chunks 1..: P itself IS at a frame boundary with nothing pending -- see `can`)
    g.tab = mine;
    ctl->gp[k] = g;
    ctl->s_begin[k] = found ? s0 : 0;
    ctl->clean[k] = 0;
    if (k == 0) {
      ctl->s_begin[K] = nslices;
      ctl->ref = P;
      if (idx >= 0) ctl->ref_entry = P.tab[idx];
      ctl->hint_idx = (uint32_t)(idx >= 0 ? idx : 0);
      ctl->n_planned++;
      ctl->ok = 0;
      ctl->K = K;
    }
  }
  if (!found) return;  // (uniform; clean[k] = 0 fails the chain)
  // (hand-offs inside one workgroup: the barrier orders them -- an agent-scope release here would write the XCD's
  // whole L2 back, dirty lines of the copy kernels included, once per workgroup)
  __syncthreads();
  const uint64_t tk2 = __builtin_amdgcn_s_memtime();
  if (s1 > s0) {
    const uint64_t stride = ctl->ev_total / K;
    h2_deframe_body(&ctl->gp[k], arena, slices + s0, s1 - s0, ctl->ev_tmp + (size_t)k * stride, stride, &ctl->res[k]);
  } else if (tid == 0) {
    // an empty chunk (its quantile and the next one's lie in front of the same message start): it ends where and as
    // it starts, which is what the next chunk assumes
    grdma_h2_deframe_result r;
    r.nevents = r.overflow = r.slices_done = 0;
    r.error = 0;
    r.bulk_steps = r.bulk_frames = r.t_wait = r.t_bulk = r.t_total = r.t_serial = r.boundary_steps = r.t_boundary = 0;
    ctl->res[k] = r;
  }
  // every wave is back (the staging waves when the parser said stop); the parser's stores to its map are its own
  // wave's: the barrier makes them visible to the comparing threads
  __syncthreads();
  const uint64_t tk3 = __builtin_amdgcn_s_memtime();
  if (h2_tab_differs(mine, P.tab, slots, tid, (uint32_t)idx)) s_dirty = 1;
  __syncthreads();
  if (tid == 0) {
    ctl->clean[k] = 1u | (s_dirty ? 0u : 2u);
    ctl->end_entry[k] = mine[idx];
    uint64_t* d = ctl->dbg[k];
    d[0] = tk0; d[1] = tk1; d[2] = tk2; d[3] = tk3; d[4] = __builtin_amdgcn_s_memtime(); d[5] = s1 - s0;
  }
}

// is the end of chunk k the start the chunk behind it was given?
__device__ __forceinline__ bool h2_chunk_link_ok(const grdma_h2_chunks* ctl, uint32_t k) {
  const grdma_h2_deframe_result& r = ctl->res[k];
  const grdma_h2_parser_dev& g = ctl->gp[k];
  const grdma_h2_parser_dev& R = ctl->ref;
  const grdma_h2_stream_dev& e = ctl->end_entry[k];
  const grdma_h2_stream_dev& re = ctl->ref_entry;
  if (ctl->clean[k] != 3u) return false;
  if (r.error != 0 || r.overflow != 0 || r.slices_done != ctl->s_begin[k + 1] - ctl->s_begin[k]) return false;
  if (g.state != ST_FH0 || g.expect_continuation != 0 || g.is_first_frame != 0 || g.error != 0) return false;
  if (g.last_new_stream_id != R.last_new_stream_id || g.live_streams != R.live_streams) return false;
  if (e.stream_id != R.hint_id || e.state != R.hint_state || (e.state == 5 && e.frame_size != R.hint_fsz)) return false;
  if (e.compressed != re.compressed || e.read_closed != re.read_closed || e.write_closed != re.write_closed ||
      e.hdr_frames != re.hdr_frames)
    return false;
  return true;
}

__global__ __launch_bounds__(H2_DEFRAME_THREADS) void k_h2_merge_or_deframe(grdma_h2_parser_dev* gp, grdma_h2_chunks* ctl,
                                                                           const uint8_t* arena, const grdma_slice_out* slices,
                                                                           uint64_t nslices, grdma_h2_event* ev, uint64_t ev_cap,
                                                                           grdma_h2_deframe_result* res) {
  const uint32_t tid = threadIdx.x;
  const uint64_t tm0 = __builtin_amdgcn_s_memtime();
  const uint32_t K = ctl->K;
  __shared__ uint64_t s_pre[H2_KMAX + 1];
  __shared__ uint32_t s_bad;
  // every workgroup verifies the chain itself (a few KB of control data): no inter-workgroup wait.  The last chunk
  // that holds slices may end in any state (it is the call's end); the ones in front of it must link.
  if (tid == 0) s_bad = K == 0 ? 1u : 0u;
  __syncthreads();
  if (K != 0) {
    __shared__ uint32_t s_last;
    __shared__ unsigned long long s_stat[4];
    if (tid < 4) s_stat[tid] = 0;
    if (tid == 64) {  // the last chunk that holds slices (chunk 0 always does)
      uint32_t last = K - 1;
      while (last > 0 && ctl->s_begin[last] == ctl->s_begin[last + 1]) last--;
      s_last = last;
    }
    if (tid <= H2_KMAX) s_pre[tid] = 0;
    __syncthreads();
    const uint32_t last = s_last;
    // what the tail of the merge needs, asked for NOW (registers): the control data was written by the chunk kernel on
    // other XCDs, every dependent load is a trip to memory -- a chain of five of them behind the event copy was half
    // of this kernel
    const bool installer = blockIdx.x == 0 && tid == 0;
    grdma_h2_parser_dev g_last;
    grdma_h2_deframe_result r_last;
    grdma_h2_stream_dev e_last;
    uint32_t clean_last = 0, hint_slot = 0;
    uint64_t n_merged0 = 0;
    grdma_h2_stream_dev* orig = nullptr;
    uint64_t st_bulk = 0, st_frames = 0, st_bsteps = 0, st_total = 0;
    if (blockIdx.x == 0) {
      clean_last = ctl->clean[last];
      orig = ctl->ref.tab;
      if (installer) {
        g_last = ctl->gp[last];
        r_last = ctl->res[last];
        e_last = ctl->end_entry[last];
        hint_slot = ctl->hint_idx;
        n_merged0 = ctl->n_merged;
      }
    }
    if (tid < K) {
      bool ok;
      if (tid < last) ok = h2_chunk_link_ok(ctl, tid);
      else if (tid == last) ok = (ctl->clean[tid] & 1u) && ctl->res[tid].error == 0 && ctl->res[tid].overflow == 0 &&
                                ctl->res[tid].slices_done == ctl->s_begin[tid + 1] - ctl->s_begin[tid];
      else ok = (ctl->clean[tid] & 1u) != 0;  // empty chunks behind the last one
      if (!ok) s_bad = 1;
      s_pre[tid + 1] = ok ? ctl->res[tid].nevents : 0;
      if (blockIdx.x == 0) {
        const grdma_h2_deframe_result& c = ctl->res[tid];
        st_bulk = c.bulk_steps;
        st_frames = c.bulk_frames;
        st_bsteps = c.boundary_steps;
        st_total = c.t_total;
      }
    }
    __syncthreads();
    if (tid < 64) {  // inclusive prefix sums of the event counts: H2_KMAX / 64 wave scans with a carry
      static_assert(H2_KMAX % 64 == 0, "whole waves");
      uint64_t carry = 0;
#pragma unroll
      for (uint32_t c = 0; c < H2_KMAX / 64; c++) {
        const uint64_t inc = wave_incl_scan(s_pre[c * 64 + tid + 1], (int)tid) + carry;
        s_pre[c * 64 + tid + 1] = inc;
        carry = __shfl(inc, 63, 64);
      }
      if (tid == 63 && carry > ev_cap) s_bad = 1;  // (entries behind K are zero: the last carry is the total)
    }
    __syncthreads();
    const uint64_t tm1 = __builtin_amdgcn_s_memtime();
    if (!s_bad) {
      const uint64_t total = s_pre[K];
      const uint64_t stride = ctl->ev_total / K;
      const grdma_h2_event* const ev_tmp = ctl->ev_tmp;
      const uint64_t per = (total + gridDim.x - 1) / gridDim.x;
      const uint64_t i0 = (uint64_t)blockIdx.x * per, i1 = i0 + per < total ? i0 + per : total;
      for (uint64_t i = i0 + tid; i < i1; i += H2_DEFRAME_THREADS) {
        uint32_t lo = 0, hi = K;  // the chunk with s_pre[k] <= i < s_pre[k + 1]
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (s_pre[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t k = lo;
        const uint64_t* src = reinterpret_cast<const uint64_t*>(ev_tmp + (size_t)k * stride + (i - s_pre[k]));
        const uint64_t w0 = src[0], w1 = src[1];
        uint64_t w2 = src[2];
        w2 += ctl->s_begin[k] << 32;  // the slice index of an event is its high half: rebase it on the chunk's first slice
        uint64_t* dst = reinterpret_cast<uint64_t*>(ev + i);
        dst[0] = w0;
        dst[1] = w1;
        dst[2] = w2;
      }
      const uint64_t tm2 = __builtin_amdgcn_s_memtime();
      if (blockIdx.x == 0) {
        // the last chunk's parser block and stream map are the connection's
        const uint32_t slots = ctl->slots;
        if (clean_last == 3u) {  // nothing but the hinted stream's entry moved in the last chunk either
          if (installer) orig[hint_slot] = e_last;
        } else {
          h2_tab_copy(orig, ctl->tabs + (size_t)last * slots, slots, tid, ~0u, 0, 0);
        }
        if (tid < K) {  // the call's counters: sums over the chunks, the longest chunk's clock (they ran side by side)
          atomicAdd(&s_stat[0], (unsigned long long)st_bulk);
          atomicAdd(&s_stat[1], (unsigned long long)st_frames);
          atomicAdd(&s_stat[2], (unsigned long long)st_bsteps);
          atomicMax(&s_stat[3], (unsigned long long)st_total);
        }
        __syncthreads();
        if (installer) {
          g_last.tab = orig;
          *gp = g_last;
          r_last.nevents = total;
          r_last.slices_done = nslices;
          r_last.bulk_steps = s_stat[0];
          r_last.bulk_frames = s_stat[1];
          r_last.boundary_steps = s_stat[2];
          r_last.t_total = s_stat[3];
          *res = r_last;
          ctl->n_merged = n_merged0 + 1;
          ctl->ok = 1;
          uint64_t* d = ctl->dbg[H2_KMAX];
          d[0] = tm0; d[1] = tm1; d[2] = tm2; d[3] = __builtin_amdgcn_s_memtime();
        }
      }
      return;
    }
  }
  // not merged: the ordinary deframer over the whole list, on the connection's own block and map
  if (blockIdx.x != 0) return;
  h2_deframe_body(gp, arena, slices, nslices, ev, ev_cap, res);
}

// What the surface does to the stream map outside the read path, batched: op 1 = a client starts
// a call on stream id (grpc_chttp2_stream_map_add), op 2 = the write side of a stream closes
// (grpc_chttp2_mark_stream_closed(close_writes)); a stream already read-closed then leaves the map.
struct grdma_h2_table_op { uint32_t op, id; int32_t rc, pad; };
__global__ void k_h2_table_ops(grdma_h2_parser_dev* gp, grdma_h2_table_op* ops, uint32_t n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  grdma_h2_stream_dev* tab = gp->tab;
  const uint32_t mask = gp->tab_mask;
  uint32_t live = gp->live_streams;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t id = ops[i].id;
    int rc = -1;
    const int idx = tab_find(tab, mask, id);
    if (ops[i].op == 1) {
      if (id != 0 && idx < 0 && 2 * (live + 1) <= mask + 1) {
        tab_insert(tab, mask, id);
        live++;
        rc = 0;
      }
    } else if (ops[i].op == 2) {
      if (idx >= 0) {
        tab[idx].write_closed = 1;
        if (tab[idx].read_closed) {
          tab_remove(tab, mask, (uint32_t)idx);
          live--;
        }
        rc = 0;
      }
    }
    ops[i].rc = rc;
  }
  gp->live_streams = live;
}

}  // namespace

#endif  // GRDMA_H2_KERNELS_H
