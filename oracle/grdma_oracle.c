/* TEST INFRASTRUCTURE ONLY -- see grdma_oracle.h for the rules.
 *
 * Plain-C restatement of the reference's CPU algorithms on the RDMA_BP/BPEV
 * endpoint hot path.  Parity is pinned against the reference-built
 * oracle/_ref/libref_ring.so (its ring_buffer.cc), against oracle/_ref/ref_pair_trace
 * (its pair.cc -- PairPollable::Send / Recv / SendZerocopy and the credit rule --
 * compiled unmodified over the software verbs of oracle/fakeverbs) and the golden
 * vectors under tests/golden/ (tests/test_oracle_vs_ref.py, tests/test_golden_*.py).
 */
#define _POSIX_C_SOURCE 200809L
#include "grdma_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ===================================================================== ring */

uint64_t orc_round_up8(uint64_t v) { /* ring_buffer.h:232-237 */
  return (v % ORC_ALIGN == 0) ? v : v - v % ORC_ALIGN + ORC_ALIGN;
}

uint64_t orc_round_down8(uint64_t v) { /* ring_buffer.h:240-245 */
  return v - v % ORC_ALIGN;
}

uint64_t orc_encoded_size(uint64_t payload) { /* ring_buffer.h:180-183 */
  return 2u * ORC_ALIGN + orc_round_up8(payload);
}

uint64_t orc_calc_writable(uint64_t space) { /* ring_buffer.h:185-189 */
  int64_t fr = (int64_t)space - (int64_t)ORC_RESERVED;
  if (fr < 0) fr = 0;
  return orc_round_down8((uint64_t)fr);
}

static uint64_t ld64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v; /* native little-endian, ring_buffer.h:84-87 */
}
static void st64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }

int orc_ring_init(orc_ring* r, uint8_t* buf, uint64_t cap) {
  /* ring_buffer.cc:20-25: power of two, larger than the reserved space */
  if ((cap & (cap - 1)) != 0 || cap <= ORC_RESERVED) return -1;
  r->buf = buf;
  r->cap = cap;
  r->mask = cap - 1;
  memset(buf, 0, cap); /* Init(), ring_buffer.cc:49-54 */
  r->head = r->moving_head = r->remain = 0;
  return 0;
}

int orc_ring_has_message(const orc_ring* r) { /* ring_buffer.cc:56-65 */
  if (r->remain > 0) return 1;
  return ld64(r->buf + r->head) > 0;
}

uint64_t orc_ring_readable(const orc_ring* r) { /* ring_buffer.cc:67-97 */
  if (r->remain > 0) return r->remain;
  uint64_t size = ld64(r->buf + r->head);
  if (size == 0) return 0;
  /* The reference spins ("goto retry") on an oversized header, treating it as
   * a torn read.  A static snapshot cannot change, so report "not ready". */
  if (size > r->cap - ORC_RESERVED) return 0;
  uint64_t foot = (r->head + ORC_ALIGN + orc_round_up8(size)) & r->mask;
  return ld64(r->buf + foot) == ORC_FOOTER ? size : 0;
}

uint64_t orc_ring_free_size(const orc_ring* r, uint64_t head, uint64_t tail) {
  uint64_t occupied = (tail + r->cap - head) & r->mask; /* ring_buffer.cc:99-104 */
  return r->cap - occupied;
}

uint64_t orc_ring_writable(const orc_ring* r, uint64_t head, uint64_t tail) {
  uint64_t remaining = orc_ring_free_size(r, head, tail); /* ring_buffer.cc:106-116 */
  return remaining > ORC_RESERVED ? remaining - ORC_RESERVED : 0;
}

uint64_t orc_ring_read(orc_ring* r, void* dst, uint64_t cap, uint64_t* internal) {
  /* ring_buffer.cc:122-191 */
  uint64_t readable = orc_ring_readable(r);
  uint64_t copy = readable < cap ? readable : cap;
  uint64_t prev_moving = r->moving_head;
  if (copy == 0) {
    if (internal) *internal = 0;
    return 0;
  }
  if (r->remain == 0) { /* opening a record: :133-146 */
    r->moving_head = (r->head + ORC_ALIGN) & r->mask;
    st64(r->buf + r->head, 0); /* clear header */
    r->head = (r->head + 2u * ORC_ALIGN + orc_round_up8(readable)) & r->mask;
  }
  uint64_t end = (r->moving_head + copy) & r->mask;
  uint64_t seg1, seg2 = 0;
  if (r->moving_head < end) {
    seg1 = copy;
  } else { /* circular case :153-156 */
    seg2 = end;
    seg1 = copy - seg2;
  }
  memcpy(dst, r->buf + r->moving_head, seg1);
  memset(r->buf + r->moving_head, 0, seg1);
  if (seg2 > 0) {
    memcpy((uint8_t*)dst + seg1, r->buf, seg2);
    memset(r->buf, 0, seg2);
  }
  r->moving_head = (r->moving_head + copy) & r->mask;
  r->remain = readable - copy;
  if (r->remain == 0) { /* record finished: :169-182 */
    for (uint64_t pos = r->moving_head; pos < orc_round_up8(r->moving_head); pos++)
      r->buf[pos & r->mask] = 0;
    r->moving_head = orc_round_up8(r->moving_head) & r->mask;
    st64(r->buf + r->moving_head, 0); /* clear footer */
    r->moving_head = (r->moving_head + ORC_ALIGN) & r->mask;
  }
  if (internal) *internal = (r->moving_head + r->cap - prev_moving) & r->mask;
  return copy;
}

uint64_t orc_ring_write(orc_ring* r, uint64_t tail, const void* src, uint64_t n) {
  /* ring_buffer.cc:193-226 */
  if (n == 0) return tail;
  st64(r->buf + tail, n);
  tail = (tail + ORC_ALIGN) & r->mask;
  uint64_t end = (tail + n) & r->mask;
  uint64_t seg1 = (tail < end) ? n : n - end;
  memcpy(r->buf + tail, src, seg1);
  if (n - seg1 > 0) memcpy(r->buf, (const uint8_t*)src + seg1, n - seg1);
  tail = (tail + orc_round_up8(n)) & r->mask;
  st64(r->buf + tail, ORC_FOOTER);
  return (tail + ORC_ALIGN) & r->mask;
}

/* ===================================================================== pair */

uint64_t orc_plan_send(uint64_t ring_cap, uint64_t staging_cap, uint64_t remote_head,
                       uint64_t remote_tail, int max_sge, const uint64_t* lens,
                       uint64_t n, uint64_t byte_idx, uint64_t* pays,
                       uint64_t* sent) {
  /* pair.cc:671-707, the arithmetic only */
  uint64_t mask = ring_cap - 1, st = 0, total = 0, k = 0;
  for (uint64_t i = 0; i < n && (int64_t)k < (int64_t)max_sge; i++) {
    uint64_t len = lens[i] - byte_idx;
    byte_idx = 0;
    uint64_t occupied = (remote_tail + ring_cap - remote_head) & mask;
    uint64_t recv_free = ring_cap - occupied;
    uint64_t send_free = staging_cap - st;
    uint64_t a = orc_calc_writable(send_free), b = orc_calc_writable(recv_free);
    uint64_t pay = len;
    if (a < pay) pay = a;
    if (b < pay) pay = b;
    if (pay == 0) break;
    uint64_t enc = orc_encoded_size(pay);
    pays[k++] = pay;
    total += pay;
    st += enc;
    remote_tail = (remote_tail + enc) & mask;
  }
  if (sent) *sent = total;
  return k;
}

int orc_pair_init(orc_pair* p, uint64_t ring_cap, int max_sge) {
  memset(p, 0, sizeof(*p));
  uint8_t* buf = (uint8_t*)malloc(ring_cap);
  if (!buf || orc_ring_init(&p->ring, buf, ring_cap) != 0) {
    free(buf);
    return -1;
  }
  p->staging_cap = ring_cap / 2; /* pair.cc:104 */
  p->staging = (uint8_t*)calloc(1, p->staging_cap);
  p->max_sge = max_sge;
  return p->staging ? 0 : -1;
}

void orc_pair_destroy(orc_pair* p) {
  free(p->ring.buf);
  free(p->staging);
  free(p->zc_buf);
  memset(p, 0, sizeof(*p));
}

void orc_pair_connect(orc_pair* a, orc_pair* b) {
  a->peer = b;
  b->peer = a;
}

uint64_t orc_pair_writable(const orc_pair* p) { /* pair.cc:294-301 */
  return orc_ring_writable(&p->peer->ring, p->status_recv.remote_head, p->remote_tail);
}

uint64_t orc_pair_send(orc_pair* p, const orc_slice* slices, uint64_t n,
                       uint64_t byte_idx) {
  /* pair.cc:645-734 */
  orc_ring* rr = &p->peer->ring;
  uint64_t remote_head = p->status_recv.remote_head; /* pair.h:229-233 */
  uint64_t remote_tail = p->remote_tail;
  uint64_t st = 0, total = 0, written = 0, k = 0;
  /* wrap bookkeeping of GetWriteRequests(sg_list), ring_buffer.cc:261-330 */
  int64_t split = -1;
  uint64_t seg1 = 0, seg2 = 0, before_split = 0;

  for (uint64_t i = 0; i < n; i++) total += slices[i].len;
  total -= byte_idx;

  for (uint64_t i = 0; i < n && (int64_t)k < (int64_t)p->max_sge; i++) {
    const uint8_t* ptr = slices[i].ptr + byte_idx;
    uint64_t len = slices[i].len - byte_idx;
    byte_idx = 0;
    uint64_t recv_free = orc_ring_free_size(rr, remote_head, remote_tail);
    uint64_t send_free = p->staging_cap - st;
    uint64_t a = orc_calc_writable(send_free), b = orc_calc_writable(recv_free);
    uint64_t pay = len;
    if (a < pay) pay = a;
    if (b < pay) pay = b;
    if (pay == 0) break;
    uint64_t enc = orc_encoded_size(pay);
    /* AppendHeader / AppendPayload / AppendFooter, ring_buffer.h:84-99: the
     * padding bytes are skipped, not written. */
    st64(p->staging + st, pay);
    memcpy(p->staging + st + ORC_ALIGN, ptr, pay);
    st64(p->staging + st + ORC_ALIGN + orc_round_up8(pay), ORC_FOOTER);
    uint64_t next_tail = (remote_tail + enc) & rr->mask;
    if (remote_tail > next_tail && split < 0) { /* ring_buffer.cc:276-283 */
      split = (int64_t)k;
      seg2 = next_tail;
      seg1 = enc - seg2;
      before_split = st;
    }
    st += enc;
    written += pay;
    remote_tail = next_tail;
    k++;
  }
  p->partial_write = written < total; /* pair.cc:709 */
  p->staging_used = st;
  p->wr_count = 0;
  if (k > 0) {
    /* Execute the (at most two) RDMA WRITE work requests in order. */
    if (split >= 0) {
      uint64_t n0 = before_split + seg1, n1 = st - n0;
      memcpy(rr->buf + p->remote_tail, p->staging, n0);
      if (n1) memcpy(rr->buf, p->staging + n0, n1);
      p->wr[0][0] = p->remote_tail; p->wr[0][1] = n0;
      p->wr[1][0] = 0;              p->wr[1][1] = n1;
      p->wr_count = 2;
    } else {
      memcpy(rr->buf + p->remote_tail, p->staging, st);
      p->wr[0][0] = p->remote_tail; p->wr[0][1] = st;
      p->wr_count = 1;
    }
    p->remote_tail = remote_tail;
  }
  return written;
}

int orc_pair_enable_zerocopy(orc_pair* p, uint64_t zc_cap) { /* pair.cc:103,113,120 */
  free(p->zc_buf);
  p->zc_buf = (uint8_t*)calloc(1, zc_cap ? zc_cap : 1);
  p->zc_cap = p->zc_buf ? zc_cap : 0;
  p->zc_tail = 0;
  return p->zc_buf ? 0 : -1;
}

uint8_t* orc_pair_allocate_send_buffer(orc_pair* p, uint64_t size) { /* pair.cc:305-323 */
  if (size == 0) return NULL;
  if (p->zc_tail != 0 || p->zc_tail + size > p->zc_cap) return NULL;
  uint64_t tail = p->zc_tail;
  p->zc_tail = tail + size;
  return p->zc_buf + tail;
}

uint64_t orc_pair_send_zerocopy(orc_pair* p, const orc_slice* slices, uint64_t n,
                                uint64_t byte_idx) {
  /* pair.cc:793-941 */
  orc_ring* rr = &p->peer->ring;
  uint64_t remote_head = p->status_recv.remote_head;
  uint64_t remote_tail = p->remote_tail;
  uint64_t st = 0, total = 0, written = 0;
  /* the scatter-gather list: {source, length}; at most max_sge entries (+1 after the wrap split) */
  uint64_t cap_sge = (uint64_t)(p->max_sge > 0 ? p->max_sge : 0) + 4;
  const uint8_t** sg_ptr = (const uint8_t**)malloc(sizeof(uint8_t*) * cap_sge);
  uint64_t* sg_len = (uint64_t*)malloc(sizeof(uint64_t) * cap_sge);
  uint64_t nsge = 0;

  for (uint64_t i = 0; i < n; i++) total += slices[i].len;
  total -= byte_idx;

  for (uint64_t i = 0; i < n && (int64_t)nsge < (int64_t)p->max_sge; i++) {
    const uint8_t* ptr = slices[i].ptr + byte_idx;
    uint64_t len = slices[i].len - byte_idx;
    uint64_t recv_free = orc_ring_free_size(rr, remote_head, remote_tail);
    uint64_t send_free = p->staging_cap - st;
    byte_idx = 0;
    if (p->zc_buf != NULL && ptr >= p->zc_buf && ptr + len <= p->zc_buf + p->zc_cap) { /* :825-826 */
      uint64_t pay = len, b = orc_calc_writable(recv_free);
      if (b < pay) pay = b;
      /* header, footer and padding are staged; four entries are needed (:830-834) */
      if (pay == 0 || send_free < 3ull * ORC_ALIGN || (int64_t)(nsge + 4) > (int64_t)p->max_sge) break;
      uint64_t enc = orc_encoded_size(pay);
      st64(p->staging + st, pay);                               /* AppendHeader into the staging buffer */
      sg_ptr[nsge] = p->staging + st; sg_len[nsge] = 8; nsge++;
      st += 8;
      sg_ptr[nsge] = ptr; sg_len[nsge] = pay; nsge++;           /* the payload where it lies */
      uint64_t pad = orc_round_up8(pay) - pay;
      if (pad > 0) {                                            /* whatever the staging buffer holds there */
        sg_ptr[nsge] = p->staging + st; sg_len[nsge] = pad; nsge++;
        st += pad;
      }
      st64(p->staging + st, ORC_FOOTER);                        /* AppendFooter */
      sg_ptr[nsge] = p->staging + st; sg_len[nsge] = 8; nsge++;
      st += 8;
      p->zc_tail = (uint32_t)(p->zc_tail - pay);                /* :876, std::atomic_uint32_t */
      written += pay;
      remote_tail = (remote_tail + enc) & rr->mask;
      p->zc_bytes += pay;
    } else {
      uint64_t a = orc_calc_writable(send_free), b = orc_calc_writable(recv_free);
      uint64_t pay = len;
      if (a < pay) pay = a;
      if (b < pay) pay = b;
      if (pay == 0) break;
      uint64_t enc = orc_encoded_size(pay);
      st64(p->staging + st, pay);
      memcpy(p->staging + st + ORC_ALIGN, ptr, pay);
      st64(p->staging + st + ORC_ALIGN + orc_round_up8(pay), ORC_FOOTER);
      sg_ptr[nsge] = p->staging + st; sg_len[nsge] = enc; nsge++;
      written += pay;
      st += enc;
      remote_tail = (remote_tail + enc) & rr->mask;
      p->copy_bytes += pay;
    }
  }
  p->partial_write = written < total; /* :908 */
  p->staging_used = st;
  p->wr_count = 0;
  p->sge_count = nsge;
  if (nsge > 0) {
    /* GetWriteRequests(sg_list) + ibv_post_send: the entries land back to back at remote_tail; the
     * entry that crosses the ring end is split and the rest goes out as a second request at offset 0
     * (ring_buffer.cc:261-330). */
    uint64_t at = p->remote_tail, first = 0, second = 0;
    int wrapped = 0;
    for (uint64_t k = 0; k < nsge; k++) {
      const uint8_t* src = sg_ptr[k];
      uint64_t left = sg_len[k];
      while (left > 0) {
        uint64_t room = rr->cap - at, m = left < room ? left : room;
        memcpy(rr->buf + at, src, m);
        if (wrapped) second += m; else first += m;
        src += m; left -= m;
        at = (at + m) & rr->mask;
        if (at == 0 && !wrapped) {
          wrapped = 1;
          if (left > 0) p->sge_count++;  /* the split entry becomes two (:296-303) */
        }
      }
    }
    p->wr[0][0] = p->remote_tail; p->wr[0][1] = first;
    p->wr_count = 1;
    if (wrapped) {
      p->wr[1][0] = 0; p->wr[1][1] = second;
      p->wr_count = 2;
    }
    p->remote_tail = remote_tail;
  }
  free(sg_ptr);
  free(sg_len);
  return written;
}

uint64_t orc_pair_recv(orc_pair* p, void* dst, uint64_t cap) {
  /* pair.cc:264-286 */
  uint64_t internal = 0;
  uint64_t n = orc_ring_read(&p->ring, dst, cap, &internal);
  p->internal_read_size += internal;
  if (p->internal_read_size >= p->ring.cap / 2) {
    p->status_send.remote_head = p->ring.moving_head; /* get_head(), .cc:334 */
    p->peer->status_recv = p->status_send;            /* updateStatus(), pair.cc:624-641 */
    p->credit_msgs++;
    p->internal_read_size = 0;
  }
  return n;
}

uint64_t orc_endpoint_read(orc_pair* p, uint8_t* dst, uint64_t* alloc_out) {
  /* rdma_bp_posix.cc:306-326: a fresh slice of max(256, readable) is allocated
   * only when incoming_buffer is empty; otherwise the unfilled tail retained in
   * last_read_buffer (:283-287, :350-351) is the read target. */
  uint64_t readable = orc_ring_readable(&p->ring);
  uint64_t alloc = p->leftover_cap;
  if (alloc == 0) alloc = readable > 256 ? readable : 256;
  if (alloc_out) *alloc_out = alloc;
  uint64_t total = 0;
  for (;;) { /* rdma_do_read loop :195-277 */
    uint64_t n = orc_pair_recv(p, dst + total, alloc - total);
    if (n == 0) break;
    total += n;
    if (total == alloc) break;
  }
  if (total == 0) { /* would block: the slice stays in incoming_buffer */
    p->leftover_cap = alloc;
    return 0;
  }
  p->leftover_cap = alloc - total; /* grpc_slice_buffer_trim_end → last_read_buffer */
  return total;
}

/* ================================================================ HTTP/2 TX */

void orc_grpc_msg_header(uint8_t out[5], int compressed, uint32_t len) {
  /* chttp2_transport.cc:1502-1510 */
  out[0] = compressed ? 1 : 0;
  out[1] = (uint8_t)(len >> 24);
  out[2] = (uint8_t)(len >> 16);
  out[3] = (uint8_t)(len >> 8);
  out[4] = (uint8_t)len;
}

void orc_h2_data_header(uint8_t out[9], uint32_t len, int end_stream, uint32_t id) {
  /* frame_data.cc:73-82 */
  out[0] = (uint8_t)(len >> 16);
  out[1] = (uint8_t)(len >> 8);
  out[2] = (uint8_t)len;
  out[3] = ORC_H2_FRAME_DATA;
  out[4] = end_stream ? ORC_H2_FLAG_END_STREAM : 0;
  out[5] = (uint8_t)(id >> 24);
  out[6] = (uint8_t)(id >> 16);
  out[7] = (uint8_t)(id >> 8);
  out[8] = (uint8_t)id;
}

/* A slice buffer modelled as (length, inlined?) entries; bytes live in the
 * flat wire image because concatenation order is all that the ring sees. */
typedef struct sbuf {
  uint64_t* lens;
  uint8_t* inl;
  uint64_t count, cap;
  int overflow;
} sbuf;

static void sb_add_indexed(sbuf* sb, uint64_t len, int inlined) {
  /* grpc_slice_buffer_add_indexed, slice_buffer.cc:127-134 */
  if (sb->count >= sb->cap) {
    sb->overflow = 1;
    return;
  }
  sb->lens[sb->count] = len;
  sb->inl[sb->count] = (uint8_t)inlined;
  sb->count++;
}

static void sb_add(sbuf* sb, uint64_t len, int inlined) {
  /* grpc_slice_buffer_add, slice_buffer.cc:136-171 */
  if (inlined && sb->count > 0) {
    uint64_t b = sb->count - 1;
    if (sb->inl[b] && sb->lens[b] < ORC_SLICE_INLINED_SIZE) {
      if (len + sb->lens[b] <= ORC_SLICE_INLINED_SIZE) {
        sb->lens[b] += len;
      } else {
        uint64_t cp1 = ORC_SLICE_INLINED_SIZE - sb->lens[b];
        sb->lens[b] = ORC_SLICE_INLINED_SIZE;
        sb_add_indexed(sb, len - cp1, 1);
      }
      return;
    }
  }
  sb_add_indexed(sb, len, inlined);
}

/* One message appended to an existing slice buffer `out` / wire image. */
static int frame_one(sbuf* out, const uint8_t* msg, uint64_t msg_len, int compressed,
                     uint32_t stream_id, uint32_t max_frame, int end_stream, uint8_t* wire,
                     uint64_t wire_cap, uint64_t* wpos) {
  /* flow_controlled_buffer after perform_stream_op_locked: [inlined 5][msg] */
  struct { uint64_t len; int inl; } fcb[2];
  int fcb_n = 0, fcb_i = 0;
  fcb[fcb_n].len = 5; fcb[fcb_n].inl = 1; fcb_n++;
  if (msg_len > 0) { fcb[fcb_n].len = msg_len; fcb[fcb_n].inl = 0; fcb_n++; }
  uint64_t fcb_len = 5 + msg_len;
  uint64_t w = *wpos, src_off = 0; /* src_off: bytes of [hdr5|msg] already moved */
  uint8_t hdr5[5];
  orc_grpc_msg_header(hdr5, compressed, (uint32_t)msg_len);

  while (fcb_len > 0) {
    /* DataSendContext::FlushUncompressedBytes, writing.cc:344-355 (windows are
     * assumed open: flow control is out of scope) */
    uint64_t send = fcb_len < max_frame ? fcb_len : max_frame;
    int is_last = end_stream && send == fcb_len;
    if (w + 9 + send > wire_cap) return -1;
    orc_h2_data_header(wire + w, (uint32_t)send, is_last, stream_id);
    w += 9;
    sb_add(out, 9, 1); /* GRPC_SLICE_MALLOC(9) is an inlined slice */
    for (uint64_t i = 0; i < send; i++) {
      uint64_t o = src_off + i;
      wire[w + i] = o < 5 ? hdr5[o] : msg[o - 5];
    }
    w += send;
    src_off += send;
    /* grpc_slice_buffer_move_first_no_ref(inbuf, send, outbuf), slice_buffer.cc:270-313 */
    uint64_t n = send;
    if (fcb_len == n) { /* move_into -> grpc_slice_buffer_add for each slice */
      for (; fcb_i < fcb_n; fcb_i++) sb_add(out, fcb[fcb_i].len, fcb[fcb_i].inl);
    } else {
      while (fcb_i < fcb_n) {
        uint64_t sl = fcb[fcb_i].len;
        if (n > sl) {
          sb_add(out, sl, fcb[fcb_i].inl);
          n -= sl;
          fcb_i++;
        } else if (n == sl) {
          sb_add(out, sl, fcb[fcb_i].inl);
          fcb_i++;
          break;
        } else { /* split: head goes in un-merged (add_indexed), tail stays */
          sb_add_indexed(out, n, fcb[fcb_i].inl);
          fcb[fcb_i].len = sl - n;
          break;
        }
      }
    }
    fcb_len -= send;
  }
  *wpos = w;
  return out->overflow ? -1 : 0;
}

int64_t orc_h2_frame_message(const uint8_t* msg, uint64_t msg_len, int compressed,
                             uint32_t stream_id, uint32_t max_frame, int end_stream,
                             uint8_t* wire, uint64_t wire_cap, uint64_t* wire_len,
                             uint64_t* lens, uint64_t lens_cap) {
  uint8_t* inl = (uint8_t*)malloc(lens_cap ? lens_cap : 1);
  if (!inl) return -1;
  sbuf out = {lens, inl, 0, lens_cap, 0};
  uint64_t w = 0;
  int rc = frame_one(&out, msg, msg_len, compressed, stream_id, max_frame, end_stream, wire,
                     wire_cap, &w);
  free(inl);
  if (rc) return -1;
  *wire_len = w;
  return (int64_t)out.count;
}

int64_t orc_h2_frame_batch(const uint8_t* const* msgs, const uint64_t* msg_lens,
                           const uint32_t* stream_ids, const uint32_t* flags, uint64_t n,
                           uint32_t max_frame, uint8_t* wire, uint64_t wire_cap,
                           uint64_t* wire_len, uint64_t* lens, uint64_t lens_cap) {
  /* several messages queued on one t->outbuf: the inlined-slice merge rule of
   * grpc_slice_buffer_add applies across message boundaries too */
  uint8_t* inl = (uint8_t*)malloc(lens_cap ? lens_cap : 1);
  if (!inl) return -1;
  sbuf out = {lens, inl, 0, lens_cap, 0};
  uint64_t w = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (frame_one(&out, msgs[i], msg_lens[i], flags[i] & 1, stream_ids[i], max_frame,
                  (flags[i] >> 1) & 1, wire, wire_cap, &w)) {
      free(inl);
      return -1;
    }
  }
  free(inl);
  *wire_len = w;
  return (int64_t)out.count;
}

/* ================================================================ HTTP/2 RX */

static const char kClientPrefix[] = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n"; /* internal.h:781 */
enum { ST_FH0 = 24, ST_FH8 = 32, ST_FRAME = 33 };
enum { PARSER_SKIP = 0, PARSER_DATA = 1, PARSER_HEADER = 2, PARSER_RST = 3, PARSER_HEADER3 = 4 };
enum { FT_DATA = 0, FT_HEADERS = 1, FT_RST_STREAM = 3, FT_SETTINGS = 4, FT_PING = 6, FT_GOAWAY = 7, FT_WINDOW_UPDATE = 8,
       FT_CONTINUATION = 9 };
enum { FL_ACK = 1 };
enum { FL_END_STREAM = 1, FL_END_HEADERS = 4 };

void orc_h2_parser_init_ex(orc_h2_parser* p, int flags, uint32_t max_frame_size,
                           uint32_t max_concurrent_streams) {
  memset(p, 0, sizeof(*p));
  p->is_server = (flags & ORC_H2_SERVER) != 0;
  p->is_first_frame = (flags & ORC_H2_FIRST_FRAME) != 0; /* chttp2_transport.cc: t->is_first_frame = true */
  p->state = p->is_server ? 0 : ST_FH0;   /* a server starts at GRPC_DTS_CLIENT_PREFIX_0 */
  p->max_frame_size = max_frame_size;     /* http2_settings.cc:56 default 16384 */
  p->check_frame_size = 1;
  p->max_concurrent_streams = max_concurrent_streams; /* http2_settings.cc:46 default 0xffffffff */
}

void orc_h2_parser_init(orc_h2_parser* p, int expect_client_prefix, uint32_t max_frame_size) {
  orc_h2_parser_init_ex(p, expect_client_prefix ? (ORC_H2_SERVER | ORC_H2_FIRST_FRAME) : 0,
                        max_frame_size, 0xffffffffu);
}

void orc_h2_parser_free(orc_h2_parser* p) {
  free(p->streams);
  p->streams = NULL;
  p->nstreams = p->streams_cap = 0;
}

uint64_t orc_h2_parser_live_streams(const orc_h2_parser* p) { return p->nstreams; }

/* grpc_chttp2_parsing_lookup_stream (chttp2_transport.cc): plain map lookup, no creation */
static orc_h2_stream* lookup_stream(orc_h2_parser* p, uint32_t id) {
  if (id == 0) return NULL;
  for (uint64_t i = 0; i < p->nstreams; i++)
    if (p->streams[i].stream_id == id) return &p->streams[i];
  return NULL;
}

static orc_h2_stream* add_stream(orc_h2_parser* p, uint32_t id) {
  if (p->nstreams == p->streams_cap) {
    uint64_t nc = p->streams_cap ? 2 * p->streams_cap : 16;
    orc_h2_stream* ns = (orc_h2_stream*)realloc(p->streams, nc * sizeof(orc_h2_stream));
    if (!ns) return NULL;
    p->streams = ns;
    p->streams_cap = nc;
  }
  orc_h2_stream* s = &p->streams[p->nstreams++];
  memset(s, 0, sizeof(*s));
  s->stream_id = id;
  return s;
}

static void remove_stream(orc_h2_parser* p, orc_h2_stream* s) {
  *s = p->streams[--p->nstreams];
}

int orc_h2_parser_open_stream(orc_h2_parser* p, uint32_t id) {
  if (id == 0 || lookup_stream(p, id)) return -1;
  return add_stream(p, id) ? 0 : -1;
}

int orc_h2_parser_close_writes(orc_h2_parser* p, uint32_t id) {
  orc_h2_stream* s = lookup_stream(p, id);
  if (!s) return -1;
  s->write_closed = 1;
  if (s->read_closed) remove_stream(p, s); /* chttp2_transport.cc:2218-2223 */
  return 0;
}

static int push_ev(orc_h2_event* ev, uint64_t cap, uint64_t* nev, uint32_t kind,
                   uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  if (*nev >= cap) return ORC_H2_ERR_EVENT_OVERFLOW;
  ev[*nev].kind = kind; ev[*nev].a = a; ev[*nev].b = b; ev[*nev].c = c; ev[*nev].d = d;
  (*nev)++;
  return 0;
}

/* grpc_chttp2_mark_stream_closed(t, s, close_reads, close_writes), chttp2_transport.cc:2194-2244 */
static int mark_closed(orc_h2_parser* p, uint32_t id, int close_reads, int close_writes,
                       orc_h2_event* ev, uint64_t cap, uint64_t* nev) {
  orc_h2_stream* s = lookup_stream(p, id);
  if (!s) return 0;
  if (close_reads) s->read_closed = 1;
  if (close_writes) s->write_closed = 1;
  int gone = s->read_closed && s->write_closed;
  if (gone) remove_stream(p, s);
  return push_ev(ev, cap, nev, ORC_EV_STREAM_CLOSED, (uint32_t)gone, 0, id, 0);
}

/* grpc_deframe_unprocessed_incoming_frames (frame_data.cc:92-276) applied to a
 * run of DATA payload bytes [beg, beg+len) of the chunk at `base`. */
static int grpc_deframe(orc_h2_stream* d, const uint8_t* base, uint64_t beg,
                        uint64_t len, orc_h2_event* ev, uint64_t cap, uint64_t* nev) {
  uint64_t cur = beg, end = beg + len;
  int rc;
  while (cur < end) {
    switch (d->state) {
      case 6: /* GRPC_CHTTP2_DATA_ERROR: swallow */
        return 0;
      case 0: { /* FH_0: compressed flag, :113-141 */
        uint8_t t = base[cur];
        if (t > 1) {
          d->state = 6;
          return ORC_H2_ERR_GRPC_FRAME_TYPE;
        }
        d->compressed = t;
        d->state = 1; cur++;
        break;
      }
      case 1: d->frame_size = (uint32_t)base[cur] << 24; d->state = 2; cur++; break;
      case 2: d->frame_size |= (uint32_t)base[cur] << 16; d->state = 3; cur++; break;
      case 3: d->frame_size |= (uint32_t)base[cur] << 8; d->state = 4; cur++; break;
      case 4: /* FH_4 :174-203 */
        d->frame_size |= (uint32_t)base[cur]; cur++;
        if ((rc = push_ev(ev, cap, nev, ORC_EV_MSG_BEGIN, (uint32_t)d->compressed,
                          d->frame_size, d->stream_id, 0))) return rc;
        if (d->frame_size == 0) {
          if ((rc = push_ev(ev, cap, nev, ORC_EV_MSG_END, 0, 0, d->stream_id, 0))) return rc;
          d->state = 0;
        } else {
          d->state = 5;
        }
        break;
      case 5: { /* FRAME :204-271 */
        uint64_t remaining = end - cur;
        uint64_t take = remaining < d->frame_size ? remaining : d->frame_size;
        if ((rc = push_ev(ev, cap, nev, ORC_EV_MSG_BYTES, (uint32_t)cur, (uint32_t)take,
                          d->stream_id, 0))) return rc;
        d->frame_size -= (uint32_t)take;
        cur += take;
        if (d->frame_size == 0) {
          if ((rc = push_ev(ev, cap, nev, ORC_EV_MSG_END, 0, 0, d->stream_id, 0))) return rc;
          d->state = 0;
        }
        break;
      }
    }
  }
  return 0;
}

/* init_header_frame_parser (parsing.cc:566-680): which stream the header block belongs to,
 * stream acceptance on a server.  The HPACK bytes themselves are control plane (skipped). */
static int begin_header_frame(orc_h2_parser* p, int is_continuation, int* parser_kind,
                              int* opened) {
  const uint32_t id = p->incoming_stream_id;
  p->header_boundary = (p->incoming_frame_flags & FL_END_HEADERS) != 0;
  p->expect_continuation_stream_id = p->header_boundary ? 0 : id;          /* :574-578 */
  if (!is_continuation) p->header_eof = (p->incoming_frame_flags & FL_END_STREAM) != 0; /* :580-583 */
  *parser_kind = PARSER_SKIP;
  orc_h2_stream* s = lookup_stream(p, id);
  if (s == NULL) {
    if (is_continuation) return 0;                      /* :590-595 */
    if (!p->is_server) return 0;                        /* :596-608 */
    if (p->last_new_stream_id >= id) return 0;          /* :609-616 */
    if ((id & 1) == 0) return 0;                        /* :617-622 */
    if (p->nstreams >= p->max_concurrent_streams) return ORC_H2_ERR_MAX_STREAMS; /* :623-627 */
    p->last_new_stream_id = id;                         /* :629-631 grpc_chttp2_parsing_accept_stream */
    s = add_stream(p, id);
    if (s == NULL) return 0;
    *opened = 1;
  }
  if (s->read_closed) return 0;                         /* :645-650 */
  if (s->header_frames_received >= 2) {
    /* :667-669 "too many header frames received": init_skip_frame_parser(t, 1) -- the header parser in skipping mode,
     * with is_boundary = (expect_continuation_stream_id != 0) (:318-327), i.e. set when the frame LACKS END_HEADERS.
     * The stream stays t->incoming_stream, so at the end of such a frame the parser finds header_frames_received == 2
     * and fails the connection ("Too many trailer frames", hpack_parser.cc:1756-1759); with END_HEADERS it skips. */
    if (!p->header_boundary) *parser_kind = PARSER_HEADER3;
    return 0;
  }
  *parser_kind = PARSER_HEADER;
  return 0;
}

static int begin_frame(orc_h2_parser* p, orc_h2_event* ev, uint64_t cap, uint64_t* nev) {
  /* init_frame_parser (parsing.cc:255-306) */
  uint32_t status = 0;
  int kind = PARSER_SKIP, opened = 0, rc;
  const uint32_t id = p->incoming_stream_id;
  if (p->is_first_frame && p->incoming_frame_type != FT_SETTINGS) return ORC_H2_ERR_FIRST_FRAME;
  p->is_first_frame = 0;
  if (p->expect_continuation_stream_id != 0) {
    if (p->incoming_frame_type != FT_CONTINUATION) return ORC_H2_ERR_EXPECTED_CONTINUATION;
    if (p->expect_continuation_stream_id != id) return ORC_H2_ERR_CONTINUATION_STREAM;
    if ((rc = begin_header_frame(p, 1, &kind, &opened))) return rc;
  } else if (p->incoming_frame_type == FT_DATA) {
    /* init_data_frame_parser (:341-397) + grpc_chttp2_data_parser_begin_frame (frame_data.cc:43-62) */
    orc_h2_stream* s = lookup_stream(p, id);
    if (s != NULL && !s->read_closed) {
      if (p->incoming_frame_flags & ~FL_END_STREAM) {
        status = ORC_H2_ERR_DATA_FLAGS; /* stream error: the stream is closed for reads, RST_STREAM queued */
      } else {
        p->received_last_frame = (p->incoming_frame_flags & FL_END_STREAM) != 0;
        kind = PARSER_DATA;
      }
    }
  } else if (p->incoming_frame_type == FT_HEADERS) {
    if ((rc = begin_header_frame(p, 0, &kind, &opened))) return rc;
  } else if (p->incoming_frame_type == FT_CONTINUATION) {
    return ORC_H2_ERR_UNEXPECTED_CONTINUATION;
  } else if (p->incoming_frame_type == FT_RST_STREAM) {
    if (p->incoming_frame_size != 4) return ORC_H2_ERR_RST_LENGTH;
    if (lookup_stream(p, id)) kind = PARSER_RST;
  } else if (p->incoming_frame_type == FT_SETTINGS) {
    /* init_settings_frame_parser (:732-757) + grpc_chttp2_settings_parser_begin_frame (frame_settings.cc:88-111) */
    if (id != 0) return ORC_H2_ERR_SETTINGS_STREAM;
    if (p->incoming_frame_flags == FL_ACK) {
      if (p->incoming_frame_size != 0) return ORC_H2_ERR_SETTINGS_ACK_LENGTH;
    } else if (p->incoming_frame_flags != 0) {
      return ORC_H2_ERR_SETTINGS_FLAGS;
    } else if (p->incoming_frame_size % 6 != 0) {
      return ORC_H2_ERR_SETTINGS_LENGTH;
    }
  } else if (p->incoming_frame_type == FT_PING) {
    /* init_ping_parser (:705-712) + grpc_chttp2_ping_parser_begin_frame (frame_ping.cc:58-69) */
    if ((p->incoming_frame_flags & 0xfe) || p->incoming_frame_size != 8) return ORC_H2_ERR_PING;
  } else if (p->incoming_frame_type == FT_GOAWAY) {
    /* init_goaway_parser (:723-730) + grpc_chttp2_goaway_parser_begin_frame (frame_goaway.cc:39-52) */
    if (p->incoming_frame_size < 8) return ORC_H2_ERR_GOAWAY;
  } else if (p->incoming_frame_type == FT_WINDOW_UPDATE) {
    /* init_window_update_frame_parser (:686-703) + ..._begin_frame (frame_window_update.cc:56-67): checked before
     * the stream is looked up */
    if (p->incoming_frame_flags || p->incoming_frame_size != 4) return ORC_H2_ERR_WINDOW_UPDATE;
  } /* the PAYLOADS of SETTINGS, WINDOW_UPDATE, PING, GOAWAY are control plane -> skipped */
  p->cur_parser = kind; /* must survive across feeds */
  if ((rc = push_ev(ev, cap, nev, ORC_EV_FRAME, p->incoming_frame_type,
                    p->incoming_frame_flags | (status << 8), id, p->incoming_frame_size)))
    return rc;
  if (opened && (rc = push_ev(ev, cap, nev, ORC_EV_STREAM_OPEN, 0, 0, id, 0))) return rc;
  if (status == ORC_H2_ERR_DATA_FLAGS) return mark_closed(p, id, 1, 0, ev, cap, nev); /* parsing.cc:388-391 */
  return 0;
}

/* what the payload parser does when it is handed the last piece of a frame */
static int end_frame(orc_h2_parser* p, orc_h2_event* ev, uint64_t cap, uint64_t* nev) {
  const uint32_t id = p->incoming_stream_id;
  if (p->cur_parser == PARSER_DATA) {
    /* grpc_chttp2_data_parser_parse, frame_data.cc:299-305 */
    if (p->received_last_frame) return mark_closed(p, id, 1, 0, ev, cap, nev);
  } else if (p->cur_parser == PARSER_HEADER) {
    /* grpc_chttp2_header_parser_parse, hpack_parser.cc:1746-1782 */
    orc_h2_stream* s = lookup_stream(p, id);
    if (s != NULL && p->header_boundary) {
      s->header_frames_received++;
      if (p->header_eof) return mark_closed(p, id, 1, 0, ev, cap, nev);
    }
  } else if (p->cur_parser == PARSER_RST) {
    /* grpc_chttp2_rst_stream_parser_parse, frame_rst_stream.cc:99-119 */
    return mark_closed(p, id, 1, 1, ev, cap, nev);
  } else if (p->cur_parser == PARSER_HEADER3) {
    return ORC_H2_ERR_TOO_MANY_TRAILERS; /* hpack_parser.cc:1756-1759 (see begin_header_frame) */
  }
  return 0;
}

int orc_h2_parser_feed(orc_h2_parser* p, const uint8_t* data, uint64_t len,
                       orc_h2_event* ev, uint64_t cap, uint64_t* nev) {
  /* grpc_chttp2_perform_read, parsing.cc:56-253 */
  uint64_t cur = 0;
  int rc;
  while (cur < len) {
    if (p->state < ST_FH0) { /* :70-109 */
      if (data[cur] != (uint8_t)kClientPrefix[p->state]) return ORC_H2_ERR_PREFIX;
      cur++; p->state++;
      continue;
    }
    uint8_t c = data[cur];
    switch (p->state) {
      case 24: p->incoming_frame_size = (uint32_t)c << 16; p->state++; cur++; break;
      case 25: p->incoming_frame_size |= (uint32_t)c << 8; p->state++; cur++; break;
      case 26: p->incoming_frame_size |= c; p->state++; cur++; break;
      case 27: p->incoming_frame_type = c; p->state++; cur++; break;
      case 28: p->incoming_frame_flags = c; p->state++; cur++; break;
      case 29: p->incoming_stream_id = ((uint32_t)c & 0x7f) << 24; p->state++; cur++; break;
      case 30: p->incoming_stream_id |= (uint32_t)c << 16; p->state++; cur++; break;
      case 31: p->incoming_stream_id |= (uint32_t)c << 8; p->state++; cur++; break;
      case 32: { /* FH_8 :177-214 */
        p->incoming_stream_id |= c;
        cur++;
        if ((rc = begin_frame(p, ev, cap, nev))) return rc;
        if (p->incoming_frame_size == 0) {
          /* parse_frame_slice(empty, is_last=1) */
          if ((rc = push_ev(ev, cap, nev, ORC_EV_PAYLOAD, (uint32_t)cur, 0, 1, 0))) return rc;
          if ((rc = end_frame(p, ev, cap, nev))) return rc;
          p->state = ST_FH0;
        } else if (p->check_frame_size && p->incoming_frame_size > p->max_frame_size) {
          return ORC_H2_ERR_FRAME_TOO_LARGE;
        } else {
          p->state = ST_FRAME;
        }
        break;
      }
      case ST_FRAME: { /* :215-250 */
        uint64_t avail = len - cur;
        uint64_t take = avail < p->incoming_frame_size ? avail : p->incoming_frame_size;
        int is_last = take == p->incoming_frame_size;
        if ((rc = push_ev(ev, cap, nev, ORC_EV_PAYLOAD, (uint32_t)cur, (uint32_t)take,
                          (uint32_t)is_last, 0))) return rc;
        if (p->cur_parser == PARSER_DATA) {
          orc_h2_stream* d = lookup_stream(p, p->incoming_stream_id);
          rc = d ? grpc_deframe(d, data, cur, take, ev, cap, nev) : 0;
          if (rc == ORC_H2_ERR_GRPC_FRAME_TYPE) {
            /* stream error: reported, connection keeps parsing */
            int rc2 = push_ev(ev, cap, nev, ORC_EV_FRAME, 0xff, 0, p->incoming_stream_id,
                              ORC_H2_ERR_GRPC_FRAME_TYPE);
            if (rc2) return rc2;
          } else if (rc) {
            return rc;
          }
        }
        p->incoming_frame_size -= (uint32_t)take;
        cur += take;
        if (is_last) {
          if ((rc = end_frame(p, ev, cap, nev))) return rc;
          p->state = ST_FH0;
        }
        break;
      }
      default:
        return -1;
    }
  }
  return ORC_H2_OK;
}

/* ============================================================ cpu baseline */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

uint64_t orc_stream_baseline(uint64_t ring_cap, int max_sge, const uint8_t* wire,
                             const uint64_t* lens, uint64_t nslices, uint64_t n_msgs,
                             double* seconds, uint64_t* checksum) {
  orc_pair a, b;
  if (orc_pair_init(&a, ring_cap, max_sge) || orc_pair_init(&b, ring_cap, max_sge)) return 0;
  orc_pair_connect(&a, &b);
  orc_slice* sl = (orc_slice*)malloc(sizeof(orc_slice) * nslices);
  uint8_t* dst = (uint8_t*)malloc(ring_cap);
  uint64_t off = 0, delivered = 0, sum = 0;
  for (uint64_t i = 0; i < nslices; i++) {
    sl[i].ptr = wire + off;
    sl[i].len = lens[i];
    off += lens[i];
  }
  double t0 = now_s();
  for (uint64_t m = 0; m < n_msgs; m++) {
    uint64_t idx = 0, byte_idx = 0;
    while (idx < nslices) { /* rdma_write/rdma_flush loop, rdma_bp_posix.cc:470-524 */
      uint64_t sent = orc_pair_send(&a, sl + idx, nslices - idx, byte_idx);
      while (sent > 0) {
        uint64_t sl_len = sl[idx].len - byte_idx;
        if (sent >= sl_len) { sent -= sl_len; idx++; byte_idx = 0; }
        else { byte_idx += sent; sent = 0; }
      }
      /* receiver drains everything that landed */
      for (;;) {
        uint64_t alloc;
        uint64_t n = orc_endpoint_read(&b, dst, &alloc);
        if (n == 0) break;
        delivered += n;
        sum += dst[0] + dst[n - 1];
      }
    }
  }
  *seconds = now_s() - t0;
  if (checksum) *checksum = sum;
  free(sl);
  free(dst);
  orc_pair_destroy(&a);
  orc_pair_destroy(&b);
  return delivered;
}

/* The sequential schedule the device-resident jobs are checked against, entirely in C so that
 * the bench-sized configurations (tens of thousands of slices) take milliseconds: per pass, one
 * Send from the rdma_flush cursor (rdma_bp_posix.cc:470-524), then endpoint reads until one
 * would block (:180-291); `passes` passes over the same link.  Reports the delivered slice
 * lengths of the LAST pass, the Sends of the FIRST pass that produced a record, whether every
 * delivered byte equals the input stream, whether the receiver's ring ended all zero, and the
 * state words {remote_tail, remote_head, partial_write, head, moving_head, remain,
 * internal_read_size, credit_msgs(rx), leftover_cap}. */
int orc_stream_rounds(uint64_t ring_cap, int max_sge, const uint8_t* wire, const uint64_t* lens,
                      uint64_t nslices, int passes, uint64_t* out_lens, uint64_t out_cap,
                      uint64_t* n_out, uint64_t* first_rounds, uint64_t state[9], int* stream_ok,
                      int* ring_zero) {
  return orc_stream_rounds_burst(ring_cap, max_sge, 1, wire, lens, nslices, passes, out_lens, out_cap, n_out,
                                 first_rounds, state, stream_ok, ring_zero);
}

/* The same with `burst` Sends back to back per round before the reader runs (a sender that is
 * ahead of its reader: rdma_write -> rdma_flush retried on every writable edge while the peer
 * has not read yet).  A Send that finds no credit accepts nothing and changes nothing but
 * partial_write.  *first_rounds counts the Sends that accepted at least one byte. */
int orc_stream_rounds_burst(uint64_t ring_cap, int max_sge, int burst, const uint8_t* wire, const uint64_t* lens,
                            uint64_t nslices, int passes, uint64_t* out_lens, uint64_t out_cap,
                            uint64_t* n_out, uint64_t* first_rounds, uint64_t state[9], int* stream_ok,
                            int* ring_zero) {
  orc_pair a, b;
  if (burst < 1) burst = 1;
  if (orc_pair_init(&a, ring_cap, max_sge) || orc_pair_init(&b, ring_cap, max_sge)) return -1;
  orc_pair_connect(&a, &b);
  orc_slice* sl = (orc_slice*)malloc(sizeof(orc_slice) * (nslices ? nslices : 1));
  uint8_t* dst = (uint8_t*)malloc(ring_cap);
  uint64_t off = 0;
  for (uint64_t i = 0; i < nslices; i++) {
    sl[i].ptr = wire + off;
    sl[i].len = lens[i];
    off += lens[i];
  }
  int ok = 1, rc = 0;
  *first_rounds = 0;
  for (int p = 0; p < passes && rc == 0; p++) {
    uint64_t idx = 0, byte_idx = 0, rounds = 0, n = 0, pos = 0;
    while (idx < nslices) {
      for (int k = 0; k < burst && idx < nslices; k++) {
        uint64_t sent = orc_pair_send(&a, sl + idx, nslices - idx, byte_idx);
        if (sent) rounds++;
        while (sent > 0) {
          uint64_t sl_len = sl[idx].len - byte_idx;
          if (sent >= sl_len) { sent -= sl_len; idx++; byte_idx = 0; }
          else { byte_idx += sent; sent = 0; }
        }
      }
      for (;;) {
        uint64_t alloc;
        uint64_t got = orc_endpoint_read(&b, dst, &alloc);
        if (got == 0) break;
        if (memcmp(dst, wire + pos, got) != 0) ok = 0;
        pos += got;
        if (p == passes - 1) {
          if (n >= out_cap) { rc = -2; break; }
          out_lens[n] = got;
        }
        n++;
      }
      if (rc) break;
      if (rounds > (1ull << 24)) { rc = -3; break; }
    }
    if (pos != off) ok = 0;
    if (p == 0) *first_rounds = rounds;
    *n_out = n;
  }
  state[0] = a.remote_tail; state[1] = a.status_recv.remote_head; state[2] = (uint64_t)a.partial_write;
  state[3] = b.ring.head; state[4] = b.ring.moving_head; state[5] = b.ring.remain;
  state[6] = b.internal_read_size; state[7] = b.credit_msgs; state[8] = b.leftover_cap;
  int zero = 1;
  for (uint64_t i = 0; i < ring_cap; i++)
    if (b.ring.buf[i]) { zero = 0; break; }
  *stream_ok = ok;
  *ring_zero = zero;
  free(sl);
  free(dst);
  orc_pair_destroy(&a);
  orc_pair_destroy(&b);
  return rc;
}
