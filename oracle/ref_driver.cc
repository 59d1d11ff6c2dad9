// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// C-ABI driver around the REFERENCE's own ring codec.  This file is compiled
// together with /root/reference/src/core/lib/ibverbs/ring_buffer.cc (taken
// where it lies, unmodified, see oracle/Makefile) into oracle/_ref/libref_ring.so.
// It contains no copy of reference source: every record byte is produced by
// calling the reference's RingBufferPollable members/statics.
//
//  * ref_ring_*  : thin wrappers over RingBufferPollable
//                  (src/core/lib/ibverbs/ring_buffer.h:41-246, .cc:12-335).
//  * ref_pair_*  : a loop-back "pair" that drives those members in the order
//                  PairPollable does (src/core/lib/ibverbs/pair.cc:645-734 Send,
//                  :264-286 Recv, :294-301 GetWritableSize, :624-641
//                  updateStatus).  This library was written when pair.cc itself
//                  had not been built here yet (it needs libibverbs; since the end
//                  of round 3 oracle/ref_pair_trace.cc builds it over the software
//                  verbs of oracle/fakeverbs and the oracle is pinned to that too):
//                  ibv_post_send(RDMA_WRITE) is replaced by executing the work requests that the
//                  reference's GetWriteRequests() built: memcpy of every SGE to
//                  remote_addr, in order -- the RC in-order placement contract.
//
// The class below is *named* grpc_core::ibverbs::PairPollable on purpose:
// ring_buffer.h forward-declares that name as a friend of RingBufferPollable,
// which lets the driver observe head_/moving_head_/remain_ without touching the
// reference header.
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <time.h>
#include <vector>

#include <grpc/support/log.h>

#include "src/core/lib/ibverbs/ring_buffer.h"

// ---- stubs for the two gpr symbols ring_buffer.cc references -------------
extern "C" void gpr_log(const char* file, int line, gpr_log_severity severity,
                        const char* format, ...) {
  (void)file;
  (void)line;
  (void)severity;
  (void)format;
}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }

namespace grpc_core {
namespace ibverbs {

class PairPollable {
 public:
  struct status_report {  // pair.h:100-103
    uint64_t remote_head;
    int peer_exit;
  };

  PairPollable(uint64_t ring_size, int max_sge)
      : ring_mem_(ring_size, 0),
        staging_(ring_size / 2, 0),  // pair.cc:104
        ring_(ring_mem_.data(), ring_size),
        max_sge_(max_sge) {
    ring_.Init();
    memset(&status_recv_, 0, sizeof(status_recv_));
    memset(&status_send_, 0, sizeof(status_send_));
  }

  void Connect(PairPollable* peer) { peer_ = peer; }

  // pair.cc:645-734, with the verbs post replaced by execute_wrs().
  uint64_t Send(grpc_slice* slices, size_t slice_count, size_t byte_idx) {
    uint64_t remote_head = status_recv_.remote_head;  // pair.h:229-233
    uint64_t remote_tail = remote_tail_;
    uint64_t send_buf_tail = 0;
    uint64_t total = 0, written = 0;
    for (size_t i = 0; i < slice_count; i++) total += GRPC_SLICE_LENGTH(slices[i]);
    total -= byte_idx;
    sg_list_.clear();
    last_wr_count_ = 0;
    for (size_t i = 0; i < slice_count && (int)sg_list_.size() < max_sge_; i++) {
      uint8_t* p = GRPC_SLICE_START_PTR(slices[i]) + byte_idx;
      uint64_t len = GRPC_SLICE_LENGTH(slices[i]) - byte_idx;
      byte_idx = 0;
      uint64_t recv_free = peer_ring().GetFreeSize(remote_head, remote_tail);
      uint64_t send_free = staging_.size() - send_buf_tail;
      uint64_t pay = std::min(
          len, std::min(RingBufferPollable::CalculateWritableSize(send_free),
                        RingBufferPollable::CalculateWritableSize(recv_free)));
      if (pay == 0) break;
      uint64_t enc = RingBufferPollable::GetEncodedSize(pay);
      uint8_t* q = RingBufferPollable::AppendHeader(
          staging_.data() + send_buf_tail, pay);
      q = RingBufferPollable::AppendPayload(q, p, pay);
      q = RingBufferPollable::AppendFooter(q);
      if ((uint64_t)(q - (staging_.data() + send_buf_tail)) != enc) abort();
      ibv_sge sge;
      sge.addr = reinterpret_cast<uint64_t>(staging_.data()) + send_buf_tail;
      sge.length = (uint32_t)enc;
      sge.lkey = 0;
      sg_list_.push_back(sge);
      written += pay;
      send_buf_tail += enc;
      remote_tail = peer_ring().NextTail(remote_tail, enc);
    }
    partial_write_ = written < total;
    last_staging_used_ = send_buf_tail;
    if (!sg_list_.empty()) {
      std::array<ibv_send_wr, 2> wrs;
      remote_tail_ = peer_ring().GetWriteRequests(
          remote_tail_, peer_->ring_mem_.data(), 0, sg_list_, wrs);
      execute_wrs(&wrs[0]);
    }
    return written;
  }

  // pair.cc:103,113,120: the zero-copy send buffer
  void EnableZerocopy(uint64_t size) {
    zerocopy_.assign(size, 0);
    zerocopy_tail_ = 0;
  }

  // pair.cc:305-323
  uint8_t* AllocateSendBuffer(size_t size) {
    if (size == 0) return nullptr;
    uint32_t tail = zerocopy_tail_;
    if (tail != 0 || tail + size > zerocopy_.size()) return nullptr;
    zerocopy_tail_ = (uint32_t)(tail + size);
    return zerocopy_.data() + tail;
  }

  // pair.cc:793-941, with the verbs post replaced by execute_wrs().
  uint64_t SendZerocopy(grpc_slice* slices, size_t slice_count, size_t byte_idx) {
    uint64_t remote_head = status_recv_.remote_head;
    uint64_t remote_tail = remote_tail_;
    uint64_t send_buf_tail = 0;
    uint64_t total = 0, written = 0;
    for (size_t i = 0; i < slice_count; i++) total += GRPC_SLICE_LENGTH(slices[i]);
    total -= byte_idx;
    sg_list_.clear();
    last_wr_count_ = 0;
    for (size_t i = 0; i < slice_count && (int)sg_list_.size() < max_sge_; i++) {
      uint8_t* slice_ptr = GRPC_SLICE_START_PTR(slices[i]) + byte_idx;
      uint64_t slice_len = GRPC_SLICE_LENGTH(slices[i]) - byte_idx;
      uint64_t recv_buf_free = peer_ring().GetFreeSize(remote_head, remote_tail);
      uint64_t send_buf_free = staging_.size() - send_buf_tail;
      byte_idx = 0;
      if (!zerocopy_.empty() && slice_ptr >= zerocopy_.data() &&
          slice_ptr + slice_len <= zerocopy_.data() + zerocopy_.size()) {
        uint64_t pay = std::min(slice_len, RingBufferPollable::CalculateWritableSize(recv_buf_free));
        if (pay == 0 || send_buf_free < 3ul * RingBufferPollable::alignment ||
            (int)sg_list_.size() + 4 > max_sge_)
          break;
        uint64_t enc = RingBufferPollable::GetEncodedSize(pay);
        ibv_sge sge;
        sge.lkey = 0;
        sge.addr = reinterpret_cast<uint64_t>(staging_.data()) + send_buf_tail;  // header
        sge.length = sizeof(uint64_t);
        RingBufferPollable::AppendHeader(reinterpret_cast<uint8_t*>(sge.addr), pay);
        send_buf_tail += sge.length;
        sg_list_.push_back(sge);
        sge.addr = reinterpret_cast<uint64_t>(slice_ptr);  // payload where it lies
        sge.length = (uint32_t)pay;
        sg_list_.push_back(sge);
        uint64_t pad = RingBufferPollable::round_up(pay) - pay;
        if (pad > 0) {
          sge.addr = reinterpret_cast<uint64_t>(staging_.data()) + send_buf_tail;
          sge.length = (uint32_t)pad;
          send_buf_tail += sge.length;
          sg_list_.push_back(sge);
        }
        sge.addr = reinterpret_cast<uint64_t>(staging_.data()) + send_buf_tail;  // footer
        sge.length = sizeof(uint64_t);
        RingBufferPollable::AppendFooter(reinterpret_cast<uint8_t*>(sge.addr));
        send_buf_tail += sge.length;
        sg_list_.push_back(sge);
        zerocopy_tail_ -= (uint32_t)pay;
        written += pay;
        remote_tail = peer_ring().NextTail(remote_tail, enc);
        zerocopy_bytes_ += pay;
      } else {
        uint64_t pay = std::min(
            slice_len, std::min(RingBufferPollable::CalculateWritableSize(send_buf_free),
                                RingBufferPollable::CalculateWritableSize(recv_buf_free)));
        if (pay == 0) break;
        uint64_t enc = RingBufferPollable::GetEncodedSize(pay);
        uint8_t* q = RingBufferPollable::AppendHeader(staging_.data() + send_buf_tail, pay);
        q = RingBufferPollable::AppendPayload(q, slice_ptr, pay);
        q = RingBufferPollable::AppendFooter(q);
        if ((uint64_t)(q - (staging_.data() + send_buf_tail)) != enc) abort();
        ibv_sge sge;
        sge.addr = reinterpret_cast<uint64_t>(staging_.data()) + send_buf_tail;
        sge.length = (uint32_t)enc;
        sge.lkey = 0;
        sg_list_.push_back(sge);
        written += pay;
        send_buf_tail += enc;
        remote_tail = peer_ring().NextTail(remote_tail, enc);
        copy_bytes_ += pay;
      }
    }
    partial_write_ = written < total;
    last_staging_used_ = send_buf_tail;
    if (!sg_list_.empty()) {
      std::array<ibv_send_wr, 2> wrs;
      remote_tail_ = peer_ring().GetWriteRequests(
          remote_tail_, peer_->ring_mem_.data(), 0, sg_list_, wrs);
      execute_wrs(&wrs[0]);
    }
    return written;
  }

  // pair.cc:264-286
  uint64_t Recv(void* buf, uint64_t capacity) {
    uint64_t internal = 0;
    uint64_t n = ring_.Read(buf, capacity, &internal);
    internal_read_size_ += internal;
    if (internal_read_size_ >= ring_.get_capacity() / 2) {
      status_send_.remote_head = ring_.get_head();
      // updateStatus(): RDMA-write the 16-byte report into the peer's status
      // receive buffer (pair.cc:624-641).
      memcpy(&peer_->status_recv_, &status_send_, sizeof(status_report));
      credit_msgs_++;
      internal_read_size_ = 0;
    }
    return n;
  }

  uint64_t GetWritableSize() const {  // pair.cc:294-301
    return peer_ring().GetWritableSize(status_recv_.remote_head, remote_tail_);
  }

  RingBufferPollable& ring() { return ring_; }
  const RingBufferPollable& peer_ring() const { return peer_->ring_; }
  RingBufferPollable& peer_ring() { return peer_->ring_; }

  uint64_t head() const { return ring_.head_.load(); }
  uint64_t moving_head() const { return ring_.moving_head_; }
  uint64_t remain() const { return ring_.remain_.load(); }

  std::vector<uint8_t> ring_mem_;
  std::vector<uint8_t> staging_;
  RingBufferPollable ring_;
  int max_sge_;
  PairPollable* peer_ = nullptr;
  uint64_t remote_tail_ = 0;
  uint64_t internal_read_size_ = 0;
  bool partial_write_ = false;
  uint64_t last_staging_used_ = 0;
  uint64_t credit_msgs_ = 0;
  status_report status_recv_;
  status_report status_send_;
  std::vector<ibv_sge> sg_list_;
  std::vector<uint8_t> zerocopy_;  // send_buffers_[kZeroCopyBuffer]
  uint32_t zerocopy_tail_ = 0;     // std::atomic_uint32_t zerocopy_buffer_tail_, pair.h:178
  uint64_t zerocopy_bytes_ = 0, copy_bytes_ = 0;
  // Trace of the last Send's work requests: (remote offset, length) per WR.
  uint64_t last_wr_[2][2] = {{0, 0}, {0, 0}};
  int last_wr_count_ = 0;

 private:
  void execute_wrs(ibv_send_wr* wr) {
    last_wr_count_ = 0;
    uint8_t* base = peer_->ring_mem_.data();
    for (; wr != nullptr; wr = wr->next) {
      uint8_t* dst = reinterpret_cast<uint8_t*>(wr->wr.rdma.remote_addr);
      uint64_t off = 0;
      for (int i = 0; i < wr->num_sge; i++) {
        memcpy(dst + off, reinterpret_cast<void*>(wr->sg_list[i].addr),
               wr->sg_list[i].length);
        off += wr->sg_list[i].length;
      }
      if (last_wr_count_ < 2) {
        last_wr_[last_wr_count_][0] = (uint64_t)(dst - base);
        last_wr_[last_wr_count_][1] = off;
        last_wr_count_++;
      }
    }
  }
};

}  // namespace ibverbs
}  // namespace grpc_core

using grpc_core::ibverbs::PairPollable;
using grpc_core::ibverbs::RingBufferPollable;

namespace {
// A non-null refcount pointer marks a slice as "refcounted" for the
// GRPC_SLICE_* accessor macros (include/grpc/impl/codegen/slice.h:96-101); it is
// never dereferenced by the ring codec.
struct grpc_slice_refcount* kFakeRefcount =
    reinterpret_cast<struct grpc_slice_refcount*>(uintptr_t{0x10});

std::vector<grpc_slice> make_slices(const uint8_t* const* ptrs,
                                    const uint64_t* lens, size_t n,
                                    int inline_small) {
  std::vector<grpc_slice> out(n);
  for (size_t i = 0; i < n; i++) {
    if (inline_small && lens[i] <= GRPC_SLICE_INLINED_SIZE) {
      out[i].refcount = nullptr;
      out[i].data.inlined.length = (uint8_t)lens[i];
      memcpy(out[i].data.inlined.bytes, ptrs[i], lens[i]);
    } else {
      out[i].refcount = kFakeRefcount;
      out[i].data.refcounted.length = lens[i];
      out[i].data.refcounted.bytes = const_cast<uint8_t*>(ptrs[i]);
    }
  }
  return out;
}
}  // namespace

extern "C" {

// ---------------- statics (ring_buffer.h:180-189) -------------------------
uint64_t ref_encoded_size(uint64_t payload) {
  return RingBufferPollable::GetEncodedSize(payload);
}
uint64_t ref_calc_writable(uint64_t space) {
  return RingBufferPollable::CalculateWritableSize(space);
}
uint64_t ref_reserved_space(void) { return RingBufferPollable::reserved_space; }
uint64_t ref_sizeof_grpc_slice(void) { return sizeof(grpc_slice); }
uint64_t ref_sizeof_grpc_slice_buffer(void) { return sizeof(grpc_slice_buffer); }
uint64_t ref_slice_inlined_size(void) { return GRPC_SLICE_INLINED_SIZE; }

// ---------------- bare ring ----------------------------------------------
struct ref_ring {
  std::vector<uint8_t> mem;
  RingBufferPollable* rb;
};

void* ref_ring_new(uint64_t size) {
  auto* r = new ref_ring;
  r->mem.assign(size, 0);
  r->rb = new RingBufferPollable(r->mem.data(), size);
  r->rb->Init();
  return r;
}
void ref_ring_free(void* h) {
  auto* r = static_cast<ref_ring*>(h);
  delete r->rb;
  delete r;
}
uint8_t* ref_ring_mem(void* h) { return static_cast<ref_ring*>(h)->mem.data(); }
uint64_t ref_ring_write(void* h, uint64_t tail, const void* src, uint64_t n) {
  return static_cast<ref_ring*>(h)->rb->Write(tail, const_cast<void*>(src), n);
}
uint64_t ref_ring_readable(void* h) {
  return static_cast<ref_ring*>(h)->rb->GetReadableSize();
}
int ref_ring_has_message(void* h) {
  return static_cast<ref_ring*>(h)->rb->HasMessage() ? 1 : 0;
}
uint64_t ref_ring_read(void* h, void* dst, uint64_t cap, uint64_t* internal) {
  return static_cast<ref_ring*>(h)->rb->Read(dst, cap, internal);
}
uint64_t ref_ring_free_size(void* h, uint64_t head, uint64_t tail) {
  return static_cast<ref_ring*>(h)->rb->GetFreeSize(head, tail);
}
uint64_t ref_ring_writable(void* h, uint64_t head, uint64_t tail) {
  return static_cast<ref_ring*>(h)->rb->GetWritableSize(head, tail);
}
uint64_t ref_ring_get_head(void* h) {  // == moving_head_ (ring_buffer.cc:334)
  return static_cast<ref_ring*>(h)->rb->get_head();
}
// GetWriteRequests(size, tail, reqs): ring_buffer.cc:233-259.  out[k] =
// {src_offset, dst_offset, size}; returns the request count, *new_tail set.
int ref_ring_write_requests(void* h, uint64_t size, uint64_t tail,
                            uint64_t out[2][3], uint64_t* new_tail) {
  std::vector<grpc_core::ibverbs::ring_buffer_write_request> reqs;
  *new_tail = static_cast<ref_ring*>(h)->rb->GetWriteRequests(size, tail, reqs);
  for (size_t i = 0; i < reqs.size() && i < 2; i++) {
    out[i][0] = reqs[i].src_offset;
    out[i][1] = reqs[i].dst_offset;
    out[i][2] = reqs[i].size;
  }
  return (int)reqs.size();
}

// ---------------- loop-back pair -----------------------------------------
struct ref_pair_link {
  PairPollable* a;
  PairPollable* b;
};

void* ref_link_new(uint64_t ring_size, int max_sge) {
  auto* l = new ref_pair_link;
  l->a = new PairPollable(ring_size, max_sge);
  l->b = new PairPollable(ring_size, max_sge);
  l->a->Connect(l->b);
  l->b->Connect(l->a);
  return l;
}
void ref_link_free(void* h) {
  auto* l = static_cast<ref_pair_link*>(h);
  delete l->a;
  delete l->b;
  delete l;
}
static PairPollable* side(void* h, int s) {
  auto* l = static_cast<ref_pair_link*>(h);
  return s == 0 ? l->a : l->b;
}
uint64_t ref_pair_send(void* h, int s, const uint8_t* const* ptrs,
                       const uint64_t* lens, uint64_t n, uint64_t byte_idx,
                       int inline_small) {
  auto slices = make_slices(ptrs, lens, n, inline_small);
  return side(h, s)->Send(slices.data(), slices.size(), byte_idx);
}
uint64_t ref_pair_recv(void* h, int s, void* dst, uint64_t cap) {
  return side(h, s)->Recv(dst, cap);
}
uint64_t ref_pair_readable(void* h, int s) {
  return side(h, s)->ring().GetReadableSize();
}
int ref_pair_has_message(void* h, int s) {
  return side(h, s)->ring().HasMessage() ? 1 : 0;
}
int ref_pair_partial_write(void* h, int s) {
  return side(h, s)->partial_write_ ? 1 : 0;
}
uint64_t ref_pair_writable(void* h, int s) { return side(h, s)->GetWritableSize(); }
uint8_t* ref_pair_ring_mem(void* h, int s) { return side(h, s)->ring_mem_.data(); }
uint8_t* ref_pair_staging_mem(void* h, int s) { return side(h, s)->staging_.data(); }
uint64_t ref_pair_staging_used(void* h, int s) { return side(h, s)->last_staging_used_; }
// state[0..8) = head_, moving_head_, remain_, remote_tail_, remote_head view,
//               internal_read_size_, credit messages sent, partial flag
void ref_pair_state(void* h, int s, uint64_t state[8]) {
  PairPollable* p = side(h, s);
  state[0] = p->head();
  state[1] = p->moving_head();
  state[2] = p->remain();
  state[3] = p->remote_tail_;
  state[4] = p->status_recv_.remote_head;
  state[5] = p->internal_read_size_;
  state[6] = p->credit_msgs_;
  state[7] = p->partial_write_ ? 1 : 0;
}
// Single-thread streaming pass over the reference-built ring codec: the same loop as
// orc_stream_baseline (oracle/grdma_oracle.c) -- rdma_write/rdma_flush on one side
// (rdma_bp_posix.cc:470-524), rdma_continue_read/rdma_do_read on the other (:180-326) --
// with every byte encoded and decoded by the reference's own ring_buffer.cc.  bench.py
// times it as the CPU baseline (kind "reference") when oracle/_ref is present.
uint64_t ref_stream_baseline(uint64_t ring_size, int max_sge, const uint8_t* wire,
                             const uint64_t* lens, uint64_t nslices, uint64_t n_msgs,
                             double* seconds, uint64_t* checksum) {
  PairPollable a(ring_size, max_sge), b(ring_size, max_sge);
  a.Connect(&b);
  b.Connect(&a);
  std::vector<const uint8_t*> ptrs(nslices);
  uint64_t off = 0;
  for (uint64_t i = 0; i < nslices; i++) {
    ptrs[i] = wire + off;
    off += lens[i];
  }
  std::vector<grpc_slice> sl = make_slices(ptrs.data(), lens, nslices, 0);
  std::vector<uint8_t> dst(ring_size);
  uint64_t delivered = 0, sum = 0, leftover = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t m = 0; m < n_msgs; m++) {
    uint64_t idx = 0, byte_idx = 0;
    while (idx < nslices) {
      uint64_t sent = a.Send(sl.data() + idx, nslices - idx, byte_idx);
      while (sent > 0) {
        const uint64_t sl_len = lens[idx] - byte_idx;
        if (sent >= sl_len) { sent -= sl_len; idx++; byte_idx = 0; }
        else { byte_idx += sent; sent = 0; }
      }
      for (;;) {  // one endpoint read per iteration
        const uint64_t readable = b.ring().GetReadableSize();
        const uint64_t alloc = leftover ? leftover : (readable > 256 ? readable : 256);
        uint64_t total = 0;
        while (total < alloc) {
          const uint64_t n = b.Recv(dst.data() + total, alloc - total);
          if (n == 0) break;
          total += n;
        }
        leftover = total ? alloc - total : alloc;
        if (total == 0) break;
        delivered += total;
        sum += dst[0] + dst[total - 1];
      }
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (checksum) *checksum = sum;
  return delivered;
}

// Zero-copy send buffer (pair.cc:103-120, 305-323, 793-941).  A slice is given as an offset into the
// zero-copy buffer (zc_off[i] >= 0) or as a plain pointer (zc_off[i] < 0, ptrs[i]).
void ref_pair_enable_zerocopy(void* h, int s, uint64_t size) { side(h, s)->EnableZerocopy(size); }
int64_t ref_pair_allocate_send_buffer(void* h, int s, uint64_t size) {
  PairPollable* p = side(h, s);
  uint8_t* q = p->AllocateSendBuffer(size);
  return q ? (int64_t)(q - p->zerocopy_.data()) : -1;
}
uint8_t* ref_pair_zerocopy_mem(void* h, int s) { return side(h, s)->zerocopy_.data(); }
uint64_t ref_pair_send_zerocopy(void* h, int s, const uint8_t* const* ptrs, const int64_t* zc_off,
                                const uint64_t* lens, uint64_t n, uint64_t byte_idx) {
  PairPollable* p = side(h, s);
  std::vector<const uint8_t*> real(n);
  for (uint64_t i = 0; i < n; i++) real[i] = zc_off[i] >= 0 ? p->zerocopy_.data() + zc_off[i] : ptrs[i];
  auto slices = make_slices(real.data(), lens, n, 0);
  return p->SendZerocopy(slices.data(), slices.size(), byte_idx);
}
// {zerocopy_buffer_tail_, zerocopy_bytes_, copy_bytes_, entries of the last scatter-gather list}
void ref_pair_zerocopy_state(void* h, int s, uint64_t out[4]) {
  PairPollable* p = side(h, s);
  out[0] = p->zerocopy_tail_;
  out[1] = p->zerocopy_bytes_;
  out[2] = p->copy_bytes_;
  out[3] = p->sg_list_.size();
}

// Work requests of the last Send: out[k] = {remote ring offset, length}.
int ref_pair_last_wrs(void* h, int s, uint64_t out[2][2]) {
  PairPollable* p = side(h, s);
  for (int i = 0; i < p->last_wr_count_; i++) {
    out[i][0] = p->last_wr_[i][0];
    out[i][1] = p->last_wr_[i][1];
  }
  return p->last_wr_count_;
}

}  // extern "C"
