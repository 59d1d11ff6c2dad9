// grdma_endpoint_impl.hpp -- the endpoint logic of src/core/lib/iomgr/rdma_bp_posix.cc over the C ABI of
// grdma_amd.h, written ONCE and instantiated twice:
//   * grpc-rdma_amd/csrc/grdma_endpoint.cc   with the plain mirror types of include/grdma_endpoint.hpp
//                                            (what the tests, tools and bench legs of this repository execute);
//   * integration/rdma_hip_posix.cc          with the iomgr types of the gRPC 1.38 + RR-Compound tree itself
//                                            (what a maintainer drops into src/core/lib/iomgr/).
// So the code a maintainer ships is the code the tests run; the two files only differ in their traits: how a slice
// buffer is walked, how a closure is run, how an error is made, how the fd is asked for the next readable /
// writable edge.
//
// Follows rdma_read / rdma_handle_read / rdma_continue_read / rdma_do_read (rdma_bp_posix.cc:180-376) and
// rdma_write / rdma_handle_write / rdma_flush (:470-586) function by function.  What differs from the reference
// is WHERE the work happens, not the contract:
//   * a write enqueues one Send on the device (gather of the caller's slices, record encode, wire write) and
//     returns; the write callback runs from the writable edge once the Send has completed -- "may complete
//     synchronously or later" is what endpoint.h:78-91 allows.  A partial Send (credit, max_sge) is continued from
//     the same edge, as rdma_handle_write does;
//   * a read that finds a message enqueues one drain -- up to kReadAhead endpoint reads in ONE device pass, each the
//     slice rdma_continue_read would have sized (max(256, readable)) and rdma_do_read would have filled -- and
//     returns; the completions are handed to the transport one read callback each, as slices that point into the
//     pinned receive window the scatter kernel wrote (no copy on the host; the window is released when the
//     transport drops the last slice);
//   * the event engine polls grdma_endpoint_readable / _writable (plain loads of pinned memory) where it polled
//     PairPollable::HasMessage / HasPendingWrites.
//
// Traits T (all static unless noted):
//   types    host (the endpoint object embedding the core), slice_buffer, closure, error (a handle; none() is "OK")
//   buffers  count(sb) length(sb) slice_ptr(sb,i) slice_len(sb,i) reset_and_unref(sb)
//            add_copied(sb, bytes, len)               a fresh slice holding a copy
//            add_window(sb, bytes, len, window)       a slice pointing at `bytes`, takes a window reference,
//                                                     grdma_window_unref when the slice is destroyed
//   errors   none() ref(e) annotate(host*, msg)       rdma_annotate_error(GRPC_ERROR_CREATE...(msg), rdma), :86-96
//   closures run(host*, closure*, error)              grpc_core::Closure::Run
//            run_read_done(host*)                     Closure::Run(&rdma->read_done_closure, NONE), :370-374
//   fd       notify_on_read(host*) notify_on_write(host*) is_shutdown(host*)
//   refs     ref(host*) unref(host*)                  RDMA_REF / RDMA_UNREF
//   scope    struct scope { scope(int op); }          GRPCProfiler; op ids OP_READ ... OP_WRITE
#ifndef GRDMA_ENDPOINT_IMPL_HPP
#define GRDMA_ENDPOINT_IMPL_HPP

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "grdma_amd.h"

namespace grdma_ep {

constexpr uint64_t kReadAhead = 1024;    // endpoint reads performed per device pass
constexpr size_t kWriteWindow = 4000;    // slices handed to one grdma_endpoint_write_begin (the ABI takes 4095)
constexpr size_t kZeroCopyMin = 512;     // shorter slices are copied out of the window (a 9-byte frame header
                                         // should not keep 64 MiB of pinned memory alive)

template <class T>
struct core {
  typedef typename T::host host_t;
  typedef typename T::slice_buffer sb_t;
  typedef typename T::closure closure_t;
  typedef typename T::error error_t;

  host_t* h = nullptr;
  grdma_pair* pair = nullptr;
  bool is_first_read = true;   // rdma_bp_posix.cc:50-52
  int inq = 1;                 // :745
  sb_t* incoming_buffer = nullptr;
  sb_t* outgoing_buffer = nullptr;
  closure_t* read_cb = nullptr;
  closure_t* write_cb = nullptr;
  // write side
  std::vector<grdma_slice> out_views;  // {ptr, len} of outgoing_buffer's slices
  size_t out_next = 0;                 // first view not yet handed to the pair
  bool window_active = false;          // the pair holds a window of views that has not gone out whole
  bool send_submitted = false;         // a Send is in flight
  // read side
  std::vector<grdma_read_slice> ahead; // completions of the last drain not yet handed to the transport
  size_t ahead_next = 0;
  grdma_window* ahead_win = nullptr;   // the receive window they lie in (this object's reference)
  bool ahead_copy = false;             // hand them out as copies (every other window is still held by the transport)
  uint64_t arm_reads = 0;              // != 0: keep a read armed with the pair while waiting (latency mode)

  void init(host_t* host, grdma_pair* p) {
    h = host;
    pair = p;
    ahead.reserve(kReadAhead);
  }
  // rdma_free: nothing of the transport's may be referenced afterwards
  void release() {
    if (ahead_win) grdma_window_unref(ahead_win);
    ahead_win = nullptr;
    ahead.clear();
  }

  // ------------------------------------------------------------------------------------------- read
  void call_read_cb(error_t error) {  // :166-172
    closure_t* cb = read_cb;
    read_cb = nullptr;
    incoming_buffer = nullptr;
    T::run(h, cb, error);
  }

  void read_fail(const char* what) {
    T::reset_and_unref(incoming_buffer);
    call_read_cb(T::annotate(h, what));
    T::unref(h);
  }

  // "We've consumed the edge, request a new one" (:241-243), or report what get_status() says (:220-238)
  void read_would_block() {
    inq = 0;
    const int status = grdma_pair_get_status(pair);
    if (status == GRDMA_PAIR_HALF_CLOSED) return read_fail("Pair closed");
    if (status == GRDMA_PAIR_ERROR) return read_fail((std::string("Pair error, ") + grdma_last_error()).c_str());
    if (arm_reads) grdma_pair_arm_read(pair, arm_reads);
    T::notify_on_read(h);
  }

  // rdma_continue_read + rdma_do_read (:306-326, :180-291)
  void do_read() {
    typename T::scope profiler(T::OP_DO_READ);
    for (;;) {
      if (ahead_next < ahead.size()) {  // a completion of the last drain: one slice, one read callback
        const grdma_read_slice s = ahead[ahead_next++];
        const uint8_t* bytes = static_cast<const uint8_t*>(grdma_window_base(ahead_win)) + s.off;
        if (ahead_copy || s.len < kZeroCopyMin) T::add_copied(incoming_buffer, bytes, (size_t)s.len);
        else T::add_window(incoming_buffer, bytes, (size_t)s.len, ahead_win);
        if (ahead_next == ahead.size()) {  // the window now belongs to the slices that point into it
          grdma_window_unref(ahead_win);
          ahead_win = nullptr;
          ahead.clear();
          ahead_next = 0;
        }
        inq = 1;
        call_read_cb(T::none());
        T::unref(h);
        return;
      }
      if (grdma_endpoint_drain_state(pair) != 0) {  // (submitted below, or by the peer's sender for an armed read)
        int would_block = 0;
        grdma_window* win = nullptr;
        ahead.resize(kReadAhead);
        const int64_t n = grdma_endpoint_read_test(pair, ahead.data(), kReadAhead, &would_block, &win);
        if (n == -(int64_t)GRDMA_ERR_AGAIN) {  // still on the device: the readable edge comes when it is done
          ahead.clear();
          T::notify_on_read(h);
          return;
        }
        if (n < 0) {
          ahead.clear();
          return read_fail((std::string("Pair error, ") + grdma_last_error()).c_str());
        }
        ahead.resize((size_t)n);
        ahead_next = 0;
        if (n > 0) {
          ahead_win = win;
          ahead_copy = !any_window_free();
          // more has arrived meanwhile?  Then the next drain runs on the device while this one's slices are handed
          // to the transport (it needs a window of its own; none free = it waits until this one has been released)
          if (!ahead_copy && grdma_pair_has_message(pair) > 0) grdma_endpoint_read_submit(pair, kReadAhead);
          continue;
        }
        grdma_window_unref(win);
        return read_would_block();  // the drain found no complete record
      }
      // nothing pending.  A record behind head_ (or a half-read one)?  That is a host load.
      if (grdma_pair_has_message(pair) <= 0) return read_would_block();
      const int rc = grdma_endpoint_read_submit(pair, kReadAhead);
      if (rc < 0) return read_fail((std::string("Pair error, ") + grdma_last_error()).c_str());
      // (rc == 1: the transport holds slices of every window; the edge stays up and the next pass tries again)
      T::notify_on_read(h);
      return;
    }
  }

  bool any_window_free() {
    // the window being handed out is held by this object; is another one free for the next drain?
    return grdma_endpoint_free_windows(pair) > 0;
  }

  void handle_read(error_t error) {  // :328-341
    typename T::scope profiler(T::OP_HANDLE_READ);
    if (T::is_error(error)) {
      T::reset_and_unref(incoming_buffer);
      call_read_cb(T::ref(error));
      T::unref(h);
      return;
    }
    typename T::scope cont(T::OP_CONTINUE_READ);
    do_read();
  }

  void read(sb_t* incoming, closure_t* cb, bool urgent) {  // :343-376
    typename T::scope profiler(T::OP_READ);
    if (read_cb != nullptr) abort();  // GPR_ASSERT(rdma->read_cb == nullptr)
    read_cb = cb;
    incoming_buffer = incoming;
    T::reset_and_unref(incoming);
    T::ref(h);
    if (is_first_read) {
      is_first_read = false;
      if (arm_reads) grdma_pair_arm_read(pair, arm_reads);
      T::notify_on_read(h);
    } else if (!urgent && inq == 0) {
      if (arm_reads) grdma_pair_arm_read(pair, arm_reads);
      T::notify_on_read(h);
    } else {
      T::run_read_done(h);
    }
  }

  // ------------------------------------------------------------------------------------------ write
  void forget_write() {  // the pair must not keep views of slices that are about to be unreffed
    grdma_endpoint_write_abort(pair);
    window_active = false;
    send_submitted = false;
    out_views.clear();
    out_next = 0;
  }

  // rdma_flush (:470-524).  true = the whole buffer went out or *error is set; false = wait for the writable edge
  // (the Send in flight completes, or the peer returns credit).
  bool flush(error_t* error) {
    typename T::scope profiler(T::OP_FLUSH);
    *error = T::none();
    auto fail_with = [&](const std::string& what) {
      *error = T::annotate(h, what.c_str());
      forget_write();
      T::reset_and_unref(outgoing_buffer);
      return true;
    };
    for (;;) {
      if (send_submitted) {
        int done = 0;
        int64_t sent = 0;
        const int r = grdma_endpoint_write_test(pair, &done, &sent);
        if (r == 0) return false;  // still on the device
        send_submitted = false;
        if (r < 0) return fail_with(std::string("RDMA Pair has an internal error, ") + grdma_last_error());
        if (done) {
          window_active = false;
          continue;
        }
        // partial send (:499-518)
        const int status = grdma_pair_get_status(pair);
        if (status == GRDMA_PAIR_HALF_CLOSED) return fail_with("Peer has been exited");
        if (status != GRDMA_PAIR_CONNECTED) return fail_with(std::string("RDMA Pair has an internal error, ") + grdma_last_error());
        if (grdma_pair_writable_size(pair) <= 0) return false;  // out of credit: the edge comes with the peer's report
        // (max_sge, or credit that has come back since: the next Send continues from the cursor right away)
      }
      if (!window_active) {
        if (out_next >= out_views.size()) break;
        const size_t cnt = out_views.size() - out_next < kWriteWindow ? out_views.size() - out_next : kWriteWindow;
        if (grdma_endpoint_write_begin(pair, out_views.data() + out_next, cnt, GRDMA_MEM_HOST) < 0)
          return fail_with(std::string("RDMA Pair has an internal error, ") + grdma_last_error());
        out_next += cnt;
        window_active = true;
      }
      if (grdma_endpoint_write_submit(pair) < 0)
        return fail_with(std::string("RDMA Pair has an internal error, ") + grdma_last_error());
      send_submitted = true;
      return false;
    }
    out_views.clear();
    out_next = 0;
    T::reset_and_unref(outgoing_buffer);  // :519-523
    return true;
  }

  void handle_write(error_t error) {  // :527-557
    typename T::scope profiler(T::OP_HANDLE_WRITE);
    if (T::is_error(error)) {
      closure_t* cb = write_cb;
      write_cb = nullptr;
      forget_write();
      T::run(h, cb, T::ref(error));
      T::unref(h);
      return;
    }
    error_t err;
    if (!flush(&err)) {
      T::notify_on_write(h);
    } else {
      closure_t* cb = write_cb;
      write_cb = nullptr;
      T::run(h, cb, err);
      T::unref(h);
    }
  }

  void write(sb_t* buf, closure_t* cb) {  // :559-586
    typename T::scope profiler(T::OP_WRITE);
    if (write_cb != nullptr) abort();  // GPR_ASSERT(rdma->write_cb == nullptr)
    if (T::length(buf) == 0) {
      T::run(h, cb, T::is_shutdown(h) ? T::annotate(h, "EOF") : T::none());
      return;
    }
    outgoing_buffer = buf;
    const size_t n = T::count(buf);
    out_views.resize(n);
    for (size_t i = 0; i < n; i++) out_views[i] = grdma_slice{T::slice_ptr(buf, i), (uint64_t)T::slice_len(buf, i)};
    out_next = 0;
    window_active = false;
    send_submitted = false;
    error_t error;
    if (!flush(&error)) {
      T::ref(h);
      write_cb = cb;
      T::notify_on_write(h);
    } else {
      T::run(h, cb, error);
    }
  }
};

}  // namespace grdma_ep
#endif  // GRDMA_ENDPOINT_IMPL_HPP
