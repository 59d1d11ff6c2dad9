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
//   * a write that fits a SEND BUFFER of the endpoint (two pinned, device-visible buffers, GRPC_RDMA_HIP_SEND_BUFFER_KB
//     each, default 4096, 0 = off) completes as soon as its bytes have been copied there -- what a socket does with
//     its send buffer, and what rdma_flush does when Send() copies the slices into the registered send buffer: the
//     caller's slices are free, the bytes are the endpoint's to deliver.  The device chain of write k then runs while
//     the transport prepares and copies write k + 1; an error of a write that has already completed is reported by the
//     next write (like a socket's).  Writes that do not fit, or that arrive while both buffers are taken, go the
//     way described above, behind the buffered ones.  While a Send is in flight the buffer that waits behind it
//     COALESCES: further writes are appended to it (slice boundaries kept) until another write of that size would
//     not fit -- the buffer, the sixteen Sends of a burst, a quarter of the ring -- and only then does its chain go
//     into the send stream.  A chain's planner, commit and launch gaps cost the same for one message as for three,
//     and so does the host's work to submit it (what a socket's send buffer does with the bytes written while the
//     NIC is busy; GRPC_RDMA_HIP_COALESCE=0: every write a chain of its own);
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
//   errors   none() ref(e) drop(e) annotate(host*, msg)   rdma_annotate_error(GRPC_ERROR_CREATE...(msg), rdma), :86-96
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
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "grdma_amd.h"

namespace grdma_ep {

constexpr uint64_t kReadAheadMax = 4096; // most endpoint reads one device pass may perform (GRPC_RDMA_HIP_READ_AHEAD)
constexpr uint64_t kReadAhead = 1024;    // ... and the default
constexpr size_t kWriteWindow = 4000;    // slices handed to one grdma_endpoint_write_begin (the ABI takes 4095)
constexpr size_t kSendBufferMin = 8192;  // shorter writes go to the pair as they are (a unary-sized write rides inline
                                         // in the latency engine's command: nothing to gain from a copy in front of it)
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
  // send buffers (early completion): a write copied into one completes at once; at most one of them is with the pair
  // (out_views point into it, outgoing_buffer is null), the other one may wait behind it
  struct send_buffer {
    uint8_t* mem = nullptr;
    std::vector<grdma_slice> views;
    size_t used = 0;                   // bytes of mem that views cover
  };
  send_buffer sbuf[2];
  size_t sbuf_cap = 0;                 // bytes per buffer, 0 = off
  size_t co_bytes = 0, co_slices = 0;  // what the waiting buffer may grow to by coalescing (0 = off): init()
  int bg_running = -1, bg_waiting = -1;
  bool bg_queued = false;              // the waiting buffer's chain is already in the pair's send stream
                                       // (grdma_endpoint_write_queue): adopted, not submitted, when its turn comes
  bool deferred = false;               // a write of the ordinary kind (deferred_buf, write_cb) waits behind them
  sb_t* deferred_buf = nullptr;
  bool bg_failed = false;
  std::string bg_error;                // what a buffered write failed with: the next write reports it
  std::string last_failure;            // the text of the last error flush() made
  int out_flags = GRDMA_MEM_HOST;      // memory kind of out_views (a send buffer is device-visible: 0)
  std::recursive_mutex wmu;            // write() and handle_write() may come from different threads once a write
                                       // has completed before its Send has
  // read side
  std::vector<grdma_read_slice> ahead; // completions of the last drain not yet handed to the transport
  size_t ahead_next = 0;
  grdma_window* ahead_win = nullptr;   // the receive window they lie in (this object's reference)
  bool ahead_copy = false;             // hand them out as copies (every other window is still held by the transport)
  bool open_read_noted = false;        // the pair knows that a read is open (a drain ended in a would-block, or
                                       // grdma_endpoint_read_idle told it): the next read fills that 256-byte slice first
  uint64_t arm_reads = 0;              // != 0: keep a read armed with the pair while waiting (latency mode)
  uint64_t read_ahead = kReadAhead;    // endpoint reads per device pass: GRPC_RDMA_HIP_READ_AHEAD, 1 ... kReadAheadMax
                                       // (1 = every read sized and filled at the moment the transport asks for it, as
                                       // rdma_continue_read / rdma_do_read do: what the reference-trace test runs with)

  void init(host_t* host, grdma_pair* p) {
    h = host;
    pair = p;
    ahead.reserve(kReadAheadMax);
    const char* e = getenv("GRPC_RDMA_HIP_SEND_BUFFER_KB");
    const long kb = e ? atol(e) : 4096;
    sbuf_cap = kb > 0 ? (size_t)kb * 1024 : 0;
    const char* co = getenv("GRPC_RDMA_HIP_COALESCE");
    uint64_t lim[2] = {0, 0};
    if (sbuf_cap != 0 && !(co && atoi(co) == 0) && grdma_endpoint_write_queue_limits(p, lim) == 0) {
      co_slices = (size_t)lim[0] < kWriteWindow ? (size_t)lim[0] : kWriteWindow;
      co_bytes = (size_t)lim[1] < sbuf_cap ? (size_t)lim[1] : sbuf_cap;
    }
    if (const char* ra = getenv("GRPC_RDMA_HIP_READ_AHEAD")) {
      const long v = atol(ra);
      if (v >= 1 && (uint64_t)v <= kReadAheadMax) read_ahead = (uint64_t)v;
    }
  }
  // rdma_free: nothing of the transport's may be referenced afterwards
  void release() {
    if (ahead_win) grdma_window_unref(ahead_win);
    ahead_win = nullptr;
    ahead.clear();
    if (pair != nullptr && (sbuf[0].mem || sbuf[1].mem)) grdma_endpoint_write_quiesce(pair);  // (a chain may still gather from them)
    for (send_buffer& b : sbuf) {
      if (b.mem) grdma_host_free_pinned(b.mem);
      b.mem = nullptr;
    }
  }
  // rdma_destroy without a shutdown in front of it: buffered writes still on their way are dropped (what closing a
  // socket with unsent bytes does), and the reference their Send holds on the endpoint goes back
  void abandon_buffered_writes() {
    std::lock_guard<std::recursive_mutex> lk(wmu);
    if (bg_running < 0) return;
    forget_write();
    bg_running = bg_waiting = -1;
    T::unref(h);
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
        ahead.resize(kReadAheadMax);
        const int64_t n = grdma_endpoint_read_test(pair, ahead.data(), kReadAheadMax, &would_block, &win);
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
        open_read_noted = would_block != 0;  // (a drain that stopped at max_reads left no read open)
        if (n > 0) {
          ahead_win = win;
          ahead_copy = !any_window_free();
          // more has arrived meanwhile?  Then the next drain runs on the device while this one's slices are handed
          // to the transport (it needs a window of its own; none free = it waits until this one has been released)
          if (!ahead_copy && read_ahead > 1 && grdma_pair_has_message(pair) > 0) grdma_endpoint_read_submit(pair, read_ahead);
          continue;
        }
        grdma_window_unref(win);
        return read_would_block();  // the drain found no complete record
      }
      // nothing pending.  A record behind head_ (or a half-read one)?  That is a host load.
      if (grdma_pair_has_message(pair) <= 0) {
        // rdma_continue_read has allocated its 256-byte slice by now and keeps it across the would-block (:283-287):
        // the pair learns of it unless its last drain ended that way already
        if (!open_read_noted && grdma_endpoint_read_idle(pair) == 0) open_read_noted = true;
        return read_would_block();
      }
      const int rc = grdma_endpoint_read_submit(pair, read_ahead);
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
    bg_queued = false;
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
      last_failure = what;
      *error = T::annotate(h, what.c_str());
      forget_write();
      if (outgoing_buffer) T::reset_and_unref(outgoing_buffer);
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
        if (grdma_endpoint_write_begin(pair, out_views.data() + out_next, cnt, out_flags) < 0)
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
    if (outgoing_buffer) T::reset_and_unref(outgoing_buffer);  // :519-523
    return true;
  }

  // ---- send buffers
  int free_send_buffer() {
    for (int i = 0; i < 2; i++)
      if (i != bg_running && i != bg_waiting) return i;
    return -1;
  }
  // hand send buffer s to the pair; false = on its way (the writable edge continues it), true = finished (*error)
  bool start_buffered(int s, error_t* error) {
    bg_running = s;
    outgoing_buffer = nullptr;
    out_views = sbuf[s].views;
    out_next = 0;
    out_flags = 0;
    window_active = false;
    send_submitted = false;
    T::ref(h);
    if (bg_queued) {
      bg_queued = false;
      if (grdma_endpoint_write_adopt(pair) == 1) {  // its Sends went into the stream behind the buffer in front
        out_next = out_views.size();
        window_active = true;
        send_submitted = true;
      }
    }
    return flush(error);
  }
  // copies buf's slices into send buffer s (boundaries kept: one ring record each), behind what it holds when
  // `append`; false = no pinned memory
  bool fill_send_buffer(int s, sb_t* buf, bool append = false) {
    send_buffer& b = sbuf[s];
    if (b.mem == nullptr) b.mem = static_cast<uint8_t*>(grdma_host_alloc_pinned(sbuf_cap));
    if (b.mem == nullptr) return false;
    const size_t n = T::count(buf);
    const size_t v0 = append ? b.views.size() : 0;
    b.views.resize(v0 + n);
    size_t off = append ? b.used : 0;
    for (size_t i = 0; i < n; i++) {
      const size_t l = T::slice_len(buf, i);
      if (l) memcpy(b.mem + off, T::slice_ptr(buf, i), l);
      b.views[v0 + i] = grdma_slice{b.mem + off, (uint64_t)l};
      off += l;
    }
    b.used = off;
    return true;
  }
  // would the buffer that waits take `len` more bytes in `count` more slices?  (Only while its chain has not gone
  // into the send stream; a buffer is never grown past what ONE queued write may hold.)
  bool waiting_takes(size_t len, size_t count) const {
    if (co_bytes == 0 || bg_waiting < 0 || bg_queued) return false;
    const send_buffer& b = sbuf[bg_waiting];
    return b.used + len <= co_bytes && b.views.size() + count <= co_slices;
  }
  // The buffer that waits has just taken a write of `len` bytes in `count` slices: its chain goes into the send stream
  // now unless another write like this one would still fit (coalescing); without coalescing, at once.
  void queue_waiting_if_full(size_t len, size_t count) {
    if (!waiting_takes(len, count)) try_queue_waiting();
  }
  bool fits_send_buffer(sb_t* buf) {
    const size_t len = T::length(buf);
    return sbuf_cap != 0 && len >= kSendBufferMin && len <= sbuf_cap && T::count(buf) <= kWriteWindow;
  }
  // A write that found both buffers taken waits as an ordinary one (deferred).  When a buffer comes free while the
  // other one is on its way, it moves into the free one after all: copied, completed, queued behind the Sends in flight.
  void buffer_the_deferred_write() {
    if (!deferred || bg_failed || bg_running < 0 || bg_waiting >= 0 || !fits_send_buffer(deferred_buf)) return;
    const int s = free_send_buffer();
    const size_t dlen = T::length(deferred_buf), dcount = T::count(deferred_buf);
    if (s < 0 || !fill_send_buffer(s, deferred_buf)) return;
    T::reset_and_unref(deferred_buf);
    deferred_buf = nullptr;
    deferred = false;
    closure_t* cb = write_cb;
    write_cb = nullptr;
    bg_waiting = s;
    queue_waiting_if_full(dlen, dcount);
    T::run(h, cb, T::none());
    T::unref(h);  // (the reference the deferred write took in write())
  }
  // the buffer that waits, into the pair's send stream right away when the pair can take it behind the Send in flight
  void try_queue_waiting() {
    if (bg_waiting < 0 || bg_queued || bg_running < 0 || !send_submitted || out_next < out_views.size()) return;
    const send_buffer& b = sbuf[bg_waiting];
    if (grdma_endpoint_write_queue(pair, b.views.data(), b.views.size()) == 0) bg_queued = true;
  }
  void start_ordinary() {  // the write held in outgoing_buffer / write_cb
    const size_t n = T::count(outgoing_buffer);
    out_views.resize(n);
    for (size_t i = 0; i < n; i++)
      out_views[i] = grdma_slice{T::slice_ptr(outgoing_buffer, i), (uint64_t)T::slice_len(outgoing_buffer, i)};
    out_next = 0;
    out_flags = GRDMA_MEM_HOST;
    window_active = false;
    send_submitted = false;
  }
  // the Send(s) of whatever the pair was working on have ended with `err`: account for it and start what waits
  void after_flush(error_t err) {
    for (;;) {
      if (bg_running >= 0) {
        bg_running = -1;
        if (T::is_error(err)) {  // a write that has completed long ago failed: remembered for the next one
          bg_failed = true;
          bg_error = last_failure.empty() ? std::string("an earlier write failed") : last_failure;
          bg_waiting = -1;
          bg_queued = false;
          T::drop(err);
        }
        const int next = bg_waiting;
        bg_waiting = -1;
        if (next >= 0 && !bg_failed) {
          error_t e2;
          const bool fin = start_buffered(next, &e2);
          T::unref(h);  // (the reference of the buffer that has just finished; the next one took its own)
          if (!fin) {
            T::notify_on_write(h);
            buffer_the_deferred_write();
            return;
          }
          err = e2;
          continue;
        }
        if (deferred) {
          deferred = false;
          if (bg_failed) {
            closure_t* cb = write_cb;
            write_cb = nullptr;
            T::reset_and_unref(deferred_buf);
            deferred_buf = nullptr;
            T::run(h, cb, T::annotate(h, bg_error.c_str()));
            T::unref(h);  // (the deferred write's)
            T::unref(h);  // (the finished buffer's)
            return;
          }
          outgoing_buffer = deferred_buf;
          deferred_buf = nullptr;
          start_ordinary();
          error_t e2;
          const bool fin = flush(&e2);
          T::unref(h);  // (the finished buffer's; the deferred write holds its own since write())
          if (!fin) {
            T::notify_on_write(h);
            return;
          }
          err = e2;
          continue;  // (bg_running < 0 now: the ordinary completion below)
        }
        T::unref(h);
        return;
      }
      closure_t* cb = write_cb;
      write_cb = nullptr;
      T::run(h, cb, err);
      T::unref(h);
      return;
    }
  }

  // (the mutex lives in this object: a reference of its own around everything that runs under it, so that the last
  // unref inside cannot take the object away from under the lock)
  void handle_write(error_t error) {  // :527-557
    T::ref(h);
    {
      std::lock_guard<std::recursive_mutex> lk(wmu);
      handle_write_locked(error);
    }
    T::unref(h);
  }
  void write(sb_t* buf, closure_t* cb) {  // :559-586
    T::ref(h);
    {
      std::lock_guard<std::recursive_mutex> lk(wmu);
      write_locked(buf, cb);
    }
    T::unref(h);
  }

  void handle_write_locked(error_t error) {
    typename T::scope profiler(T::OP_HANDLE_WRITE);
    if (T::is_error(error)) {
      forget_write();
      if (bg_running >= 0) {  // buffered writes die with the endpoint (their callbacks have run)
        bg_running = bg_waiting = -1;
        bg_queued = false;
        bg_failed = true;
        bg_error = "Endpoint shutdown";
        T::unref(h);
      }
      if (write_cb != nullptr) {
        closure_t* cb = write_cb;
        write_cb = nullptr;
        deferred = false;
        deferred_buf = nullptr;
        T::run(h, cb, T::ref(error));
        T::unref(h);
      }
      return;
    }
    if (bg_running < 0 && write_cb == nullptr) return;  // (an edge nobody waits for any more)
    error_t err;
    if (!flush(&err)) {
      T::notify_on_write(h);
    } else {
      after_flush(err);
    }
  }

  void write_locked(sb_t* buf, closure_t* cb) {
    typename T::scope profiler(T::OP_WRITE);
    if (write_cb != nullptr) abort();  // GPR_ASSERT(rdma->write_cb == nullptr)
    if (T::length(buf) == 0) {
      T::run(h, cb, T::is_shutdown(h) ? T::annotate(h, "EOF") : T::none());
      return;
    }
    if (bg_failed) {  // what a socket reports on the write after the one that failed
      T::reset_and_unref(buf);
      T::run(h, cb, T::annotate(h, bg_error.c_str()));
      return;
    }
    const size_t wlen = T::length(buf), wcount = T::count(buf);
    if (!deferred && !T::is_shutdown(h) && waiting_takes(wlen, wcount) && fill_send_buffer(bg_waiting, buf, true)) {
      // behind the bytes that wait for the Send in flight to finish: one chain for all of them
      T::reset_and_unref(buf);
      queue_waiting_if_full(wlen, wcount);
      T::run(h, cb, T::none());
      return;
    }
    const int s = free_send_buffer();
    if (fits_send_buffer(buf) && s >= 0 && !deferred && !T::is_shutdown(h) && fill_send_buffer(s, buf)) {
      // the slices keep their boundaries, their bytes are the endpoint's now
      T::reset_and_unref(buf);
      if (bg_running >= 0) {
        bg_waiting = s;  // behind the buffer that is on its way; handle_write starts it
        queue_waiting_if_full(wlen, wcount);
      } else {
        error_t err;
        if (!start_buffered(s, &err)) T::notify_on_write(h);
        else after_flush(err);
      }
      T::run(h, cb, T::none());
      return;
    }
    if (bg_running >= 0) {  // behind the buffered writes
      T::ref(h);
      write_cb = cb;
      deferred = true;
      deferred_buf = buf;
      return;
    }
    outgoing_buffer = buf;
    start_ordinary();
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
