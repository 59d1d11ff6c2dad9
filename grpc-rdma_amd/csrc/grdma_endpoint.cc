// grpc_endpoint implementation for RDMA_BP / RDMA_BPEV on top of the C ABI.
// Mirrors src/core/lib/iomgr/rdma_bp_posix.cc function by function; the byte work
// happens in the HIP kernels behind grdma_endpoint_write_* / grdma_endpoint_read.
#include "../../include/grdma_endpoint.hpp"
#include "../../include/grdma_profiler.hpp"

#include <sys/epoll.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace grdma_core {

// ------------------------------------------------------------------------- slices
namespace {
struct malloc_refcount {
  grpc_slice_refcount base;
  // payload follows
};
void malloc_destroy(grpc_slice_refcount* r) { free(r); }
}  // namespace

grpc_slice grpc_slice_malloc(size_t length) {
  grpc_slice s;
  if (length <= GRDMA_SLICE_INLINED_SIZE) {  // small slices are inlined (slice.h:47-48)
    s.refcount = nullptr;
    s.data.inlined.length = static_cast<uint8_t>(length);
  } else {
    auto* rc = static_cast<malloc_refcount*>(malloc(sizeof(malloc_refcount) + length));
    rc->base.refs.store(1);
    rc->base.destroy = malloc_destroy;
    s.refcount = &rc->base;
    s.data.refcounted.bytes = reinterpret_cast<uint8_t*>(rc + 1);
    s.data.refcounted.length = length;
  }
  return s;
}

grpc_slice grpc_slice_from_copied_buffer(const char* src, size_t len) {
  grpc_slice s = grpc_slice_malloc(len);
  if (len) memcpy(GRPC_SLICE_START_PTR(s), src, len);
  return s;
}

void grpc_slice_unref(grpc_slice s) {
  if (s.refcount && s.refcount->refs.fetch_sub(1) == 1) s.refcount->destroy(s.refcount);
}

void grpc_slice_buffer_init(grpc_slice_buffer* sb) {
  sb->count = 0;
  sb->length = 0;
  sb->capacity = GRDMA_SLICE_BUFFER_INLINE_ELEMENTS;
  sb->base_slices = sb->slices = sb->inlined;
}

void grpc_slice_buffer_destroy(grpc_slice_buffer* sb) {
  grpc_slice_buffer_reset_and_unref(sb);
  if (sb->base_slices != sb->inlined) free(sb->base_slices);
  sb->base_slices = sb->slices = sb->inlined;
  sb->capacity = GRDMA_SLICE_BUFFER_INLINE_ELEMENTS;
}

static void maybe_embiggen(grpc_slice_buffer* sb) {
  if (sb->count < sb->capacity) return;
  size_t ncap = sb->capacity * 3 / 2 + 8;
  auto* n = static_cast<grpc_slice*>(malloc(ncap * sizeof(grpc_slice)));
  memcpy(n, sb->slices, sb->count * sizeof(grpc_slice));
  if (sb->base_slices != sb->inlined) free(sb->base_slices);
  sb->base_slices = sb->slices = n;
  sb->capacity = ncap;
}

size_t grpc_slice_buffer_add_indexed(grpc_slice_buffer* sb, grpc_slice s) {
  size_t out = sb->count;
  maybe_embiggen(sb);
  sb->slices[out] = s;
  sb->length += GRPC_SLICE_LENGTH(s);
  sb->count = out + 1;
  return out;
}

void grpc_slice_buffer_add(grpc_slice_buffer* sb, grpc_slice s) {
  size_t n = sb->count;
  // two consecutive inlined slices are concatenated (slice_buffer.cc:136-171)
  if (!s.refcount && n) {
    grpc_slice* back = &sb->slices[n - 1];
    if (!back->refcount && back->data.inlined.length < GRDMA_SLICE_INLINED_SIZE) {
      if (s.data.inlined.length + back->data.inlined.length <= GRDMA_SLICE_INLINED_SIZE) {
        memcpy(back->data.inlined.bytes + back->data.inlined.length, s.data.inlined.bytes,
               s.data.inlined.length);
        back->data.inlined.length =
            static_cast<uint8_t>(back->data.inlined.length + s.data.inlined.length);
      } else {
        size_t cp1 = GRDMA_SLICE_INLINED_SIZE - back->data.inlined.length;
        memcpy(back->data.inlined.bytes + back->data.inlined.length, s.data.inlined.bytes, cp1);
        back->data.inlined.length = GRDMA_SLICE_INLINED_SIZE;
        maybe_embiggen(sb);
        back = &sb->slices[n];
        sb->count = n + 1;
        back->refcount = nullptr;
        back->data.inlined.length = static_cast<uint8_t>(s.data.inlined.length - cp1);
        memcpy(back->data.inlined.bytes, s.data.inlined.bytes + cp1, s.data.inlined.length - cp1);
      }
      sb->length += s.data.inlined.length;
      return;
    }
  }
  grpc_slice_buffer_add_indexed(sb, s);
}

void grpc_slice_buffer_reset_and_unref(grpc_slice_buffer* sb) {
  for (size_t i = 0; i < sb->count; i++) grpc_slice_unref(sb->slices[i]);
  sb->count = 0;
  sb->length = 0;
  sb->slices = sb->base_slices;
}

void grpc_slice_buffer_swap(grpc_slice_buffer* a, grpc_slice_buffer* b) {
  // both may use their inlined storage: move through temporaries
  std::vector<grpc_slice> sa(a->slices, a->slices + a->count), sb_(b->slices, b->slices + b->count);
  size_t la = a->length, lb = b->length;
  a->count = 0; a->length = 0; a->slices = a->base_slices;
  b->count = 0; b->length = 0; b->slices = b->base_slices;
  for (auto& s : sb_) grpc_slice_buffer_add_indexed(a, s);
  for (auto& s : sa) grpc_slice_buffer_add_indexed(b, s);
  (void)la; (void)lb;
}

// ------------------------------------------------------------------------- errors
grpc_error_handle GRPC_ERROR_CREATE_FROM_STATIC_STRING(const char* desc) {
  auto* e = new grpc_error();
  e->description = desc;
  e->fd = -1;
  e->grpc_status = 0;
  e->refs.store(1);
  return e;
}
grpc_error_handle GRPC_ERROR_REF(grpc_error_handle e) {
  if (e) e->refs.fetch_add(1);
  return e;
}
void GRPC_ERROR_UNREF(grpc_error_handle e) {
  if (e && e->refs.fetch_sub(1) == 1) delete e;
}

// ----------------------------------------------------------------------- endpoint
namespace {

struct grpc_rdma;
}  // namespace

// `pollable` of the RDMA event engines (ev_epollex_rdma_bpev_linux.cc:208-246): the fds of the
// set, the epoll fd their wakeup fds are registered with, the busy-polling budget
struct grpc_pollset {
  bool bpev;
  int polling_timeout_us;
  int epfd;
  std::vector<grpc_rdma*> rdma_fds;
  std::vector<grdma_pair*> pairs;      // scratch of one pass
  std::vector<uint64_t> readable;
  std::vector<uint8_t> has_message;
  grdma_pollset_stats stats;
};

namespace {
struct grpc_rdma {  // rdma_bp_posix.cc:45-88
  grpc_endpoint base;  // must be first (endpoint.h:112-114)
  int fd;
  bool is_first_read;
  std::atomic<int> refcount;
  bool shutdown;
  grpc_error_handle shutdown_error;
  grdma_pair* pair;
  bool enable_poller;
  grpc_slice_buffer* incoming_buffer;
  int inq;
  grpc_slice_buffer* outgoing_buffer;
  std::vector<grdma_slice> out_views;  // {ptr,len} of outgoing_buffer, windowed to the ABI cap
  size_t out_next;                     // first slice not yet handed to the pair
  bool window_active;                  // the pair holds a window that has not gone out whole yet
  grpc_closure* read_cb;
  grpc_closure* write_cb;
  bool read_armed;   // notify_on_read pending
  bool write_armed;  // notify_on_write pending
  std::string peer_string;
  std::string local_address;
  grpc_pollset* pollset;  // the set this endpoint was added to (grpc_pollset_add_fd)
  // Read-ahead: ONE device pass performs many endpoint reads (grdma_endpoint_read, max_reads);
  // their slices wait here, copied to the host in one transfer, and the following
  // grpc_endpoint_read calls are served without touching the device.  Every completion handed
  // out this way filled its slice, so what it holds cannot depend on records that arrive later --
  // the chain of reads stops at the first one that would block (rdma_do_read :180-291).
  std::vector<grdma_read_slice> ahead;
  size_t ahead_next;
  uint8_t* ahead_bytes;  // pinned host memory (one device-to-host transfer per pass)
  uint64_t ahead_cap;
  uint64_t ahead_base;   // arena offset of ahead_bytes[0]
};

grdma_poller* g_poller = nullptr;  // Poller::Get(): one per process, created with the first BPEV endpoint

const size_t kWindow = 4000;   // slices handed to one grdma_endpoint_write_begin (ABI cap 4095)
const size_t kReadAhead = 1024;  // endpoint reads performed per device pass

void run_closure(grpc_closure* c, grpc_error_handle err) {  // grpc_core::Closure::Run
  c->cb(c->cb_arg, err);
  GRPC_ERROR_UNREF(err);
}

grpc_error_handle rdma_annotate_error(grpc_error_handle src, grpc_rdma* rdma) {  // :86-96
  src->fd = rdma->fd;
  src->grpc_status = GRPC_STATUS_UNAVAILABLE;  // "so that application may choose to retry"
  src->target_address = rdma->peer_string;
  return src;
}

void rdma_unref(grpc_rdma* rdma);
void pollset_del_fd(grpc_pollset* ps, grpc_rdma* rdma);

void call_read_cb(grpc_rdma* rdma, grpc_error_handle error) {  // :161-176
  grpc_closure* cb = rdma->read_cb;
  rdma->read_cb = nullptr;
  rdma->incoming_buffer = nullptr;
  run_closure(cb, error);
}

// rdma_continue_read + rdma_do_read (:306-326, :180-291): the device performs the read
// (slice sizing, Recv loop, credit return); here the slice is materialised.
void rdma_handle_read(grpc_rdma* rdma, grpc_error_handle error) {
  grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_HANDLE_READ);  // :330
  if (error != GRPC_ERROR_NONE) {  // :333-338
    grpc_slice_buffer_reset_and_unref(rdma->incoming_buffer);
    call_read_cb(rdma, GRPC_ERROR_REF(error));
    rdma_unref(rdma);
    return;
  }
  // rdma_continue_read (:307) sizes the slice and calls rdma_do_read (:181); here both are the
  // device pass below, recorded under the two names the reference uses
  grdma_profiler cont(GRDMA_STATS_TIME_TRANSPORT_CONTINUE_READ);
  grdma_profiler do_read(GRDMA_STATS_TIME_TRANSPORT_DO_READ);
  int would_block = 0;
  int64_t n = 0;
  if (rdma->ahead_next >= rdma->ahead.size()) {
    rdma->ahead.resize(kReadAhead);
    rdma->ahead_next = 0;
    n = grdma_endpoint_read(rdma->pair, kReadAhead, rdma->ahead.data(), kReadAhead, &would_block);
    rdma->ahead.resize(n > 0 ? (size_t)n : 0);
    if (n > 0) {
      uint64_t lo = ~0ull, hi = 0;
      for (const grdma_read_slice& a : rdma->ahead) {
        if (a.off < lo) lo = a.off;
        if (a.off + a.len > hi) hi = a.off + a.len;
      }
      if (hi - lo > rdma->ahead_cap) {
        grdma_host_free_pinned(rdma->ahead_bytes);
        rdma->ahead_cap = (hi - lo) * 2;
        rdma->ahead_bytes = static_cast<uint8_t*>(grdma_host_alloc_pinned(rdma->ahead_cap));
        if (rdma->ahead_bytes == nullptr) rdma->ahead_cap = 0;
      }
      rdma->ahead_base = lo;
      if (rdma->ahead_bytes == nullptr ||
          grdma_pair_arena_copy_out(rdma->pair, lo, rdma->ahead_bytes, hi - lo) != 0) {
        rdma->ahead.clear();
        n = -1;
      }
    }
  }
  if (rdma->ahead_next < rdma->ahead.size()) {
    const grdma_read_slice s = rdma->ahead[rdma->ahead_next++];
    grpc_slice out = grpc_slice_malloc(s.len);
    memcpy(GRPC_SLICE_START_PTR(out), rdma->ahead_bytes + (s.off - rdma->ahead_base), s.len);
    grpc_slice_buffer_add_indexed(rdma->incoming_buffer, out);
    rdma->inq = 1;
    call_read_cb(rdma, GRPC_ERROR_NONE);
    rdma_unref(rdma);
    return;
  }
  if (n < 0) {
    call_read_cb(rdma, rdma_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING(grdma_last_error()), rdma));
    rdma_unref(rdma);
    return;
  }
  rdma->inq = 0;
  const int status = grdma_pair_get_status(rdma->pair);
  if (status == 3 /* kHalfClosed */) {  // :220-228
    grpc_slice_buffer_reset_and_unref(rdma->incoming_buffer);
    call_read_cb(rdma, rdma_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("Pair closed"), rdma));
    rdma_unref(rdma);
  } else if (status == 5 /* kError */) {  // :229-238
    grpc_slice_buffer_reset_and_unref(rdma->incoming_buffer);
    call_read_cb(rdma, rdma_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("Pair error"), rdma));
    rdma_unref(rdma);
  } else {
    rdma->read_armed = true;  // "We've consumed the edge, request a new one" :241-243
  }
}

void rdma_read(grpc_endpoint* ep, grpc_slice_buffer* incoming_buffer, grpc_closure* cb, bool urgent) {
  grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_READ);  // :345
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (rdma->read_cb != nullptr) abort();  // GPR_ASSERT(rdma->read_cb == nullptr) :347
  rdma->read_cb = cb;
  rdma->incoming_buffer = incoming_buffer;
  grpc_slice_buffer_reset_and_unref(incoming_buffer);
  rdma->refcount.fetch_add(1);  // RDMA_REF(rdma, "read")
  if (rdma->shutdown) {  // a shut-down fd runs the closure with the shutdown error
    rdma_handle_read(rdma, rdma->shutdown_error);
  } else if (rdma->is_first_read) {  // :353-358
    rdma->is_first_read = false;
    rdma->read_armed = true;
  } else if (!urgent && rdma->inq == 0) {  // :359-363
    rdma->read_armed = true;
  } else {  // :364-375
    rdma_handle_read(rdma, GRPC_ERROR_NONE);
  }
}

// rdma_flush (:470-524): ONE Send from the cursor, then the status switch of :499-518.
// The pair's write context holds one window of the buffer at a time (the ABI takes at most
// 4095 slices); a window that went out whole is followed by the next one right away, as the
// single Send of the reference would have carried on into those slices.
// Returns true when the whole buffer has been written or an error is set.
bool rdma_flush(grpc_rdma* rdma, grpc_error_handle* error) {
  grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_FLUSH);  // :471
  *error = GRPC_ERROR_NONE;
  auto fail_with = [&](const char* what) {
    *error = rdma_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING(what), rdma);
    grdma_endpoint_write_abort(rdma->pair);  // the pair must not keep views of slices about to be unreffed
    rdma->window_active = false;
    rdma->out_views.clear();
    rdma->out_next = 0;
    grpc_slice_buffer_reset_and_unref(rdma->outgoing_buffer);
    return true;
  };
  for (;;) {
    if (!rdma->window_active) {
      if (rdma->out_next >= rdma->out_views.size()) break;
      size_t cnt = rdma->out_views.size() - rdma->out_next;
      if (cnt > kWindow) cnt = kWindow;
      if (grdma_endpoint_write_begin(rdma->pair, rdma->out_views.data() + rdma->out_next, cnt,
                                     GRDMA_MEM_HOST) < 0)
        return fail_with("RDMA Pair has an internal error");
      rdma->out_next += cnt;
      rdma->window_active = true;
    }
    int done = 0;
    const int64_t n = grdma_endpoint_write_step(rdma->pair, &done);
    if (n < 0) return fail_with("RDMA Pair has an internal error");  // :511-517
    if (done) {
      rdma->window_active = false;
      continue;
    }
    // partial send, :499-518
    const int status = grdma_pair_get_status(rdma->pair);
    if (status == 2 /* kConnected */) return false;  // wait for the writable edge
    if (status == 3 /* kHalfClosed */) return fail_with("Peer has been exited");
    return fail_with("RDMA Pair has an internal error");
  }
  rdma->out_views.clear();
  rdma->out_next = 0;
  grpc_slice_buffer_reset_and_unref(rdma->outgoing_buffer);  // :519-523
  return true;
}

void rdma_handle_write(grpc_rdma* rdma, grpc_error_handle error) {  // :527-557
  grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_HANDLE_WRITE);  // :529
  if (error != GRPC_ERROR_NONE) {
    grpc_closure* cb = rdma->write_cb;
    rdma->write_cb = nullptr;
    run_closure(cb, GRPC_ERROR_REF(error));
    rdma_unref(rdma);
    return;
  }
  grpc_error_handle err;
  if (!rdma_flush(rdma, &err)) {
    rdma->write_armed = true;  // notify_on_write
  } else {
    grpc_closure* cb = rdma->write_cb;
    rdma->write_cb = nullptr;
    run_closure(cb, err);
    rdma_unref(rdma);
  }
}

void rdma_write(grpc_endpoint* ep, grpc_slice_buffer* buf, grpc_closure* cb, void* /*arg*/) {
  grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_WRITE);  // :561
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (rdma->write_cb != nullptr) abort();  // GPR_ASSERT :563
  if (buf->length == 0) {  // :565-574
    run_closure(cb, rdma->shutdown
                        ? rdma_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("EOF"), rdma)
                        : GRPC_ERROR_NONE);
    return;
  }
  if (rdma->shutdown) {
    run_closure(cb, GRPC_ERROR_REF(rdma->shutdown_error));
    return;
  }
  rdma->outgoing_buffer = buf;
  rdma->out_views.clear();
  for (size_t i = 0; i < buf->count; i++)
    rdma->out_views.push_back({GRPC_SLICE_START_PTR(buf->slices[i]), GRPC_SLICE_LENGTH(buf->slices[i])});
  rdma->out_next = 0;
  rdma->window_active = false;
  grpc_error_handle error;
  if (!rdma_flush(rdma, &error)) {  // :577-583
    rdma->refcount.fetch_add(1);
    rdma->write_cb = cb;
    rdma->write_armed = true;
  } else {
    run_closure(cb, error);
  }
}

void rdma_shutdown(grpc_endpoint* ep, grpc_error_handle why) {  // :106-110
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (!rdma->shutdown) {
    rdma->shutdown = true;
    rdma->shutdown_error = why ? why : GRPC_ERROR_CREATE_FROM_STATIC_STRING("Endpoint shutdown");
    // grpc_fd_shutdown: pending notify_on_* closures run with the error
    if (rdma->read_armed) {
      rdma->read_armed = false;
      rdma_handle_read(rdma, rdma->shutdown_error);
    }
    if (rdma->write_armed) {
      rdma->write_armed = false;
      rdma_handle_write(rdma, rdma->shutdown_error);
    }
  } else {
    GRPC_ERROR_UNREF(why);
  }
}

void rdma_free(grpc_rdma* rdma) {  // :112-132
  if (rdma->pollset != nullptr) pollset_del_fd(rdma->pollset, rdma);  // grpc_fd_orphan
  if (rdma->pair != nullptr) {
    if (rdma->enable_poller && g_poller != nullptr) grdma_poller_remove(g_poller, rdma->pair);
    grdma_pair_disconnect(rdma->pair);
    grdma_pair_destroy(rdma->pair);  // PairPool::Putback in the reference
    rdma->pair = nullptr;
  }
  GRPC_ERROR_UNREF(rdma->shutdown_error);
  grdma_host_free_pinned(rdma->ahead_bytes);
  delete rdma;
}

void rdma_unref(grpc_rdma* rdma) {
  if (rdma->refcount.fetch_sub(1) == 1) rdma_free(rdma);
}

void rdma_destroy(grpc_endpoint* ep) { rdma_unref(reinterpret_cast<grpc_rdma*>(ep)); }  // :134-139
void pollset_del_fd(grpc_pollset* ps, grpc_rdma* rdma) {
  auto it = std::find(ps->rdma_fds.begin(), ps->rdma_fds.end(), rdma);
  if (it == ps->rdma_fds.end()) return;
  ps->rdma_fds.erase(it);
  if (ps->bpev && rdma->pair != nullptr) {
    const int wfd = grdma_pair_get_wakeup_fd(rdma->pair);
    if (wfd >= 0) epoll_ctl(ps->epfd, EPOLL_CTL_DEL, wfd, nullptr);
  }
}

// grpc_pollset_add_fd -> pollable_add_fd (ev_epollex_rdma_bpev_linux.cc:705-745): the fd joins
// p->rdma_fds; in BPEV mode the pair's wakeup fd joins the epoll set, its data pointer tagged
// with bit 1 so that process_events can tell it from a socket (:725-741)
void rdma_add_to_pollset(grpc_endpoint* ep, grpc_pollset* ps) {
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (ps == nullptr || rdma->pollset == ps) return;
  if (rdma->pollset != nullptr) pollset_del_fd(rdma->pollset, rdma);
  rdma->pollset = ps;
  ps->rdma_fds.push_back(rdma);
  if (ps->bpev) {
    const int wfd = grdma_pair_get_wakeup_fd(rdma->pair);
    if (wfd >= 0) {
      struct epoll_event ev;
      memset(&ev, 0, sizeof ev);
      ev.events = EPOLLIN | EPOLLET;
      ev.data.ptr = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(rdma) | 2);
      epoll_ctl(ps->epfd, EPOLL_CTL_ADD, wfd, &ev);
    }
  }
}
void rdma_add_to_pollset_set(grpc_endpoint*, grpc_pollset_set*) {}
void rdma_delete_from_pollset_set(grpc_endpoint*, grpc_pollset_set*) {}
grpc_resource_user* rdma_get_resource_user(grpc_endpoint*) { return nullptr; }
const char* rdma_get_peer(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma*>(ep)->peer_string.c_str(); }
const char* rdma_get_local_address(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma*>(ep)->local_address.c_str(); }
int rdma_get_fd(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma*>(ep)->fd; }
bool rdma_can_track_err(grpc_endpoint*) { return false; }

const grpc_endpoint_vtable vtable = {rdma_read,            rdma_write,
                                     rdma_add_to_pollset,  rdma_add_to_pollset_set,
                                     rdma_delete_from_pollset_set, rdma_shutdown,
                                     rdma_destroy,         rdma_get_resource_user,
                                     rdma_get_peer,        rdma_get_local_address,
                                     rdma_get_fd,          rdma_can_track_err};  // :694-705
}  // namespace

grpc_endpoint* grpc_rdma_bp_create(int fd, const char* peer_string, bool enable_poller) {
  grdma_config cfg;
  if (grdma_config_from_env(&cfg) < 0) return nullptr;
  if (grdma_init(cfg.hip_device) < 0) return nullptr;
  grdma_pair* pair = grdma_pair_create(static_cast<uint64_t>(cfg.ring_buffer_size_kb) * 1024,
                                       cfg.max_sge, GRDMA_WIRE_STAGED);
  if (pair == nullptr) return nullptr;  // "Connection failed" path :777-784
  grpc_rdma* rdma = new grpc_rdma();
  rdma->base.vtable = &vtable;
  rdma->fd = fd;
  rdma->peer_string = peer_string ? peer_string : "";
  rdma->local_address = "";
  rdma->is_first_read = true;
  rdma->refcount.store(1);
  rdma->shutdown = false;
  rdma->shutdown_error = nullptr;
  rdma->pair = pair;
  rdma->enable_poller = enable_poller;
  rdma->incoming_buffer = nullptr;
  rdma->outgoing_buffer = nullptr;
  rdma->out_next = 0;
  rdma->window_active = false;
  rdma->read_cb = rdma->write_cb = nullptr;
  rdma->read_armed = rdma->write_armed = false;
  rdma->inq = 1;  // :745
  rdma->pollset = nullptr;
  if (enable_poller) {  // RDMA_BPEV: Poller::Get().AddPollable(pair), :789-791
    if (g_poller == nullptr) g_poller = grdma_poller_create(cfg.poller_thread_num, cfg.poller_sleep_timeout_ms);
    if (g_poller != nullptr) grdma_poller_add(g_poller, pair);
  }
  return &rdma->base;
}

bool grpc_rdma_bp_connect_loopback(grpc_endpoint* a, grpc_endpoint* b) {
  return grdma_pair_connect(reinterpret_cast<grpc_rdma*>(a)->pair,
                            reinterpret_cast<grpc_rdma*>(b)->pair) == 0;
}

grpc_endpoint* grpc_endpoint_create(int fd, const char* peer_string, bool /*server*/) {
  switch (grdma_determine_platform()) {  // endpoint.cc:36-52
    case GRDMA_IOMGR_RDMA_BP:
      return grpc_rdma_bp_create(fd, peer_string, false);
    case GRDMA_IOMGR_RDMA_BPEV:
      return grpc_rdma_bp_create(fd, peer_string, true);
    default:
      return nullptr;  // TCP / RDMA_EVENT are not this library's path
  }
}

grdma_pair* grdma_endpoint_pair(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma*>(ep)->pair; }

int grdma_endpoint_poll(grpc_endpoint* ep) {
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  int ran = 0;
  // pollable_epoll busy-poll body, ev_epollex_rdma_bpev_linux.cc:1105-1149
  if (rdma->read_armed) {
    // (an armed read that has completed behind the peer's send is visible in host memory: grdma_pair_arm_read)
    const bool armed_ready = grdma_pair_armed_ready(rdma->pair) > 0;
    const int status = armed_ready ? 2 : grdma_pair_get_status(rdma->pair);
    if (armed_ready || grdma_pair_has_message(rdma->pair) > 0 || status == 3 || status == 5) {
      rdma->read_armed = false;  // fd_become_readable
      rdma_handle_read(rdma, GRPC_ERROR_NONE);
      ran++;
    }
  }
  if (rdma->write_armed) {
    // HasPendingWrites(): the last Send came up short; retry once credit is back
    if (grdma_pair_writable_size(rdma->pair) > 0 || grdma_pair_get_status(rdma->pair) != 2) {
      rdma->write_armed = false;  // fd_become_writable
      rdma_handle_write(rdma, GRPC_ERROR_NONE);
      ran++;
    }
  }
  return ran;
}

grpc_pollset* grdma_pollset_create(bool bpev, int busy_polling_timeout_us) {
  grpc_pollset* ps = new grpc_pollset();
  ps->bpev = bpev;
  ps->polling_timeout_us = busy_polling_timeout_us;
  ps->epfd = -1;
  memset(&ps->stats, 0, sizeof ps->stats);
  if (bpev) {
    ps->epfd = epoll_create1(EPOLL_CLOEXEC);
    if (ps->epfd < 0) {
      delete ps;
      return nullptr;
    }
  }
  return ps;
}

void grdma_pollset_destroy(grpc_pollset* ps) {
  if (ps == nullptr) return;
  for (grpc_rdma* r : ps->rdma_fds) r->pollset = nullptr;
  if (ps->epfd >= 0) close(ps->epfd);
  delete ps;
}

size_t grdma_pollset_size(const grpc_pollset* ps) { return ps->rdma_fds.size(); }
void grdma_pollset_get_stats(const grpc_pollset* ps, grdma_pollset_stats* out) { *out = ps->stats; }

// fd_become_readable / fd_become_writable for one endpoint of the set.  `in` / `out`: the
// synthesised EPOLLIN / EPOLLOUT of pollable_epoll (:1105-1149).
static int pollset_deliver(grpc_rdma* rdma, bool in, bool out) {
  int ran = 0;
  rdma->refcount.fetch_add(1);  // a closure may destroy the endpoint
  if (in && rdma->read_armed) {
    rdma->read_armed = false;
    rdma_handle_read(rdma, GRPC_ERROR_NONE);
    ran++;
  }
  if (out && rdma->write_armed) {
    rdma->write_armed = false;
    rdma_handle_write(rdma, GRPC_ERROR_NONE);
    ran++;
  }
  rdma_unref(rdma);
  return ran;
}

int grdma_pollset_work(grpc_pollset* ps, int timeout_ms) {
  grdma_profiler profiler(GRDMA_STATS_TIME_POLLSET_WORK);  // ev_epollex_rdma_bp_linux.cc pollset_work
  using clock = std::chrono::steady_clock;
  ps->stats.passes++;
  const auto t_begin = clock::now();
  // the busy-polling budget: RDMA_BP polls for the whole timeout, RDMA_BPEV for at most
  // polling_timeout_us (:1093-1099)
  int64_t budget_us = ps->bpev ? ps->polling_timeout_us : INT64_MAX;
  if (timeout_ms >= 0) budget_us = std::min<int64_t>(budget_us, (int64_t)timeout_ms * 1000);
  int ran = 0;
  int64_t elapsed_us = 0;
  do {
    const std::vector<grpc_rdma*> fds = ps->rdma_fds;  // closures may add / remove endpoints
    const size_t n = fds.size();
    if (n == 0) break;
    ps->pairs.resize(n);
    ps->readable.resize(n);
    ps->has_message.resize(n);
    bool any_armed = false;
    for (size_t i = 0; i < n; i++) {
      ps->pairs[i] = fds[i]->pair;
      any_armed |= fds[i]->read_armed || fds[i]->write_armed;
    }
    // nothing is armed and passes are serialised by the caller: nothing can become ready in
    // this pass, so an unbounded busy-poll would never return
    if (!any_armed && !ps->bpev && timeout_ms < 0) break;
    if (any_armed) {
      // HasMessage() of every fd of the set: one launch
      if (grdma_poll_pairs(ps->pairs.data(), (uint32_t)n, ps->readable.data(), ps->has_message.data()) < 0) return -1;
      ps->stats.device_polls++;
      for (size_t i = 0; i < n; i++) {
        grpc_rdma* r = fds[i];
        if (std::find(ps->rdma_fds.begin(), ps->rdma_fds.end(), r) == ps->rdma_fds.end()) continue;  // destroyed by a closure
        if (!r->read_armed && !r->write_armed) continue;
        const int status = grdma_pair_get_status(r->pair);
        bool in = false, out = false;
        if (status == GRDMA_PAIR_CONNECTED) {
          in = ps->has_message[i] != 0;
          // HasPendingWrites(): the last Send came up short and there is room again
          out = r->write_armed && grdma_pair_writable_size(r->pair) > 0;
        } else if (status == GRDMA_PAIR_HALF_CLOSED || status == GRDMA_PAIR_ERROR) {
          in = true;  // "Generate an event, so do_read will handle connection close"
          out = r->write_armed;
        }
        if (in || out) ran += pollset_deliver(r, in, out);
      }
    }
    elapsed_us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t_begin).count();
  } while (ran == 0 && elapsed_us < budget_us);
  if (ran == 0 && ps->bpev && !ps->rdma_fds.empty()) {
    // busy-polling timed out: switch to epoll on the wakeup fds (:1155-1170)
    int left_ms = timeout_ms;
    if (timeout_ms > 0) left_ms = (int)std::max<int64_t>(0, timeout_ms - elapsed_us / 1000);
    struct epoll_event evs[100];  // MAX_EPOLL_EVENTS
    int r;
    do {
      r = epoll_wait(ps->epfd, evs, 100, left_ms);
    } while (r < 0 && errno == EINTR);
    ps->stats.epoll_waits++;
    if (r < 0) return -1;
    for (int k = 0; k < r; k++) {
      const uintptr_t tagged = reinterpret_cast<uintptr_t>(evs[k].data.ptr);
      if (!(tagged & 2)) continue;
      grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(tagged & ~(uintptr_t)2);
      if (std::find(ps->rdma_fds.begin(), ps->rdma_fds.end(), rdma) == ps->rdma_fds.end()) continue;
      // pollable_process_events for a wakeup fd (:1010-1037): consume, then look at the pair
      if (grdma_pair_consume_wakeup(rdma->pair) > 0) ps->stats.wakeups_consumed++;
      const int status = grdma_pair_get_status(rdma->pair);
      bool in, out;
      if (status == GRDMA_PAIR_CONNECTED) {
        in = grdma_pair_has_message(rdma->pair) > 0;
        out = rdma->write_armed && grdma_pair_writable_size(rdma->pair) > 0;
      } else {
        in = true;
        out = rdma->write_armed;
      }
      if (in || out) ran += pollset_deliver(rdma, in, out);
    }
  }
  ps->stats.closures_run += (uint64_t)ran;
  return ran;
}

void grpc_endpoint_read(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, bool urgent) {
  ep->vtable->read(ep, slices, cb, urgent);
}
void grpc_endpoint_write(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, void* arg) {
  ep->vtable->write(ep, slices, cb, arg);
}
void grpc_endpoint_shutdown(grpc_endpoint* ep, grpc_error_handle why) { ep->vtable->shutdown(ep, why); }
void grpc_endpoint_destroy(grpc_endpoint* ep) { ep->vtable->destroy(ep); }
const char* grpc_endpoint_get_peer(grpc_endpoint* ep) { return ep->vtable->get_peer(ep); }
int grpc_endpoint_get_fd(grpc_endpoint* ep) { return ep->vtable->get_fd(ep); }

}  // namespace grdma_core
