// rdma_hip_posix.cc -- the endpoint of the RDMA_BP / RDMA_BPEV platforms over libgrdma_amd.so.
//
// Drop-in for src/core/lib/iomgr/rdma_bp_posix.cc of the reference tree (gRPC 1.38 + RR-Compound):
// the SAME entry point, grpc_rdma_bp_create(grpc_fd*, const grpc_channel_args*, const char*, bool)
// (rdma_bp_posix.h:41-44), the same grpc_endpoint_vtable slots in the same order
// (endpoint.h:42-57), the same callbacks-with-error contract (one read and one write outstanding,
// UNAVAILABLE-annotated errors, rdma_bp_posix.cc:86-96).  What changes is what stands behind it:
// where the reference keeps a grpc_core::ibverbs::PairPollable* and runs ring_buffer.cc on the
// host, this file keeps a grdma_pair* and every byte-moving step is a call through the C ABI
// of include/grdma_amd.h (the rings live in HBM, the codec runs on the GPU).
//
// This file is compiled INSIDE the gRPC tree (it needs iomgr's private headers).  In this
// repository it is syntax-checked against the reference's own headers:
//     integration/check.sh        (g++ -fsyntax-only, absl / verbs stand-ins from integration/shim)
// and tests/test_capi_and_host.py runs that check wherever /root/reference exists.
#include <grpc/support/port_platform.h>

#include "src/core/lib/iomgr/port.h"

#ifdef GRPC_POSIX_SOCKET_TCP

#include <errno.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include <grpc/slice.h>
#include <grpc/support/log.h>
#include <grpcpp/stats_time.h>

#include "src/core/lib/address_utils/sockaddr_utils.h"
#include "src/core/lib/channel/channel_args.h"
#include "src/core/lib/debug/trace.h"
#include "src/core/lib/gprpp/ref_counted.h"
#include "src/core/lib/iomgr/ev_posix.h"
#include "src/core/lib/iomgr/rdma_bp_posix.h"
#include "src/core/lib/iomgr/resource_quota.h"
#include "src/core/lib/slice/slice_internal.h"

#include "grdma_amd.h"

extern grpc_core::TraceFlag grpc_rdma_trace;

namespace {

// how many endpoint reads one device pass performs ahead of the transport asking for them
constexpr uint64_t kReadAhead = 1024;
// slices handed to one grdma_endpoint_write_begin (the ABI takes at most 4095)
constexpr size_t kWriteWindow = 4000;

struct grpc_rdma_hip {
  grpc_endpoint base;  // first member: endpoint.h:112-114
  grpc_fd* em_fd;
  int fd;
  bool is_first_read;  // rdma_bp_posix.cc:50-52
  grpc_core::RefCount refcount;

  grdma_pair* pair;     // stands where PairPollable* stood
  bool enable_poller;   // RDMA_BPEV: the pair is registered with the background poller

  grpc_slice_buffer* incoming_buffer;
  int inq;
  grpc_slice_buffer* outgoing_buffer;
  std::vector<grdma_slice> out_views;  // {ptr, len} of outgoing_buffer's slices
  size_t out_next;                     // first view not yet handed to the pair
  bool window_active;                  // the pair holds a window that has not gone out whole

  grpc_closure* read_cb;
  grpc_closure* write_cb;
  grpc_closure read_done_closure;
  grpc_closure write_done_closure;

  // read-ahead: completions of ONE device pass, served to the following endpoint reads
  std::vector<grdma_read_slice> ahead;
  size_t ahead_next;
  uint8_t* ahead_bytes;  // pinned host memory
  uint64_t ahead_cap;
  uint64_t ahead_base;

  std::string peer_string;
  std::string local_address;
  grpc_resource_user* resource_user;
};

grdma_poller* g_poller = nullptr;  // Poller::Get() (poller.h:16-71): one per process
gpr_once g_poller_once = GPR_ONCE_INIT;
grdma_config g_cfg;
void init_process_state() {
  grdma_config_from_env(&g_cfg);  // Config::init, config.cc:45-115
  g_poller = grdma_poller_create(g_cfg.poller_thread_num, g_cfg.poller_sleep_timeout_ms);
}

grpc_error_handle hip_annotate_error(grpc_error_handle src, grpc_rdma_hip* rdma) {  // :86-96
  return grpc_error_set_str(
      grpc_error_set_int(grpc_error_set_int(src, GRPC_ERROR_INT_FD, rdma->fd), GRPC_ERROR_INT_GRPC_STATUS,
                         GRPC_STATUS_UNAVAILABLE),
      GRPC_ERROR_STR_TARGET_ADDRESS, grpc_slice_from_copied_string(rdma->peer_string.c_str()));
}

void notify_on_read(grpc_rdma_hip* rdma) { grpc_fd_notify_on_read(rdma->em_fd, &rdma->read_done_closure); }
void notify_on_write(grpc_rdma_hip* rdma) { grpc_fd_notify_on_write(rdma->em_fd, &rdma->write_done_closure); }

void hip_free(grpc_rdma_hip* rdma) {  // rdma_free, :112-131
  grpc_fd_orphan(rdma->em_fd, nullptr, nullptr, "rdma_unref_orphan");
  grpc_resource_user_unref(rdma->resource_user);
  if (rdma->pair != nullptr) {
    if (rdma->enable_poller && g_poller != nullptr) grdma_poller_remove(g_poller, rdma->pair);
    grdma_pair_disconnect(rdma->pair);
    grdma_pair_destroy(rdma->pair);  // PairPool::Putback
    rdma->pair = nullptr;
  }
  grdma_host_free_pinned(rdma->ahead_bytes);
  delete rdma;
}
void hip_unref(grpc_rdma_hip* rdma) {
  if (GPR_UNLIKELY(rdma->refcount.Unref())) hip_free(rdma);
}
void hip_ref(grpc_rdma_hip* rdma) { rdma->refcount.Ref(); }

void hip_shutdown(grpc_endpoint* ep, grpc_error_handle why) {  // :106-110
  grpc_rdma_hip* rdma = reinterpret_cast<grpc_rdma_hip*>(ep);
  grpc_fd_shutdown(rdma->em_fd, why);
  grpc_resource_user_shutdown(rdma->resource_user);
}

void hip_destroy(grpc_endpoint* ep) {  // :156-164
  hip_unref(reinterpret_cast<grpc_rdma_hip*>(ep));
}

void call_read_cb(grpc_rdma_hip* rdma, grpc_error_handle error) {  // :166-172
  grpc_closure* cb = rdma->read_cb;
  rdma->read_cb = nullptr;
  rdma->incoming_buffer = nullptr;
  grpc_core::Closure::Run(DEBUG_LOCATION, cb, error);
}

// rdma_continue_read + rdma_do_read (:306-326, :180-291).  The slice sizing (max(256, readable)),
// the Recv loop and the credit return run on the device; one pass performs up to kReadAhead
// endpoint reads, each of which filled its slice (the chain stops at the first read that would
// block), so serving them one by one later gives the transport exactly the slices the reference's
// loop would have produced.
void hip_do_read(grpc_rdma_hip* rdma) {
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_DO_READ);
  int would_block = 0;
  int64_t n = 0;
  if (rdma->ahead_next >= rdma->ahead.size()) {
    rdma->ahead.resize(kReadAhead);
    rdma->ahead_next = 0;
    n = grdma_endpoint_read(rdma->pair, kReadAhead, rdma->ahead.data(), kReadAhead, &would_block);
    rdma->ahead.resize(n > 0 ? static_cast<size_t>(n) : 0);
    if (n > 0) {
      uint64_t lo = ~0ull, hi = 0;
      for (const grdma_read_slice& a : rdma->ahead) {
        lo = std::min(lo, a.off);
        hi = std::max(hi, a.off + a.len);
      }
      if (hi - lo > rdma->ahead_cap) {
        grdma_host_free_pinned(rdma->ahead_bytes);
        rdma->ahead_cap = 2 * (hi - lo);
        rdma->ahead_bytes = static_cast<uint8_t*>(grdma_host_alloc_pinned(rdma->ahead_cap));
        if (rdma->ahead_bytes == nullptr) rdma->ahead_cap = 0;
      }
      rdma->ahead_base = lo;
      if (rdma->ahead_bytes == nullptr || grdma_pair_arena_copy_out(rdma->pair, lo, rdma->ahead_bytes, hi - lo) != 0) {
        rdma->ahead.clear();
        n = -1;
      }
    }
  }
  if (rdma->ahead_next < rdma->ahead.size()) {
    const grdma_read_slice s = rdma->ahead[rdma->ahead_next++];
    grpc_slice out = GRPC_SLICE_MALLOC(s.len);
    memcpy(GRPC_SLICE_START_PTR(out), rdma->ahead_bytes + (s.off - rdma->ahead_base), s.len);
    grpc_slice_buffer_add_indexed(rdma->incoming_buffer, out);
    rdma->inq = 1;
    call_read_cb(rdma, GRPC_ERROR_NONE);
    hip_unref(rdma);
    return;
  }
  if (n < 0) {
    grpc_slice_buffer_reset_and_unref_internal(rdma->incoming_buffer);
    std::string err = std::string("Pair error, ") + grdma_last_error();
    call_read_cb(rdma, hip_annotate_error(GRPC_ERROR_CREATE_FROM_COPIED_STRING(err.c_str()), rdma));
    hip_unref(rdma);
    return;
  }
  rdma->inq = 0;
  const int status = grdma_pair_get_status(rdma->pair);
  if (status == GRDMA_PAIR_HALF_CLOSED) {  // :220-228
    grpc_slice_buffer_reset_and_unref_internal(rdma->incoming_buffer);
    call_read_cb(rdma, hip_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("Pair closed"), rdma));
    hip_unref(rdma);
  } else if (status == GRDMA_PAIR_ERROR) {  // :229-238
    grpc_slice_buffer_reset_and_unref_internal(rdma->incoming_buffer);
    std::string err = std::string("Pair error, ") + grdma_last_error();
    call_read_cb(rdma, hip_annotate_error(GRPC_ERROR_CREATE_FROM_COPIED_STRING(err.c_str()), rdma));
    hip_unref(rdma);
  } else {
    notify_on_read(rdma);  // the edge is consumed: ask for a new one, :241-243
  }
}

void hip_handle_read(void* arg, grpc_error_handle error) {  // :328-341
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_HANDLE_READ);
  grpc_rdma_hip* rdma = static_cast<grpc_rdma_hip*>(arg);
  if (GPR_UNLIKELY(error != GRPC_ERROR_NONE)) {
    grpc_slice_buffer_reset_and_unref_internal(rdma->incoming_buffer);
    call_read_cb(rdma, GRPC_ERROR_REF(error));
    hip_unref(rdma);
    return;
  }
  GRPCProfiler cont(GRPC_STATS_TIME_TRANSPORT_CONTINUE_READ);
  hip_do_read(rdma);
}

void hip_read(grpc_endpoint* ep, grpc_slice_buffer* incoming_buffer, grpc_closure* cb, bool urgent) {  // :343-376
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_READ);
  grpc_rdma_hip* rdma = reinterpret_cast<grpc_rdma_hip*>(ep);
  GPR_ASSERT(rdma->read_cb == nullptr);
  rdma->read_cb = cb;
  rdma->incoming_buffer = incoming_buffer;
  grpc_slice_buffer_reset_and_unref_internal(incoming_buffer);
  hip_ref(rdma);
  if (rdma->is_first_read) {
    rdma->is_first_read = false;
    notify_on_read(rdma);
  } else if (!urgent && rdma->inq == 0) {
    notify_on_read(rdma);
  } else {
    grpc_core::Closure::Run(DEBUG_LOCATION, &rdma->read_done_closure, GRPC_ERROR_NONE);
  }
}

// rdma_flush (:470-524): Send from the cursor; the cursor walk (outgoing_byte_idx) runs on the
// device.  true = the whole buffer went out or *error is set; false = wait for the writable edge.
bool hip_flush(grpc_rdma_hip* rdma, grpc_error_handle* error) {
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_FLUSH);
  *error = GRPC_ERROR_NONE;
  auto fail_with = [&](const char* what) {
    std::string err = std::string(what) + ", " + grdma_last_error();
    *error = hip_annotate_error(GRPC_ERROR_CREATE_FROM_COPIED_STRING(err.c_str()), rdma);
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
      const size_t cnt = std::min(kWriteWindow, rdma->out_views.size() - rdma->out_next);
      if (grdma_endpoint_write_begin(rdma->pair, rdma->out_views.data() + rdma->out_next, cnt, GRDMA_MEM_HOST) < 0)
        return fail_with("RDMA Pair has an internal error");
      rdma->out_next += cnt;
      rdma->window_active = true;
    }
    int done = 0;
    if (grdma_endpoint_write_step(rdma->pair, &done) < 0) return fail_with("RDMA Pair has an internal error");
    if (done) {
      rdma->window_active = false;
      continue;
    }
    const int status = grdma_pair_get_status(rdma->pair);  // partial send, :499-518
    if (status == GRDMA_PAIR_CONNECTED) return false;
    if (status == GRDMA_PAIR_HALF_CLOSED) {
      *error = hip_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("Peer has been exited"), rdma);
      grdma_endpoint_write_abort(rdma->pair);
      rdma->window_active = false;
      rdma->out_views.clear();
      rdma->out_next = 0;
      grpc_slice_buffer_reset_and_unref(rdma->outgoing_buffer);
      return true;
    }
    return fail_with("RDMA Pair has an internal error");
  }
  rdma->out_views.clear();
  rdma->out_next = 0;
  grpc_slice_buffer_reset_and_unref(rdma->outgoing_buffer);  // :519-523
  return true;
}

void hip_handle_write(void* arg, grpc_error_handle error) {  // :527-557
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_HANDLE_WRITE);
  grpc_rdma_hip* rdma = static_cast<grpc_rdma_hip*>(arg);
  if (error != GRPC_ERROR_NONE) {
    grpc_closure* cb = rdma->write_cb;
    rdma->write_cb = nullptr;
    grdma_endpoint_write_abort(rdma->pair);
    rdma->window_active = false;
    rdma->out_views.clear();
    rdma->out_next = 0;
    grpc_core::Closure::Run(DEBUG_LOCATION, cb, GRPC_ERROR_REF(error));
    hip_unref(rdma);
    return;
  }
  if (!hip_flush(rdma, &error)) {
    notify_on_write(rdma);
  } else {
    grpc_closure* cb = rdma->write_cb;
    rdma->write_cb = nullptr;
    grpc_core::Closure::Run(DEBUG_LOCATION, cb, error);
    hip_unref(rdma);
  }
}

void hip_write(grpc_endpoint* ep, grpc_slice_buffer* buf, grpc_closure* cb, void* /*arg*/) {  // :559-586
  GRPCProfiler profiler(GRPC_STATS_TIME_TRANSPORT_WRITE);
  grpc_rdma_hip* rdma = reinterpret_cast<grpc_rdma_hip*>(ep);
  grpc_error_handle error = GRPC_ERROR_NONE;
  GPR_ASSERT(rdma->write_cb == nullptr);
  if (buf->length == 0) {
    grpc_core::Closure::Run(DEBUG_LOCATION, cb,
                            grpc_fd_is_shutdown(rdma->em_fd)
                                ? hip_annotate_error(GRPC_ERROR_CREATE_FROM_STATIC_STRING("EOF"), rdma)
                                : GRPC_ERROR_NONE);
    return;
  }
  rdma->outgoing_buffer = buf;
  rdma->out_views.resize(buf->count);
  for (size_t i = 0; i < buf->count; i++)
    rdma->out_views[i] = {GRPC_SLICE_START_PTR(buf->slices[i]), GRPC_SLICE_LENGTH(buf->slices[i])};
  rdma->out_next = 0;
  rdma->window_active = false;
  if (!hip_flush(rdma, &error)) {
    hip_ref(rdma);
    rdma->write_cb = cb;
    notify_on_write(rdma);
  } else {
    grpc_core::Closure::Run(DEBUG_LOCATION, cb, error);
  }
}

void hip_add_to_pollset(grpc_endpoint* ep, grpc_pollset* pollset) {
  grpc_pollset_add_fd(pollset, reinterpret_cast<grpc_rdma_hip*>(ep)->em_fd);
}
void hip_add_to_pollset_set(grpc_endpoint* ep, grpc_pollset_set* pollset_set) {
  grpc_pollset_set_add_fd(pollset_set, reinterpret_cast<grpc_rdma_hip*>(ep)->em_fd);
}
void hip_delete_from_pollset_set(grpc_endpoint* ep, grpc_pollset_set* pollset_set) {
  grpc_pollset_set_del_fd(pollset_set, reinterpret_cast<grpc_rdma_hip*>(ep)->em_fd);
}
absl::string_view hip_get_peer(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma_hip*>(ep)->peer_string; }
absl::string_view hip_get_local_address(grpc_endpoint* ep) {
  return reinterpret_cast<grpc_rdma_hip*>(ep)->local_address;
}
int hip_get_fd(grpc_endpoint* ep) { return reinterpret_cast<grpc_rdma_hip*>(ep)->fd; }
grpc_resource_user* hip_get_resource_user(grpc_endpoint* ep) {
  return reinterpret_cast<grpc_rdma_hip*>(ep)->resource_user;
}
bool hip_can_track_err(grpc_endpoint* ep) {  // :621-633: only IP sockets on an engine that tracks errors
  if (!grpc_event_engine_can_track_errors()) return false;
  struct sockaddr sa;
  socklen_t sa_len = sizeof sa;
  const int fd = reinterpret_cast<grpc_rdma_hip*>(ep)->fd;
  return getsockname(fd, &sa, &sa_len) == 0 && (sa.sa_family == AF_INET || sa.sa_family == AF_INET6);
}

// the URI of the bootstrap socket's local end ("" when the fd has no name), :716-726
std::string local_uri_of(int fd) {
  grpc_resolved_address a;
  memset(&a, 0, sizeof a);
  a.len = sizeof a.addr;
  if (getsockname(fd, reinterpret_cast<sockaddr*>(a.addr), &a.len) < 0) return std::string();
  return grpc_sockaddr_to_uri(&a);
}

const grpc_endpoint_vtable vtable = {hip_read,
                                     hip_write,
                                     hip_add_to_pollset,
                                     hip_add_to_pollset_set,
                                     hip_delete_from_pollset_set,
                                     hip_shutdown,
                                     hip_destroy,
                                     hip_get_resource_user,
                                     hip_get_peer,
                                     hip_get_local_address,
                                     hip_get_fd,
                                     hip_can_track_err};  // order: endpoint.h:42-57

}  // namespace

// rdma_bp_posix.h:41-44.  Takes ownership of em_fd; nullptr when the pair cannot be brought up
// (the caller then closes the fd, tcp_server_posix.cc:269-273 / tcp_client_posix.cc).
grpc_endpoint* grpc_rdma_bp_create(grpc_fd* em_fd, const grpc_channel_args* /*channel_args*/,
                                   const char* peer_string, bool enable_poller) {
  gpr_once_init(&g_poller_once, init_process_state);
  if (grdma_init(g_cfg.hip_device) < 0) {  // Device::Get, device.cc:45-101
    gpr_log(GPR_ERROR, "grdma_init: %s", grdma_last_error());
    return nullptr;
  }
  grpc_resource_quota* resource_quota = grpc_resource_quota_create(nullptr);
  grpc_rdma_hip* rdma = new grpc_rdma_hip();
  rdma->base.vtable = &vtable;
  rdma->peer_string = peer_string;
  rdma->fd = grpc_fd_wrapped_fd(em_fd);
  rdma->local_address = local_uri_of(rdma->fd);
  rdma->read_cb = nullptr;
  rdma->write_cb = nullptr;
  rdma->incoming_buffer = nullptr;
  rdma->outgoing_buffer = nullptr;
  rdma->is_first_read = true;
  new (&rdma->refcount) grpc_core::RefCount(1, GRPC_TRACE_FLAG_ENABLED(grpc_rdma_trace) ? "rdma" : nullptr);
  rdma->em_fd = em_fd;
  rdma->resource_user = grpc_resource_user_create(resource_quota, peer_string);
  grpc_resource_quota_unref_internal(resource_quota);
  GRPC_CLOSURE_INIT(&rdma->read_done_closure, hip_handle_read, rdma, grpc_schedule_on_exec_ctx);
  GRPC_CLOSURE_INIT(&rdma->write_done_closure, hip_handle_write, rdma, grpc_schedule_on_exec_ctx);
  rdma->inq = 1;
  rdma->out_next = 0;
  rdma->window_active = false;
  rdma->ahead_next = 0;
  rdma->ahead_bytes = nullptr;
  rdma->ahead_cap = 0;
  rdma->ahead_base = 0;

  // PairPool::Take + PairPollable::Init (pair.h:288-296, pair.cc:85-141)
  grdma_pair* pair = grdma_pair_create(static_cast<uint64_t>(g_cfg.ring_buffer_size_kb) * 1024, g_cfg.max_sge,
                                       GRDMA_WIRE_STAGED);
  if (pair == nullptr) {
    gpr_log(GPR_ERROR, "grdma_pair_create: %s", grdma_last_error());
    grpc_resource_user_unref(rdma->resource_user);
    delete rdma;
    return nullptr;
  }
  // exchange_data + PairPollable::Connect (:640-692, :767-771; pair.cc:143-168): both ends write
  // their 48-byte Address (plus the memory handles of ring and status block) to the bootstrap
  // socket and read the peer's, full duplex, then map the peer's ring
  if (grdma_pair_bootstrap_fd(pair, rdma->fd) != 0) {
    gpr_log(GPR_ERROR, "Connection failed: %s", grdma_last_error());
    grdma_pair_disconnect(pair);
    grdma_pair_destroy(pair);
    grpc_resource_user_unref(rdma->resource_user);
    delete rdma;
    return nullptr;
  }
  rdma->pair = pair;
  rdma->enable_poller = enable_poller;
  grpc_fd_set_arg(em_fd, pair);  // :788 -- the event engine polls the pair through grdma_poll_pairs()
  if (enable_poller && g_poller != nullptr) grdma_poller_add(g_poller, pair);  // RDMA_BPEV, :789-791
  return &rdma->base;
}

#endif /* GRPC_POSIX_SOCKET_TCP */
