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
// The endpoint logic -- the read / write state machines of rdma_bp_posix.cc -- is NOT in this file: it is
// include/grdma_endpoint_impl.hpp, one template that this file instantiates with iomgr's types and
// grpc-rdma_amd/csrc/grdma_endpoint.cc with the mirror types the repository's tests, tools and bench legs execute.
// What is left here is glue: the traits, the vtable, the constructor.  The event engines keep working unmodified:
// integration/ibverbs_facade/ puts a PairPollable / Poller / Config with the reference's names over the library.
//
// This file is compiled INSIDE the gRPC tree (it needs iomgr's private headers).  In this
// repository it is syntax-checked against the reference's own headers, together with the reference's two RDMA
// event engines compiled UNMODIFIED against the facade:
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

#include <new>
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

#include "src/core/lib/ibverbs/pair.h"    // integration/ibverbs_facade: PairPollable / Poller over libgrdma_amd.so
#include "src/core/lib/ibverbs/poller.h"

#include "grdma_amd.h"
#include "grdma_endpoint_impl.hpp"  // the endpoint logic, shared with the vtable mirror this repository's tests run

extern grpc_core::TraceFlag grpc_rdma_trace;

namespace {

struct grpc_rdma_hip;
void hip_unref(grpc_rdma_hip* rdma);
void hip_handle_read(void* arg, grpc_error_handle error);

void window_slice_destroy(void* user_data) { grdma_window_unref(static_cast<grdma_window*>(user_data)); }

// iomgr's types behind the shared endpoint logic (include/grdma_endpoint_impl.hpp lists what a traits type provides)
struct iomgr_traits {
  typedef grpc_rdma_hip host;
  typedef grpc_slice_buffer slice_buffer;
  typedef grpc_closure closure;
  typedef grpc_error_handle error;
  static size_t count(const slice_buffer* sb) { return sb->count; }
  static size_t length(const slice_buffer* sb) { return sb->length; }
  static const void* slice_ptr(const slice_buffer* sb, size_t i) { return GRPC_SLICE_START_PTR(sb->slices[i]); }
  static size_t slice_len(const slice_buffer* sb, size_t i) { return GRPC_SLICE_LENGTH(sb->slices[i]); }
  static void reset_and_unref(slice_buffer* sb) { grpc_slice_buffer_reset_and_unref_internal(sb); }
  static void add_copied(slice_buffer* sb, const uint8_t* bytes, size_t len) {
    grpc_slice_buffer_add_indexed(sb, grpc_slice_from_copied_buffer(reinterpret_cast<const char*>(bytes), len));
  }
  static void add_window(slice_buffer* sb, const uint8_t* bytes, size_t len, grdma_window* w) {
    grdma_window_ref(w);  // dropped by window_slice_destroy when the transport unrefs the slice
    grpc_slice_buffer_add_indexed(sb, grpc_slice_new_with_user_data(const_cast<uint8_t*>(bytes), len, window_slice_destroy, w));
  }
  static error none() { return GRPC_ERROR_NONE; }
  static bool is_error(error e) { return e != GRPC_ERROR_NONE; }
  static error ref(error e) { return GRPC_ERROR_REF(e); }
  static void drop(error e) { GRPC_ERROR_UNREF(e); }
  static error annotate(host* rdma, const char* msg);  // rdma_annotate_error, :86-96
  static void run(host*, closure* c, error err) { grpc_core::Closure::Run(DEBUG_LOCATION, c, err); }
  static void run_read_done(host* rdma);
  static void notify_on_read(host* rdma);
  static void notify_on_write(host* rdma);
  static bool is_shutdown(host* rdma);
  static void ref(host* rdma);
  static void unref(host* rdma) { hip_unref(rdma); }
  typedef GRPCProfiler scope;
  static constexpr grpc_stats_time OP_DO_READ = GRPC_STATS_TIME_TRANSPORT_DO_READ;
  static constexpr grpc_stats_time OP_CONTINUE_READ = GRPC_STATS_TIME_TRANSPORT_CONTINUE_READ;
  static constexpr grpc_stats_time OP_HANDLE_READ = GRPC_STATS_TIME_TRANSPORT_HANDLE_READ;
  static constexpr grpc_stats_time OP_READ = GRPC_STATS_TIME_TRANSPORT_READ;
  static constexpr grpc_stats_time OP_FLUSH = GRPC_STATS_TIME_TRANSPORT_FLUSH;
  static constexpr grpc_stats_time OP_HANDLE_WRITE = GRPC_STATS_TIME_TRANSPORT_HANDLE_WRITE;
  static constexpr grpc_stats_time OP_WRITE = GRPC_STATS_TIME_TRANSPORT_WRITE;
};

struct grpc_rdma_hip {
  grpc_endpoint base;  // first member: endpoint.h:112-114
  grpc_fd* em_fd;
  int fd;
  grpc_core::RefCount refcount;

  grdma_pair* pair;                            // the HBM rings and the protocol state
  grpc_core::ibverbs::PairPollable* pollable;  // what the event engine holds (grpc_fd_set_arg): the facade over `pair`
  bool enable_poller;                          // RDMA_BPEV: registered with Poller::Get()

  grdma_ep::core<iomgr_traits> core;           // rdma_read / rdma_do_read / rdma_write / rdma_flush ...

  grpc_closure read_done_closure;
  grpc_closure write_done_closure;

  std::string peer_string;
  std::string local_address;
  grpc_resource_user* resource_user;
};

grpc_error_handle iomgr_traits::annotate(grpc_rdma_hip* rdma, const char* msg) {
  return grpc_error_set_str(
      grpc_error_set_int(grpc_error_set_int(GRPC_ERROR_CREATE_FROM_COPIED_STRING(msg), GRPC_ERROR_INT_FD, rdma->fd),
                         GRPC_ERROR_INT_GRPC_STATUS, GRPC_STATUS_UNAVAILABLE),
      GRPC_ERROR_STR_TARGET_ADDRESS, grpc_slice_from_copied_string(rdma->peer_string.c_str()));
}
void iomgr_traits::run_read_done(grpc_rdma_hip* rdma) {
  grpc_core::Closure::Run(DEBUG_LOCATION, &rdma->read_done_closure, GRPC_ERROR_NONE);
}
void iomgr_traits::notify_on_read(grpc_rdma_hip* rdma) { grpc_fd_notify_on_read(rdma->em_fd, &rdma->read_done_closure); }
void iomgr_traits::notify_on_write(grpc_rdma_hip* rdma) { grpc_fd_notify_on_write(rdma->em_fd, &rdma->write_done_closure); }
bool iomgr_traits::is_shutdown(grpc_rdma_hip* rdma) { return grpc_fd_is_shutdown(rdma->em_fd); }
void iomgr_traits::ref(grpc_rdma_hip* rdma) { rdma->refcount.Ref(); }

gpr_once g_cfg_once = GPR_ONCE_INIT;
grdma_config g_cfg;
void init_process_state() { grdma_config_from_env(&g_cfg); }  // Config::init, config.cc:45-115

void hip_free(grpc_rdma_hip* rdma) {  // rdma_free, :112-131
  grpc_fd_orphan(rdma->em_fd, nullptr, nullptr, "rdma_unref_orphan");
  grpc_resource_user_unref(rdma->resource_user);
  rdma->core.release();
  if (rdma->pair != nullptr) {
    if (rdma->enable_poller) grpc_core::ibverbs::Poller::Get().RemovePollable(rdma->pollable);
    grdma_pair_disconnect(rdma->pair);
    grpc_core::ibverbs::PairPool::Get().Putback(rdma->pollable);  // :128 (deletes the facade object, returns the memory)
    rdma->pollable = nullptr;
    rdma->pair = nullptr;
  }
  delete rdma;
}
void hip_unref(grpc_rdma_hip* rdma) {
  if (GPR_UNLIKELY(rdma->refcount.Unref())) hip_free(rdma);
}

void hip_shutdown(grpc_endpoint* ep, grpc_error_handle why) {  // :106-110
  grpc_rdma_hip* rdma = reinterpret_cast<grpc_rdma_hip*>(ep);
  grpc_fd_shutdown(rdma->em_fd, why);
  grpc_resource_user_shutdown(rdma->resource_user);
}

void hip_destroy(grpc_endpoint* ep) {  // :156-164
  grpc_rdma_hip* rdma = reinterpret_cast<grpc_rdma_hip*>(ep);
  rdma->refcount.Ref();
  rdma->core.abandon_buffered_writes();  // (writes that completed into the send buffer and have not gone out yet)
  hip_unref(rdma);
  hip_unref(rdma);
}

void hip_handle_read(void* arg, grpc_error_handle error) {  // :328-341
  static_cast<grpc_rdma_hip*>(arg)->core.handle_read(error);
}
void hip_handle_write(void* arg, grpc_error_handle error) {  // :527-557
  static_cast<grpc_rdma_hip*>(arg)->core.handle_write(error);
}
void hip_read(grpc_endpoint* ep, grpc_slice_buffer* incoming_buffer, grpc_closure* cb, bool urgent) {  // :343-376
  reinterpret_cast<grpc_rdma_hip*>(ep)->core.read(incoming_buffer, cb, urgent);
}
void hip_write(grpc_endpoint* ep, grpc_slice_buffer* buf, grpc_closure* cb, void* /*arg*/) {  // :559-586
  reinterpret_cast<grpc_rdma_hip*>(ep)->core.write(buf, cb);
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
grpc_endpoint* grpc_rdma_bp_create(grpc_fd* em_fd, const grpc_channel_args* channel_args,
                                   const char* peer_string, bool enable_poller) {
  gpr_once_init(&g_cfg_once, init_process_state);
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
  new (&rdma->refcount) grpc_core::RefCount(1, GRPC_TRACE_FLAG_ENABLED(grpc_rdma_trace) ? "rdma" : nullptr);
  rdma->em_fd = em_fd;
  rdma->pair = nullptr;
  rdma->pollable = nullptr;
  rdma->resource_user = grpc_resource_user_create(resource_quota, peer_string);
  grpc_resource_quota_unref_internal(resource_quota);
  GRPC_CLOSURE_INIT(&rdma->read_done_closure, hip_handle_read, rdma, grpc_schedule_on_exec_ctx);
  GRPC_CLOSURE_INIT(&rdma->write_done_closure, hip_handle_write, rdma, grpc_schedule_on_exec_ctx);

  // PairPool::Take + PairPollable::Init (pair.h:288-296, pair.cc:85-141).  Fine-grained: the peer -- another
  // process -- writes this pair's ring and status block through an IPC mapping.
#ifndef GRPC_ARG_SERVER_URI  // src/core/ext/filters/client_channel/client_channel.h:61 (not pulled in: it drags the resolver stack along)
#define GRPC_ARG_SERVER_URI "grpc.server_uri"
#endif
  std::string pair_id(rdma->peer_string);  // :745-759: a client registers its pair under the target of the server URI
  if (const char* server_uri = grpc_channel_args_find_string(channel_args, GRPC_ARG_SERVER_URI)) {
    std::string uri(server_uri);
    const size_t pos = uri.find_last_of('/');  // get rid of prefix "dns:///"
    if (pos != std::string::npos) pair_id = uri.substr(pos + 1);
  }
  // (the facade's PairPool keeps the id -> PairPollable* table the zero-copy hook looks the call's pair up in,
  //  core_codegen.cc:130-139; the pair itself comes out of the library's pool of pair memory)
  grpc_core::ibverbs::PairPollable* pollable = grpc_core::ibverbs::PairPool::Get().Take(pair_id);
  grdma_pair* pair = pollable != nullptr ? pollable->hip_pair() : nullptr;
  // exchange_data + PairPollable::Connect (:640-692, :767-771; pair.cc:143-168): both ends write
  // their 48-byte Address (plus the memory handles of ring and status block) to the bootstrap
  // socket and read the peer's, full duplex, then map the peer's ring
  if (pair == nullptr || grdma_endpoint_set_async(pair, 0, 0) != 0 || grdma_pair_bootstrap_fd(pair, rdma->fd) != 0) {
    gpr_log(GPR_ERROR, "Connection failed: %s", grdma_last_error());
    if (pair != nullptr) {
      grdma_pair_disconnect(pair);
      grpc_core::ibverbs::PairPool::Get().Putback(pollable);  // :780
    }
    grpc_resource_user_unref(rdma->resource_user);
    delete rdma;
    return nullptr;
  }
  rdma->pair = pair;
  rdma->pollable = pollable;
  rdma->enable_poller = enable_poller;
  rdma->core.init(rdma, pair);
  // :788 -- the event engines keep this pointer as a PairPollable* and poll HasMessage() / HasPendingWrites() /
  // get_status() on it (ev_epollex_rdma_bpev_linux.cc:478, 1111-1116): the facade answers with host loads
  grpc_fd_set_arg(em_fd, rdma->pollable);
  if (enable_poller) grpc_core::ibverbs::Poller::Get().AddPollable(rdma->pollable);  // RDMA_BPEV, :789-791
  return &rdma->base;
}

#endif /* GRPC_POSIX_SOCKET_TCP */
