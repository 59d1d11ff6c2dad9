// grdma_endpoint.hpp -- C++ host mirror of the reference's endpoint interface for the
// RDMA_BP / RDMA_BPEV path, layered on the C ABI (grdma_amd.h).
//
// The real grpc_endpoint_vtable cannot be compiled in this repository (gRPC 1.38
// needs abseil, upb, ...), so this header restates the few
// plain types that cross the boundary with the SAME names, field order and
// semantics, inside namespace grdma_core to stay ODR-safe when linked next to a real
// gRPC:
//   grpc_slice / grpc_slice_buffer   include/grpc/impl/codegen/slice.h:60-101
//   grpc_closure                     src/core/lib/iomgr/closure.h:56-89
//   grpc_endpoint / _vtable          src/core/lib/iomgr/endpoint.h:42-57,112-114
//   grpc_endpoint_read/write/...     src/core/lib/iomgr/endpoint.cc:56-108
//   grpc_rdma_bp_create              src/core/lib/iomgr/rdma_bp_posix.h:41-44
// The behaviour contract is the one of rdma_bp_posix.cc (one outstanding read, one
// outstanding write, callbacks with an error handle, UNAVAILABLE-annotated errors).
#ifndef GRDMA_ENDPOINT_HPP
#define GRDMA_ENDPOINT_HPP

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>

#include "grdma_amd.h"

namespace grdma_core {

// ---- slices: include/grpc/impl/codegen/slice.h ------------------------------------
struct grpc_slice_refcount {
  std::atomic<int> refs;
  void (*destroy)(grpc_slice_refcount*);
};

#define GRDMA_SLICE_INLINED_SIZE (sizeof(size_t) + sizeof(uint8_t*) - 1 + sizeof(void*))  // 23

struct grpc_slice {
  grpc_slice_refcount* refcount;
  union grpc_slice_data {
    struct grpc_slice_refcounted {
      size_t length;
      uint8_t* bytes;
    } refcounted;
    struct grpc_slice_inlined {
      uint8_t length;
      uint8_t bytes[GRDMA_SLICE_INLINED_SIZE];
    } inlined;
  } data;
};
static_assert(sizeof(grpc_slice) == 32, "grpc_slice layout (slice.h:60-75)");

#define GRDMA_SLICE_BUFFER_INLINE_ELEMENTS 8
struct grpc_slice_buffer {
  grpc_slice* base_slices;
  grpc_slice* slices;
  size_t count;
  size_t capacity;
  size_t length;
  grpc_slice inlined[GRDMA_SLICE_BUFFER_INLINE_ELEMENTS];
};
static_assert(sizeof(grpc_slice_buffer) == 296, "grpc_slice_buffer layout (slice.h:82-94)");

#define GRPC_SLICE_START_PTR(slice) \
  ((slice).refcount ? (slice).data.refcounted.bytes : (slice).data.inlined.bytes)
#define GRPC_SLICE_LENGTH(slice) \
  ((slice).refcount ? (slice).data.refcounted.length : (slice).data.inlined.length)

grpc_slice grpc_slice_malloc(size_t length);              // slice.cc grpc_slice_malloc
grpc_slice grpc_slice_from_copied_buffer(const char* src, size_t len);
void grpc_slice_unref(grpc_slice s);
void grpc_slice_buffer_init(grpc_slice_buffer* sb);
void grpc_slice_buffer_destroy(grpc_slice_buffer* sb);
void grpc_slice_buffer_add(grpc_slice_buffer* sb, grpc_slice s);          // slice_buffer.cc:136-171
size_t grpc_slice_buffer_add_indexed(grpc_slice_buffer* sb, grpc_slice s);
void grpc_slice_buffer_reset_and_unref(grpc_slice_buffer* sb);
void grpc_slice_buffer_swap(grpc_slice_buffer* a, grpc_slice_buffer* b);

// ---- errors: src/core/lib/iomgr/error.h (GRPC_ERROR_NONE == nullptr in 1.38) --------
struct grpc_error {
  std::string description;
  int fd;                      // GRPC_ERROR_INT_FD
  int grpc_status;             // GRPC_ERROR_INT_GRPC_STATUS (14 = UNAVAILABLE)
  std::string target_address;  // GRPC_ERROR_STR_TARGET_ADDRESS
  std::atomic<int> refs;
};
typedef grpc_error* grpc_error_handle;
#define GRPC_ERROR_NONE nullptr
#define GRPC_STATUS_UNAVAILABLE 14
grpc_error_handle GRPC_ERROR_CREATE_FROM_STATIC_STRING(const char* desc);
grpc_error_handle GRPC_ERROR_REF(grpc_error_handle e);
void GRPC_ERROR_UNREF(grpc_error_handle e);

// ---- closures: src/core/lib/iomgr/closure.h ---------------------------------------------
typedef void (*grpc_iomgr_cb_func)(void* arg, grpc_error_handle error);
struct grpc_closure {
  grpc_closure* next;
  grpc_iomgr_cb_func cb;
  void* cb_arg;
  grpc_error_handle error;
};
inline grpc_closure* GRPC_CLOSURE_INIT(grpc_closure* c, grpc_iomgr_cb_func cb, void* arg, void*) {
  c->next = nullptr; c->cb = cb; c->cb_arg = arg; c->error = nullptr;
  return c;
}

// ---- endpoint: src/core/lib/iomgr/endpoint.h ---------------------------------------------
struct grpc_endpoint;
struct grpc_pollset;
struct grpc_pollset_set;
struct grpc_resource_user;
struct grpc_endpoint_vtable {  // order: endpoint.h:42-57
  void (*read)(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, bool urgent);
  void (*write)(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, void* arg);
  void (*add_to_pollset)(grpc_endpoint* ep, grpc_pollset* pollset);
  void (*add_to_pollset_set)(grpc_endpoint* ep, grpc_pollset_set* pollset);
  void (*delete_from_pollset_set)(grpc_endpoint* ep, grpc_pollset_set* pollset);
  void (*shutdown)(grpc_endpoint* ep, grpc_error_handle why);
  void (*destroy)(grpc_endpoint* ep);
  grpc_resource_user* (*get_resource_user)(grpc_endpoint* ep);
  const char* (*get_peer)(grpc_endpoint* ep);           // absl::string_view in the reference
  const char* (*get_local_address)(grpc_endpoint* ep);  //   "
  int (*get_fd)(grpc_endpoint* ep);
  bool (*can_track_err)(grpc_endpoint* ep);
};
struct grpc_endpoint {
  const grpc_endpoint_vtable* vtable;
};

// endpoint.cc:56-108
void grpc_endpoint_read(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, bool urgent);
void grpc_endpoint_write(grpc_endpoint* ep, grpc_slice_buffer* slices, grpc_closure* cb, void* arg);
void grpc_endpoint_shutdown(grpc_endpoint* ep, grpc_error_handle why);
void grpc_endpoint_destroy(grpc_endpoint* ep);
const char* grpc_endpoint_get_peer(grpc_endpoint* ep);
int grpc_endpoint_get_fd(grpc_endpoint* ep);

// rdma_bp_posix.h:41-44.  `fd` stands in for the grpc_fd wrapping the bootstrap socket
// (kept only as an identity here: the wire is the loop-back backend).  Returns
// nullptr on failure, like the reference (the caller then closes the fd,
// tcp_server_posix.cc:269-273).
grpc_endpoint* grpc_rdma_bp_create(int fd, const char* peer_string, bool enable_poller);
// Loop-back stand-in for exchange_data + PairPollable::Connect
// (rdma_bp_posix.cc:767-771): joins two freshly created endpoints.
bool grpc_rdma_bp_connect_loopback(grpc_endpoint* a, grpc_endpoint* b);
// exchange_data + PairPollable::Connect over the endpoint's own socket (rdma_bp_posix.cc:763-784): what
// grpc_rdma_bp_create does with the fd it was given, for a peer in ANOTHER process.
bool grpc_rdma_bp_connect_fd(grpc_endpoint* ep);
// Small-message mode of this build (no reference counterpart): every write / read of the endpoint becomes one
// command to the resident latency engine instead of a launch chain; arm_reads != 0 keeps a read armed with the pair
// whenever the endpoint waits for the readable edge, so that an in-process peer's small sends carry the drain.
bool grdma_endpoint_set_latency_mode(grpc_endpoint* ep, bool on, uint64_t arm_reads);
// grpc_endpoint_create (endpoint.cc:33-54): switches on GRPC_PLATFORM_TYPE.  TCP is
// out of scope here and yields nullptr.
grpc_endpoint* grpc_endpoint_create(int fd, const char* peer_string, bool server);

// What pollset_work does for RDMA fds (ev_epollex_rdma_bpev_linux.cc:1105-1149,
// 1010-1037): turn ring state into readable / writable edges and run the armed
// closures.  Returns the number of closures run.
int grdma_endpoint_poll(grpc_endpoint* ep);

grdma_pair* grdma_endpoint_pair(grpc_endpoint* ep);  // the PairPollable handed to grpc_fd_set_arg

// ---- the pollset of the RDMA event engines --------------------------------------------------
// Mirror of `pollable` + pollable_epoll + pollable_process_events of
// ev_epollex_rdma_bp_linux.cc / ev_epollex_rdma_bpev_linux.cc (:1079-1172, :977-1075): endpoints
// join through the vtable's add_to_pollset; one grdma_pollset_work() call is one pollset_work pass:
//   * busy-poll: walks the fds of the set calling get_status / readable / writable -- plain loads of each pair's
//     host-visible state line, no device call (the reference's HasMessage / HasPendingWrites are host loads
//     too) --, synthesises EPOLLIN / EPOLLOUT and runs the armed closures; a half-closed or failed pair yields
//     EPOLLIN so that do_read reports the close;
//   * RDMA_BP (bpev = false): keeps busy-polling until an event or the timeout;
//   * RDMA_BPEV (bpev = true): busy-polls for at most busy_polling_timeout_us
//     (GRPC_RDMA_BUSY_POLLING_TIMEOUT_US), then sleeps in epoll_wait on the wakeup fds of its pairs
//     -- signalled by the background poller (grdma_poller, poller.cc:52-106) -- and consumes the
//     wakeup (:1010-1037) before it looks at the pair again.
// Returns the number of closures run, < 0 on error.  Any number of threads may call it on one set: the pass over
// the fds runs under the set's rdma_mu (ev_epollex_rdma_bpev_linux.cc:1103-1145), closures run outside it.
grpc_pollset* grdma_pollset_create(bool bpev, int busy_polling_timeout_us);
void grdma_pollset_destroy(grpc_pollset* ps);
int grdma_pollset_work(grpc_pollset* ps, int timeout_ms);
size_t grdma_pollset_size(const grpc_pollset* ps);
struct grdma_pollset_stats { uint64_t passes, device_polls, epoll_waits, wakeups_consumed, closures_run; };
void grdma_pollset_get_stats(const grpc_pollset* ps, grdma_pollset_stats* out);

}  // namespace grdma_core
#endif  // GRDMA_ENDPOINT_HPP
