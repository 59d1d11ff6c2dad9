// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// ref_endpoint_trace: the reference's OWN endpoint -- the write path rdma_write / rdma_flush / rdma_handle_write
// (src/core/lib/iomgr/rdma_bp_posix.cc:470-586: one Send from the cursor per flush, the slices that went out whole dropped,
// the rest waits for the writable edge) and the read path rdma_read / rdma_handle_read / rdma_continue_read /
// rdma_do_read (:180-376: the read sized max(256, readable), the slice kept
// across a would-block, grpc_slice_buffer_trim_end into last_read_buffer and the swap on the next read) -- compiled
// unmodified over the reference's own pair.cc / ring_buffer.cc (software verbs, oracle/fakeverbs) and its own slice
// layer.  Two endpoints are created by grpc_rdma_bp_create itself, which exchanges the pair addresses over a socket
// (here: the two ends of a socketpair, one thread per end, as two processes would).  What the file needs from iomgr is
// stubbed below: an fd object that remembers the closure notify_on_read was given (the driver fires it: "the fd
// became readable"), a slice allocator that allocates at once, errors as opaque handles.
//
// The SAME driver, compiled with -DGRDMA_HIP_ADAPTER over integration/rdma_hip_posix.cc + integration/ibverbs_facade
// and linked against libgrdma_amd.so (oracle/_ref/hip_endpoint_trace; libgrdma_emu.so for the CPU suite:
// hip_endpoint_trace_emu), replays the same operation lists on the file a maintainer SHIPS: the adapter's
// grpc_rdma_bp_create makes both endpoints (the pair addresses exchanged over the same socketpair, both ends in this
// process), and the driver stands in for the event engine the way the reference's engines poll the facade -- it fires
// the readable / writable edge while a drain or a Send of the endpoint is on the device (grdma_endpoint_busy /
// _readable / _writable), so that every operation is run to the quiescent point at which the reference's synchronous
// endpoint stands when its call returns.  tests/test_adapter_trace.py compares the two outputs line by line.
//
//   ops:  S <side> <byte_idx> <seed> <n> <len_1> ... <len_n>    PairPollable::Send on the pair of endpoint <side>
//         E <side>                                             one endpoint read on <side>: grpc_endpoint_read if none is
//                                                              outstanding, then (if the endpoint asked for the readable
//                                                              edge) the edge, once
//         W <side> <seed> <n> <len_1> ... <len_n>               grpc_endpoint_write of a fresh slice buffer on <side>
//         F <side>                                             the writable edge for the write of <side> that waits
//         C <side>                                             grpc_endpoint_shutdown + grpc_endpoint_destroy of <side>
//                                                              (rdma_free: Disconnect -- the peer becomes half closed)
//         Z <side> <seed> <hdr_len> <msg_len>                  the zero-copy hook as the reference's caller would run it:
//                                                              the body of CoreCodegen::grpc_call_allocate_send_buffer
//                                                              (src/cpp/common/core_codegen.cc:126-142: Config's threshold,
//                                                              PairPool::Get().Get(peer id), get_status() == kConnected,
//                                                              AllocateSendBuffer), the HOST writes the message through
//                                                              the pointer (GenericSerialize, proto_utils.h:78-84), then
//                                                              SendZerocopy([header: host memory][the message: a static
//                                                              slice over the buffer]) from the cursor until nothing more
//                                                              goes out
//   out:  S <sent>
//         W | F <1 = the write completed, 0 = it waits for the writable edge> <readable size of the peer> <writable size>
//               <HasPendingWrites>            ("F -" = nothing was waiting)
//         Z <1 = the pool knew the id> <1 = a buffer was handed out> <bytes of every SendZerocopy ...> | <readable size of
//               the peer> <writable size> <HasPendingWrites>
//         E <-1 | bytes delivered> <crc32> <slices> <readable after> <writable of the peer after>      (-1 = would block)
//         a read or write that FAILS prints -2 in place of the count and, behind the line, the error as the endpoint
//         built it: "| <text> | fd <0 or 1: was GRPC_ERROR_INT_FD this endpoint's fd> | status <GRPC_ERROR_INT_GRPC_STATUS>
//         | target <GRPC_ERROR_STR_TARGET_ADDRESS>" (rdma_annotate_error, rdma_bp_posix.cc:86-96)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/socket.h>
#include <string>
#include <thread>
#include <vector>

#include <grpc/slice.h>
#include <grpc/slice_buffer.h>
#include <grpc/support/alloc.h>
#include <grpc/support/log.h>

#include "grpcpp/stats_time.h"
#include "src/core/lib/address_utils/sockaddr_utils.h"
#include "src/core/lib/channel/channel_args.h"
#include "src/core/lib/debug/trace.h"
#include "src/core/lib/gprpp/fork.h"
#include "src/core/lib/ibverbs/config.h"
#include "src/core/lib/ibverbs/pair.h"
#include "src/core/lib/ibverbs/poller.h"
#include "src/core/lib/iomgr/endpoint.h"
#include "src/core/lib/iomgr/ev_posix.h"
#include "src/core/lib/iomgr/exec_ctx.h"
#include "src/core/lib/iomgr/rdma_bp_posix.h"
#include "src/core/lib/iomgr/resource_quota.h"
#include "src/core/lib/slice/slice_internal.h"

// ---- gpr -----------------------------------------------------------------------------------------------------------
extern "C" void gpr_log(const char*, int, gpr_log_severity, const char*, ...) {}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }
extern "C" void* gpr_malloc(size_t n) { return malloc(n ? n : 1); }
extern "C" void* gpr_zalloc(size_t n) { return calloc(n ? n : 1, 1); }
extern "C" void* gpr_realloc(void* p, size_t n) { return realloc(p, n ? n : 1); }
extern "C" void gpr_free(void* p) { free(p); }
#ifdef GRDMA_HIP_ADAPTER
#include <pthread.h>
#include <grpc/support/sync.h>
void gpr_once_init(gpr_once* once, void (*init_function)(void)) { pthread_once(once, init_function); }  // (sync_posix.cc)
#endif
char* gpr_getenv(const char* name) {
  const char* v = getenv(name);
  return v ? strdup(v) : nullptr;
}
// ---- errors: what the endpoint put into them ---------------------------------------------------------------------
struct trace_error {
  std::string text, target;
  intptr_t fd = -1, status = -1;
};
static std::string slice_text(const grpc_slice& s) {
  return std::string(reinterpret_cast<const char*>(GRPC_SLICE_START_PTR(s)), GRPC_SLICE_LENGTH(s));
}
grpc_error_handle grpc_error_create(const char*, int, const grpc_slice& desc, grpc_error_handle*, size_t) {
  trace_error* e = new trace_error();  // (never freed: a test driver)
  e->text = slice_text(desc);
  return reinterpret_cast<grpc_error_handle>(e);
}
grpc_error_handle grpc_error_do_ref(grpc_error_handle e) { return e; }
void grpc_error_do_unref(grpc_error_handle) {}
grpc_error_handle grpc_error_set_int(grpc_error_handle e, grpc_error_ints which, intptr_t v) {
  trace_error* t = reinterpret_cast<trace_error*>(e);
  if (which == GRPC_ERROR_INT_FD) t->fd = v;
  if (which == GRPC_ERROR_INT_GRPC_STATUS) t->status = v;
  return e;
}
grpc_error_handle grpc_error_set_str(grpc_error_handle e, grpc_error_strs which, const grpc_slice& str) {
  if (which == GRPC_ERROR_STR_TARGET_ADDRESS) reinterpret_cast<trace_error*>(e)->target = slice_text(str);
  return e;
}
std::string grpc_error_std_string(grpc_error_handle e) {
  return e == GRPC_ERROR_NONE ? "OK" : reinterpret_cast<trace_error*>(e)->text;
}
// ---- the fd object of the event engine ----------------------------------------------------------------------------
struct grpc_fd {
  int fd;
  void* arg;  // the PairPollable grpc_rdma_bp_create hands over (grpc_fd_set_arg)
  grpc_closure* on_read;
  grpc_closure* on_write;
};
int grpc_fd_wrapped_fd(grpc_fd* f) { return f->fd; }
void grpc_fd_set_arg(grpc_fd* f, void* arg) { f->arg = arg; }
void grpc_fd_notify_on_read(grpc_fd* f, grpc_closure* c) { f->on_read = c; }
void grpc_fd_notify_on_write(grpc_fd* f, grpc_closure* c) { f->on_write = c; }
void grpc_fd_shutdown(grpc_fd*, grpc_error_handle) {}
void grpc_fd_orphan(grpc_fd*, grpc_closure*, int*, const char*) {}
bool grpc_fd_is_shutdown(grpc_fd*) { return false; }
void grpc_pollset_add_fd(grpc_pollset*, grpc_fd*) {}
void grpc_pollset_set_add_fd(grpc_pollset_set*, grpc_fd*) {}
void grpc_pollset_set_del_fd(grpc_pollset_set*, grpc_fd*) {}
bool grpc_event_engine_can_track_errors() { return false; }
std::string grpc_sockaddr_to_uri(const grpc_resolved_address*) { return ""; }
char* grpc_channel_args_find_string(const grpc_channel_args*, const char*) { return nullptr; }
// ---- resource quota: every allocation is granted at once ----------------------------------------------------------
grpc_resource_quota* grpc_resource_quota_create(const char*) { return reinterpret_cast<grpc_resource_quota*>(uintptr_t{8}); }
void grpc_resource_quota_unref_internal(grpc_resource_quota*) {}
grpc_resource_user* grpc_resource_user_create(grpc_resource_quota*, const char*) {
  return reinterpret_cast<grpc_resource_user*>(uintptr_t{8});
}
void grpc_resource_user_unref(grpc_resource_user*) {}
void grpc_resource_user_shutdown(grpc_resource_user*) {}
void grpc_resource_user_slice_allocator_init(grpc_resource_user_slice_allocator* a, grpc_resource_user* u, grpc_iomgr_cb_func,
                                             void*) {
  memset(a, 0, sizeof(*a));
  a->resource_user = u;
}
bool grpc_resource_user_alloc_slices(grpc_resource_user_slice_allocator*, size_t length, size_t count,
                                     grpc_slice_buffer* dest) {
  for (size_t i = 0; i < count; i++) grpc_slice_buffer_add_indexed(dest, GRPC_SLICE_MALLOC(length));
  return true;  // (resource_quota.cc:  true = the slices are there, the caller goes on)
}
// ---- the rest --------------------------------------------------------------------------------------------------------
GRPCProfiler::GRPCProfiler(grpc_stats_time op) : op_(op) {}
GRPCProfiler::~GRPCProfiler() {}
grpc_error_handle grpc_wakeup_fd_init(grpc_wakeup_fd* fd) {
  fd->read_fd = fd->write_fd = -1;
  return GRPC_ERROR_NONE;
}
void grpc_wakeup_fd_destroy(grpc_wakeup_fd*) {}
namespace grpc_core {
TraceFlag::TraceFlag(bool, const char* name) : name_(name), value_(false) {}
GPR_TLS_CLASS_DEF(ExecCtx::exec_ctx_);
Atomic<bool> Fork::support_enabled_(false);
void Fork::DoIncExecCtxCount() {}
void Fork::DoDecExecCtxCount() {}
bool ExecCtx::Flush() { return false; }
#ifndef GRDMA_HIP_ADAPTER  // (the facade's Poller is a header over grdma_poller_*; this driver never enables it)
namespace ibverbs {
void Poller::AddPollable(PairPollable*) {}
void Poller::RemovePollable(PairPollable*) {}
void Poller::begin_polling(int) {}
}  // namespace ibverbs
#endif
}  // namespace grpc_core
grpc_core::TraceFlag grpc_rdma_trace(false, "rdma");

namespace {
using grpc_core::ibverbs::PairPollable;
uint32_t crc32_of(const uint8_t* p, uint64_t n, uint32_t c) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t v = i;
      for (int k = 0; k < 8; k++) v = (v & 1) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
      table[i] = v;
    }
    init = true;
  }
  for (uint64_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}
inline uint8_t pat(uint64_t seed, uint64_t i, uint64_t j) { return (uint8_t)(seed * 131 + i * 17 + j * 7 + (j >> 8)); }
grpc_slice_refcount* const kFakeRefcount = reinterpret_cast<grpc_slice_refcount*>(uintptr_t{0x10});

struct side_state {
  grpc_fd fd;
  grpc_endpoint* ep = nullptr;
  grpc_slice_buffer incoming;
  grpc_closure on_read_done;
  bool outstanding = false, completed = false, failed = false;
  std::string read_error;
  // write side
  grpc_slice_buffer outgoing;
  grpc_closure on_write_done;
  bool write_outstanding = false, write_completed = false, write_failed = false;
  std::string write_error;
};
side_state g_side[2];
std::string error_line(side_state* s, grpc_error_handle error) {
  const trace_error* e = reinterpret_cast<const trace_error*>(error);
  return " | " + e->text + " | fd " + (e->fd == s->fd.fd ? "1" : "0") + " | status " + std::to_string((long long)e->status) +
         " | target " + e->target;
}
void read_done(void* arg, grpc_error_handle error) {
  side_state* s = static_cast<side_state*>(arg);
  s->completed = true;
  s->failed = error != GRPC_ERROR_NONE;
  s->read_error = s->failed ? error_line(s, error) : "";
}
void write_done(void* arg, grpc_error_handle error) {
  side_state* s = static_cast<side_state*>(arg);
  s->write_completed = true;
  s->write_failed = error != GRPC_ERROR_NONE;
  s->write_error = s->write_failed ? error_line(s, error) : "";
}

// ---- what differs between the two builds: how the pair behind an endpoint is asked, and who plays the event engine
#ifdef GRDMA_HIP_ADAPTER
grdma_pair* hip(PairPollable* p) { return p->hip_pair(); }
uint64_t pair_send(PairPollable* p, grpc_slice* sl, size_t n, size_t byte_idx) {
  std::vector<grdma_slice> v(n);
  for (size_t i = 0; i < n; i++) v[i] = grdma_slice{GRPC_SLICE_START_PTR(sl[i]), GRPC_SLICE_LENGTH(sl[i])};
  const int64_t r = grdma_pair_send(hip(p), v.data(), n, byte_idx, GRDMA_MEM_HOST);
  return r > 0 ? (uint64_t)r : 0;
}
bool pending_writes(PairPollable* p) { return grdma_pair_has_pending_writes(hip(p)) > 0; }  // partial_write_, pair.cc:303
// The endpoint's Sends and drains are asynchronous: while one is on the device the endpoint has asked for the edge
// that says "it has completed" -- what the event engines fire when HasMessage() / HasPendingWrites() of the facade
// turn true.  Fired here until the operation has either completed or waits for the PEER (credit, data).
void wait_not_busy(PairPollable* p) {
  for (long spins = 0; grdma_endpoint_busy(hip(p)) > 0; spins++) {
    if (spins > 200000000L) {
      fprintf(stderr, "hip_endpoint_trace: an operation of the endpoint never completed\n");
      _exit(5);
    }
  }
}
bool engine_sees_readable(PairPollable* p) { return grdma_endpoint_readable(hip(p)) > 0; }  // the facade's HasMessage()
bool engine_sees_writable(PairPollable* p) { return grdma_endpoint_writable(hip(p)) > 0; }  // the facade's HasPendingWrites()
#else
uint64_t pair_send(PairPollable* p, grpc_slice* sl, size_t n, size_t byte_idx) { return p->Send(sl, n, byte_idx); }
bool pending_writes(PairPollable* p) { return p->HasPendingWrites(); }
void wait_not_busy(PairPollable*) {}  // (the reference's Sends and reads are synchronous)
bool engine_sees_readable(PairPollable*) { return false; }  // (a read that found nothing would find nothing again)
// what the engines' next pass acts on: HasPendingWrites() (ev_epollex_rdma_bpev_linux.cc:1116) -- with no room in the
// peer's ring the flush it triggers sends nothing and waits again, so only a pair that can take bytes is worth firing
// (TRACE_ENGINE_SETTLE=1: the comparison with the shipped endpoint, whose flush goes on by itself while there is room;
// unset: one rdma_flush per W / F, what tests/test_oracle_vs_ref.py models Send by Send)
bool engine_sees_writable(PairPollable* p) {
  static const bool on = getenv("TRACE_ENGINE_SETTLE") != nullptr;
  return on && p->HasPendingWrites() && p->GetWritableSize() > 0;
}
#endif
// Runs the operation to the quiescent point: the edge is fired as long as the event engine would see a reason to --
// a drain or a Send that has completed on the device (the shipped endpoint), a partial write with room in the peer's
// ring (both) -- and no longer (bounded: an edge that changes nothing is not fired for ever).
void settle_read(side_state& s, PairPollable* p) {
  for (int k = 0; k < 4096 && !s.completed && s.fd.on_read != nullptr; k++) {
    wait_not_busy(p);
    if (!engine_sees_readable(p)) break;
    grpc_closure* c = s.fd.on_read;
    s.fd.on_read = nullptr;
    grpc_core::Closure::Run(DEBUG_LOCATION, c, GRPC_ERROR_NONE);
  }
}
void settle_write(side_state& s, PairPollable* p) {
  for (int k = 0; k < 4096 && !s.write_completed && s.fd.on_write != nullptr; k++) {
    wait_not_busy(p);
    if (!engine_sees_writable(p)) break;
    grpc_closure* c = s.fd.on_write;
    s.fd.on_write = nullptr;
    grpc_core::Closure::Run(DEBUG_LOCATION, c, GRPC_ERROR_NONE);
  }
}
}  // namespace

int main() {
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 2;
  for (int i = 0; i < 2; i++) {
    memset(&g_side[i].fd, 0, sizeof(grpc_fd));
    g_side[i].fd.fd = sv[i];
    grpc_slice_buffer_init(&g_side[i].incoming);
    GRPC_CLOSURE_INIT(&g_side[i].on_read_done, read_done, &g_side[i], grpc_schedule_on_exec_ctx);
    grpc_slice_buffer_init(&g_side[i].outgoing);
    GRPC_CLOSURE_INIT(&g_side[i].on_write_done, write_done, &g_side[i], grpc_schedule_on_exec_ctx);
  }
  {
    std::thread t([&] { g_side[1].ep = grpc_rdma_bp_create(&g_side[1].fd, nullptr, "peer-of-1", false); });
    g_side[0].ep = grpc_rdma_bp_create(&g_side[0].fd, nullptr, "peer-of-0", false);
    t.join();
  }
  if (!g_side[0].ep || !g_side[1].ep) {
    fprintf(stderr, "grpc_rdma_bp_create failed\n");
    return 2;
  }
  PairPollable* pair[2] = {static_cast<PairPollable*>(g_side[0].fd.arg), static_cast<PairPollable*>(g_side[1].fd.arg)};
  bool closed[2] = {false, false};
  char op;
  while (scanf(" %c", &op) == 1) {
    if (op == 'S') {
      int side;
      unsigned long long byte_idx, seed, n;
      if (scanf("%d %llu %llu %llu", &side, &byte_idx, &seed, &n) != 4) return 3;
      std::vector<std::vector<uint8_t>> mem(n);
      std::vector<grpc_slice> sl(n);
      const bool busy = g_side[side].write_outstanding || closed[side];  // (a raw Send would move a waiting write's cursor)
      for (unsigned long long i = 0; i < n; i++) {
        unsigned long long len;
        if (scanf("%llu", &len) != 1) return 3;
        if (busy) continue;
        mem[i].resize(len ? len : 1);
        for (unsigned long long j = 0; j < len; j++) mem[i][j] = pat(seed, i, j);
        memset(&sl[i], 0, sizeof(grpc_slice));
        sl[i].refcount = kFakeRefcount;
        sl[i].data.refcounted.length = len;
        sl[i].data.refcounted.bytes = mem[i].data();
      }
      if (busy) {
        printf("S busy\n");
        continue;
      }
      printf("S %llu\n", (unsigned long long)pair_send(pair[side], sl.data(), n, byte_idx));
    } else if (op == 'Z') {
      int side;
      unsigned long long seed, hdr_len, msg_len;
      if (scanf("%d %llu %llu %llu", &side, &seed, &hdr_len, &msg_len) != 4) return 3;
      if (g_side[side].write_outstanding || closed[side]) {  // (as S: a raw Send would move a waiting write's cursor)
        printf("Z busy\n");
        continue;
      }
      // ---- core_codegen.cc:126-142, as it stands there (the `call` is replaced by the id its peer lookup returns:
      //      grpc_call_get_peer_id = the peer string the endpoint registered its pair under, surface/call.cc:663-672)
      void* buffer = nullptr;
      bool known = false;
      {
        auto& config = grpc_core::ibverbs::Config::Get();
        if (msg_len / 1024 >= config.get_zerocopy_threshold_kb()) {
          auto& pair_pool = grpc_core::ibverbs::PairPool::Get();
          const std::string peer = std::string("peer-of-") + (side ? "1" : "0");
          auto* found = pair_pool.Get(peer);
          known = found == pair[side];
          if (found != nullptr && found->get_status() == grpc_core::ibverbs::PairStatus::kConnected) {
            buffer = found->AllocateSendBuffer(msg_len);
          }
        }
      }
      printf("Z %d %d", known ? 1 : 0, buffer != nullptr ? 1 : 0);
      if (buffer != nullptr) {
        uint8_t* q = static_cast<uint8_t*>(buffer);
        for (unsigned long long j = 0; j < msg_len; j++) q[j] = pat(seed, 1, j);  // <- the host serialises into the buffer
        std::vector<uint8_t> hdr(hdr_len ? hdr_len : 1);
        for (unsigned long long j = 0; j < hdr_len; j++) hdr[j] = pat(seed, 0, j);
        grpc_slice sl[2];
        memset(sl, 0, sizeof(sl));
        sl[0].refcount = sl[1].refcount = kFakeRefcount;
        sl[0].data.refcounted.length = hdr_len;
        sl[0].data.refcounted.bytes = hdr.data();
        sl[1].data.refcounted.length = msg_len;
        sl[1].data.refcounted.bytes = q;
        size_t idx = hdr_len ? 0 : 1, byte = 0;
        for (int k = 0; k < 64 && idx < 2; k++) {
          uint64_t sent = pair[side]->SendZerocopy(sl + idx, 2 - idx, byte);
          printf(" %llu", (unsigned long long)sent);
          if (sent == 0) break;
          while (sent > 0) {  // the cursor walk of rdma_flush (rdma_bp_posix.cc:480-493)
            const uint64_t room = GRPC_SLICE_LENGTH(sl[idx]) - byte;
            if (sent >= room) { sent -= room; idx++; byte = 0; }
            else { byte += sent; sent = 0; }
          }
        }
      }
      printf(" | %llu %llu %d\n", closed[1 - side] ? 0ull : (unsigned long long)pair[1 - side]->GetReadableSize(),
             (unsigned long long)pair[side]->GetWritableSize(), pending_writes(pair[side]) ? 1 : 0);
    } else if (op == 'C') {
      // grpc_endpoint_shutdown + grpc_endpoint_destroy (rdma_shutdown / rdma_destroy -> rdma_free, :106-164): the pair
      // is disconnected (peer_exit in the peer's status buffer) and goes back to the pool
      int side;
      if (scanf("%d", &side) != 1) return 3;
      side_state& s = g_side[side];
      // (a read or write still outstanding would be failed by the fd's shutdown in a real engine; this driver's fd
      // does nothing, so only an endpoint with nothing outstanding is closed)
      if (closed[side] || s.outstanding || s.write_outstanding) {
        printf("C busy\n");
        continue;
      }
      grpc_error_handle why = GRPC_ERROR_CREATE_FROM_STATIC_STRING("endpoint closed by the driver");
      s.ep->vtable->shutdown(s.ep, why);
      s.ep->vtable->destroy(s.ep);
      closed[side] = true;
      printf("C %d\n", side);
    } else if (op == 'W' || op == 'F') {
      // W: grpc_endpoint_write of a fresh slice buffer (rdma_write -> rdma_flush: ONE Send from the cursor; what is left
      //    waits for the writable edge).  F: the writable edge for a write that waits (rdma_handle_write -> rdma_flush).
      int side;
      if (scanf("%d", &side) != 1) return 3;
      side_state& s = g_side[side];
      if (closed[side]) return 4;
      if (op == 'W') {
        unsigned long long seed, n;
        if (scanf("%llu %llu", &seed, &n) != 2) return 3;
        if (s.write_outstanding) {  // (one write at a time, GPR_ASSERT(rdma->write_cb == nullptr): the list's author need not know)
          for (unsigned long long i = 0, len; i < n; i++)
            if (scanf("%llu", &len) != 1) return 3;
          printf("W busy\n");
          continue;
        }
        grpc_slice_buffer_reset_and_unref(&s.outgoing);
        for (unsigned long long i = 0; i < n; i++) {
          unsigned long long len;
          if (scanf("%llu", &len) != 1) return 3;
          grpc_slice m = grpc_slice_malloc_large(len ? len : 1);
          uint8_t* q = GRPC_SLICE_START_PTR(m);
          for (unsigned long long j = 0; j < len; j++) q[j] = pat(seed, i, j);
          if (len == 0) { grpc_slice_unref(m); m = grpc_empty_slice(); }
          grpc_slice_buffer_add_indexed(&s.outgoing, m);
        }
        s.write_completed = false;
        s.write_outstanding = true;
        s.fd.on_write = nullptr;
        s.ep->vtable->write(s.ep, &s.outgoing, &s.on_write_done, nullptr);
      } else {
        if (!s.write_outstanding || s.fd.on_write == nullptr) {
          printf("F -\n");
          continue;
        }
        grpc_closure* c = s.fd.on_write;
        s.fd.on_write = nullptr;
        grpc_core::Closure::Run(DEBUG_LOCATION, c, GRPC_ERROR_NONE);
      }
      settle_write(s, pair[side]);
      if (s.write_completed) s.write_outstanding = false;
      if (s.write_completed && s.write_failed) {
        printf("%c -2%s\n", op, s.write_error.c_str());
        continue;
      }
      printf("%c %d %llu %llu %d\n", op, s.write_completed ? 1 : 0,
             closed[1 - side] ? 0ull : (unsigned long long)pair[1 - side]->GetReadableSize(),
             (unsigned long long)pair[side]->GetWritableSize(), pending_writes(pair[side]) ? 1 : 0);
    } else if (op == 'E') {
      int side;
      if (scanf("%d", &side) != 1) return 3;
      side_state& s = g_side[side];
      if (closed[side]) return 4;
      s.completed = false;
      s.fd.on_read = nullptr;
      static grpc_closure* pending[2] = {nullptr, nullptr};
      if (!s.outstanding) {
        s.outstanding = true;
        s.ep->vtable->read(s.ep, &s.incoming, &s.on_read_done, /*urgent=*/false);
      }
      if (!s.completed) {
        // the endpoint asked for the readable edge (first read, inq == 0, or a would-block): the fd is readable, once
        if (s.fd.on_read != nullptr) pending[side] = s.fd.on_read;
        grpc_closure* c = pending[side];
        pending[side] = nullptr;
        s.fd.on_read = nullptr;
        if (c != nullptr) grpc_core::Closure::Run(DEBUG_LOCATION, c, GRPC_ERROR_NONE);
        settle_read(s, pair[side]);
        if (!s.completed && s.fd.on_read != nullptr) pending[side] = s.fd.on_read;  // would block: armed again
      }
      if (s.completed) {
        s.outstanding = false;
        if (s.failed) {
          printf("E -2%s\n", s.read_error.c_str());
          continue;
        }
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = 0; i < s.incoming.count; i++)
          c = crc32_of(GRPC_SLICE_START_PTR(s.incoming.slices[i]), GRPC_SLICE_LENGTH(s.incoming.slices[i]), c);
        printf("E %lld %u %zu %llu %llu\n", (long long)s.incoming.length, c ^ 0xFFFFFFFFu, s.incoming.count,
               (unsigned long long)pair[side]->GetReadableSize(),
               closed[1 - side] ? 0ull : (unsigned long long)pair[1 - side]->GetWritableSize());
      } else {
        printf("E -1 0 0 %llu %llu\n", (unsigned long long)pair[side]->GetReadableSize(),
               closed[1 - side] ? 0ull : (unsigned long long)pair[1 - side]->GetWritableSize());
      }
    } else {
      return 3;
    }
  }
  fflush(stdout);
  _exit(0);
}
