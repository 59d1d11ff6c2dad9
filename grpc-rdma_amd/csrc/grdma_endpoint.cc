// grpc_endpoint implementation for RDMA_BP / RDMA_BPEV on top of the C ABI.
// Mirrors src/core/lib/iomgr/rdma_bp_posix.cc function by function; the byte work
// happens in the HIP kernels behind grdma_endpoint_write_* / grdma_endpoint_read.
#include "../../include/grdma_endpoint.hpp"
#include "../../include/grdma_endpoint_impl.hpp"
#include "../../include/grdma_profiler.hpp"

#include <sys/epoll.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
// The endpoint logic itself -- rdma_read / rdma_handle_read / rdma_do_read, rdma_write / rdma_handle_write /
// rdma_flush -- is include/grdma_endpoint_impl.hpp, shared with the drop-in for the gRPC tree
// (integration/rdma_hip_posix.cc).  This file supplies the mirror types' traits and what stands in for grpc_fd and
// the event engine here: the armed flags, the shutdown switch, the pollset.
namespace {
struct grpc_rdma;
}  // namespace

// `pollable` of the RDMA event engines (ev_epollex_rdma_bpev_linux.cc:208-246): the fds of the
// set, the epoll fd their wakeup fds are registered with, the busy-polling budget
struct grpc_pollset {
  bool bpev;
  int polling_timeout_us;
  int epfd;
  std::mutex rdma_mu;                  // p->rdma_mu: the passes of N worker threads are serialised (:1103)
  std::vector<grpc_rdma*> rdma_fds;
  grdma_pollset_stats stats;
};

namespace {

struct window_slice_refcount {  // a slice that points into a receive window (grpc_slice_new_with_user_data)
  grpc_slice_refcount base;
  grdma_window* window;
};
void window_slice_destroy(grpc_slice_refcount* r) {
  auto* w = reinterpret_cast<window_slice_refcount*>(r);
  grdma_window_unref(w->window);
  free(w);
}

void rdma_unref(grpc_rdma* rdma);
void rdma_handle_read(grpc_rdma* rdma, grpc_error_handle error);
void rdma_handle_write(grpc_rdma* rdma, grpc_error_handle error);

struct mirror_traits {
  typedef grpc_rdma host;
  typedef grpc_slice_buffer slice_buffer;
  typedef grpc_closure closure;
  typedef grpc_error_handle error;
  static size_t count(const slice_buffer* sb) { return sb->count; }
  static size_t length(const slice_buffer* sb) { return sb->length; }
  static const void* slice_ptr(const slice_buffer* sb, size_t i) { return GRPC_SLICE_START_PTR(sb->slices[i]); }
  static size_t slice_len(const slice_buffer* sb, size_t i) { return GRPC_SLICE_LENGTH(sb->slices[i]); }
  static void reset_and_unref(slice_buffer* sb) { grpc_slice_buffer_reset_and_unref(sb); }
  static void add_copied(slice_buffer* sb, const uint8_t* bytes, size_t len) {
    grpc_slice_buffer_add_indexed(sb, grpc_slice_from_copied_buffer(reinterpret_cast<const char*>(bytes), len));
  }
  static void add_window(slice_buffer* sb, const uint8_t* bytes, size_t len, grdma_window* w) {
    auto* rc = static_cast<window_slice_refcount*>(malloc(sizeof(window_slice_refcount)));
    rc->base.refs.store(1);
    rc->base.destroy = window_slice_destroy;
    rc->window = w;
    grdma_window_ref(w);
    grpc_slice s;
    s.refcount = &rc->base;
    s.data.refcounted.bytes = const_cast<uint8_t*>(bytes);
    s.data.refcounted.length = len;
    grpc_slice_buffer_add_indexed(sb, s);
  }
  static error none() { return GRPC_ERROR_NONE; }
  static bool is_error(error e) { return e != GRPC_ERROR_NONE; }
  static error ref(error e) { return GRPC_ERROR_REF(e); }
  static void drop(error e) { GRPC_ERROR_UNREF(e); }
  static error annotate(host* rdma, const char* msg);
  static void run(host*, closure* c, error err) {  // grpc_core::Closure::Run
    c->cb(c->cb_arg, err);
    GRPC_ERROR_UNREF(err);
  }
  static void run_read_done(host* rdma) { rdma_handle_read(rdma, GRPC_ERROR_NONE); }
  // grpc_fd_notify_on_read / _write: a shut-down fd runs the closure with the shutdown error at once
  static void notify_on_read(host* rdma);
  static void notify_on_write(host* rdma);
  static bool is_shutdown(host* rdma);
  static void ref(host* rdma);
  static void unref(host* rdma) { rdma_unref(rdma); }
  typedef grdma_profiler scope;
  static constexpr int OP_DO_READ = GRDMA_STATS_TIME_TRANSPORT_DO_READ;
  static constexpr int OP_CONTINUE_READ = GRDMA_STATS_TIME_TRANSPORT_CONTINUE_READ;
  static constexpr int OP_HANDLE_READ = GRDMA_STATS_TIME_TRANSPORT_HANDLE_READ;
  static constexpr int OP_READ = GRDMA_STATS_TIME_TRANSPORT_READ;
  static constexpr int OP_FLUSH = GRDMA_STATS_TIME_TRANSPORT_FLUSH;
  static constexpr int OP_HANDLE_WRITE = GRDMA_STATS_TIME_TRANSPORT_HANDLE_WRITE;
  static constexpr int OP_WRITE = GRDMA_STATS_TIME_TRANSPORT_WRITE;
};


struct grpc_rdma {  // rdma_bp_posix.cc:45-88
  grpc_endpoint base;  // must be first (endpoint.h:112-114)
  int fd;
  std::atomic<int> refcount;
  std::atomic<bool> shutdown;
  std::atomic<bool> shutdown_claimed{false};  // the one rdma_shutdown that sets shutdown_error and publishes `shutdown`
  grpc_error_handle shutdown_error;  // (written before `shutdown` is set, read after it has been seen set)
  grdma_pair* pair;
  bool enable_poller;
  grdma_ep::core<mirror_traits> core;
  std::atomic<bool> read_armed;   // notify_on_read pending (set by the closure that asks, taken by the thread
  std::atomic<bool> write_armed;  // notify_on_write pending        that finds the edge)
  std::string peer_string;
  std::string local_address;
  grpc_pollset* pollset;  // the set this endpoint was added to (grpc_pollset_add_fd)
};

grdma_poller* g_poller = nullptr;  // Poller::Get(): one per process, created with the first BPEV endpoint
std::mutex g_poller_mu;

void pollset_del_fd(grpc_pollset* ps, grpc_rdma* rdma);

grpc_error_handle mirror_traits::annotate(grpc_rdma* rdma, const char* msg) {  // rdma_annotate_error, :86-96
    grpc_error_handle src = GRPC_ERROR_CREATE_FROM_STATIC_STRING(msg);
    src->fd = rdma->fd;
    src->grpc_status = GRPC_STATUS_UNAVAILABLE;  // "so that application may choose to retry"
    src->target_address = rdma->peer_string;
    return src;
  }
// Any number of threads run grdma_pollset_work while another one may shut the endpoint down: the flag is armed FIRST
// and the shutdown switch looked at afterwards, so that either rdma_shutdown's exchange finds the armed flag or this
// thread finds the switch -- whoever takes the flag back runs the closure with the shutdown error, exactly once
// (what grpc_fd's lock-free event does for notify_on vs. shutdown).
void mirror_traits::notify_on_read(grpc_rdma* rdma) {
    rdma->read_armed.store(true, std::memory_order_seq_cst);
    if (rdma->shutdown.load(std::memory_order_seq_cst) && rdma->read_armed.exchange(false)) rdma_handle_read(rdma, rdma->shutdown_error);
  }
void mirror_traits::notify_on_write(grpc_rdma* rdma) {
    rdma->write_armed.store(true, std::memory_order_seq_cst);
    if (rdma->shutdown.load(std::memory_order_seq_cst) && rdma->write_armed.exchange(false)) rdma_handle_write(rdma, rdma->shutdown_error);
  }
bool mirror_traits::is_shutdown(grpc_rdma* rdma) { return rdma->shutdown; }
void mirror_traits::ref(grpc_rdma* rdma) { rdma->refcount.fetch_add(1); }

void rdma_handle_read(grpc_rdma* rdma, grpc_error_handle error) { rdma->core.handle_read(error); }
void rdma_handle_write(grpc_rdma* rdma, grpc_error_handle error) { rdma->core.handle_write(error); }

void rdma_read(grpc_endpoint* ep, grpc_slice_buffer* incoming_buffer, grpc_closure* cb, bool urgent) {
  reinterpret_cast<grpc_rdma*>(ep)->core.read(incoming_buffer, cb, urgent);
}

void rdma_write(grpc_endpoint* ep, grpc_slice_buffer* buf, grpc_closure* cb, void* /*arg*/) {
  reinterpret_cast<grpc_rdma*>(ep)->core.write(buf, cb);
}

void rdma_shutdown(grpc_endpoint* ep, grpc_error_handle why) {  // :106-110
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  // (one winner: two concurrent shutdowns must not both assign shutdown_error and both run the armed closures --
  //  grpc_fd_shutdown resolves the same race atomically)
  if (!rdma->shutdown_claimed.exchange(true, std::memory_order_acq_rel)) {
    rdma->shutdown_error = why ? why : GRPC_ERROR_CREATE_FROM_STATIC_STRING("Endpoint shutdown");
    rdma->shutdown.store(true, std::memory_order_seq_cst);
    // grpc_fd_shutdown: pending notify_on_* closures run with the error
    rdma->refcount.fetch_add(1);
    if (rdma->read_armed.exchange(false)) rdma_handle_read(rdma, rdma->shutdown_error);
    if (rdma->write_armed.exchange(false)) rdma_handle_write(rdma, rdma->shutdown_error);
    rdma_unref(rdma);
  } else {
    GRPC_ERROR_UNREF(why);
  }
}

void rdma_free(grpc_rdma* rdma) {  // :112-132
  if (rdma->pollset != nullptr) pollset_del_fd(rdma->pollset, rdma);  // grpc_fd_orphan
  rdma->core.release();
  if (rdma->pair != nullptr) {
    if (rdma->enable_poller && g_poller != nullptr) grdma_poller_remove(g_poller, rdma->pair);
    grdma_pair_disconnect(rdma->pair);
    grdma_pair_pool_putback(rdma->pair);  // PairPool::Putback (rdma_bp_posix.cc:128): its memory goes back to the pool
    rdma->pair = nullptr;
  }
  GRPC_ERROR_UNREF(rdma->shutdown_error);
  delete rdma;
}

void rdma_unref(grpc_rdma* rdma) {
  if (rdma->refcount.fetch_sub(1) == 1) rdma_free(rdma);
}

void rdma_destroy(grpc_endpoint* ep) {  // :134-139
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  rdma->refcount.fetch_add(1);
  rdma->core.abandon_buffered_writes();
  rdma_unref(rdma);
  rdma_unref(rdma);
}
void pollset_del_fd(grpc_pollset* ps, grpc_rdma* rdma) {
  std::lock_guard<std::mutex> lk(ps->rdma_mu);
  auto it = std::find(ps->rdma_fds.begin(), ps->rdma_fds.end(), rdma);
  if (it == ps->rdma_fds.end()) return;
  ps->rdma_fds.erase(it);
  if (ps->bpev && rdma->pair != nullptr) {
    const int wfd = grdma_pair_get_wakeup_fd(rdma->pair);
    if (wfd >= 0) epoll_ctl(ps->epfd, EPOLL_CTL_DEL, wfd, nullptr);
    if (rdma->fd >= 0) epoll_ctl(ps->epfd, EPOLL_CTL_DEL, rdma->fd, nullptr);
  }
}

// grpc_pollset_add_fd -> pollable_add_fd (ev_epollex_rdma_bpev_linux.cc:705-745): the fd joins
// p->rdma_fds; in BPEV mode the pair's wakeup fd joins the epoll set, its data pointer tagged
// with bit 1 so that process_events can tell it from a socket (:725-741), and so does the TCP fd
// itself (untagged: a hang-up on it is how a vanished peer is noticed while the thread sleeps)
void rdma_add_to_pollset(grpc_endpoint* ep, grpc_pollset* ps) {
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (ps == nullptr || rdma->pollset == ps) return;
  if (rdma->pollset != nullptr) pollset_del_fd(rdma->pollset, rdma);
  rdma->pollset = ps;
  std::lock_guard<std::mutex> lk(ps->rdma_mu);
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
    if (rdma->fd > 2) {  // (a real socket: the stand-in fds 0..2 of tests that pass none are skipped)
      struct epoll_event ev;
      memset(&ev, 0, sizeof ev);
      ev.events = EPOLLRDHUP | EPOLLET;
      ev.data.ptr = reinterpret_cast<void*>(rdma);
      epoll_ctl(ps->epfd, EPOLL_CTL_ADD, rdma->fd, &ev);  // (fails harmlessly for an fd that is not pollable)
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
  // (fine-grained: the pair may be handed to a peer in another process through grdma_pair_bootstrap_fd)
  static const int pool_on = grdma_pair_pool_reserve(0, 0, 0, 0, static_cast<uint64_t>(cfg.hip_pair_pool_mb) << 20);
  (void)pool_on;
  // PairPool::Take(pair_id), rdma_bp_posix.cc:745-761: the id is what follows the last '/' of the peer string
  std::string pair_id = peer_string ? peer_string : "";
  {
    const size_t pos = pair_id.find_last_of('/');
    if (pos != std::string::npos) pair_id = pair_id.substr(pos + 1);
  }
  grdma_pair* pair = grdma_pair_pool_take(pair_id.c_str(), static_cast<uint64_t>(cfg.ring_buffer_size_kb) * 1024,
                                          cfg.max_sge,
                                          (cfg.hip_wire_direct ? GRDMA_WIRE_DIRECT : GRDMA_WIRE_STAGED) | GRDMA_RING_FINE_GRAINED);
  if (pair == nullptr) return nullptr;  // "Connection failed" path :777-784
  if (grdma_endpoint_set_async(pair, 0, 0) < 0) {
    grdma_pair_pool_putback(pair);
    return nullptr;
  }
  grpc_rdma* rdma = new grpc_rdma();
  rdma->base.vtable = &vtable;
  rdma->fd = fd;
  rdma->peer_string = peer_string ? peer_string : "";
  rdma->local_address = "";
  rdma->refcount.store(1);
  rdma->shutdown = false;
  rdma->shutdown_error = nullptr;
  rdma->pair = pair;
  rdma->enable_poller = enable_poller;
  rdma->core.init(rdma, pair);
  rdma->read_armed.store(false);
  rdma->write_armed.store(false);
  rdma->pollset = nullptr;
  if (enable_poller) {  // RDMA_BPEV: Poller::Get().AddPollable(pair), :789-791
    std::lock_guard<std::mutex> lk(g_poller_mu);
    if (g_poller == nullptr) g_poller = grdma_poller_create(cfg.poller_thread_num, cfg.poller_sleep_timeout_ms);
    if (g_poller != nullptr) grdma_poller_add(g_poller, pair);
  }
  return &rdma->base;
}

bool grpc_rdma_bp_connect_loopback(grpc_endpoint* a, grpc_endpoint* b) {
  return grdma_pair_connect(reinterpret_cast<grpc_rdma*>(a)->pair,
                            reinterpret_cast<grpc_rdma*>(b)->pair) == 0;
}

bool grpc_rdma_bp_connect_fd(grpc_endpoint* ep) {  // exchange_data + Connect over the endpoint's socket, :763-784
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  return grdma_pair_bootstrap_fd(rdma->pair, rdma->fd) == 0;
}

bool grdma_endpoint_set_latency_mode(grpc_endpoint* ep, bool on, uint64_t arm_reads) {
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (grdma_pair_set_latency_mode(rdma->pair, on ? 1 : 0) != 0) return false;
  rdma->core.arm_reads = on ? arm_reads : 0;
  if (!on || arm_reads == 0) grdma_pair_arm_read(rdma->pair, 0);
  return true;
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

namespace {
// The readable / writable edge of one endpoint, as pollable_epoll synthesises it (:1105-1149): plain host loads.
inline void edges_of(grpc_rdma* r, bool* in, bool* out) {
  const int status = grdma_pair_get_status(r->pair);
  *in = *out = false;
  if (status == GRDMA_PAIR_CONNECTED) {
    *in = r->read_armed && grdma_endpoint_readable(r->pair) > 0;
    *out = r->write_armed && grdma_endpoint_writable(r->pair) > 0;
  } else if (status == GRDMA_PAIR_HALF_CLOSED || status == GRDMA_PAIR_ERROR) {
    *in = r->read_armed;  // "Generate an event, so do_read will handle connection close"
    *out = r->write_armed;
  }
}

// fd_become_readable / fd_become_writable for one endpoint
int deliver(grpc_rdma* rdma, bool in, bool out) {
  int ran = 0;
  rdma->refcount.fetch_add(1);  // a closure may destroy the endpoint
  if (in && rdma->read_armed.exchange(false)) {
    rdma_handle_read(rdma, GRPC_ERROR_NONE);
    ran++;
  }
  if (out && rdma->write_armed.exchange(false)) {
    rdma_handle_write(rdma, GRPC_ERROR_NONE);
    ran++;
  }
  rdma_unref(rdma);
  return ran;
}
}  // namespace

int grdma_endpoint_poll(grpc_endpoint* ep) {
  grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(ep);
  if (!rdma->read_armed && !rdma->write_armed) return 0;
  bool in, out;
  edges_of(rdma, &in, &out);
  return (in || out) ? deliver(rdma, in, out) : 0;
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

// pollset_work for the RDMA platforms.  Any number of threads may be inside: the busy-poll pass over the fds runs
// under rdma_mu, as pollable_epoll does (ev_epollex_rdma_bpev_linux.cc:1103-1145), the events it found are
// delivered outside the lock (closures re-enter the endpoint and may add / destroy fds); an endpoint's
// read side and write side each belong to one closure at a time (the armed flags are taken under the lock).
int grdma_pollset_work(grpc_pollset* ps, int timeout_ms) {
  grdma_profiler profiler(GRDMA_STATS_TIME_POLLSET_WORK);  // ev_epollex_rdma_bp_linux.cc pollset_work
  using clock = std::chrono::steady_clock;
  const auto t_begin = clock::now();
  // the busy-polling budget: RDMA_BP polls for the whole timeout, RDMA_BPEV for at most
  // polling_timeout_us (:1093-1099)
  int64_t budget_us = ps->bpev ? ps->polling_timeout_us : INT64_MAX;
  if (timeout_ms >= 0) budget_us = std::min<int64_t>(budget_us, (int64_t)timeout_ms * 1000);
  struct event { grpc_rdma* r; bool in, out; };
  std::vector<event> evs;
  int ran = 0;
  int64_t elapsed_us = 0;
  bool any_fd = false;
  auto run_events = [&]() {
    for (event& e : evs) {
      // (the flags were cleared under the lock: this thread owns the edges it found)
      if (e.in) { rdma_handle_read(e.r, GRPC_ERROR_NONE); ran++; }
      if (e.out) { rdma_handle_write(e.r, GRPC_ERROR_NONE); ran++; }
      rdma_unref(e.r);
    }
    evs.clear();
  };
  do {
    bool any_armed = false;
    {
      std::lock_guard<std::mutex> lk(ps->rdma_mu);
      ps->stats.passes++;
      any_fd = !ps->rdma_fds.empty();
      for (grpc_rdma* r : ps->rdma_fds) {
        if (evs.size() >= 100) break;  // MAX_EPOLL_EVENTS
        if (!r->read_armed && !r->write_armed) continue;
        any_armed = true;
        bool in, out;
        edges_of(r, &in, &out);
        if (!in && !out) continue;
        if (in) in = r->read_armed.exchange(false);    // fd_become_readable: the closure is taken
        if (out) out = r->write_armed.exchange(false);
        if (!in && !out) continue;
        r->refcount.fetch_add(1);         // a closure may destroy the endpoint
        evs.push_back(event{r, in, out});
      }
    }
    if (!any_fd) break;
    run_events();
    // nothing is armed and nobody else can arm: an unbounded busy-poll would never return
    if (ran == 0 && !any_armed && !ps->bpev && timeout_ms < 0) break;
    elapsed_us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t_begin).count();
  } while (ran == 0 && elapsed_us < budget_us);
  if (ran == 0 && ps->bpev && any_fd) {
    // busy-polling timed out: switch to epoll on the wakeup fds (:1155-1170)
    int left_ms = timeout_ms;
    if (timeout_ms > 0) left_ms = (int)std::max<int64_t>(0, timeout_ms - elapsed_us / 1000);
    struct epoll_event eev[100];  // MAX_EPOLL_EVENTS
    int r;
    do {
      r = epoll_wait(ps->epfd, eev, 100, left_ms);
    } while (r < 0 && errno == EINTR);
    if (r < 0) return -1;
    {
      std::lock_guard<std::mutex> lk(ps->rdma_mu);
      ps->stats.epoll_waits++;
      for (int k = 0; k < r; k++) {
        const uintptr_t tagged = reinterpret_cast<uintptr_t>(eev[k].data.ptr);
        grpc_rdma* rdma = reinterpret_cast<grpc_rdma*>(tagged & ~(uintptr_t)2);
        if (std::find(ps->rdma_fds.begin(), ps->rdma_fds.end(), rdma) == ps->rdma_fds.end()) continue;
        bool in, out;
        if (tagged & 2) {
          // pollable_process_events for a wakeup fd (:1010-1037): consume, then look at the pair
          if (grdma_pair_consume_wakeup(rdma->pair) > 0) ps->stats.wakeups_consumed++;
          edges_of(rdma, &in, &out);
        } else {
          // the socket itself: a hang-up makes the fd readable and writable (:1038-1064); get_status() has
          // noticed the dead peer by then (socket check), so the closures report the close
          grdma_pair_get_status(rdma->pair);
          in = rdma->read_armed;
          out = rdma->write_armed;
        }
        if (in) in = rdma->read_armed.exchange(false);
        if (out) out = rdma->write_armed.exchange(false);
        if (!in && !out) continue;
        rdma->refcount.fetch_add(1);
        evs.push_back(event{rdma, in, out});
      }
    }
    run_events();
  }
  {
    std::lock_guard<std::mutex> lk(ps->rdma_mu);
    ps->stats.closures_run += (uint64_t)ran;
  }
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
