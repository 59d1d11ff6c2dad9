// Host side of the data plane: the PairPollable analogue and the C ABI declared
// in include/grdma_amd.h.  This layer owns HBM allocations and enqueues the
// kernels of grdma_kernels.hip; it never computes on payload bytes itself and
// has no CPU fallback -- without a HIP device every entry point fails.
//
// Reference counterparts: src/core/lib/ibverbs/pair.{h,cc} (PairPollable),
// src/core/lib/iomgr/rdma_bp_posix.cc (endpoint read/write loops).
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <sched.h>

#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <unordered_map>
#include <map>
#include <shared_mutex>
#include <thread>
#include <poll.h>
#include <signal.h>
#include <sys/socket.h>
#include <errno.h>
#include <sys/eventfd.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "../../include/grdma_amd.h"
#include "../../include/grdma_profiler.hpp"
#include "grdma_dev.h"
#include "grdma_host.h"
#include "grdma_ops.h"
#include "grdma_wire_verbs.h"
#include "grdma_link.h"

extern "C" {
hipError_t grdma_launch_link(lk_ctl* const*, uint32_t, uint32_t, uint64_t, hipStream_t);
uint32_t grdma_link_resident_blocks(void);
hipError_t grdma_launch_tx_plan(const grdma_tx_op*, uint32_t, hipStream_t);
hipError_t grdma_launch_tx_plan_seq(const grdma_tx_op*, uint32_t, uint32_t, hipStream_t);
hipError_t grdma_launch_tx_plan_zc(const grdma_zc_op*, uint32_t, hipStream_t);
hipError_t grdma_launch_tx_commit(grdma_conn* const*, const uint64_t*, uint32_t, hipStream_t);
hipError_t grdma_launch_tx_commit1(grdma_conn*, uint64_t, hipStream_t);
hipError_t grdma_launch_rx_commit1(grdma_conn*, uint64_t, hipStream_t);
hipError_t grdma_launch_rx_idle(grdma_conn*, hipStream_t);
hipError_t grdma_launch_copy(const grdma_plan* const*, uint32_t, uint32_t, hipStream_t);
hipError_t grdma_launch_rx_plan(const grdma_rx_op*, uint32_t, hipStream_t);
hipError_t grdma_launch_rx_apply(const grdma_rx_op*, uint32_t, uint32_t, hipStream_t);
hipError_t grdma_launch_poll(grdma_conn* const*, uint32_t, uint64_t*, uint64_t*, uint64_t*, uint64_t*,
                             hipStream_t);
hipError_t grdma_launch_engine(grdma_engine_mbox*, grdma_watch_ctl*, uint64_t epoch, uint32_t groups, uint32_t flags, hipStream_t, hipStream_t);
const void* grdma_kernel_fn(int which);          // 0 tx_plan, 1 copy, 3 rx_apply, 4 tx_plan_seq
const void* grdma_kernel_fn_rx_plan(void);
const void* grdma_kernel_fn_rx_plan_job(void);
hipError_t grdma_launch_rx_plan_job(const grdma_rx_op*, uint32_t, hipStream_t);
uint32_t grdma_rx_plan_job_threads(void);
uint32_t grdma_tx_plan_job_threads(void);
const void* grdma_kernel_fn_tx_index(void);
uint32_t grdma_tx_index_threads(void);
hipError_t grdma_launch_tx_index(grdma_txf_ctl*, uint32_t, uint32_t, hipStream_t);
hipError_t grdma_launch_tx_plan_job(const grdma_tx_op*, const grdma_txf_ctl*, uint32_t, hipStream_t);
const void* grdma_kernel_fn_plan_pair(void);
const void* grdma_kernel_fn_plan_pair_job(void);
const void* grdma_kernel_fn_plan_pair_mw(void);
uint32_t grdma_rx_multi_groups(void);
hipError_t grdma_launch_rx_plan_mw(const grdma_rx_op*, uint32_t, hipStream_t);
uint32_t grdma_tx_multi_groups(void);
uint32_t grdma_tx_multi_max_sends(void);
uint32_t grdma_tx_multi_seq_sends(void);
const void* grdma_kernel_fn_round_xag(void);
uint64_t grdma_rx_scratch_bytes(void);
uint32_t grdma_round_xag_resident_blocks(void);
const void* grdma_kernel_fn_rxplan_gather_job(void);
uint32_t grdma_kernel_threads(int which);
uint32_t grdma_copy_resident_blocks(void);
}

namespace {

struct grdma_ctx {
  std::mutex mu;
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  // pinned scratch for k_poll
  grdma_conn** h_conns = nullptr;
  uint64_t* h_readable = nullptr;
  uint64_t* h_masks = nullptr;
  uint32_t poll_cap = 0;
};

grdma_ctx g_ctx;
thread_local std::string g_err;

// persistent latency engine (k_engine): mailbox in pinned host memory
struct grdma_engine {
  std::mutex mu;         // one command in the mailbox at a time: callers on different threads queue here
  grdma_engine_mbox* mb = nullptr;
  hipStream_t stream = nullptr;
  bool wanted = false;   // grdma_engine_start() was called
  uint64_t seq = 0;
  uint64_t posted = 0;   // last command written into the mailbox (== seq while one may still be running)
  // the read side (k_watch): watcher workgroups resident beside the command workgroup, and the slots they serve
  hipStream_t wstream = nullptr;
  grdma_watch_ctl* d_watch = nullptr;   // device memory
  grdma_watch_cmd* h_wcmd = nullptr;    // pinned: the GRDMA_ENGINE_WATCH command in flight
  uint32_t groups = 0;                  // watcher workgroups per incarnation
  uint64_t epoch = 0;                   // engine incarnations launched
  uint64_t gen = 0;                     // armings handed out
  grdma_pair* slot_owner[GRDMA_WATCH_SLOTS] = {};
  bool slot_posted[GRDMA_WATCH_SLOTS] = {};   // the device slot holds the owner's standing order
  bool watch_dirty = false;             // some owner's order has yet to be posted
};
grdma_engine g_engine;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return -code;
}

#define HIP_TRY(expr)                                                               \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess)                                                           \
      return fail(GRDMA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                              \
  } while (0)

int require_ctx() {
  if (!g_ctx.ready) return fail(GRDMA_ERR_NO_DEVICE, "grdma_init() has not succeeded: no HIP device");
  return 0;
}

// Host-visible state lines (grdma_hostline): pinned coherent memory, carved from slabs of 512 lines so that a
// connection does not cost a hipHostMalloc of its own.
struct line_slab {
  std::mutex mu;
  std::vector<grdma_hostline*> blocks;
  std::vector<grdma_hostline*> free_list;
};
line_slab g_lines;
static_assert(sizeof(grdma_hostline) == 128, "one state line = two cache lines");

grdma_hostline* line_alloc() {
  std::lock_guard<std::mutex> lk(g_lines.mu);
  if (g_lines.free_list.empty()) {
    grdma_hostline* blk = nullptr;
    if (hipHostMalloc((void**)&blk, sizeof(grdma_hostline) * 512, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
      return nullptr;
    g_lines.blocks.push_back(blk);
    for (int i = 511; i >= 0; i--) g_lines.free_list.push_back(blk + i);
  }
  grdma_hostline* l = g_lines.free_list.back();
  g_lines.free_list.pop_back();
  memset(l, 0, sizeof(*l));
  return l;
}
void line_free(grdma_hostline* l) {
  if (!l) return;
  std::lock_guard<std::mutex> lk(g_lines.mu);
  g_lines.free_list.push_back(l);
}

uint32_t copy_blocks_for(uint64_t bytes) {
  uint64_t tiles = (bytes + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
  uint64_t blocks = (tiles + 3) / 4;  // 4 waves per block, one tile per wave pass
  if (blocks < 1) blocks = 1;
  static const uint64_t cap_blocks = [] {
    const char* e = getenv("GRDMA_COPY_BLOCKS");  // tuning knob, tools/ experiments
    const long v = e ? atol(e) : 0;
    return (uint64_t)(v > 0 ? v : grdma_copy_resident_blocks());  // the rest is grid-strided
  }();
  if (blocks > cap_blocks) blocks = cap_blocks;
  return (uint32_t)blocks;
}

}  // namespace

// Host control block shared with the device (pinned, device-visible).
struct grdma_hostblk {
  grdma_tx_op txop;
  grdma_rx_op rxop;
  grdma_tx_result txres;
  grdma_rx_result rxres;
  const grdma_plan* plan_ptrs[4];  // [0] tx gather, [1] wire, [2] rx scatter
  grdma_zc_op zcop;                // SendZerocopy
  // refresh pass of a pair whose peer lives in another process (k_poll over this one connection)
  grdma_conn* refresh_conn;
  uint64_t refresh_out[4];         // readable, ready mask, has-message mask, trigger mask
  // how far the sender's completed writes reached when the drain in flight was submitted (the state line's wire_tail):
  // ONE reading for all the workgroups that lay the drain out -- the sender goes on committing while they run, and two
  // workgroups that read the connection's arrival report for themselves could see two different tails
  uint64_t rx_limit;
};

// One receive window of an asynchronous endpoint: the slices of one drain, in pinned host memory the scatter kernel
// writes over PCIe.  Reference-counted: the pair holds one reference, every slice handed to the transport another
// (it may outlive the endpoint).
struct grdma_window {
  uint8_t* base = nullptr;
  uint64_t bytes = 0;
  std::atomic<int> refs{1};
};

struct grdma_pair {
  uint64_t ring_size = 0;
  int max_sge = 0;
  int flags = 0;
  grdma_conn* d_conn = nullptr;
  uint8_t* d_ring = nullptr;
  uint8_t* d_staging = nullptr;
  grdma_plan* d_txplan = nullptr;
  grdma_plan* d_wireplan = nullptr;
  grdma_plan* d_rxplan = nullptr;
  uint8_t* d_arena = nullptr;
  uint64_t arena_cap = 0;
  uint32_t* d_hist = nullptr;
  bool latency = false;              // fused single-launch kernels + spin on pinned seq words
  bool cmd_inline = false;           // the current slice list lives in h_cmd (small host message)
  uint8_t* h_arena = nullptr;        // pinned receive arena used in latency mode
  grdma_engine_cmd* h_cmd = nullptr; // pinned self-contained command block (latency engine): sends
  grdma_engine_cmd* h_cmd_rx = nullptr;  // the same for this pair's own drains (the read side may run on another thread)
  uint64_t h_arena_cap = 0;
  grdma_hostblk* h = nullptr;        // pinned
  grdma_hostline* line = nullptr;    // pinned: what the read-only queries load (grdma_hostline)
  std::atomic<uint32_t> status{GRDMA_PAIR_UNINITIALIZED};  // status_, pair.h:166 (the device copy gates the kernels)
  uint64_t tx_seq = 0;               // sends committed so far (k_tx_commit publishes it in line->tx_seq)
  uint64_t refresh_launched = 0;     // refresh passes launched (remote peers only)
  hipStream_t refresh_stream = nullptr;
  uint64_t peer_pid = 0;             // remote peer: its process id (liveness check of get_status)
  int boot_fd = -1;                  // remote peer: the bootstrap socket (not owned), for the hang-up check
  std::atomic<int64_t> liveness_checked_ns{0};
  grdma_sge* h_sges = nullptr;       // pinned, GRDMA_MAX_SEGS entries
  grdma_slice_out* h_slices = nullptr;  // pinned, GRDMA_MAX_SLICES entries
  uint8_t* h_bounce = nullptr;       // pinned, staging-sized, lazily allocated
  grdma_pair* peer = nullptr;
  // a peer in another process (or on another GPU): its ring and its connection block are
  // mapped here through HIP IPC handles exchanged at bootstrap (grdma_pair_connect_remote)
  bool remote = false;
  void* ipc_ring = nullptr;             // base of the peer's ring mapping
  void* ipc_conn = nullptr;             // base of the peer's grdma_conn mapping
  grdma_status_report* remote_status = nullptr;  // peer's status_recv inside ipc_conn
  uint32_t serial = 0;                  // "queue pair number" of this pair in its process
  bool ipc_export_failed = false;       // hipIpcGetMemHandle refused: only peers in this process can connect
  bool bounce_truncated = false;        // stage_slices copied less than the slices hold (the bounce buffer was full)
  hipStream_t stream = nullptr;
  int wakeup_fd = -1;                // grpc_wakeup_fd of the pair (pair.h:187): an eventfd
  std::mutex fd_mu;                  // creation of wakeup_fd
  // zero-copy send buffer (send_buffers_[kZeroCopyBuffer], pair.h:96,178,195)
  uint8_t* d_zc = nullptr;
  uint64_t zc_cap = 0;
  uint32_t zc_tail = 0;              // zerocopy_buffer_tail_ (std::atomic_uint32_t)
  uint64_t zc_bytes = 0, zc_copy_bytes = 0, zc_last_sges = 0;
  std::mutex zc_mu;
  // armed read (grdma_pair_arm_read): the LOCAL peer's small sends carry this pair's drain in the same
  // engine command; the next grdma_endpoint_read picks the completion up instead of asking for one
  uint64_t armed_reads = 0;          // max_reads of the armed drain, 0 = not armed
  std::atomic<bool> armed_done{false};  // a chained drain has completed and nobody has consumed it yet (set by the
                                     // sender's thread, consumed by the receiver's)
  uint64_t armed_hits = 0;
  // ... or, by default, a watcher workgroup of the engine carries the standing order out when bytes land in this
  // pair's ring, whoever wrote them (k_watch): the slot it was handed, the sequence word of the next completion,
  // completions taken since the arming (what grdma_engine_mbox::consumed tells the device), completions taken in all
  grdma_verbs_wire* verbs = nullptr; // != NULL: a NIC writes the peer's ring and mine (csrc/grdma_wire_verbs.cc)
  int watch_slot = -1;
  uint64_t watch_expect = 0, watch_taken = 0, watch_hits = 0;
  uint64_t armed_half = 0;           // which half of the arena the completion `armed_done` stands for lies in (a
                                     // watcher's completion that was left behind when its order was taken back)
  bool watch_parked = false;         // asynchronous endpoint: the last completion was taken while the transport held
                                     // every other window -- the watcher waits for the word that names the next one
  // asynchronous endpoint operations (grdma_endpoint_set_async): the send and the receive direction on streams of
  // their own, one Send and one drain in flight at most, completions in pinned host memory
  bool async = false;
  hipStream_t s_tx = nullptr, s_rx = nullptr;
  std::vector<grdma_window*> windows;   // receive windows (pinned host memory), one drain each
  std::atomic<int> rx_inflight{-1};     // window of the drain in flight, -1 = none
  std::atomic<uint64_t> rx_expect{0};   // the sequence number that says it has completed: line->rx_seq (launch chain,
                                        // written by k_rx_commit1 behind the scatter) or rxres.seq (engine command)
  std::atomic<int> rx_by_engine{0};
  uint64_t rx_seq = 0;                  // drains committed through k_rx_commit1
  std::atomic<int> tx_inflight{0};
  uint64_t tx_expect = 0;               // line->tx_seq (launch chain) or txres.seq (engine command) of the Send in flight
  bool tx_by_engine = false;
  std::mutex rx_mu;                     // receive-side submission state: the armed read lets the PEER's sender post my drain
  std::atomic<uint64_t> armed_async{0}; // standing order of an asynchronous endpoint: max_reads, 0 = none
  uint64_t test_calls = 0, test_calls_rx = 0;
  // several Sends per submit (a write of more slices than max_sge, wire direct): rdma_flush retried back to back on
  // the device from the cursor, every Send with a gather plan of its own, ONE planning launch and ONE gather launch
  std::vector<grdma_plan*> b_plans;
  grdma_tx_op* h_bops = nullptr;              // pinned, kBurstMax ops
  grdma_tx_result* h_bres = nullptr;          // pinned, kBurstMax results
  const grdma_plan** h_bplan_ptrs = nullptr;  // pinned
  uint32_t tx_burst = 0;                      // Sends of the submit in flight (0 = a single Send through h->txop)
  uint64_t tx_burst_byte0 = 0;                // byte offset the first slice of that submit was entered at
  // A write QUEUED behind the burst in flight (grdma_endpoint_write_queue): its chain is already in the send stream,
  // reading a second set of pinned ops / results / slice table, and its first Send is gated on the device -- it runs
  // only if the last Send of the write in front took everything (grdma_tx_op::use_cursor 3); otherwise the whole
  // chain is skipped and the host continues the write in front first.  Set 0 is {h_bops, h_bres, h_bplan_ptrs, h_sges}.
  grdma_tx_op* q_ops = nullptr;
  grdma_tx_result* q_res = nullptr;
  const grdma_plan** q_plan_ptrs = nullptr;
  grdma_sge* q_sges = nullptr;
  int inflight_set = 0;                       // which set the burst in flight reads
  bool inflight_covers_all = false;           // ... and whether it offers every remaining slice of its write
  int q_state = 0;                            // 1 = a write is queued
  int q_set = 0;
  uint32_t q_burst = 0;
  uint64_t q_expect = 0;                      // tx_seq of the queued chain's commit
  uint64_t q_pending_seq = 0;                 // a skipped chain still in the stream commits this sequence number
  std::vector<grdma_slice> q_slices;
  uint64_t q_queued = 0, q_promoted = 0, q_skipped = 0;  // counters (grdma_endpoint_write_queue_stats)
  // endpoint_write context
  std::vector<grdma_slice> w_slices;
  uint64_t w_idx = 0, w_byte = 0;
  int w_flags = 0;
  bool w_active = false;
  int pool_keep_open = 0;
};

namespace {

int fetch_conn(grdma_pair* p, grdma_conn* out) {
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(out, p->d_conn, sizeof(grdma_conn), hipMemcpyDeviceToHost));
  return 0;
}

// Registration cache for large host slices (grdma_set_host_register_min): page ranges registered with the device
// once, looked up by address afterwards.  Small, MRU-ordered, shared by all pairs of the process.
struct reg_entry { uintptr_t lo, hi; uint8_t* dev; };
struct reg_cache {
  std::mutex mu;
  std::vector<reg_entry> e;
  std::atomic<uint64_t> min_bytes{0};
  bool env_read = false;
};
reg_cache g_reg;

uint64_t register_min() {
  if (!g_reg.env_read) {
    std::lock_guard<std::mutex> lk(g_reg.mu);
    if (!g_reg.env_read) {
      if (const char* v = getenv("GRPC_RDMA_HIP_REGISTER_MIN")) g_reg.min_bytes.store(strtoull(v, nullptr, 10));
      g_reg.env_read = true;
    }
  }
  return g_reg.min_bytes.load(std::memory_order_relaxed);
}

// device-visible address of [ptr, ptr + len), registering the pages if need be; nullptr = copy instead
const uint8_t* registered_view(const void* ptr, uint64_t len) {
  const uintptr_t a = (uintptr_t)ptr, lo = a & ~(uintptr_t)4095, hi = (a + len + 4095) & ~(uintptr_t)4095;
  std::lock_guard<std::mutex> lk(g_reg.mu);
  for (size_t i = 0; i < g_reg.e.size(); i++) {
    if (g_reg.e[i].lo <= a && a + len <= g_reg.e[i].hi) {
      const reg_entry hit = g_reg.e[i];
      if (i) {  // move to front
        g_reg.e.erase(g_reg.e.begin() + (long)i);
        g_reg.e.insert(g_reg.e.begin(), hit);
      }
      return hit.dev + (a - hit.lo);
    }
  }
  // (a range that overlaps a registered one cannot be registered again: such a slice is copied)
  for (const reg_entry& r : g_reg.e)
    if (lo < r.hi && r.lo < hi) return nullptr;
  if (hipHostRegister((void*)lo, hi - lo, hipHostRegisterMapped) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, (void*)lo, 0) != hipSuccess || !dev) {
    hipHostUnregister((void*)lo);
    return nullptr;
  }
  if (g_reg.e.size() >= 256) {  // forget the least recently used range
    hipHostUnregister((void*)g_reg.e.back().lo);
    g_reg.e.pop_back();
  }
  g_reg.e.insert(g_reg.e.begin(), reg_entry{lo, hi, static_cast<uint8_t*>(dev)});
  return static_cast<uint8_t*>(dev) + (a - lo);
}

// Fill the device-visible slice table.  For GRDMA_MEM_HOST the first
// staging_cap bytes of the list are copied into the pinned bounce buffer (a Send
// can never consume more than the staging budget, pair.cc:676-685).
int stage_slices(grdma_pair* p, const grdma_slice* slices, uint64_t count, uint64_t skip_first,
                 int flags) {
  if (count > GRDMA_TX_MAX_RECORDS - 1)
    return fail(GRDMA_ERR_CAPACITY, "slice list of %llu entries exceeds %d",
                (unsigned long long)count, GRDMA_TX_MAX_RECORDS - 1);
  p->cmd_inline = false;
  p->bounce_truncated = false;
  if ((flags & GRDMA_MEM_HOST) && p->latency && p->h_cmd && count <= GRDMA_CMD_MAX_SGES) {
    uint64_t total = 0;
    for (uint64_t i = 0; i < count; i++) total += slices[i].len;
    if (total <= GRDMA_CMD_INLINE_BYTES) {
      uint64_t off = 0;
      for (uint64_t i = 0; i < count; i++) {
        if (slices[i].len) memcpy(p->h_cmd->inline_data + off, slices[i].ptr, slices[i].len);
        p->h_cmd->sges[i].ptr = reinterpret_cast<const uint8_t*>(off);  // rebased by the engine
        p->h_cmd->sges[i].len = slices[i].len;
        // keep the ordinary table valid too (byte_idx handling reads lengths from it)
        p->h_sges[i].ptr = p->h_cmd->inline_data + off;
        p->h_sges[i].len = slices[i].len;
        off += slices[i].len;
      }
      p->cmd_inline = true;
      return 0;
    }
  }
  if (flags & GRDMA_MEM_HOST) {
    // (a Send takes at most ring / 2; a burst of Sends of an asynchronous endpoint at most what the ring holds)
    const uint64_t cap = p->async ? p->ring_size : p->ring_size / 2;
    if (!p->h_bounce) HIP_TRY(hipHostMalloc((void**)&p->h_bounce, cap + 64, hipHostMallocCoherent | hipHostMallocMapped));
    const uint64_t reg_min = register_min();
    uint64_t off = 0;
    for (uint64_t i = 0; i < count; i++) {
      const uint8_t* src = static_cast<const uint8_t*>(slices[i].ptr);
      uint64_t len = slices[i].len;
      if (reg_min && len >= reg_min) {  // read where it lies: the gather kernel pulls it over PCIe
        if (const uint8_t* dv = registered_view(src, len)) {
          p->h_sges[i].ptr = dv;
          p->h_sges[i].len = len;
          continue;
        }
      }
      uint64_t sk = (i == 0) ? skip_first : 0;  // bytes before byte_idx are never read
      uint64_t room = cap > off ? cap - off : 0;
      uint64_t n = len > sk ? len - sk : 0;
      if (n > room) {
        n = room;
        p->bounce_truncated = true;  // (the Sends of this submit must not accept more than what was staged)
      }
      if (n) memcpy(p->h_bounce + off, src + sk, n);
      p->h_sges[i].ptr = p->h_bounce + off - sk;  // so that ptr + byte_idx lands on the copy
      p->h_sges[i].len = len;
      off += n;
    }
  } else {
    for (uint64_t i = 0; i < count; i++) {
      p->h_sges[i].ptr = static_cast<const uint8_t*>(slices[i].ptr);
      p->h_sges[i].len = slices[i].len;
    }
  }
  return 0;
}

// GRDMA_PROFILE_TICKS=1: the latency paths take their phase stamps (grdma_tx_small_ticks, grdma_rx_express_ticks,
// grdma_watch_ticks, grdma_engine_debug); read once per process.
inline bool profile_ticks() {
  static const bool on = [] { const char* e = getenv("GRDMA_PROFILE_TICKS"); return e && atoi(e) != 0; }();
  return on;
}
inline uint32_t latency_op_bits() { return 1u | (profile_ticks() ? 16u : 0u); }

// Watcher workgroups per engine incarnation (GRDMA_ENGINE_WATCHERS, 1 .. 8; slot s is served by workgroup s % n).
uint32_t engine_watch_groups() {
#ifdef GRDMA_WAVE_EMU
  return 1;  // (the emulator runs the workgroups of a launch one after the other: a resident one never ends)
#else
  static const uint32_t n = [] {
    const char* e = getenv("GRDMA_ENGINE_WATCHERS");
    long v = e ? atol(e) : 4;
    if (v < 1) v = 1;
    if (v > GRDMA_WATCH_MAX_GROUPS) v = GRDMA_WATCH_MAX_GROUPS;
    return (uint32_t)v;
  }();
  return n;
#endif
}

int engine_launch() {
  grdma_engine& e = g_engine;
  if (!e.mb) {
    HIP_TRY(hipHostMalloc((void**)&e.mb, sizeof(grdma_engine_mbox), hipHostMallocCoherent | hipHostMallocMapped));
    memset(e.mb, 0, sizeof(*e.mb));
    HIP_TRY(hipHostMalloc((void**)&e.h_wcmd, sizeof(grdma_watch_cmd), hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&e.wstream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&e.d_watch, sizeof(grdma_watch_ctl)));
    HIP_TRY(hipMemsetAsync(e.d_watch, 0, sizeof(grdma_watch_ctl), e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
    e.groups = engine_watch_groups();
  }
  volatile uint64_t* alive = &e.mb->alive;
  if (*alive) return 0;
  // the incarnation before has left (or is leaving: its watchers follow the command workgroup out)
  HIP_TRY(hipStreamSynchronize(e.stream));
  HIP_TRY(hipStreamSynchronize(e.wstream));
  e.mb->exit_flag = 0;
  const uint64_t epoch = ++e.epoch;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  // (GRDMA_WATCH_FAST=0: every drain of a watcher through the plan body -- the A/B of tests/test_zzz_gpu_watch_read.py;
  //  GRDMA_PROFILE_TICKS=1: the phase stamps of the latency paths, off by default -- see prof_time in grdma_devfn.h)
  const char* wf = getenv("GRDMA_WATCH_FAST");
  const uint32_t kflags = ((wf && atoi(wf) == 0) ? 0u : 1u) | (profile_ticks() ? 2u : 0u);
  HIP_TRY(grdma_launch_engine(e.mb, e.d_watch, epoch, e.groups, kflags, e.stream, e.wstream));
  const auto t0 = std::chrono::steady_clock::now();
  auto up = [&] {
    if (!*alive) return false;
    for (uint32_t w = 0; w < e.groups; w++)
      if (*(volatile uint64_t*)&e.mb->watch_alive[w] != epoch) return false;
    return true;
  };
  while (!up()) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5))
      return fail(GRDMA_ERR_HIP, "latency engine did not come up (command workgroup %s, watchers %llu %llu %llu %llu of incarnation %llu)",
                  *alive ? "resident" : "not resident", (unsigned long long)e.mb->watch_alive[0], (unsigned long long)e.mb->watch_alive[1],
                  (unsigned long long)e.mb->watch_alive[2], (unsigned long long)e.mb->watch_alive[3], (unsigned long long)epoch);
  }
  return 0;
}

// Pack a small command for the mailbox's fast lane (see grdma_engine_mbox); 0 = does not fit.
size_t pack_fast(uint64_t type, const grdma_engine_cmd* blk, uint64_t* words) {
  size_t nsges = 0, dbytes = 0, nw = 1;
  if (type == GRDMA_ENGINE_SEND_INLINE || type == GRDMA_ENGINE_SEND_INLINE_DRAIN) {
    const size_t rxw = type == GRDMA_ENGINE_SEND_INLINE_DRAIN ? sizeof(grdma_rx_op) / 8 : 0;
    nsges = (size_t)blk->tx.nslices;
    if (nsges > GRDMA_CMD_MAX_SGES) return 0;
    for (size_t i = 0; i < nsges; i++) dbytes += (size_t)blk->sges[i].len;
    const size_t total = 1 + sizeof(grdma_tx_op) / 8 + 2 * nsges + (dbytes + 7) / 8 + rxw;
    if (total > GRDMA_FAST_WORDS || dbytes > GRDMA_CMD_INLINE_BYTES) return 0;
    memcpy(words + nw, &blk->tx, sizeof(grdma_tx_op));
    nw += sizeof(grdma_tx_op) / 8;
    for (size_t i = 0; i < nsges; i++) {
      words[nw++] = (uint64_t)blk->sges[i].ptr;  // offset into the data
      words[nw++] = blk->sges[i].len;
    }
    if (dbytes) {
      words[nw + (dbytes - 1) / 8] = 0;
      memcpy(words + nw, blk->inline_data, dbytes);
      nw += (dbytes + 7) / 8;
    }
    if (rxw) {
      memcpy(words + nw, &blk->rx, sizeof(grdma_rx_op));
      nw += rxw;
    }
  } else if (type == GRDMA_ENGINE_DRAIN_BLOCK) {
    memcpy(words + nw, &blk->rx, sizeof(grdma_rx_op));
    nw += sizeof(grdma_rx_op) / 8;
  } else {
    return 0;
  }
  words[0] = type | ((uint64_t)nsges << 8) | ((uint64_t)dbytes << 16);
  return nw;
}
static_assert(sizeof(grdma_tx_op) % 8 == 0 && sizeof(grdma_rx_op) % 8 == 0, "ops are packed as 8-byte words");

// Wait until the engine has acknowledged command `seq` (engine.mu held).
int engine_wait_locked(uint64_t seq) {
  grdma_engine& e = g_engine;
  volatile uint64_t* ack = &e.mb->ack_seq;
  volatile uint64_t* alive = &e.mb->alive;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0;; spins++) {
    if (*ack == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return 0;
    }
    if ((spins & 0x3FF) == 0x3FF) {
      if (!*alive && *ack != seq) {  // the engine timed out just before the doorbell
        if (int rc = engine_launch()) return rc;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3)) {
        *(volatile uint64_t*)&e.mb->exit_flag = 1;
        const uint64_t a = *ack, polls = e.mb->pad1[0];
        // Every incarnation of the engine resumes behind ack_seq and only takes the command numbered ack_seq + 1:
        // once it has left, forget the command that was never answered, or every later one would time out too.
        if (hipStreamSynchronize(e.stream) == hipSuccess) {
          *(volatile uint64_t*)&e.mb->ack_seq = e.seq;
          e.posted = e.seq;
        }
        return fail(GRDMA_ERR_HIP, "latency engine did not answer: alive=%llu ack=%llu seq=%llu polls=%llu",
                    (unsigned long long)*alive, (unsigned long long)a, (unsigned long long)seq, (unsigned long long)polls);
      }
    }
  }
}

// GRDMA_ENGINE_CUT_THROUGH=0 (tests: the A/B of tests/test_zzz_gpu_armed_read.py): an armed send + drain command runs its
// two bodies the way separate commands do -- records through the ring, a probe, three releases -- instead of handing
// the sizes over in LDS and cutting unary-sized records through (k_engine).  The switch travels in the command: a
// sizes_out of 1 is "do not chain" (the engine clears it).  Read per command: a test flips it inside one process.
inline grdma_size_hint* engine_chain_marker() {
  const char* e = getenv("GRDMA_ENGINE_CUT_THROUGH");
  return (e && atoi(e) == 0) ? reinterpret_cast<grdma_size_hint*>(1) : nullptr;
}

// GRDMA_ENGINE_CHAIN=1 (the A/B of tests/test_zzz_gpu_armed_read.py): an armed read is carried by the in-process
// peer's send command (GRDMA_ENGINE_SEND_INLINE_DRAIN, round 4) instead of by a watcher workgroup.  Read per call.
inline bool engine_chain_mode() {
  const char* e = getenv("GRDMA_ENGINE_CHAIN");
  return e && atoi(e) != 0;
}

int engine_post_locked(uint64_t type, const void* op);
void fill_rxop(grdma_pair* p, uint8_t* arena, uint64_t arena_cap, uint64_t max_reads, uint64_t raw_cap);

// The standing order of `p` as its watcher runs it: grdma_endpoint_read(max_reads) into the pinned arena, the sequence
// word continuing where the pair's result block stands.  engine.mu held; nothing of this pair's drains is in flight.
// Asynchronous endpoint, standing order posted, the last completion taken: name the window the next drain delivers
// into -- the word the watcher is waiting for -- once the transport has let one go.  p->rx_mu held.
void watch_unpark(grdma_pair* p) {
  if (!p->watch_parked || p->watch_slot < 0 || p->watch_expect == 0) return;
  int w = -1;
  for (size_t i = 0; i < p->windows.size() && i < GRDMA_WATCH_WINDOWS; i++)
    if (p->windows[i]->refs.load(std::memory_order_acquire) == 1) { w = (int)i; break; }
  if (w < 0) return;
  p->rx_expect.store(p->watch_expect, std::memory_order_relaxed);
  p->rx_by_engine.store(1, std::memory_order_relaxed);
  p->rx_inflight.store(w, std::memory_order_release);
  p->watch_parked = false;
  __atomic_store_n(&g_engine.mb->consumed[p->watch_slot], p->watch_taken | ((uint64_t)w << 56), __ATOMIC_RELEASE);
}

// false: not now (an asynchronous endpoint whose windows are all held by the transport, or with a drain of its own
// still in flight) -- the order stays pending and is tried again with the next command / arming.
bool watch_fill_cmd(grdma_pair* p, uint32_t sidx, grdma_watch_cmd* cmd) {
  memset(cmd->win_base, 0, sizeof(cmd->win_base));
  int w0 = 0;
  std::unique_lock<std::mutex> rxl(p->rx_mu, std::defer_lock);
  if (p->async) {
    // the standing order is "a drain in flight" to the endpoint (grdma_endpoint_drain_state / _read_test): into the
    // window named with it, completing by itself when bytes land
    if (!rxl.try_lock()) return false;   // (the endpoint's reading thread is submitting or taking a drain right now)
    if (p->rx_inflight.load(std::memory_order_acquire) >= 0) return false;
    w0 = -1;
    for (size_t i = 0; i < p->windows.size() && i < GRDMA_WATCH_WINDOWS; i++) {
      cmd->win_base[i] = p->windows[i]->base;
      if (w0 < 0 && p->windows[i]->refs.load(std::memory_order_acquire) == 1) w0 = (int)i;
    }
    if (w0 < 0) return false;
    fill_rxop(p, p->windows[w0]->base, p->windows[w0]->bytes, p->armed_reads, 0);
  } else {
    // (a blocking pair: the two halves of its pinned arena take turns, so that the slices of the completion just
    //  taken stay where they are while the next drain delivers)
    cmd->win_base[0] = p->h_arena;
    cmd->win_base[1] = p->h_arena + p->h_arena_cap;
    fill_rxop(p, p->h_arena, p->h_arena_cap, p->armed_reads, 0);
  }
  cmd->slot = sidx;
  cmd->gen = ++g_engine.gen;
  cmd->op = p->h->rxop;
  cmd->op.inline_apply = latency_op_bits() | 8u;  // (8: a watcher's drain -- rx_plan_body reads the ring past the caches)
  cmd->consumed0 = (uint64_t)w0 << 56;
  p->watch_expect = cmd->op.seq_next;
  p->watch_taken = 0;
  p->watch_parked = false;
  *(volatile uint64_t*)&g_engine.mb->consumed[sidx] = cmd->consumed0;
  if (p->async) {
    p->rx_expect.store(cmd->op.seq_next, std::memory_order_relaxed);
    p->rx_by_engine.store(1, std::memory_order_relaxed);
    p->rx_inflight.store(w0, std::memory_order_release);
  }
  return true;
}

// Standing orders that have not reached their device slot yet (armed before the engine was started, or cleared
// by grdma_engine_stop): one GRDMA_ENGINE_WATCH command each.  engine.mu held, engine resident.
int watch_flush_locked() {
  grdma_engine& e = g_engine;
  if (!e.watch_dirty) return 0;
  e.watch_dirty = false;
  for (uint32_t sidx = 0; sidx < GRDMA_WATCH_SLOTS; sidx++) {
    grdma_pair* p = e.slot_owner[sidx];
    if (!p || e.slot_posted[sidx]) continue;
    if (!watch_fill_cmd(p, sidx, e.h_wcmd)) {
      e.watch_dirty = true;   // (not now: tried again with the next command)
      continue;
    }
    if (int rc = engine_post_locked(GRDMA_ENGINE_WATCH, e.h_wcmd)) return rc;
    if (int rc = engine_wait_locked(e.posted)) return rc;
    e.slot_posted[sidx] = true;
  }
  return 0;
}

// Hand one command to the resident engine WITHOUT waiting for it (the mailbox holds one command: the one posted
// before must have been acknowledged, which this waits for).  The caller finds the completion in its result block.
int engine_post(uint64_t type, const void* op) {
  grdma_engine& e = g_engine;
  std::lock_guard<std::mutex> lk(e.mu);
  if (int rc = engine_launch()) return rc;
  if (int rc = watch_flush_locked()) return rc;
  return engine_post_locked(type, op);
}

int engine_post_locked(uint64_t type, const void* op) {
  grdma_engine& e = g_engine;
  if (e.posted != 0 && e.posted == e.seq) {
    if (int rc = engine_wait_locked(e.posted)) return rc;
  }
  uint64_t words[GRDMA_FAST_WORDS];
  const size_t nw = (type == GRDMA_ENGINE_SEND_INLINE || type == GRDMA_ENGINE_DRAIN_BLOCK ||
                     type == GRDMA_ENGINE_SEND_INLINE_DRAIN)
                        ? pack_fast(type, static_cast<const grdma_engine_cmd*>(op), words) : 0;
  const uint64_t seq = ++e.seq;
  if (nw) {
    volatile uint64_t* f = e.mb->fast;
    const size_t lines = (nw + 6) / 7;
    for (size_t j = lines; j-- > 0;) {  // line 0 -- the one that announces the command -- last
      for (size_t k = 0; k < 7 && 7 * j + k < nw; k++) f[8 * j + k] = words[7 * j + k];
      std::atomic_thread_fence(std::memory_order_release);
      f[8 * j + 7] = seq;
    }
  } else {
    e.mb->cmd_type = type;
    e.mb->op = op;
    std::atomic_thread_fence(std::memory_order_release);
    *(volatile uint64_t*)&e.mb->cmd_seq = seq;
  }
  e.posted = seq;
  return 0;
}

// Hand one command to the resident engine and wait for it.
int engine_submit(uint64_t type, const void* op) {
  if (int rc = engine_post(type, op)) return rc;
  grdma_engine& e = g_engine;
  std::lock_guard<std::mutex> lk(e.mu);
  return engine_wait_locked(e.posted);
}

int engine_stop() {
  grdma_engine& e = g_engine;
  std::lock_guard<std::mutex> lk(e.mu);
  e.wanted = false;
  if (!e.mb) return 0;
  *(volatile uint64_t*)&e.mb->exit_flag = 1;
  HIP_TRY(hipStreamSynchronize(e.stream));
  HIP_TRY(hipStreamSynchronize(e.wstream));
  // Nothing is resident now: a pair whose read is armed drains through launches of its own until the engine is
  // started again, so the device slots are emptied and the standing orders posted afresh (with the sequence
  // numbers of that moment) by the first command of the next start.
  bool any = false;
  for (uint32_t sidx = 0; sidx < GRDMA_WATCH_SLOTS; sidx++) {
    grdma_pair* p = e.slot_owner[sidx];
    if (e.slot_posted[sidx]) {
      any = true;
      // a completion the watcher produced and nobody has taken stays for the next grdma_endpoint_read
      if (p && p->watch_expect != 0 && __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) >= p->watch_expect) {
        p->armed_half = p->watch_taken & 1;
        p->armed_done = true;
      }
    }
    if (p && p->async && e.slot_posted[sidx] && p->rx_inflight.load() >= 0 &&
        __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) < p->rx_expect.load())
      p->rx_inflight.store(-1);   // (the standing order leaves with the engine; a completion that is there stays)
    if (p && p->async) p->armed_done = false;
    if (p) p->watch_expect = 0;
    e.slot_posted[sidx] = false;
    if (p) e.watch_dirty = true;
  }
  if (any) {
    HIP_TRY(hipMemsetAsync(e.d_watch, 0, sizeof(grdma_watch_ctl), e.stream));
    HIP_TRY(hipStreamSynchronize(e.stream));
  }
  return 0;
}

// ---- arrival-triggered reads: the host side of k_watch (grdma_watch_slot, grdma_ops.h) ------------------------
// (engine.mu held)
int watch_slot_of(const grdma_pair* p) {
  for (uint32_t sidx = 0; sidx < GRDMA_WATCH_SLOTS; sidx++)
    if (g_engine.slot_owner[sidx] == p) return (int)sidx;
  return -1;
}

// Take the standing order of `p` back from its watcher (slot freed).  engine.mu held.
int watch_release_locked(grdma_pair* p) {
  grdma_engine& e = g_engine;
  const int sidx = watch_slot_of(p);
  if (sidx < 0) return 0;
  int rc = 0;
  if (e.slot_posted[sidx]) {
    // (a posted slot means an incarnation that has not been stopped: resident, or retired by itself -- the command
    //  brings it back, and its command workgroup waits until the watcher has let the connection go)
    e.h_wcmd->slot = (uint64_t)sidx;
    e.h_wcmd->gen = 0;
    rc = engine_launch();
    if (!rc) rc = engine_post_locked(GRDMA_ENGINE_WATCH, e.h_wcmd);
    if (!rc) rc = engine_wait_locked(e.posted);
    e.slot_posted[sidx] = false;
  }
  e.slot_owner[sidx] = nullptr;
  return rc;
}

// Wait for a plan kernel by watching its sequence word in pinned host memory
// (the kernel bumps it last, with a system-scope release); falls back to a
// stream synchronize so that a failed launch is still reported.
int wait_seq(grdma_pair* p, volatile uint64_t* seq, uint64_t old) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0;; spins++) {
    if (*seq != old) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return 0;
    }
    if ((spins & 0xFFF) == 0xFFF &&
        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
      break;
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  return *seq != old ? 0 : fail(GRDMA_ERR_HIP, "plan kernel did not complete");
}

void fill_rxop(grdma_pair* p, uint8_t* arena, uint64_t arena_cap, uint64_t max_reads, uint64_t raw_cap);

int run_send(grdma_pair* p, uint64_t count, uint64_t byte_idx, uint32_t use_cursor) {
  grdma_profiler profiler(GRDMA_STATS_TIME_PAIR_SEND);  // pair.cc:647
  grdma_hostblk* h = p->h;
  h->txop.conn = p->d_conn;
  h->txop.slices = p->h_sges;
  h->txop.nslices = count;
  h->txop.byte_idx = byte_idx;
  h->txop.plan = p->d_txplan;
  h->txop.wire_plan = p->d_wireplan;
  h->txop.result = &h->txres;
  h->txop.use_cursor = use_cursor;
  h->txop.inline_copy = p->latency ? latency_op_bits() : 0;
  h->txop.seq_next = p->latency ? h->txres.seq + 1 : 0;
  const uint32_t blocks = copy_blocks_for(p->ring_size / 2);
  if (p->latency) {
    if (g_engine.wanted && p->h_cmd && p->cmd_inline) {
      // the slice table and payload were staged into the command block (stage_slices)
      p->h_cmd->tx = h->txop;
      grdma_pair* q = p->peer;
      if (q && !p->remote && q->armed_reads && q->watch_slot < 0 && !q->armed_done && q->latency && q->h_arena) {
        // the peer has a read armed: its drain rides in this command (the engine runs it right after
        // the send, in the same workgroup), and its completion waits in the peer's result block
        fill_rxop(q, q->h_arena, q->h_arena_cap, q->armed_reads, 0);
        p->h_cmd->rx = q->h->rxop;
        p->h_cmd->tx.sizes_out = engine_chain_marker();
        if (int rc = engine_submit(GRDMA_ENGINE_SEND_INLINE_DRAIN, p->h_cmd)) return rc;
        q->armed_done = true;
        q->armed_hits++;
        return 0;
      }
      return engine_submit(GRDMA_ENGINE_SEND_INLINE, p->h_cmd);
    }
    if (g_engine.wanted) return engine_submit(GRDMA_ENGINE_SEND, &h->txop);
    const uint64_t old = h->txres.seq;
    HIP_TRY(grdma_launch_tx_plan(&h->txop, 1, p->stream));
    return wait_seq(p, &h->txres.seq, old);
  }
  HIP_TRY(grdma_launch_tx_plan(&h->txop, 1, p->stream));
  HIP_TRY(grdma_launch_copy(&h->plan_ptrs[0], 1, blocks, p->stream));
  if (p->verbs) {
    // the wire is a NIC: the records lie encoded in the staging buffer, the planner has left the Send's <= 2 write
    // requests in the result block (K2) -- posted as chained RDMA WRITEs and reaped (pair.cc:709-734, waitDataWrites)
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::string err;
    if (grdma_verbs_post_data(p->verbs, h->txres.wr_off, h->txres.wr_len, h->txres.wr_count, &err) != 0)
      return fail(GRDMA_ERR_HIP, "NIC wire: %s", err.c_str());
  } else if (!(p->flags & GRDMA_WIRE_DIRECT)) {
    HIP_TRY(grdma_launch_copy(&h->plan_ptrs[1], 1, blocks, p->stream));
  }
  // behind the wire write, as a kernel of its own: the arrival report (and the state lines)
  HIP_TRY(grdma_launch_tx_commit1(p->d_conn, ++p->tx_seq, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return 0;
}

int run_recv(grdma_pair* p, uint8_t* arena, uint64_t arena_cap, uint64_t max_reads,
             uint64_t raw_cap) {
  grdma_profiler profiler(GRDMA_STATS_TIME_PAIR_RECV);  // pair.cc:265
  // a drain that rode behind the peer's send has delivered into the result block and the slice table: those bytes have
  // left the ring, their credit is out -- another drain now would overwrite the only record of them
  if (p->armed_done.load(std::memory_order_acquire))
    return fail(GRDMA_ERR_INVALID, "an armed read has completed: grdma_endpoint_read takes it before anything else drains");
  if (p->watch_slot >= 0 && g_engine.wanted)
    return fail(GRDMA_ERR_INVALID, "the read side of this pair belongs to its watcher (grdma_pair_arm_read): grdma_endpoint_read takes its completions");
  grdma_hostblk* h = p->h;
  fill_rxop(p, arena, arena_cap, max_reads, raw_cap);
  const uint32_t blocks = copy_blocks_for(p->ring_size);
  if (p->latency) {
    if (g_engine.wanted && p->h_cmd) {
      p->h_cmd->rx = h->rxop;
      return engine_submit(GRDMA_ENGINE_DRAIN_BLOCK, p->h_cmd);
    }
    if (g_engine.wanted) return engine_submit(GRDMA_ENGINE_DRAIN, &h->rxop);
    const uint64_t old = h->rxres.seq;
    HIP_TRY(grdma_launch_rx_plan(&h->rxop, 1, p->stream));
    return wait_seq(p, &h->rxres.seq, old);
  }
  HIP_TRY(grdma_launch_rx_plan(&h->rxop, 1, p->stream));
  HIP_TRY(grdma_launch_rx_apply(&h->rxop, 1, blocks, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->verbs && h->rxres.credit_sent) {
    // updateStatus() (pair.cc:624-641): the drain's planner has left remote_head in status_send; the copy-out and the
    // zero-fill of the bytes it grants are done (the synchronize above) -- the 16-byte report goes out as an RDMA WRITE
    std::string err;
    if (grdma_verbs_post_status(p->verbs, &err) != 0) return fail(GRDMA_ERR_HIP, "NIC wire: %s", err.c_str());
  }
  return 0;
}

void fill_rxop(grdma_pair* p, uint8_t* arena, uint64_t arena_cap, uint64_t max_reads, uint64_t raw_cap) {
  grdma_hostblk* h = p->h;
  h->rxop.conn = p->d_conn;
  h->rxop.plan = p->d_rxplan;
  h->rxop.result = &h->rxres;
  h->rxop.slices = p->h_slices;
  h->rxop.arena = arena;
  h->rxop.arena_cap = arena_cap;
  h->rxop.max_reads = max_reads;
  h->rxop.raw_cap = raw_cap;
  h->rxop.append = 0;
  h->rxop.slices_cap = GRDMA_MAX_SLICES;
  h->rxop.inline_apply = p->latency ? latency_op_bits() : 0;
  h->rxop.seq_next = p->latency ? h->rxres.seq + 1 : 0;
  h->rxop.limit_ptr = nullptr;
  h->rxop.sizes_in = nullptr;
}

}  // namespace

extern "C" {

int grdma_abi_version(void) { return GRDMA_ABI_VERSION; }

const char* grdma_last_error(void) { return g_err.c_str(); }

int grdma_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int grdma_init(int hip_device) {
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(GRDMA_ERR_NO_DEVICE, "no HIP device visible (%s); this data plane has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (hip_device < 0 || hip_device >= n)
    return fail(GRDMA_ERR_INVALID, "hip_device %d out of range [0,%d)", hip_device, n);
  if (g_ctx.ready && g_ctx.device == hip_device) return 0;
  HIP_TRY(hipSetDevice(hip_device));
  if (!g_ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
  g_ctx.device = hip_device;
  g_ctx.ready = true;
  return 0;
}

// ---- PairPool (src/core/lib/ibverbs/pair.h:273-333) --------------------------------------------------------
// The reference keeps 128 pre-built PairPollable objects in a queue and maps connection ids to the pairs handed
// out (Take(id) / Get(id) / Putback).  What is expensive to build here is not the object but its memory: ring,
// staging buffer, three plans, arena, pinned blocks -- about a dozen hipMalloc / hipHostMalloc calls per
// connection.  So the pool keeps the BLOCKS: grdma_pair_destroy hands them back by {size, kind}, grdma_pair_create
// takes them from there (and zeroes what Init() zeroes, as it always does); the id map sits on top.
namespace {
struct block_pool {
  std::mutex mu;
  std::multimap<std::pair<size_t, int>, void*> free_blocks;  // kind: 0 device, 1 device fine-grained, 2 pinned host
  size_t cached_bytes[3] = {0, 0, 0};
  size_t cap_bytes = 0;  // 0 = pooling off: blocks go back to the runtime
  uint64_t hits = 0, misses = 0;
  std::shared_timed_mutex id_mu;
  std::unordered_map<std::string, grdma_pair*> id_pair;
  std::unordered_map<grdma_pair*, std::string> pair_id;
};
block_pool g_pool;

hipError_t pool_alloc(void** ptr, size_t n, int kind) {
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    auto it = g_pool.free_blocks.find({n, kind});
    if (it != g_pool.free_blocks.end()) {
      *ptr = it->second;
      g_pool.free_blocks.erase(it);
      g_pool.cached_bytes[kind] -= n;
      g_pool.hits++;
      return hipSuccess;
    }
    if (g_pool.cap_bytes) g_pool.misses++;
  }
  if (kind == 2) return hipHostMalloc(ptr, n, hipHostMallocCoherent | hipHostMallocMapped);
  return kind == 1 ? hipExtMallocWithFlags(ptr, n, hipDeviceMallocFinegrained) : hipMalloc(ptr, n);
}
void pool_free(void* ptr, size_t n, int kind) {
  if (!ptr) return;
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    const size_t total = g_pool.cached_bytes[0] + g_pool.cached_bytes[1] + g_pool.cached_bytes[2];
    if (g_pool.cap_bytes && total + n <= g_pool.cap_bytes) {
      g_pool.free_blocks.insert({{n, kind}, ptr});
      g_pool.cached_bytes[kind] += n;
      return;
    }
  }
  if (kind == 2) hipHostFree(ptr); else hipFree(ptr);
}
}  // namespace

namespace {
// pairs that have exported an address, by serial ("queue pair number"): a peer that turns out to live in THIS process
// (client and server of one test binary: what gRPC's own end2end tests are) is looked up here instead of being
// mapped through an IPC handle, which cannot be opened where it was made
std::mutex g_exported_mu;
std::unordered_map<uint32_t, grdma_pair*> g_exported;
}  // namespace

grdma_pair* grdma_pair_create(uint64_t ring_size, int max_sge, int flags) {
  if (require_ctx()) return nullptr;
  // ring_buffer.cc:22-24
  if (ring_size <= GRDMA_RESERVED || (ring_size & (ring_size - 1)) != 0 || ring_size < 64) {
    fail(GRDMA_ERR_INVALID, "ring size %llu is not a power of two >= 64",
         (unsigned long long)ring_size);
    return nullptr;
  }
  if (max_sge <= 0) max_sge = 30;
  if (max_sge > GRDMA_TX_MAX_RECORDS - 1) max_sge = GRDMA_TX_MAX_RECORDS - 1;
  grdma_pair* p = new grdma_pair();
  p->ring_size = ring_size;
  p->max_sge = max_sge;
  p->flags = flags;
  p->stream = g_ctx.stream;
  p->arena_cap = 2 * ring_size + 4096;
  // the two things a REMOTE writer touches: the ring and the connection block (status report)
  const bool fine = (flags & GRDMA_RING_FINE_GRAINED) != 0;
  const int rk = fine ? 1 : 0;  // (blocks come from the pool when it holds one of the size, see block_pool)
  bool ok = pool_alloc((void**)&p->d_conn, sizeof(grdma_conn), rk) == hipSuccess &&
            pool_alloc((void**)&p->d_ring, ring_size, rk) == hipSuccess &&
            pool_alloc((void**)&p->d_staging, ring_size / 2 + 64, 0) == hipSuccess &&
            pool_alloc((void**)&p->d_txplan, sizeof(grdma_plan), 0) == hipSuccess &&
            pool_alloc((void**)&p->d_wireplan, sizeof(grdma_plan), 0) == hipSuccess &&
            pool_alloc((void**)&p->d_rxplan, sizeof(grdma_plan), 0) == hipSuccess &&
            pool_alloc((void**)&p->d_arena, p->arena_cap, 0) == hipSuccess &&
            pool_alloc((void**)&p->d_hist, sizeof(uint32_t) * GRDMA_RX_HIST, 0) == hipSuccess &&
            pool_alloc((void**)&p->h, sizeof(grdma_hostblk), 2) == hipSuccess &&
            pool_alloc((void**)&p->h_sges, sizeof(grdma_sge) * GRDMA_TX_MAX_RECORDS, 2) == hipSuccess &&
            pool_alloc((void**)&p->h_slices, sizeof(grdma_slice_out) * GRDMA_MAX_SLICES, 2) == hipSuccess;
  if (ok) ok = (p->line = line_alloc()) != nullptr;
  if (!ok) {
    fail(GRDMA_ERR_HIP, "device allocation failed for a %llu-byte ring",
         (unsigned long long)ring_size);
    grdma_pair_destroy(p);
    return nullptr;
  }
  // Init(): zero the ring (ring_buffer.cc:49-54, pair.cc:117-118) and the state
  hipMemsetAsync(p->d_ring, 0, ring_size, p->stream);
  hipMemsetAsync(p->d_staging, 0, ring_size / 2 + 64, p->stream);
  hipMemsetAsync(p->d_txplan, 0, sizeof(grdma_plan), p->stream);
  hipMemsetAsync(p->d_wireplan, 0, sizeof(grdma_plan), p->stream);
  hipMemsetAsync(p->d_rxplan, 0, sizeof(grdma_plan), p->stream);
  hipMemsetAsync(p->d_hist, 0, sizeof(uint32_t) * GRDMA_RX_HIST, p->stream);
  memset(p->h, 0, sizeof(grdma_hostblk));
  p->h->plan_ptrs[0] = p->d_txplan;
  p->h->plan_ptrs[1] = p->d_wireplan;
  p->h->plan_ptrs[2] = p->d_rxplan;
  grdma_conn c;
  memset(&c, 0, sizeof(c));
  c.ring = p->d_ring;
  c.cap = ring_size;
  c.staging = p->d_staging;
  c.staging_cap = ring_size / 2;  // pair.cc:104
  c.max_sge = (uint32_t)max_sge;
  c.status = GRDMA_PAIR_INITIALIZED;
  c.wire_direct = (flags & GRDMA_WIRE_DIRECT) ? 1 : 0;
  c.rx_hist = p->d_hist;
  c.line = p->line;
  // every wire of this build is a parallel copy: the receiver honours the arrival report -- unless the
  // caller says an ordered writer (a NIC) fills this ring
  c.wire_limit = (flags & GRDMA_WIRE_ORDERED) ? 0 : 1;
  p->h->refresh_conn = p->d_conn;
  p->status.store(GRDMA_PAIR_INITIALIZED);
  hipMemcpyAsync(p->d_conn, &c, sizeof(c), hipMemcpyHostToDevice, p->stream);
  if (hipStreamSynchronize(p->stream) != hipSuccess) {
    fail(GRDMA_ERR_HIP, "pair initialisation failed");
    grdma_pair_destroy(p);
    return nullptr;
  }
  return p;
}

void grdma_pair_destroy(grdma_pair* p) {
  if (!p) return;
  if (p->serial != 0) {
    std::lock_guard<std::mutex> lk(g_exported_mu);
    auto it = g_exported.find(p->serial);
    if (it != g_exported.end() && it->second == p) g_exported.erase(it);
  }
  if (p->verbs) {
    grdma_verbs_close(p->verbs);
    p->verbs = nullptr;
  }
  if (p->watch_slot >= 0) {  // (its watcher lets the connection go before the memory does)
    std::lock_guard<std::mutex> lk(g_engine.mu);
    watch_release_locked(p);
    p->watch_slot = -1;
  }
  if (p->stream) hipStreamSynchronize(p->stream);
  // (a queued or skipped write chain, a drain in flight: nothing of this pair's may still run when its memory goes back)
  if (p->s_tx) hipStreamSynchronize(p->s_tx);
  if (p->s_rx) hipStreamSynchronize(p->s_rx);
  if (p->ipc_ring) hipIpcCloseMemHandle(p->ipc_ring);
  if (p->ipc_conn) hipIpcCloseMemHandle(p->ipc_conn);
  {
    const int rk = (p->flags & GRDMA_RING_FINE_GRAINED) ? 1 : 0;
    pool_free(p->d_conn, sizeof(grdma_conn), rk);
    pool_free(p->d_ring, p->ring_size, rk);
    pool_free(p->d_staging, p->ring_size / 2 + 64, 0);
    pool_free(p->d_txplan, sizeof(grdma_plan), 0);
    pool_free(p->d_wireplan, sizeof(grdma_plan), 0);
    pool_free(p->d_rxplan, sizeof(grdma_plan), 0);
    pool_free(p->d_arena, p->arena_cap, 0);
    pool_free(p->d_hist, sizeof(uint32_t) * GRDMA_RX_HIST, 0);
  }
  hipFree(p->d_zc);
  if (p->s_tx) { hipStreamSynchronize(p->s_tx); hipStreamDestroy(p->s_tx); }
  if (p->s_rx) { hipStreamSynchronize(p->s_rx); hipStreamDestroy(p->s_rx); }
  for (grdma_window* w : p->windows) grdma_window_unref(w);  // (slices the transport still holds keep theirs)
  for (grdma_plan* pl : p->b_plans) hipFree(pl);
  if (p->h_bops) hipHostFree(p->h_bops);
  if (p->h_bres) hipHostFree(p->h_bres);
  if (p->h_bplan_ptrs) hipHostFree(p->h_bplan_ptrs);
  if (p->q_ops) hipHostFree(p->q_ops);
  if (p->q_res) hipHostFree(p->q_res);
  if (p->q_plan_ptrs) hipHostFree(p->q_plan_ptrs);
  if (p->q_sges) hipHostFree(p->q_sges);
  if (p->refresh_stream) {
    hipStreamSynchronize(p->refresh_stream);  // a refresh pass in flight writes the line
    hipStreamDestroy(p->refresh_stream);
  }
  pool_free(p->h, sizeof(grdma_hostblk), 2);
  line_free(p->line);
  pool_free(p->h_sges, sizeof(grdma_sge) * GRDMA_TX_MAX_RECORDS, 2);
  pool_free(p->h_slices, sizeof(grdma_slice_out) * GRDMA_MAX_SLICES, 2);
  if (p->h_bounce) hipHostFree(p->h_bounce);
  if (p->wakeup_fd >= 0) close(p->wakeup_fd);
  if (p->h_arena) hipHostFree(p->h_arena);
  if (p->h_cmd) hipHostFree(p->h_cmd);
  if (p->peer && p->peer->peer == p) p->peer->peer = nullptr;
  {  // (a pair destroyed without Putback leaves the id map too)
    std::unique_lock<std::shared_timed_mutex> lk(g_pool.id_mu);
    auto it = g_pool.pair_id.find(p);
    if (it != g_pool.pair_id.end()) {
      g_pool.id_pair.erase(it->second);
      g_pool.pair_id.erase(it);
    }
  }
  delete p;
}

// PairPool::createPairs (pair.h:323-327): memory for `pairs` connections of this shape is set aside now, so that
// the Takes that follow do not call into the allocator.  cap_bytes = how much the pool may hold (0: what this
// call sets aside).
int grdma_pair_pool_reserve(uint32_t pairs, uint64_t ring_size, int max_sge, int flags, uint64_t cap_bytes) {
  if (int rc = require_ctx()) return rc;
  const uint64_t per_pair = ring_size + ring_size / 2 + 64 + 3 * sizeof(grdma_plan) + 2 * ring_size + 4096 + (1 << 20);
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    const uint64_t want = cap_bytes ? cap_bytes : (uint64_t)pairs * per_pair;
    if (want > g_pool.cap_bytes) g_pool.cap_bytes = want;
  }
  std::vector<grdma_pair*> made;
  for (uint32_t i = 0; i < pairs; i++) {
    grdma_pair* p = grdma_pair_create(ring_size, max_sge, flags);
    if (!p) break;
    made.push_back(p);
  }
  const bool all = made.size() == pairs;
  for (grdma_pair* p : made) grdma_pair_destroy(p);  // (their blocks stay in the pool)
  return all ? 0 : fail(GRDMA_ERR_HIP, "pair pool: only %zu of %u pairs could be set aside", made.size(), pairs);
}

// Take(id): a pair of this shape, built from pooled blocks when there are any, registered under `id`.
grdma_pair* grdma_pair_pool_take(const char* id, uint64_t ring_size, int max_sge, int flags) {
  if (!id) {
    fail(GRDMA_ERR_INVALID, "pair pool: null id");
    return nullptr;
  }
  grdma_pair* p = grdma_pair_create(ring_size, max_sge, flags);
  if (!p) return nullptr;
  std::unique_lock<std::shared_timed_mutex> lk(g_pool.id_mu);
  g_pool.id_pair[id] = p;
  g_pool.pair_id[p] = id;
  return p;
}

// Get(id): the pair a connection id was handed (what the zero-copy hook of the reference looks up,
// src/core/lib/surface/call.cc:663-672); NULL when the id is unknown.
grdma_pair* grdma_pair_pool_get(const char* id) {
  if (!id) return nullptr;
  std::shared_lock<std::shared_timed_mutex> lk(g_pool.id_mu);
  auto it = g_pool.id_pair.find(id);
  return it == g_pool.id_pair.end() ? nullptr : it->second;
}

// Putback(pair): the id is forgotten, the pair's blocks go back to the pool (rdma_bp_posix.cc:128,780).
void grdma_pair_pool_putback(grdma_pair* p) {
  if (!p) return;
  if (p->pool_keep_open == 0 && p->status.load() == GRDMA_PAIR_CONNECTED) grdma_pair_disconnect(p);
  grdma_pair_destroy(p);
}

// {blocks held, bytes held, allocations served from the pool, allocations that went to the runtime, ids registered}
int grdma_pair_pool_stats(uint64_t out[5]) {
  if (!out) return fail(GRDMA_ERR_INVALID, "null argument");
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    out[0] = g_pool.free_blocks.size();
    out[1] = g_pool.cached_bytes[0] + g_pool.cached_bytes[1] + g_pool.cached_bytes[2];
    out[2] = g_pool.hits;
    out[3] = g_pool.misses;
  }
  std::shared_lock<std::shared_timed_mutex> lk(g_pool.id_mu);
  out[4] = g_pool.id_pair.size();
  return 0;
}

// Returns every pooled block to the runtime (process shutdown, tests).
void grdma_pair_pool_trim(void) {
  std::vector<std::pair<void*, int>> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto& kv : g_pool.free_blocks) blocks.push_back({kv.second, kv.first.second});
    g_pool.free_blocks.clear();
    g_pool.cached_bytes[0] = g_pool.cached_bytes[1] = g_pool.cached_bytes[2] = 0;
    g_pool.cap_bytes = 0;
  }
  for (auto& b : blocks) {
    if (b.second == 2) hipHostFree(b.first); else hipFree(b.first);
  }
}

int grdma_pair_connect(grdma_pair* a, grdma_pair* b) {
  if (int rc = require_ctx()) return rc;
  if (!a || !b) return fail(GRDMA_ERR_INVALID, "null pair");
  if (a->ring_size != b->ring_size)  // pair.cc:149
    return fail(GRDMA_ERR_INVALID, "ring sizes differ (%llu vs %llu)",
                (unsigned long long)a->ring_size, (unsigned long long)b->ring_size);
  grdma_pair* ends[2] = {a, b};
  for (int i = 0; i < 2; i++) {
    grdma_pair* me = ends[i];
    grdma_pair* other = ends[1 - i];
    grdma_conn c;
    if (int rc = fetch_conn(me, &c)) return rc;
    c.peer_ring = other->d_ring;
    c.peer_status = reinterpret_cast<grdma_status_report*>(
        reinterpret_cast<uint8_t*>(other->d_conn) + offsetof(grdma_conn, status_recv));
    c.peer_wire = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(other->d_conn) +
                                              offsetof(grdma_conn, wire_recv) + offsetof(grdma_wire_report, wire_tail));
    c.peer_line = other->line;  // both ends in this process: the kernels push into each other's state line
    c.line_remote = 0;
    c.peer_limited = (other->flags & GRDMA_WIRE_ORDERED) ? 0 : 1;
    c.status = GRDMA_PAIR_CONNECTED;
    HIP_TRY(hipMemcpy(me->d_conn, &c, sizeof(c), hipMemcpyHostToDevice));
    me->status.store(GRDMA_PAIR_CONNECTED);
    me->peer = other;
    me->stream = a->stream;  // one in-order queue per loop-back link
  }
  return 0;
}

// ---- bootstrap across processes / GPUs ------------------------------------------------------
// The reference exchanges a 48-byte Address over the TCP fd (exchange_data,
// rdma_bp_posix.cc:640-692, 767-771), brings the queue pair up (pair.cc:143-168, 413-457) and then
// swaps memory-region descriptors {addr, rkey} of the ring and of the status buffer over the
// new QP (syncMemoryRegion).  Here the "memory region" of a ring in HBM is a HIP IPC handle: the
// peer maps it and its one-sided writes (the wire kernel, or a NIC given the dma-buf of the same
// allocation) land in my ring.  Both travel in one blob whose first 48 bytes are the reference's
// Address, so the tag / ring-size checks of Connect() read the same fields.
namespace {
std::atomic<uint32_t> g_pair_serial{1};
const uint32_t kBlobMagic = 0x4d445247u;  // "GRDM"
static_assert(sizeof(grdma_address) == 48, "Address layout (address.h:24-31)");
static_assert(sizeof(grdma_bootstrap_blob) == 208, "bootstrap blob layout");
}  // namespace

int grdma_pair_export_address(grdma_pair* p, grdma_bootstrap_blob* out) {
  if (int rc = require_ctx()) return rc;
  if (!p || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  memset(out, 0, sizeof(*out));
  if (p->serial == 0) p->serial = g_pair_serial.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(g_exported_mu);
    g_exported[p->serial] = p;
  }
  out->addr.lid = (uint32_t)g_ctx.device;          // "local id": the HIP device ordinal
  out->addr.qpn = p->serial;
  out->addr.psn = (uint32_t)(((uint64_t)getpid() * 2654435761u) ^ p->serial) & 0xffffffu;  // 24 bits, like lrand48() & 0xffffff
  char bus[32] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), g_ctx.device) == hipSuccess) memcpy(out->addr.gid, bus, 16);
  out->addr.tag = 0xa0;                            // IBVERBS_PAIR_TAG_POLLABLE, pair.h:26, pair.cc:72
  out->addr.ring_buffer_size = p->ring_size;       // pair.cc:107
  out->magic = kBlobMagic;
  out->version = GRDMA_ABI_VERSION;
  out->hip_device = g_ctx.device;
  out->pid = (uint64_t)getpid();
  out->status_off = offsetof(grdma_conn, status_recv);
  out->wire_off = (uint32_t)offsetof(grdma_conn, wire_recv);
  // What another process (or a NIC) writes must be fine-grained device memory: its stores are then at
  // memory once acknowledged, and mine (the reader's zero-fill, the credit word) are write-through --
  // with a coarse-grained ring the zero-fill of one XCD's L2 could be written back OVER records the peer
  // has placed since (k_rx_apply counts its workgroups in with a relaxed atomic on that assumption).
  if (!(p->flags & GRDMA_RING_FINE_GRAINED))
    return fail(GRDMA_ERR_INVALID, "a pair exported to a remote peer must be created with GRDMA_RING_FINE_GRAINED");
  static_assert(sizeof(hipIpcMemHandle_t) == sizeof(out->ring_handle), "HIP IPC handle size");
  // (a peer in this process never opens them; a process whose runtime cannot export memory can still serve those)
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p->d_ring) == hipSuccess) memcpy(out->ring_handle, &h, sizeof(h));
  else p->ipc_export_failed = true;
  if (hipIpcGetMemHandle(&h, p->d_conn) == hipSuccess) memcpy(out->conn_handle, &h, sizeof(h));
  else p->ipc_export_failed = true;
  if (p->ipc_export_failed) {
    (void)hipGetLastError();
    memset(out->ring_handle, 0, sizeof(out->ring_handle));
    memset(out->conn_handle, 0, sizeof(out->conn_handle));
  }
  return 0;
}

int grdma_pair_connect_remote(grdma_pair* p, const grdma_bootstrap_blob* peer) {
  if (int rc = require_ctx()) return rc;
  if (!p || !peer) return fail(GRDMA_ERR_INVALID, "null argument");
  grdma_conn c;
  if (int rc = fetch_conn(p, &c)) return rc;
  if (c.status != GRDMA_PAIR_INITIALIZED)  // Connect() only acts on kInitialized, pair.cc:144
    return fail(GRDMA_ERR_INVALID, "pair is not in the initialized state (status %u)", c.status);
  // GPR_ASSERT(peer_.addr_.tag == self_.addr_.tag), pair.cc:146
  if (peer->addr.tag != 0xa0) return fail(GRDMA_ERR_INVALID, "peer address tag 0x%x, expected 0xa0", peer->addr.tag);
  // GPR_ASSERT(peer_.addr_.ring_buffer_size == self_.addr_.ring_buffer_size), pair.cc:147-149
  if (peer->addr.ring_buffer_size != p->ring_size)
    return fail(GRDMA_ERR_INVALID, "ring sizes differ (mine %llu, peer %llu)", (unsigned long long)p->ring_size,
                (unsigned long long)peer->addr.ring_buffer_size);
  if (peer->magic != kBlobMagic || peer->version != GRDMA_ABI_VERSION)
    return fail(GRDMA_ERR_INVALID, "peer is not a HIP data-plane endpoint of this ABI version");
  // the offsets come off a socket: they must be the ones of this build's connection block
  if (peer->status_off != offsetof(grdma_conn, status_recv) || peer->wire_off != offsetof(grdma_conn, wire_recv))
    return fail(GRDMA_ERR_INVALID, "peer's connection block layout differs (status %llu, wire %u)",
                (unsigned long long)peer->status_off, peer->wire_off);
  if (!(p->flags & GRDMA_RING_FINE_GRAINED))
    return fail(GRDMA_ERR_INVALID, "a pair connected to a remote peer must be created with GRDMA_RING_FINE_GRAINED");
  if (peer->pid == (uint64_t)getpid()) {
    // The peer lives in this process (an IPC handle cannot be opened where it was made): its half of
    // grdma_pair_connect.  Both ends come through here, each for itself, possibly at the same time on two threads;
    // the loop-back link's in-order queue is the stream of the end with the smaller serial on both sides.
    grdma_pair* other = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_exported_mu);
      auto it = g_exported.find(peer->addr.qpn);
      if (it != g_exported.end()) other = it->second;
    }
    if (other == nullptr || other == p)
      return fail(GRDMA_ERR_INVALID, "peer address names pair %u of this process, which does not exist", peer->addr.qpn);
    c.peer_ring = other->d_ring;
    c.peer_status = reinterpret_cast<grdma_status_report*>(reinterpret_cast<uint8_t*>(other->d_conn) + offsetof(grdma_conn, status_recv));
    c.peer_wire = reinterpret_cast<uint64_t*>(reinterpret_cast<uint8_t*>(other->d_conn) + offsetof(grdma_conn, wire_recv) +
                                              offsetof(grdma_wire_report, wire_tail));
    c.peer_line = other->line;
    c.line_remote = 0;
    c.peer_limited = (other->flags & GRDMA_WIRE_ORDERED) ? 0 : 1;
    c.status = GRDMA_PAIR_CONNECTED;
    HIP_TRY(hipMemcpy(p->d_conn, &c, sizeof(c), hipMemcpyHostToDevice));
    p->peer = other;
    if (other->serial != 0 && other->serial < p->serial) p->stream = other->stream;
    p->status.store(GRDMA_PAIR_CONNECTED);
    return 0;
  }
  if (peer->ring_handle[0] == 0 && memcmp(peer->ring_handle, peer->ring_handle + 1, sizeof(peer->ring_handle) - 1) == 0)
    return fail(GRDMA_ERR_INVALID, "the peer could not export its ring (no IPC handle in its address)");
  hipIpcMemHandle_t h;
  void* ring = nullptr;
  void* conn = nullptr;
  memcpy(&h, peer->ring_handle, sizeof(h));
  HIP_TRY(hipIpcOpenMemHandle(&ring, h, hipIpcMemLazyEnablePeerAccess));
  memcpy(&h, peer->conn_handle, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(&conn, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    hipIpcCloseMemHandle(ring);
    return fail(GRDMA_ERR_HIP, "hipIpcOpenMemHandle(peer connection block) failed: %s", hipGetErrorString(e));
  }
  p->ipc_ring = ring;
  p->ipc_conn = conn;
  p->remote = true;
  p->remote_status = reinterpret_cast<grdma_status_report*>(static_cast<uint8_t*>(conn) + peer->status_off);
  c.peer_ring = static_cast<uint8_t*>(ring);
  c.peer_status = p->remote_status;
  c.peer_wire = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(conn) + peer->wire_off + offsetof(grdma_wire_report, wire_tail));
  c.peer_line = nullptr;  // pinned host memory of another process is out of reach:
  c.line_remote = 1;      // my line is refreshed from my own connection block (k_poll's refresh pass)
  c.peer_limited = 0;     // (the address blob does not say how the peer reads: footers last)
  c.status = GRDMA_PAIR_CONNECTED;
  HIP_TRY(hipMemcpy(p->d_conn, &c, sizeof(c), hipMemcpyHostToDevice));
  p->peer_pid = peer->pid;
  p->status.store(GRDMA_PAIR_CONNECTED);
  return 0;
}

// exchange_data (rdma_bp_posix.cc:640-692): full-duplex swap of sz bytes over a connected socket
static int exchange_blob(int fd, const char* buf_in, char* buf_out, size_t sz) {
  size_t sent = 0, got = 0;
  if (fd < 0) return -1;
  struct pollfd pfd = {fd, 0, 0};
  while (got < sz || sent < sz) {
    pfd.events = (short)((got < sz ? POLLIN : 0) | (sent < sz ? POLLOUT : 0));
    const int r = poll(&pfd, 1, 30000);
    if (r == 0) return -2;  // the peer never answered
    if (r < 0) {
      if (errno == EINTR) continue;
      return -1;
    }
    if (sent < sz && (pfd.revents & POLLOUT)) {
      const ssize_t n = ::send(fd, buf_in + sent, sz - sent, MSG_NOSIGNAL);
      if (n < 0 && errno != EINTR && errno != EAGAIN) return -1;
      if (n > 0) sent += (size_t)n;
    }
    if (got < sz && (pfd.revents & (POLLIN | POLLHUP))) {
      const ssize_t n = ::recv(fd, buf_out + got, sz - got, 0);
      if (n == 0) return -3;  // closed before the whole address arrived
      if (n < 0 && errno != EINTR && errno != EAGAIN) return -1;
      if (n > 0) got += (size_t)n;
    }
  }
  return 0;
}

int grdma_pair_bootstrap_fd(grdma_pair* p, int fd) {
  grdma_bootstrap_blob mine, theirs;
  if (int rc = grdma_pair_export_address(p, &mine)) return rc;
  const int x = exchange_blob(fd, reinterpret_cast<const char*>(&mine), reinterpret_cast<char*>(&theirs), sizeof(mine));
  if (x != 0)
    return fail(GRDMA_ERR_NOT_CONNECTED, "address exchange over fd %d failed (%s)", fd,
                x == -2 ? "timeout" : x == -3 ? "peer closed" : strerror(errno));
  if (int rc = grdma_pair_connect_remote(p, &theirs)) return rc;
  p->boot_fd = fd;  // (not owned) a hang-up on it means the peer is gone: get_status()
  return 0;
}

int grdma_pair_disconnect(grdma_pair* p) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  HIP_TRY(hipStreamSynchronize(p->stream));
  const uint32_t st0 = p->status.load();
  if (st0 == GRDMA_PAIR_CONNECTED && (p->peer || p->remote_status)) {
    // peer_exit = 1 in the peer's status buffer, pair.cc:332-336
    int32_t one = 1;
    uint8_t* dst = p->peer ? reinterpret_cast<uint8_t*>(p->peer->d_conn) + offsetof(grdma_conn, status_recv)
                           : reinterpret_cast<uint8_t*>(p->remote_status);
    HIP_TRY(hipMemcpy(dst + offsetof(grdma_status_report, peer_exit), &one, sizeof(one), hipMemcpyHostToDevice));
    if (p->peer && p->peer->line)  // an in-process peer sees it in its state line at once
      __atomic_store_n(&p->peer->line->peer_exit, 1, __ATOMIC_RELEASE);
  }
  if (st0 == GRDMA_PAIR_CONNECTED && p->verbs) {
    // Disconnect() over a NIC wire (pair.cc:332-336): peer_exit = 1 in my status_send, and the report goes out
    int32_t one = 1;
    HIP_TRY(hipMemcpy(reinterpret_cast<uint8_t*>(p->d_conn) + offsetof(grdma_conn, status_send) + offsetof(grdma_status_report, peer_exit),
                      &one, sizeof(one), hipMemcpyHostToDevice));
    std::string err;
    (void)grdma_verbs_post_status(p->verbs, &err);  // (a peer that is gone already: nothing to tell)
  }
  uint32_t st = GRDMA_PAIR_DISCONNECTED;
  HIP_TRY(hipMemcpy(reinterpret_cast<uint8_t*>(p->d_conn) + offsetof(grdma_conn, status), &st,
                    sizeof(st), hipMemcpyHostToDevice));
  p->status.store(st);
  return 0;
}

namespace {
int64_t mono_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// A peer in another process writes into my connection block, not into my host line: k_poll's refresh pass
// copies arrival report, credit report and peer_exit over.  At most one pass is in flight per pair, launched
// without waiting from whichever read-only query finds the previous one done -- the queries themselves stay
// plain host loads (they see the line as of the last pass).
void refresh_async(grdma_pair* p) {
  if (!(p->remote || (p->flags & GRDMA_WIRE_ORDERED)) || !p->line) return;
  const uint64_t done = __atomic_load_n(&p->line->refresh_seq, __ATOMIC_ACQUIRE);
  if (done != p->refresh_launched) return;  // the last pass has not written the line yet
  std::lock_guard<std::mutex> lk(p->fd_mu);
  if (__atomic_load_n(&p->line->refresh_seq, __ATOMIC_ACQUIRE) != p->refresh_launched) return;
  if (!p->refresh_stream && hipStreamCreateWithFlags(&p->refresh_stream, hipStreamNonBlocking) != hipSuccess) return;
  grdma_hostblk* h = p->h;
  if (grdma_launch_poll(&h->refresh_conn, 1, &h->refresh_out[0], &h->refresh_out[1], &h->refresh_out[2],
                        &h->refresh_out[3], p->refresh_stream) == hipSuccess)
    p->refresh_launched++;
}

// get_status() (pair.cc:349-375).  The reference asks the queue pair every 500 ms whether it is still in a
// working state (ibv_query_qp, :358-372) and reports kHalfClosed when it is not; here the peer of a remote pair
// is a process: it is gone when its pid is, or when the bootstrap socket reports a hang-up.
uint32_t status_now(grdma_pair* p) {
  uint32_t st = p->status.load(std::memory_order_acquire);
  if (st != GRDMA_PAIR_CONNECTED) return st;
  bool closed = p->line && __atomic_load_n(&p->line->peer_exit, __ATOMIC_ACQUIRE) == 1;
  if (!closed && p->remote) {
    refresh_async(p);
    const int64_t now = mono_ns(), last = p->liveness_checked_ns.load(std::memory_order_relaxed);
    if (now - last > 100 * 1000 * 1000 &&
        p->liveness_checked_ns.compare_exchange_strong(const_cast<int64_t&>(last), now)) {
      if (p->peer_pid && kill((pid_t)p->peer_pid, 0) != 0 && errno == ESRCH) closed = true;
      if (!closed && p->boot_fd >= 0) {
        struct pollfd pfd = {p->boot_fd, POLLRDHUP, 0};
        if (poll(&pfd, 1, 0) > 0 && (pfd.revents & (POLLHUP | POLLRDHUP | POLLERR))) closed = true;
      }
    }
  }
  if (closed) {
    uint32_t expect = GRDMA_PAIR_CONNECTED;
    if (p->status.compare_exchange_strong(expect, GRDMA_PAIR_HALF_CLOSED)) {
      // the device copy gates the kernels (a Send on a pair that is not connected takes nothing, pair.cc:652-655)
      std::lock_guard<std::mutex> lk(p->fd_mu);
      const uint32_t hc = GRDMA_PAIR_HALF_CLOSED;
      hipMemcpyAsync(reinterpret_cast<uint8_t*>(p->d_conn) + offsetof(grdma_conn, status), &hc, sizeof(hc),
                     hipMemcpyHostToDevice, p->refresh_stream ? p->refresh_stream : p->stream);
      hipStreamSynchronize(p->refresh_stream ? p->refresh_stream : p->stream);
    }
    return p->status.load();
  }
  return st;
}
}  // namespace

int grdma_pair_get_status(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  return (int)status_now(p);
}

int64_t grdma_pair_send(grdma_pair* p, const grdma_slice* slices, uint64_t count,
                        uint64_t byte_idx, int flags) {
  if (int rc = require_ctx()) return rc;
  if (!p || (!slices && count)) return fail(GRDMA_ERR_INVALID, "null argument");
  if (count == 0) return 0;
  if (byte_idx >= slices[0].len && slices[0].len > 0)
    return fail(GRDMA_ERR_INVALID, "byte_idx %llu beyond the first slice",
                (unsigned long long)byte_idx);
  if (int rc = stage_slices(p, slices, count, byte_idx, flags)) return rc;
  if (int rc = run_send(p, count, byte_idx, 0)) return rc;
  return (int64_t)p->h->txres.sent;
}

int64_t grdma_pair_recv(grdma_pair* p, void* dst, uint64_t capacity, int flags) {
  if (int rc = require_ctx()) return rc;
  if (!p || !dst) return fail(GRDMA_ERR_INVALID, "null argument");
  if (capacity == 0) return 0;
  uint8_t* target = static_cast<uint8_t*>(dst);
  bool bounce = (flags & GRDMA_MEM_HOST) != 0;
  if (bounce) {
    if (capacity > p->arena_cap) capacity = p->arena_cap;
    target = p->d_arena;
  }
  if (int rc = run_recv(p, target, capacity, 1, capacity)) return rc;
  uint64_t n = p->h->rxres.bytes;
  if (bounce && n) HIP_TRY(hipMemcpy(dst, p->d_arena, n, hipMemcpyDeviceToHost));
  return (int64_t)n;
}

int grdma_poll_pairs(grdma_pair* const* pairs, uint32_t n, uint64_t* readable,
                     uint8_t* has_message) {
  if (int rc = require_ctx()) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  if (n > g_ctx.poll_cap) {
    if (g_ctx.h_conns) hipHostFree(g_ctx.h_conns);
    if (g_ctx.h_readable) hipHostFree(g_ctx.h_readable);
    if (g_ctx.h_masks) hipHostFree(g_ctx.h_masks);
    uint32_t cap = (n + 63) & ~63u;
    HIP_TRY(hipHostMalloc((void**)&g_ctx.h_conns, sizeof(void*) * cap, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&g_ctx.h_readable, sizeof(uint64_t) * cap, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&g_ctx.h_masks, sizeof(uint64_t) * 2 * (cap / 64),
                          hipHostMallocCoherent | hipHostMallocMapped));
    g_ctx.poll_cap = cap;
  }
  hipStream_t s = pairs[0]->stream;
  for (uint32_t i = 0; i < n; i++) {
    g_ctx.h_conns[i] = pairs[i]->d_conn;
    if (pairs[i]->stream != s) HIP_TRY(hipStreamSynchronize(pairs[i]->stream));
  }
  uint32_t words = (n + 63) / 64;
  HIP_TRY(grdma_launch_poll(g_ctx.h_conns, n, g_ctx.h_readable, g_ctx.h_masks,
                            g_ctx.h_masks + words, nullptr, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (uint32_t i = 0; i < n; i++) {
    if (readable) readable[i] = g_ctx.h_readable[i];
    if (has_message) has_message[i] = (g_ctx.h_masks[words + i / 64] >> (i % 64)) & 1;
  }
  return 0;
}

// ---- background poller (RDMA_BPEV): src/core/lib/ibverbs/poller.{h,cc} ---------------------
// The reference runs GRPC_RDMA_POLLER_THREAD_NUM threads that visit the registered pairs
// one by one and kick a pair's wakeup fd when it has a message, pending writes or a dead
// peer (poller.cc:52-106).  Here ONE thread covers every registered pair with one k_poll
// launch per pass (64 connections per wavefront, the tests done on the device where the
// rings live), on a stream of its own so that it never queues behind the data path.
}  // extern "C"

struct grdma_poller {
  // the slot table of Poller (poller.h:60-66): a hole is a free slot (RemovePollable leaves one, poller.cc:45-54)
  std::shared_mutex mu;             // shared: a thread looks at one slot; exclusive: add / remove
  std::vector<grdma_pair*> pairs;
  std::atomic<uint32_t> curr{0};    // curr_: the threads share one round-robin cursor (poller.cc:66-69)
  std::atomic<uint32_t> n_pairs{0};
  std::mutex cv_mu;
  std::condition_variable cv;
  std::atomic<bool> running{true};
  std::atomic<uint64_t> wakeups{0}, passes{0};
  std::vector<std::thread> threads; // GRPC_RDMA_POLLER_THREAD_NUM of them
  int sleep_ms = 1000;
  int device = 0;
};

namespace {

// Poller::begin_polling (poller.cc:52-106): every thread takes the next slot of the shared cursor and kicks the
// pair's wakeup fd when the pair is connected and readable or writable, or half-closed / in error -- unless the
// fd is still signalled (the consumer has not read it: poller.cc:76-78).  What it looks at is the pair's host-visible
// state (grdma_endpoint_readable / _writable / get_status: plain loads), so a pass costs no device work for a pair
// whose peer lives in this process; for a remote peer the queries keep one refresh pass of k_poll in flight.
void poller_loop(grdma_poller* pl, int /*poller_id*/) {
  hipSetDevice(pl->device);  // (the refresh pass of a remote pair is launched from here)
  uint32_t idle = 0;
  while (pl->running.load(std::memory_order_acquire)) {
    if (pl->n_pairs.load(std::memory_order_acquire) == 0) {  // poller.cc:58-63
      std::unique_lock<std::mutex> lk(pl->cv_mu);
      pl->cv.wait_for(lk, std::chrono::milliseconds(pl->sleep_ms),
                      [&] { return !pl->running.load() || pl->n_pairs.load() != 0; });
      continue;
    }
    bool kicked = false;
    {
      std::shared_lock<std::shared_mutex> lk(pl->mu);
      const size_t n = pl->pairs.size();
      if (n == 0) continue;
      const uint32_t at = pl->curr.fetch_add(1, std::memory_order_relaxed);
      grdma_pair* p = pl->pairs[at % n];
      if (at % n == 0) pl->passes.fetch_add(1, std::memory_order_relaxed);
      if (p != nullptr && p->wakeup_fd >= 0) {
        const uint32_t st = status_now(p);
        const bool trigger = st == GRDMA_PAIR_CONNECTED
                                 ? (grdma_endpoint_readable(p) > 0 || grdma_endpoint_writable(p) > 0)
                                 : (st == GRDMA_PAIR_HALF_CLOSED || st == GRDMA_PAIR_ERROR);
        if (trigger) {
          struct pollfd pfd = {p->wakeup_fd, POLLIN, 0};
          if (poll(&pfd, 1, 0) <= 0) {  // not signalled yet
            const uint64_t one = 1;
            if (write(p->wakeup_fd, &one, sizeof(one)) == (ssize_t)sizeof(one)) {
              pl->wakeups.fetch_add(1, std::memory_order_relaxed);
              kicked = true;
            }
          }
        }
      }
    }
    // (the reference spins at full speed; a poller that has found nothing for a while yields its core)
    if (kicked) idle = 0;
    else if (++idle > 4096) std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

}  // namespace

extern "C" {

int grdma_pair_get_wakeup_fd(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  std::lock_guard<std::mutex> lk(p->fd_mu);
  if (p->wakeup_fd < 0) {
    p->wakeup_fd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);  // grpc_wakeup_fd_init, pair.cc:74
    if (p->wakeup_fd < 0) return fail(GRDMA_ERR_INVALID, "eventfd failed");
  }
  return p->wakeup_fd;
}

int grdma_pair_consume_wakeup(grdma_pair* p) {  // grpc_wakeup_fd_consume_wakeup
  if (!p || p->wakeup_fd < 0) return fail(GRDMA_ERR_INVALID, "pair has no wakeup fd");
  uint64_t v = 0;
  const ssize_t r = read(p->wakeup_fd, &v, sizeof(v));
  return r == (ssize_t)sizeof(v) ? 1 : 0;
}

grdma_poller* grdma_poller_create(int n_threads, int sleep_timeout_ms) {
  if (require_ctx()) return nullptr;
  if (n_threads <= 0) {  // GPR_ASSERT(poller_thread_num_ > 0), config.cc
    fail(GRDMA_ERR_CONFIG, "poller thread count must be positive");
    return nullptr;
  }
  if (n_threads > 64) n_threads = 64;
  grdma_poller* pl = new grdma_poller();
  pl->sleep_ms = sleep_timeout_ms > 0 ? sleep_timeout_ms : 1000;
  pl->device = g_ctx.device;
  for (int i = 0; i < n_threads; i++) pl->threads.emplace_back(poller_loop, pl, i);  // poller.h:24-28
  return pl;
}

int grdma_poller_add(grdma_poller* pl, grdma_pair* p) {  // Poller::AddPollable, poller.cc:12-43
  if (!pl || !p) return fail(GRDMA_ERR_INVALID, "null argument");
  const int fd = grdma_pair_get_wakeup_fd(p);
  if (fd < 0) return fd;
  {
    std::unique_lock<std::shared_mutex> lk(pl->mu);
    size_t slot = 0;
    for (; slot < pl->pairs.size(); slot++)
      if (pl->pairs[slot] == nullptr) break;
    if (slot == pl->pairs.size()) {
      if (pl->pairs.size() >= 4096) return fail(GRDMA_ERR_CAPACITY, "poller is full");  // GRPC_IBVERBS_POLLER_CAPACITY
      pl->pairs.push_back(p);
    } else {
      pl->pairs[slot] = p;
    }
    pl->n_pairs.fetch_add(1, std::memory_order_release);
  }
  pl->cv.notify_all();
  return fd;
}

int grdma_poller_remove(grdma_poller* pl, grdma_pair* p) {  // Poller::RemovePollable, poller.cc:45-54
  if (!pl || !p) return fail(GRDMA_ERR_INVALID, "null argument");
  // exclusive: no thread is looking at the pair when this returns, so it may be destroyed
  std::unique_lock<std::shared_mutex> lk(pl->mu);
  for (auto& q : pl->pairs)
    if (q == p) {
      q = nullptr;
      pl->n_pairs.fetch_sub(1, std::memory_order_release);
      return 0;
    }
  return fail(GRDMA_ERR_INVALID, "pair is not registered with this poller");
}

int grdma_poller_stats(grdma_poller* pl, uint64_t* passes, uint64_t* wakeups) {
  if (!pl) return fail(GRDMA_ERR_INVALID, "null poller");
  if (passes) *passes = pl->passes.load();
  if (wakeups) *wakeups = pl->wakeups.load();
  return 0;
}

int grdma_poller_threads(grdma_poller* pl) { return pl ? (int)pl->threads.size() : -1; }

void grdma_poller_destroy(grdma_poller* pl) {  // Poller::Shutdown, poller.h:37-50
  if (!pl) return;
  pl->running.store(false, std::memory_order_release);
  pl->cv.notify_all();
  for (std::thread& t : pl->threads)
    if (t.joinable()) t.join();
  delete pl;
}

// The read-only queries of PairPollable: plain loads of the pair's host-visible state line, no device
// call, no lock, any number of threads (the contract of ring_buffer.cc:56-65 and pair.cc:294-303, which the
// event engines rely on: ev_epollex_rdma_bpev_linux.cc:1103-1145 calls them for every fd on every pass).
int grdma_pair_has_message(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  const grdma_hostline* l = p->line;
  refresh_async(p);  // (remote peers only; never waits)
  // HasMessage(), ring_buffer.cc:56-65: remain_ > 0, or a record behind head_ -- the sender's arrival
  // report has moved past the position my drains have opened up to
  const uint64_t wt = __atomic_load_n(&l->wire_tail, __ATOMIC_ACQUIRE);
  const uint64_t rh = __atomic_load_n(&l->rx_head, __ATOMIC_RELAXED);
  const uint64_t rem = __atomic_load_n(&l->rx_remain, __ATOMIC_RELAXED);
  return (rem > 0 || wt != rh) ? 1 : 0;
}

int64_t grdma_pair_readable_size(grdma_pair* p) {  // GetReadableSize(): the header itself is in HBM -> one k_poll launch
  uint64_t r = 0;
  grdma_pair* arr[1] = {p};
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  int rc = grdma_poll_pairs(arr, 1, &r, nullptr);
  if (rc < 0) return rc;
  return status_now(p) == GRDMA_PAIR_CONNECTED ? (int64_t)r : 0;  // pair.cc:290-292
}

int grdma_pair_has_pending_writes(grdma_pair* p) {  // HasPendingWrites(), pair.cc:303: partial_write_
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  return __atomic_load_n(&p->line->partial_write, __ATOMIC_ACQUIRE) ? 1 : 0;
}

int64_t grdma_pair_writable_size(grdma_pair* p) {  // GetWritableSize(), pair.cc:294-301
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  refresh_async(p);
  const grdma_hostline* l = p->line;
  return (int64_t)grdma_host_writable(p->ring_size, __atomic_load_n(&l->remote_head, __ATOMIC_ACQUIRE),
                                      __atomic_load_n(&l->remote_tail, __ATOMIC_ACQUIRE));
}

int grdma_pair_state_get(grdma_pair* p, grdma_pair_state* out) {
  if (int rc = require_ctx()) return rc;
  if (!p || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  grdma_conn c;
  if (int rc = fetch_conn(p, &c)) return rc;
  out->head = c.head;
  out->moving_head = c.moving_head;
  out->remain = c.remain;
  out->remote_tail = c.remote_tail;
  out->remote_head = c.status_recv.remote_head;
  out->internal_read_size = c.internal_read_size;
  out->credit_msgs = c.credit_msgs;
  out->partial_write = c.partial_write;
  out->total_read = c.total_read;
  out->total_written = c.total_written;
  out->leftover_cap = c.leftover_cap;
  return 0;
}

int grdma_pair_peek_ring(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len) {
  if (int rc = require_ctx()) return rc;
  if (!p || off + len > p->ring_size) return fail(GRDMA_ERR_INVALID, "range outside the ring");
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(host_dst, p->d_ring + off, len, hipMemcpyDeviceToHost));
  return 0;
}

int grdma_pair_peek_staging(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len) {
  if (int rc = require_ctx()) return rc;
  if (!p || off + len > p->ring_size / 2) return fail(GRDMA_ERR_INVALID, "range outside staging");
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipMemcpy(host_dst, p->d_staging + off, len, hipMemcpyDeviceToHost));
  return 0;
}

// ---- zero-copy send buffer: pair.cc:103-120 (initSendBuffer(kZeroCopyBuffer)), :305-323
// (AllocateSendBuffer), :793-941 (SendZerocopy) ------------------------------------------------
int grdma_pair_enable_zerocopy(grdma_pair* p, uint64_t bytes) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (bytes == 0) {  // the reference sizes it from its Config (config.cc:100-106)
    grdma_config cfg;
    if (grdma_config_from_env(&cfg) < 0) return fail(GRDMA_ERR_INVALID, "bad configuration");
    bytes = (uint64_t)cfg.zerocopy_buffer_size_kb * 1024;
  }
  if (bytes >= (1ull << 32)) return fail(GRDMA_ERR_INVALID, "zero-copy buffer of %llu bytes: the tail is 32 bits",
                                         (unsigned long long)bytes);
  std::lock_guard<std::mutex> lk(p->zc_mu);
  if (p->d_zc && p->zc_cap == bytes) return 0;
  if (p->zc_tail != 0) return fail(GRDMA_ERR_INVALID, "zero-copy buffer in use");
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->d_zc) hipFree(p->d_zc);
  p->d_zc = nullptr;
  p->zc_cap = 0;
  HIP_TRY(hipMalloc((void**)&p->d_zc, bytes));
  p->zc_cap = bytes;
  return 0;
}

void* grdma_pair_allocate_send_buffer(grdma_pair* p, uint64_t size) {
  if (require_ctx() != 0 || !p || size == 0) return nullptr;                  // :306-308
  if (!p->d_zc && grdma_pair_enable_zerocopy(p, 0) != 0) return nullptr;
  std::lock_guard<std::mutex> lk(p->zc_mu);
  const uint32_t tail = p->zc_tail;
  if (tail != 0 || (uint64_t)tail + size > p->zc_cap) return nullptr;        // :315-318: only an empty buffer serves
  p->zc_tail = (uint32_t)(tail + size);
  return p->d_zc + tail;
}

int64_t grdma_pair_send_zerocopy(grdma_pair* p, const grdma_slice* slices, uint64_t count, uint64_t byte_idx,
                                 int flags) {
  if (int rc = require_ctx()) return rc;
  if (!p || (!slices && count)) return fail(GRDMA_ERR_INVALID, "null argument");
  if (flags & GRDMA_MEM_HOST)
    return fail(GRDMA_ERR_INVALID, "SendZerocopy takes device-accessible slices (the payload is read where it lies)");
  if (count == 0) return 0;
  if (byte_idx >= slices[0].len && slices[0].len > 0)
    return fail(GRDMA_ERR_INVALID, "byte_idx %llu beyond the first slice", (unsigned long long)byte_idx);
  grdma_profiler profiler(GRDMA_STATS_TIME_PAIR_SEND);  // pair.cc:795
  if (int rc = stage_slices(p, slices, count, byte_idx, GRDMA_MEM_DEVICE)) return rc;
  grdma_hostblk* h = p->h;
  h->zcop.conn = p->d_conn;
  h->zcop.slices = p->h_sges;
  h->zcop.nslices = count;
  h->zcop.byte_idx = byte_idx;
  h->zcop.plan = p->d_txplan;
  h->zcop.result = &h->txres;
  h->zcop.zc_base = p->d_zc;
  h->zcop.zc_cap = p->zc_cap;
  HIP_TRY(grdma_launch_tx_plan_zc(&h->zcop, 1, p->stream));
  // the records go straight into the peer ring: one gather launch, no wire launch
  HIP_TRY(grdma_launch_copy(&h->plan_ptrs[0], 1, copy_blocks_for(p->ring_size), p->stream));
  HIP_TRY(grdma_launch_tx_commit1(p->d_conn, ++p->tx_seq, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  {
    std::lock_guard<std::mutex> lk(p->zc_mu);
    p->zc_tail = (uint32_t)(p->zc_tail - (uint32_t)h->txres.dbg[0]);  // :876
    p->zc_bytes += h->txres.dbg[0];
    p->zc_copy_bytes += h->txres.dbg[1];
    p->zc_last_sges = h->txres.dbg[2];
  }
  return (int64_t)h->txres.sent;
}

// out = {zerocopy_buffer_tail_, zerocopy_bytes_, copy_bytes_, scatter-gather entries of the last SendZerocopy}
int grdma_pair_zerocopy_state(grdma_pair* p, uint64_t out[4]) {
  if (!p || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk(p->zc_mu);
  out[0] = p->zc_tail;
  out[1] = p->zc_bytes;
  out[2] = p->zc_copy_bytes;
  out[3] = p->zc_last_sges;
  return 0;
}

int grdma_pair_last_wrs(grdma_pair* p, uint64_t out[2][2]) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  int n = (int)p->h->txres.wr_count;
  for (int i = 0; i < n && i < 2; i++) {
    out[i][0] = p->h->txres.wr_off[i];
    out[i][1] = p->h->txres.wr_len[i];
  }
  return n;
}

void* grdma_pair_ring_device_ptr(grdma_pair* p) { return p ? p->d_ring : nullptr; }

int grdma_pair_export_ring_dmabuf(grdma_pair* p) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  int fd = -1;
  const hipError_t e = hipMemGetHandleForAddressRange(&fd, (hipDeviceptr_t)p->d_ring, p->ring_size,
                                                      hipMemRangeHandleTypeDmaBufFd, 0);
  if (e != hipSuccess || fd < 0) return fail(GRDMA_ERR_HIP, "dma-buf export of the ring failed: %s", hipGetErrorString(e));
  return fd;
}
void* grdma_pair_arena_device_ptr(grdma_pair* p) {
  return p ? (p->latency ? p->h_arena : p->d_arena) : nullptr;
}
uint64_t grdma_pair_arena_size(grdma_pair* p) { return p ? p->arena_cap : 0; }

int grdma_pair_arena_copy_out(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->latency) {  // the arena is pinned host memory: the kernel already wrote it there
    if (off + len > 2 * p->h_arena_cap) return fail(GRDMA_ERR_INVALID, "range outside the arena");
    memcpy(host_dst, p->h_arena + off, len);
    return 0;
  }
  if (off + len > p->arena_cap) return fail(GRDMA_ERR_INVALID, "range outside the arena");
  HIP_TRY(hipMemcpy(host_dst, p->d_arena + off, len, hipMemcpyDeviceToHost));
  return 0;
}

int grdma_engine_start(void) {
  if (int rc = require_ctx()) return rc;
  std::lock_guard<std::mutex> lk(g_engine.mu);
  g_engine.wanted = true;
  return engine_launch();
}

int grdma_pair_last_dbg(grdma_pair* p, uint64_t* tx_dbg, uint64_t* rx_dbg) {
  if (!p) return -1;
  memcpy(tx_dbg, p->h->txres.dbg, sizeof(uint64_t) * 16);
  memcpy(rx_dbg, p->h->rxres.dbg, sizeof(uint64_t) * 16);
  return 0;
}

int grdma_engine_debug(uint64_t out[5]) {
  if (!g_engine.mb) return -1;
  for (int i = 0; i < 5; i++) out[i] = g_engine.mb->pad1[i];
  return 0;
}

int grdma_engine_stop(void) {
  if (int rc = require_ctx()) return rc;
  return engine_stop();
}

int grdma_pair_set_latency_mode(grdma_pair* p, int on) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (on && !p->h_cmd) {
    HIP_TRY(hipHostMalloc((void**)&p->h_cmd, 2 * sizeof(grdma_engine_cmd), hipHostMallocCoherent | hipHostMallocMapped));
    memset(p->h_cmd, 0, 2 * sizeof(grdma_engine_cmd));
    p->h_cmd_rx = p->h_cmd + 1;
  }
  if (on && !p->h_arena) {
    p->h_arena_cap = 2 * p->ring_size + 4096;
    if (p->h_arena_cap > (64ull << 20)) p->h_arena_cap = 64ull << 20;
    // (twice what one drain may fill: a watcher's drains alternate between the halves, so the slices of a completion
    //  stay where they are while the next drain -- which no longer waits for a call -- delivers; k_watch)
    HIP_TRY(hipHostMalloc((void**)&p->h_arena, 2 * p->h_arena_cap, hipHostMallocCoherent | hipHostMallocMapped));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (!on && p->watch_slot >= 0) {
    if (int rc = grdma_pair_arm_read(p, 0)) return rc;
  }
  if (!on && p->armed_done) return fail(GRDMA_ERR_INVALID, "an armed read has completed and was not consumed");
  if (!on) p->armed_reads = 0;
  p->latency = on != 0;
  return 0;
}

// ---- NIC wire (ibverbs): see include/grdma_amd.h and csrc/grdma_wire_verbs.cc -------------------------------------
int grdma_verbs_supported(void) { return grdma_verbs_available() ? 1 : 0; }

int grdma_pair_verbs_open(grdma_pair* p, const char* device, int port, int gid_index) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->verbs) return fail(GRDMA_ERR_INVALID, "the pair has a NIC wire already");
  if (!(p->flags & GRDMA_WIRE_ORDERED) || (p->flags & GRDMA_WIRE_DIRECT))
    return fail(GRDMA_ERR_INVALID, "a NIC-written ring needs GRDMA_WIRE_ORDERED and a staged wire (no GRDMA_WIRE_DIRECT)");
  if (p->status.load() == GRDMA_PAIR_CONNECTED) return fail(GRDMA_ERR_INVALID, "the pair is connected already");
  // the ring's dma-buf, where the runtime exports one (what ibv_reg_dmabuf_mr takes): -1 = register by address
  int fd = grdma_pair_export_ring_dmabuf(p);
  if (fd < 0) fd = -1;
  uint8_t* conn = reinterpret_cast<uint8_t*>(p->d_conn);
  std::string err;
  p->verbs = grdma_verbs_open(device, port, gid_index, p->d_ring, p->ring_size, fd, p->d_staging, p->ring_size / 2 + 64,
                              conn + offsetof(grdma_conn, status_send), conn + offsetof(grdma_conn, status_recv),
                              sizeof(grdma_status_report), &err);
  if (fd >= 0) close(fd);  // (the registration holds its own reference)
  if (!p->verbs) return fail(GRDMA_ERR_HIP, "NIC wire: %s", err.c_str());
  return 0;
}

int grdma_pair_verbs_address(grdma_pair* p, grdma_verbs_address* out) {
  if (!p || !out || !p->verbs) return fail(GRDMA_ERR_INVALID, "no NIC wire on this pair");
  return grdma_verbs_address_of(p->verbs, out) == 0 ? 0 : fail(GRDMA_ERR_HIP, "NIC wire: no address");
}

int grdma_pair_verbs_connect(grdma_pair* p, const grdma_verbs_address* peer) {
  if (int rc = require_ctx()) return rc;
  if (!p || !peer || !p->verbs) return fail(GRDMA_ERR_INVALID, "no NIC wire on this pair");
  if (peer->ring_size != p->ring_size)  // pair.cc:149
    return fail(GRDMA_ERR_INVALID, "ring sizes differ (%llu vs %llu)", (unsigned long long)p->ring_size, (unsigned long long)peer->ring_size);
  std::string err;
  if (grdma_verbs_connect(p->verbs, peer, &err) != 0) return fail(GRDMA_ERR_HIP, "NIC wire: %s", err.c_str());
  grdma_conn c;
  if (int rc = fetch_conn(p, &c)) return rc;
  c.peer_ring = nullptr;      // what this end writes leaves through the queue pair, never through a pointer
  c.peer_status = nullptr;
  c.peer_wire = nullptr;      // (a NIC places bytes in order: the peer reads by the records' tags, no arrival report)
  c.peer_line = nullptr;
  c.line_remote = 1;          // credit lands in status_recv by DMA: the state line is refreshed from the connection block
  c.peer_limited = 0;
  c.wire_limit = 0;
  c.status = GRDMA_PAIR_CONNECTED;
  HIP_TRY(hipMemcpy(p->d_conn, &c, sizeof(c), hipMemcpyHostToDevice));
  p->remote = true;           // (host-side queries take the paths of a peer that is not in this process)
  p->status.store(GRDMA_PAIR_CONNECTED);
  return 0;
}

int grdma_pair_verbs_counts(grdma_pair* p, uint64_t out[3]) {
  if (!p || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  grdma_verbs_counts(p->verbs, out);
  return 0;
}

// Armed read.  gRPC keeps an endpoint_read outstanding on every connection (the transport re-arms it
// from read_action_locked, chttp2_transport.cc:2508-2596), and the reference's busy-polling thread
// completes it the moment a record lands.  Here, when both ends of a link live in this process and
// run through the latency engine, a small send from the peer carries this pair's drain in the same
// engine command, so the completion costs no doorbell round trip of its own; the next
// grdma_endpoint_read (max_reads >= the armed value) returns it.  Bytes, order and connection state
// are those of the two separate commands.  max_reads = 0 disarms.  Single-threaded use per link.
int grdma_pair_arm_read(grdma_pair* p, uint64_t max_reads) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (max_reads && !p->latency) return fail(GRDMA_ERR_INVALID, "armed reads need latency mode");
  if (max_reads > GRDMA_MAX_SLICES) max_reads = GRDMA_MAX_SLICES;
  if (engine_chain_mode() && p->watch_slot < 0) {  // (round 4's way: the in-process peer's send command carries the drain)
    if (p->async) p->armed_async.store(max_reads, std::memory_order_release);
    else p->armed_reads = max_reads;
    return 0;
  }
  grdma_engine& e = g_engine;
  // (the endpoint arms before every wait: nothing to do when the order stands)
  if (p->watch_slot >= 0 && max_reads == p->armed_reads && !e.watch_dirty && !p->watch_parked) return 0;
  std::lock_guard<std::mutex> lk(e.mu);
  if (p->watch_slot >= 0 && max_reads == p->armed_reads) {
    if (p->watch_parked) {
      std::lock_guard<std::mutex> rl(p->rx_mu);
      watch_unpark(p);
    }
    if (e.wanted && e.watch_dirty) {
      if (int rc = engine_launch()) return rc;
      if (int rc = watch_flush_locked()) return rc;
    }
    return 0;
  }
  if (p->watch_slot >= 0 && (max_reads == 0 || max_reads != p->armed_reads)) {
    // back from the watcher; a completion it produced and nobody has taken stays for the next grdma_endpoint_read
    // (watch_expect is non-zero exactly while the order sits in its device slot)
    int rc = watch_release_locked(p);
    if (!p->async && p->watch_expect != 0 && __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) >= p->watch_expect) {
      p->armed_half = p->watch_taken & 1;
      p->armed_done = true;
    }
    if (p->async && p->rx_inflight.load() >= 0 && __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) < p->rx_expect.load())
      p->rx_inflight.store(-1);   // (the standing order is withdrawn; a completion that is there stays to be taken)
    p->watch_slot = -1;
    p->watch_expect = 0;
    if (rc) return rc;
  }
  p->armed_reads = max_reads;
  if (max_reads == 0 || p->watch_slot >= 0) return 0;
  int sidx = -1;
  for (uint32_t k = 0; k < GRDMA_WATCH_SLOTS; k++)
    if (!e.slot_owner[k]) { sidx = (int)k; break; }
  if (sidx < 0) return fail(GRDMA_ERR_CAPACITY, "all %d watch slots of the latency engine are taken", GRDMA_WATCH_SLOTS);
  e.slot_owner[sidx] = p;
  e.slot_posted[sidx] = false;
  e.watch_dirty = true;
  p->watch_slot = sidx;
  p->watch_expect = 0;
  if (e.wanted) {
    if (int rc = engine_launch()) return rc;
    if (int rc = watch_flush_locked()) return rc;
  }
  return 0;
}
int64_t grdma_pair_armed_hits(const grdma_pair* p) { return p ? (int64_t)p->armed_hits : -1; }
int64_t grdma_pair_watch_hits(const grdma_pair* p) { return p ? (int64_t)p->watch_hits : -1; }
int grdma_engine_watchers(void) { return (int)engine_watch_groups(); }
int grdma_pair_armed_ready(const grdma_pair* p) {
  if (!p) return 0;
  if (p->armed_done) return 1;
  return p->watch_slot >= 0 && p->watch_expect != 0 && __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) >= p->watch_expect ? 1 : 0;
}

// Unary ping-pong over a connected loop-back link, host in the loop exactly
// where gRPC's consumer is: a = client end, b = server end.  Per iteration:
// client endpoint_write(req) -> server endpoint_read -> server
// endpoint_write(resp) -> client endpoint_read.  rtt_ns[i] = wall time of
// iteration i; phase_ns[0..4) = summed time of the four phases.
int grdma_pingpong(grdma_pair* a, grdma_pair* b, const grdma_slice* req, uint64_t nreq,
                   const grdma_slice* resp, uint64_t nresp, int mem_flags, uint64_t iters,
                   uint64_t warmup, uint64_t* rtt_ns, uint64_t phase_ns[4]) {
  if (int rc = require_ctx()) return rc;
  if (!a || !b || a->peer != b || !req || !resp || !rtt_ns) return fail(GRDMA_ERR_INVALID, "bad argument");
  uint64_t req_bytes = 0, resp_bytes = 0;
  for (uint64_t i = 0; i < nreq; i++) req_bytes += req[i].len;
  for (uint64_t i = 0; i < nresp; i++) resp_bytes += resp[i].len;
  grdma_read_slice sl[64];
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ns = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(y - x).count();
  };
  auto write_all = [&](grdma_pair* p, const grdma_slice* s, uint64_t n) -> int {
    grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_WRITE);  // rdma_write, rdma_bp_posix.cc:561
    if (int64_t rc = grdma_endpoint_write_begin(p, s, n, mem_flags); rc < 0) return (int)rc;
    int done = 0;
    for (int tries = 0; !done && tries < 1000; tries++) {
      int64_t rc = grdma_endpoint_write_step(p, &done);
      if (rc < 0) return (int)rc;
    }
    return done ? 0 : fail(GRDMA_ERR_HIP, "write did not complete");
  };
  auto read_all = [&](grdma_pair* p, uint64_t want) -> int {
    grdma_profiler profiler(GRDMA_STATS_TIME_TRANSPORT_READ);  // rdma_read, rdma_bp_posix.cc:345
    uint64_t got = 0;
    // (a watched pair's read is a look at host memory: bounded by time, not by tries)
    const bool watched = p->watch_slot >= 0;
    const auto r0 = watched ? now() : std::chrono::steady_clock::time_point();
    for (uint64_t tries = 0; got < want && (tries < 100000 || (watched && ns(r0, now()) < 5000000000ull)); tries++) {
      int wb = 0;
      int64_t n = grdma_endpoint_read(p, 64, sl, 64, &wb);
      if (n < 0) return (int)n;
      for (int64_t i = 0; i < n; i++) got += sl[i].len;
    }
    return got == want ? 0 : fail(GRDMA_ERR_HIP, "short read in ping-pong");
  };
  if (phase_ns) phase_ns[0] = phase_ns[1] = phase_ns[2] = phase_ns[3] = 0;
  for (uint64_t it = 0; it < warmup + iters; it++) {
    const auto t0 = now();
    if (int rc = write_all(a, req, nreq)) return rc;
    const auto t1 = now();
    if (int rc = read_all(b, req_bytes)) return rc;
    const auto t2 = now();
    if (int rc = write_all(b, resp, nresp)) return rc;
    const auto t3 = now();
    if (int rc = read_all(a, resp_bytes)) return rc;
    const auto t4 = now();
    if (it >= warmup) {
      rtt_ns[it - warmup] = ns(t0, t4);
      if (phase_ns) {
        phase_ns[0] += ns(t0, t1); phase_ns[1] += ns(t1, t2);
        phase_ns[2] += ns(t2, t3); phase_ns[3] += ns(t3, t4);
      }
    }
  }
  return 0;
}

// ONE end of a unary ping-pong whose other end lives elsewhere -- another process, through the IPC mapping of the
// rings -- the client's loop of examples/cpp/micro-bench/mb_client.cc or the server's echo side: client = write `out`,
// then read until in_bytes have arrived; server = the other way round.  The read side is whatever the pair is set up
// for: with a standing order (grdma_pair_arm_read) the watcher workgroup of THIS process's engine finds what the other
// process wrote into this pair's ring and the loop only looks at host memory.  rtt_ns (client: per iteration, first
// write to last byte read; server: read to read), byte_sum = sum of every byte received (both ends check it).
int grdma_pingpong_end(grdma_pair* p, int is_client, const grdma_slice* out, uint64_t nout, uint64_t in_bytes, int mem_flags,
                       uint64_t iters, uint64_t warmup, uint64_t* rtt_ns, uint64_t* byte_sum) {
  if (int rc = require_ctx()) return rc;
  if (!p || !out || !rtt_ns) return fail(GRDMA_ERR_INVALID, "bad argument");
  grdma_read_slice sl[64];
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ns = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(y - x).count();
  };
  uint64_t sum = 0;
  const uint8_t* arena = static_cast<const uint8_t*>(grdma_pair_arena_device_ptr(p));
  auto write_all = [&]() -> int {
    if (int64_t rc = grdma_endpoint_write_begin(p, out, nout, mem_flags); rc < 0) return (int)rc;
    int done = 0;
    for (int tries = 0; !done && tries < 100000; tries++) {
      int64_t rc = grdma_endpoint_write_step(p, &done);
      if (rc < 0) return (int)rc;
    }
    return done ? 0 : fail(GRDMA_ERR_HIP, "write did not complete");
  };
  auto read_all = [&]() -> int {
    uint64_t got = 0;
    const auto r0 = now();
    while (got < in_bytes) {
      int wb = 0;
      int64_t n = grdma_endpoint_read(p, 64, sl, 64, &wb);
      if (n < 0) return (int)n;
      for (int64_t i = 0; i < n; i++) {
        got += sl[i].len;
        if (p->latency && arena)
          for (uint64_t k = 0; k < sl[i].len; k++) sum += arena[sl[i].off + k];
      }
      if (n == 0 && ns(r0, now()) > 20000000000ull) return fail(GRDMA_ERR_HIP, "no message within 20 s");
    }
    return got == in_bytes ? 0 : fail(GRDMA_ERR_HIP, "a message of %llu bytes instead of %llu", (unsigned long long)got,
                                      (unsigned long long)in_bytes);
  };
  auto t_prev = now();
  for (uint64_t it = 0; it < warmup + iters; it++) {
    const auto t0 = now();
    if (is_client) {
      if (int rc = write_all()) return rc;
      if (int rc = read_all()) return rc;
    } else {
      if (int rc = read_all()) return rc;
      if (int rc = write_all()) return rc;
    }
    const auto t1 = now();
    if (it >= warmup) rtt_ns[it - warmup] = is_client ? ns(t0, t1) : ns(t_prev, t1);
    t_prev = t1;
  }
  if (byte_sum) *byte_sum = sum;
  return 0;
}

// ---- endpoint write: rdma_write / rdma_flush / rdma_handle_write -------------
int64_t grdma_endpoint_write_begin(grdma_pair* p, const grdma_slice* slices, uint64_t count,
                                   int flags) {
  if (int rc = require_ctx()) return rc;
  if (!p || (!slices && count)) return fail(GRDMA_ERR_INVALID, "null argument");
  if (p->w_active) return fail(GRDMA_ERR_INVALID, "a write is already outstanding");  // :563
  if (count > GRDMA_TX_MAX_RECORDS - 1)
    return fail(GRDMA_ERR_CAPACITY, "slice buffer of %llu slices exceeds %d",
                (unsigned long long)count, GRDMA_TX_MAX_RECORDS - 1);
  p->w_slices.assign(slices, slices + count);
  p->w_idx = 0;
  p->w_byte = 0;
  p->w_flags = flags;
  p->w_active = count > 0;
  return 0;
}

// Drops the write context (error exits of rdma_flush, rdma_bp_posix.cc:505-517: the slice
// buffer is unreffed there, so no view of it may survive in the pair).
int grdma_endpoint_write_abort(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  // A Send in flight may be gathering straight from the caller's pages (GRPC_RDMA_HIP_REGISTER_MIN: registered host
  // slices are read where they lie) and the caller is about to unref them: the send stream is drained first.  The
  // reference's Send is synchronous and has no such window.
  if (p->async && p->tx_inflight.load(std::memory_order_acquire)) {
    // the Send in flight reads the pinned tables (h_sges, the bounce buffer, the command block) -- and, with
    // registered host slices, the caller's pages -- until it has completed: wait for ITS completion word (an engine
    // command publishes txres.seq, a launch chain line->tx_seq) before the flag comes down and the next submit may
    // overwrite them.  Bounded: a wedged device must not turn an error exit into a hang.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const uint64_t seen = p->tx_by_engine ? __atomic_load_n(&p->h->txres.seq, __ATOMIC_ACQUIRE)
                                            : __atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE);
      if (seen >= p->tx_expect) break;
      if (!p->tx_by_engine && p->s_tx) {
        (void)hipStreamSynchronize(p->s_tx);
        break;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
    p->tx_inflight.store(0, std::memory_order_release);
  }
  p->w_active = false;
  p->w_slices.clear();
  p->w_idx = p->w_byte = 0;
  if (p->q_state != 0) {  // (a queued chain may still run: the connection is being given up)
    p->q_pending_seq = p->q_expect;
    p->q_state = 0;
    p->q_slices.clear();
  }
  return 0;
}

int64_t grdma_endpoint_write_step(grdma_pair* p, int* done) {
  if (int rc = require_ctx()) return rc;
  if (!p || !done) return fail(GRDMA_ERR_INVALID, "null argument");
  if (!p->w_active) {
    *done = 1;
    return 0;
  }
  // One rdma_flush: Send(slices + idx, count - idx, byte_idx), then walk the
  // cursor over what was accepted (rdma_bp_posix.cc:476-493).  The walk itself
  // happens on the device (k_tx_plan); the host mirrors it from the result.
  const uint64_t n = p->w_slices.size() - p->w_idx;
  if (int rc = stage_slices(p, p->w_slices.data() + p->w_idx, n, p->w_byte, p->w_flags)) return rc;
  if (int rc = run_send(p, n, p->w_byte, 0)) return rc;
  const grdma_tx_result& r = p->h->txres;
  p->w_idx += r.slice_idx;
  p->w_byte = r.byte_idx;
  *done = r.done ? 1 : 0;
  if (r.done) {
    p->w_active = false;
    p->w_slices.clear();
  }
  return (int64_t)r.sent;
}

// ---- endpoint read: rdma_read / rdma_continue_read / rdma_do_read -----------
int64_t grdma_endpoint_read(grdma_pair* p, uint64_t max_reads, grdma_read_slice* slices,
                            uint64_t slices_cap, int* would_block) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (max_reads > slices_cap) max_reads = slices_cap;
  if (max_reads > GRDMA_MAX_SLICES) max_reads = GRDMA_MAX_SLICES;
  if (max_reads == 0) return 0;
  uint8_t* arena = p->latency ? p->h_arena : p->d_arena;
  const uint64_t acap = p->latency ? p->h_arena_cap : p->arena_cap;
  bool watched = false;
  if (p->armed_done && p->latency) {
    // the drain already ran behind the peer's send (grdma_pair_arm_read)
    if (p->h->rxres.nslices > max_reads)   // (the completion stays: a call with room for it still gets it)
      return fail(GRDMA_ERR_INVALID, "the armed read delivered %llu slices, this call takes %llu",
                  (unsigned long long)p->h->rxres.nslices, (unsigned long long)max_reads);
    p->armed_done = false;
  } else if (p->latency && p->watch_slot >= 0 && g_engine.wanted) {
    // The standing order is with a watcher workgroup of the engine: it drains when bytes land, this call only looks
    // at the result block in pinned host memory.  Nothing there: the read stays outstanding -- no device work, no
    // change of state (what a pending grpc_endpoint_read is, rdma_bp_posix.cc:345-372).
    grdma_engine& e = g_engine;
    if (e.watch_dirty || !e.mb || !*(volatile uint64_t*)&e.mb->alive) {
      std::lock_guard<std::mutex> lk(e.mu);
      if (int rc = engine_launch()) return rc;   // (an engine that retired by itself comes back with its slots)
      if (int rc = watch_flush_locked()) return rc;
    }
    if (p->watch_expect == 0 || __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE) < p->watch_expect) {
      if (would_block) *would_block = 1;
      return 0;
    }
    if (p->h->rxres.nslices > max_reads)
      return fail(GRDMA_ERR_INVALID, "the armed read delivered %llu slices, this call takes %llu",
                  (unsigned long long)p->h->rxres.nslices, (unsigned long long)max_reads);
    watched = true;
  } else if (int rc = run_recv(p, arena, acap, max_reads, 0)) {
    return rc;
  }
  const grdma_rx_result& r = p->h->rxres;
  const uint64_t n = r.nslices;
  // (a watcher's drains alternate between the two halves of the arena: completion k of an arming lies in half k & 1)
  const uint64_t half_off = (watched ? (p->watch_taken & 1) : p->armed_half) * p->h_arena_cap;
  p->armed_half = 0;
  for (uint64_t i = 0; i < n; i++) {
    slices[i].off = p->h_slices[i].off + half_off;
    slices[i].len = p->h_slices[i].len;
  }
  if (would_block) *would_block = (int)r.would_block;
  if (watched) {
    // taken: the watcher may run the next drain (it overwrites result block, slice table and arena)
    p->watch_expect++;
    p->watch_hits++;
    ++p->watch_taken;   // (the next drain delivers into the other half of the arena)
    __atomic_store_n(&g_engine.mb->consumed[p->watch_slot], p->watch_taken | ((p->watch_taken & 1) << 56), __ATOMIC_RELEASE);
  }
  return (int64_t)n;
}

// ---- asynchronous endpoint operations (see include/grdma_amd.h) ----------------------------------
const void* grdma_window_base(const grdma_window* w) { return w ? w->base : nullptr; }
void grdma_window_ref(grdma_window* w) { if (w) w->refs.fetch_add(1, std::memory_order_relaxed); }
void grdma_window_unref(grdma_window* w) {
  if (w && w->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    if (w->base) hipHostFree(w->base);
    delete w;
  }
}

int grdma_set_host_register_min(uint64_t bytes) {
  register_min();  // (the environment is read first, an explicit call wins)
  g_reg.min_bytes.store(bytes);
  return 0;
}
int grdma_forget_host_range(const void* ptr, uint64_t len) {
  const uintptr_t a = (uintptr_t)ptr;
  std::lock_guard<std::mutex> lk(g_reg.mu);
  int n = 0;
  for (size_t i = 0; i < g_reg.e.size();) {
    if (g_reg.e[i].lo < a + len && a < g_reg.e[i].hi) {
      hipHostUnregister((void*)g_reg.e[i].lo);
      g_reg.e.erase(g_reg.e.begin() + (long)i);
      n++;
    } else {
      i++;
    }
  }
  return n;
}

int grdma_endpoint_set_async(grdma_pair* p, int windows, uint64_t window_bytes) {
  if (int rc = require_ctx()) return rc;
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->async) return 0;
  if (windows <= 0) windows = 3;
  if (windows < 2) return fail(GRDMA_ERR_INVALID, "an asynchronous endpoint needs at least two receive windows");
  if (window_bytes == 0) {
    // a drain delivers at most what the ring holds (plus the open 256-byte read and 16-byte slice alignment); the
    // planner stops gracefully where a window ends, so a window only has to hold the largest record (< ring / 2)
    const uint64_t full = 2 * p->ring_size + 4096, floor_ = p->ring_size / 2 + 4096;
    window_bytes = std::min<uint64_t>(full, std::max<uint64_t>(floor_, 64ull << 20));
  }
  if (window_bytes < p->ring_size / 2 + 4096) return fail(GRDMA_ERR_INVALID, "receive window smaller than the largest record");
  HIP_TRY(hipStreamSynchronize(p->stream));
  HIP_TRY(hipStreamCreateWithFlags(&p->s_tx, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&p->s_rx, hipStreamNonBlocking));
  for (int i = 0; i < windows; i++) {
    grdma_window* w = new grdma_window();
    w->bytes = window_bytes;
    if (hipHostMalloc((void**)&w->base, window_bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      delete w;
      return fail(GRDMA_ERR_HIP, "receive window of %llu bytes: pinned allocation failed", (unsigned long long)window_bytes);
    }
    p->windows.push_back(w);
  }
  p->async = true;
  return 0;
}

namespace {
// the Send of one rdma_flush step, enqueued on the send stream (or posted to the engine) and not waited for
int submit_send(grdma_pair* p, uint64_t count, uint64_t byte_idx) {
  grdma_hostblk* h = p->h;
  h->txop.conn = p->d_conn;
  h->txop.slices = p->h_sges;
  h->txop.nslices = count;
  h->txop.byte_idx = byte_idx;
  h->txop.plan = p->d_txplan;
  h->txop.wire_plan = p->d_wireplan;
  h->txop.result = &h->txres;
  h->txop.use_cursor = 0;
  h->txop.inline_copy = p->latency ? latency_op_bits() : 0;
  h->txop.seq_next = p->latency ? h->txres.seq + 1 : 0;
  if (p->latency && g_engine.wanted) {
    p->tx_by_engine = true;
    p->tx_expect = h->txop.seq_next;
    if (p->h_cmd && p->cmd_inline) {
      p->h_cmd->tx = h->txop;
      grdma_pair* q = p->peer;
      // the in-process peer keeps a read armed: its drain rides in this command (k_engine runs it right behind the
      // Send), and shows up on the peer's side as a drain in flight that completes by itself
      if (q && !p->remote && q->async && q->latency && q->armed_async.load(std::memory_order_acquire) &&
          q->rx_mu.try_lock()) {
        std::lock_guard<std::mutex> lk(q->rx_mu, std::adopt_lock);
        int w = -1;
        if (q->rx_inflight.load(std::memory_order_acquire) < 0)
          for (size_t i = 0; i < q->windows.size(); i++)
            if (q->windows[i]->refs.load(std::memory_order_acquire) == 1) { w = (int)i; break; }
        if (w >= 0) {
          fill_rxop(q, q->windows[w]->base, q->windows[w]->bytes, q->armed_async.load(), 0);
          p->h_cmd->rx = q->h->rxop;
          q->rx_expect.store(q->h->rxop.seq_next, std::memory_order_relaxed);
          q->rx_by_engine.store(1, std::memory_order_relaxed);
          q->rx_inflight.store(w, std::memory_order_release);
          p->h_cmd->tx.sizes_out = engine_chain_marker();
          if (int rc = engine_post(GRDMA_ENGINE_SEND_INLINE_DRAIN, p->h_cmd)) {
            q->rx_inflight.store(-1);
            return rc;
          }
          q->armed_hits++;
          return 0;
        }
      }
      return engine_post(GRDMA_ENGINE_SEND_INLINE, p->h_cmd);
    }
    return engine_post(GRDMA_ENGINE_SEND, &h->txop);
  }
  p->tx_by_engine = false;
  p->tx_burst = 0;
  if (!p->latency && (p->flags & GRDMA_WIRE_DIRECT) && count > (uint64_t)p->max_sge) {
    // More slices than one Send takes (max_sge): the reference comes back through the writable edge for every
    // Send (HasPendingWrites() -> rdma_handle_write -> rdma_flush, rdma_bp_posix.cc:527-557); here up to kBurstMax
    // of those Sends are planned by one launch (k_tx_plan_seq: the cursor and the ring tail stay on the device) and
    // gathered by one.  Records go straight into the peer ring, so the Sends need no staging buffers of their own.
    constexpr uint32_t kBurstMax = 16;
    uint32_t B = (uint32_t)std::min<uint64_t>(kBurstMax, (count + p->max_sge - 1) / p->max_sge);
    // The bounce buffer holds at most ring_size bytes of the slices.  The burst wave prices all Sends of a burst against
    // ONE reading of the peer's head, so it accepts no more than the ring had room for; the fallback (max_sge > 64, or a
    // ring above 1 GiB) runs the block-wide plan once per Send, each with a fresh reading, and with a reader draining
    // meanwhile B Sends could take up to B * ring / 2 bytes -- past what was staged.  Two Sends cannot.
    if (p->bounce_truncated && (p->max_sge > 64 || p->ring_size > (1ull << 30)) && B > 2) B = 2;
    // (a queued chain that was skipped may still sit in the stream and will read set 0's tables when it runs: it is
    // right behind the chain whose completion brought us here -- a few tens of microseconds, a rare path)
    while (__atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE) < p->q_pending_seq) {
      const hipError_t qe = hipStreamQuery(p->s_tx);
      if (qe != hipSuccess && qe != hipErrorNotReady) return fail(GRDMA_ERR_HIP, "send stream: %s", hipGetErrorString(qe));
    }
    p->inflight_set = 0;
    p->inflight_covers_all = (uint64_t)B * (uint64_t)p->max_sge >= count;
    if (!p->h_bops) {
      HIP_TRY(hipHostMalloc((void**)&p->h_bops, sizeof(grdma_tx_op) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
      HIP_TRY(hipHostMalloc((void**)&p->h_bres, sizeof(grdma_tx_result) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
      HIP_TRY(hipHostMalloc((void**)&p->h_bplan_ptrs, sizeof(grdma_plan*) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
      memset(p->h_bres, 0, sizeof(grdma_tx_result) * kBurstMax);
    }
    while (p->b_plans.size() < B) {
      grdma_plan* pl = nullptr;
      HIP_TRY(hipMalloc((void**)&pl, sizeof(grdma_plan)));
      HIP_TRY(hipMemsetAsync(pl, 0, sizeof(grdma_plan), p->s_tx));
      p->b_plans.push_back(pl);
    }
    // the cursor starts at slice 0 of the table: fold the byte offset into the table's first entry
    p->tx_burst_byte0 = byte_idx;
    if (byte_idx) {
      p->h_sges[0].ptr += byte_idx;
      p->h_sges[0].len -= byte_idx;
    }
    for (uint32_t k = 0; k < B; k++) {
      grdma_tx_op& t = p->h_bops[k];
      memset(&t, 0, sizeof(t));
      t.conn = p->d_conn;
      t.slices = p->h_sges;
      t.nslices = count;
      t.plan = p->b_plans[k];
      t.wire_plan = nullptr;
      t.result = &p->h_bres[k];
      t.use_cursor = k == 0 ? 2 : 1;
      p->h_bplan_ptrs[k] = p->b_plans[k];
    }
    const uint32_t blocks_b = std::max<uint32_t>(1, copy_blocks_for(p->ring_size) / B + 1);
    HIP_TRY(grdma_launch_tx_plan_seq(p->h_bops, 1, B, p->s_tx));
    HIP_TRY(grdma_launch_copy(p->h_bplan_ptrs, B, blocks_b, p->s_tx));
    HIP_TRY(grdma_launch_tx_commit1(p->d_conn, ++p->tx_seq, p->s_tx));
    p->tx_expect = p->tx_seq;
    p->tx_burst = B;
    return 0;
  }
  p->inflight_covers_all = false;
  const uint32_t blocks = copy_blocks_for(p->ring_size / 2);
  HIP_TRY(grdma_launch_tx_plan(&h->txop, 1, p->s_tx));
  if (!p->latency) {
    HIP_TRY(grdma_launch_copy(&h->plan_ptrs[0], 1, blocks, p->s_tx));
    if (!(p->flags & GRDMA_WIRE_DIRECT)) HIP_TRY(grdma_launch_copy(&h->plan_ptrs[1], 1, blocks, p->s_tx));
  }
  HIP_TRY(grdma_launch_tx_commit1(p->d_conn, ++p->tx_seq, p->s_tx));
  p->tx_expect = p->tx_seq;
  return 0;
}

struct burst_set {
  grdma_tx_op* ops;
  grdma_tx_result* res;
  const grdma_plan** plan_ptrs;
  grdma_sge* sges;
};
inline burst_set bset(grdma_pair* p, int i) {
  return i == 0 ? burst_set{p->h_bops, p->h_bres, p->h_bplan_ptrs, p->h_sges}
                : burst_set{p->q_ops, p->q_res, p->q_plan_ptrs, p->q_sges};
}
}  // namespace

// A write behind the burst in flight, submitted NOW: 0 = its chain is in the send stream (the caller treats it as
// submitted once grdma_endpoint_write_adopt says it was promoted), 1 = not possible at the moment (the caller keeps it
// and submits it the ordinary way later).  slices must be device-visible memory that stays valid until the write has
// completed (the endpoint's send buffers).  See the field comments of grdma_pair.
int grdma_endpoint_write_queue(grdma_pair* p, const grdma_slice* slices, uint64_t count) {
  if (int rc = require_ctx()) return rc;
  if (!p || !slices) return fail(GRDMA_ERR_INVALID, "null argument");
  constexpr uint32_t kBurstMax = 16;
  if (!p->async || p->latency || !(p->flags & GRDMA_WIRE_DIRECT) || p->remote) return 1;
  if (!p->tx_inflight.load(std::memory_order_acquire) || p->tx_by_engine || p->tx_burst == 0 || !p->inflight_covers_all)
    return 1;
  if (p->q_state != 0 || count <= (uint64_t)p->max_sge || count > (uint64_t)kBurstMax * (uint64_t)p->max_sge ||
      count > GRDMA_TX_MAX_RECORDS - 1)
    return 1;
  if (__atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE) < p->q_pending_seq) return 1;  // a skipped chain has yet to drain
  {
    // Will the write in front go out whole, and this one behind it?  A queued chain that is skipped costs three empty
    // launches, so it is only queued when the ring -- as the state line showed it at the last commit, minus what is in
    // flight -- has room for both (a guess: the device decides, the gate keeps the order either way).
    uint64_t front = 0, mine = 0;
    for (uint64_t i = p->w_idx; i < p->w_slices.size(); i++) front += 24 + p->w_slices[i].len;
    for (uint64_t i = 0; i < count; i++) mine += 24 + slices[i].len;
    const uint64_t tail = __atomic_load_n(&p->line->remote_tail, __ATOMIC_RELAXED);
    const uint64_t head = __atomic_load_n(&p->line->remote_head, __ATOMIC_RELAXED);
    const uint64_t used = (tail - head) & (p->ring_size - 1);
    static const bool always = [] { const char* e = getenv("GRDMA_WRITE_QUEUE_ALWAYS"); return e && atoi(e) != 0; }();  // (tests: the skip path)
    if (!always && used + front + mine + 4096 > p->ring_size) return 1;
  }
  const int set = 1 - p->inflight_set;
  if (!p->q_ops) {
    HIP_TRY(hipHostMalloc((void**)&p->q_ops, sizeof(grdma_tx_op) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&p->q_res, sizeof(grdma_tx_result) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&p->q_plan_ptrs, sizeof(grdma_plan*) * kBurstMax, hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void**)&p->q_sges, sizeof(grdma_sge) * GRDMA_TX_MAX_RECORDS, hipHostMallocCoherent | hipHostMallocMapped));
    memset(p->q_res, 0, sizeof(grdma_tx_result) * kBurstMax);
  }
  const uint32_t B = (uint32_t)((count + p->max_sge - 1) / p->max_sge);
  if (p->b_plans.size() < B) return 1;  // (the plans of a burst are allocated by the ordinary path: no allocation here,
                                        // it would wait for the stream)
  const burst_set S = bset(p, set), F = bset(p, p->inflight_set);
  for (uint64_t i = 0; i < count; i++) {
    S.sges[i].ptr = static_cast<const uint8_t*>(slices[i].ptr);
    S.sges[i].len = slices[i].len;
  }
  for (uint32_t k = 0; k < B; k++) {
    grdma_tx_op& t = S.ops[k];
    memset(&t, 0, sizeof(t));
    t.conn = p->d_conn;
    t.slices = S.sges;
    t.nslices = count;
    t.plan = p->b_plans[k];
    t.wire_plan = nullptr;
    t.result = &S.res[k];
    t.use_cursor = k == 0 ? 3 : 1;
    if (k == 0) t.byte_idx = (uint64_t)(uintptr_t)&F.res[p->tx_burst - 1];  // the gate: the last Send in front
    S.plan_ptrs[k] = p->b_plans[k];
  }
  const uint32_t blocks_b = std::max<uint32_t>(1, copy_blocks_for(p->ring_size) / B + 1);
  HIP_TRY(grdma_launch_tx_plan_seq(S.ops, 1, B, p->s_tx));
  HIP_TRY(grdma_launch_copy(S.plan_ptrs, B, blocks_b, p->s_tx));
  HIP_TRY(grdma_launch_tx_commit1(p->d_conn, ++p->tx_seq, p->s_tx));
  p->q_expect = p->tx_seq;
  p->q_burst = B;
  p->q_set = set;
  p->q_slices.assign(slices, slices + count);
  p->q_state = 1;
  p->q_queued++;
  return 0;
}

// After grdma_endpoint_write_test reported the write in front complete: 1 = the queued write was promoted -- it is the
// write outstanding now, its burst in flight (grdma_endpoint_write_test / _writable as usual) --, 0 = there is none
// (never queued, or skipped on the device: submit it the ordinary way).
int grdma_endpoint_write_adopt(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->q_state != 2 || !p->w_active || !p->tx_inflight.load(std::memory_order_acquire)) return 0;
  p->q_state = 0;
  return 1;
}

// Waits until nothing of this pair's is left in its send stream (a queued chain, a chain that skipped itself): the
// memory its Sends gather from may be released afterwards.
int grdma_endpoint_write_quiesce(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->s_tx) HIP_TRY(hipStreamSynchronize(p->s_tx));
  return 0;
}

int grdma_endpoint_write_queue_stats(grdma_pair* p, uint64_t out[3]) {
  if (!p || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  out[0] = p->q_queued;
  out[1] = p->q_promoted;
  out[2] = p->q_skipped;
  return 0;
}

namespace {
inline bool drain_complete(grdma_pair* p) {
  const uint64_t seen = p->rx_by_engine.load(std::memory_order_acquire)
                            ? __atomic_load_n(&p->h->rxres.seq, __ATOMIC_ACQUIRE)   // (the release store of the plan body)
                            : __atomic_load_n(&p->line->rx_seq, __ATOMIC_ACQUIRE);
  return seen >= p->rx_expect.load(std::memory_order_acquire);
}
}  // namespace

int grdma_endpoint_write_submit(grdma_pair* p) {
  if (int rc = require_ctx()) return rc;
  if (!p || !p->async) return fail(GRDMA_ERR_INVALID, "not an asynchronous endpoint (grdma_endpoint_set_async)");
  if (!p->w_active) return fail(GRDMA_ERR_INVALID, "no write has been begun");
  if (p->tx_inflight.load(std::memory_order_acquire)) return fail(GRDMA_ERR_INVALID, "a Send is still in flight");
  grdma_profiler profiler(GRDMA_STATS_TIME_PAIR_SEND);
  // (a queued chain that was skipped may still sit in the stream and will look at the tables this submit rewrites: it
  // is right behind the chain whose completion brought us here -- a few tens of microseconds, a rare path)
  while (!p->tx_by_engine && p->q_pending_seq && __atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE) < p->q_pending_seq) {
    const hipError_t qe = hipStreamQuery(p->s_tx);
    if (qe != hipSuccess && qe != hipErrorNotReady) return fail(GRDMA_ERR_HIP, "send stream: %s", hipGetErrorString(qe));
  }
  const uint64_t n = p->w_slices.size() - p->w_idx;
  if (int rc = stage_slices(p, p->w_slices.data() + p->w_idx, n, p->w_byte, p->w_flags)) return rc;
  p->tx_inflight.store(1, std::memory_order_release);
  if (int rc = submit_send(p, n, p->w_byte)) {
    p->tx_inflight.store(0);
    return rc;
  }
  return 0;
}

int grdma_endpoint_write_test(grdma_pair* p, int* done, int64_t* sent) {
  if (!p || !done) return fail(GRDMA_ERR_INVALID, "null argument");
  if (!p->tx_inflight.load(std::memory_order_acquire)) return fail(GRDMA_ERR_INVALID, "no Send in flight");
  const uint64_t seen = p->tx_by_engine ? __atomic_load_n(&p->h->txres.seq, __ATOMIC_ACQUIRE)
                                        : __atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE);
  if (seen < p->tx_expect) {
    if (!p->tx_by_engine && (++p->test_calls & 0xFFFF) == 0) {  // a failed launch would otherwise never be noticed
      const hipError_t e = hipStreamQuery(p->s_tx);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(GRDMA_ERR_HIP, "send stream: %s", hipGetErrorString(e));
    }
    return 0;
  }
  grdma_tx_result r = p->h->txres;
  if (p->tx_burst) {  // the last Send's result holds the cursor; what was sent is the sum over the Sends
    const grdma_tx_result* res = bset(p, p->inflight_set).res;
    r = res[p->tx_burst - 1];
    r.sent = 0;
    for (uint32_t k = 0; k < p->tx_burst; k++) r.sent += res[k].sent;
    if (r.slice_idx == 0) r.byte_idx += p->tx_burst_byte0;  // (still inside the slice the offset was folded into)
    p->tx_burst = 0;
  }
  p->w_idx += r.slice_idx;
  p->w_byte = r.byte_idx;
  *done = r.done == 1 ? 1 : 0;
  if (sent) *sent = (int64_t)r.sent;
  if (r.done == 1) {
    p->w_active = false;
    p->w_slices.clear();
  }
  if (p->q_state == 1) {
    if (r.done == 1) {
      // the write queued behind this one ran (its gate saw the same `done`): it is the write outstanding now
      p->w_slices.swap(p->q_slices);
      p->q_slices.clear();
      p->w_idx = 0;
      p->w_byte = 0;
      p->w_flags = 0;
      p->w_active = true;
      p->tx_expect = p->q_expect;
      p->tx_burst = p->q_burst;
      p->tx_burst_byte0 = 0;
      p->inflight_set = p->q_set;
      p->inflight_covers_all = true;
      p->q_state = 2;  // promoted, not yet adopted by the caller
      p->q_promoted++;
      return 1;        // (tx_inflight stays up: the promoted burst is in flight)
    }
    // the write in front came up short: the queued chain skips itself on the device; it is submitted again later
    p->q_state = 0;
    p->q_slices.clear();
    p->q_pending_seq = p->q_expect;
    p->q_skipped++;
  }
  p->tx_inflight.store(0, std::memory_order_release);
  return 1;
}

int grdma_endpoint_read_submit(grdma_pair* p, uint64_t max_reads) {
  if (int rc = require_ctx()) return rc;
  if (!p || !p->async) return fail(GRDMA_ERR_INVALID, "not an asynchronous endpoint (grdma_endpoint_set_async)");
  if (max_reads == 0) return fail(GRDMA_ERR_INVALID, "max_reads is zero");
  if (max_reads > GRDMA_MAX_SLICES) max_reads = GRDMA_MAX_SLICES;
  std::lock_guard<std::mutex> lk(p->rx_mu);
  // A drain is in flight already: the in-process peer's sender may have posted this pair's ARMED drain between the
  // caller's look at grdma_endpoint_drain_state and this call.  Not an error -- the readable edge comes when it is done.
  if (p->rx_inflight.load(std::memory_order_acquire) >= 0) return 0;
  if (p->watch_slot >= 0 && p->watch_expect != 0 && g_engine.wanted) {
    // the standing order is with a watcher and parked for want of a window: name one, or report that none is free
    watch_unpark(p);
    return p->watch_parked ? 1 : 0;
  }
  int w = -1;
  for (size_t i = 0; i < p->windows.size(); i++)
    if (p->windows[i]->refs.load(std::memory_order_acquire) == 1) { w = (int)i; break; }
  if (w < 0) return 1;  // the transport still holds slices of every window
  grdma_profiler profiler(GRDMA_STATS_TIME_PAIR_RECV);
  grdma_hostblk* h = p->h;
  fill_rxop(p, p->windows[w]->base, p->windows[w]->bytes, max_reads, 0);
  if (p->latency && g_engine.wanted && p->h_cmd_rx) {
    p->h_cmd_rx->rx = h->rxop;
    p->rx_expect.store(h->rxop.seq_next, std::memory_order_relaxed);
    p->rx_by_engine.store(1, std::memory_order_relaxed);
    p->rx_inflight.store(w, std::memory_order_release);
    if (int rc = engine_post(GRDMA_ENGINE_DRAIN_BLOCK, p->h_cmd_rx)) {
      p->rx_inflight.store(-1);
      return rc;
    }
    return 0;
  }
  // the launch chain: plan, scatter (not in latency mode, where the plan kernel scatters by itself), and -- as a
  // kernel of its own -- the word the host polls
  p->rx_expect.store(++p->rx_seq, std::memory_order_relaxed);
  p->rx_by_engine.store(0, std::memory_order_relaxed);
  p->rx_inflight.store(w, std::memory_order_release);
  // (GRDMA_ENDPOINT_RX_MULTI=1: the drain plan as many small workgroups -- a periodic stream's pass predicted and
  // verified instead of walked.  Off by default: measured through the vtable with host slices it changes nothing,
  // 14.8 / 11.3 against 15.1 GiB/s -- what bounds that path is the scatter into the pinned window and the PCIe
  // round trips of the host loop, not the planner, profiles/r04_notes)
  static const bool rx_mw = getenv("GRDMA_ENDPOINT_RX_MULTI") && atoi(getenv("GRDMA_ENDPOINT_RX_MULTI")) != 0;
  const bool use_mw = rx_mw && !p->latency && !(p->flags & GRDMA_WIRE_ORDERED);
  if (use_mw) {
    h->rx_limit = __atomic_load_n(&p->line->wire_tail, __ATOMIC_ACQUIRE);
    h->rxop.limit_ptr = &h->rx_limit;
  }
  hipError_t e = use_mw ? grdma_launch_rx_plan_mw(&h->rxop, 1, p->s_rx) : grdma_launch_rx_plan(&h->rxop, 1, p->s_rx);
  if (e == hipSuccess && !p->latency) e = grdma_launch_rx_apply(&h->rxop, 1, copy_blocks_for(p->ring_size), p->s_rx);
  if (e == hipSuccess) e = grdma_launch_rx_commit1(p->d_conn, p->rx_seq, p->s_rx);
  if (e != hipSuccess) {
    p->rx_inflight.store(-1);
    return fail(GRDMA_ERR_HIP, "drain launch failed: %s", hipGetErrorString(e));
  }
  return 0;
}

// An endpoint read that would block without a drain having been submitted for it (the host saw no message): the
// connection's read state takes note, in stream order with the drains (k_rx_idle).  Latency-mode pairs, whose drains
// go through the resident engine's mailbox, do not: their next read is sized afresh.
int grdma_endpoint_read_idle(grdma_pair* p) {
  if (int rc = require_ctx()) return rc;
  if (!p || !p->async) return fail(GRDMA_ERR_INVALID, "not an asynchronous endpoint (grdma_endpoint_set_async)");
  if (p->latency) return 0;
  std::lock_guard<std::mutex> lk(p->rx_mu);
  if (p->rx_inflight.load(std::memory_order_acquire) >= 0) return 0;  // (a drain is in flight: it will find what there is)
  HIP_TRY(grdma_launch_rx_idle(p->d_conn, p->s_rx));
  return 0;
}

int64_t grdma_endpoint_read_test(grdma_pair* p, grdma_read_slice* slices, uint64_t slices_cap, int* would_block,
                                 grdma_window** window) {
  if (!p || !slices || !window) return fail(GRDMA_ERR_INVALID, "null argument");
  const int w = p->rx_inflight.load(std::memory_order_acquire);
  if (w < 0) return fail(GRDMA_ERR_INVALID, "no drain in flight");
  if (!drain_complete(p)) {
    if ((++p->test_calls_rx & 0xFFFF) == 0 && !(p->latency && g_engine.wanted)) {
      const hipError_t e = hipStreamQuery(p->s_rx);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(GRDMA_ERR_HIP, "receive stream: %s", hipGetErrorString(e));
    }
    return -(int64_t)GRDMA_ERR_AGAIN;
  }
  const grdma_rx_result& r = p->h->rxres;
  if (r.nslices > slices_cap) return fail(GRDMA_ERR_CAPACITY, "the drain delivered %llu slices, the caller takes %llu",
                                          (unsigned long long)r.nslices, (unsigned long long)slices_cap);
  for (uint64_t i = 0; i < r.nslices; i++) {
    slices[i].off = p->h_slices[i].off;
    slices[i].len = p->h_slices[i].len;
  }
  if (would_block) *would_block = (int)r.would_block;
  grdma_window* win = p->windows[w];
  win->refs.fetch_add(1, std::memory_order_relaxed);  // the caller's reference
  *window = win;
  const int64_t n = (int64_t)r.nslices;
  if (p->watch_slot >= 0 && p->watch_expect != 0 && p->rx_by_engine.load() && p->rx_expect.load() == p->watch_expect) {
    // a completion of the standing order (k_watch): the order goes on -- into the next free window, named to the
    // watcher with the count of completions taken; none free: parked until the transport lets one go
    std::lock_guard<std::mutex> rl(p->rx_mu);
    p->watch_expect++;
    p->watch_taken++;
    p->watch_hits++;
    p->rx_inflight.store(-1, std::memory_order_release);
    p->watch_parked = true;
    watch_unpark(p);
    return n;
  }
  p->rx_inflight.store(-1, std::memory_order_release);
  return n;
}

int grdma_endpoint_readable(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->rx_inflight.load(std::memory_order_acquire) >= 0)  // a drain in flight: readable when it has completed
    return drain_complete(p) ? 1 : 0;
  return grdma_pair_has_message(p);
}

int grdma_endpoint_drain_state(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->rx_inflight.load(std::memory_order_acquire) < 0) return 0;
  return drain_complete(p) ? 2 : 1;
}

// 1 while an asynchronous Send or drain of this pair is on the device and has not completed yet: what an event loop
// that wants to run a connection to quiescence waits for (the edges themselves are grdma_endpoint_readable / _writable)
int grdma_endpoint_busy(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  // (a standing order waiting for bytes is not work in progress)
  const bool standing = p->watch_slot >= 0 && p->watch_expect != 0 && p->rx_expect.load() == p->watch_expect &&
                        grdma_pair_has_message(p) <= 0;
  if (p->rx_inflight.load(std::memory_order_acquire) >= 0 && !drain_complete(p) && !standing) return 1;
  if (p->tx_inflight.load(std::memory_order_acquire)) {
    const uint64_t seen = p->tx_by_engine ? __atomic_load_n(&p->h->txres.seq, __ATOMIC_ACQUIRE)
                                          : __atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE);
    if (seen < p->tx_expect) return 1;
  }
  return 0;
}

int grdma_endpoint_free_windows(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  int n = 0;
  for (grdma_window* w : p->windows)
    if (w->refs.load(std::memory_order_acquire) == 1) n++;
  return n;
}

int grdma_endpoint_writable(grdma_pair* p) {
  if (!p) return fail(GRDMA_ERR_INVALID, "null pair");
  if (p->tx_inflight.load(std::memory_order_acquire)) {
    const uint64_t seen = p->tx_by_engine ? __atomic_load_n(&p->h->txres.seq, __ATOMIC_ACQUIRE)
                                          : __atomic_load_n(&p->line->tx_seq, __ATOMIC_ACQUIRE);
    return seen >= p->tx_expect ? 1 : 0;
  }
  // HasPendingWrites(): the last Send came up short -- worth another one as soon as the peer has returned credit
  return grdma_pair_has_pending_writes(p) > 0 && grdma_pair_writable_size(p) > 0 ? 1 : 0;
}

// ---- small device helpers -------------------------------------------------------
void* grdma_device_alloc(uint64_t bytes) {
  if (require_ctx()) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    fail(GRDMA_ERR_HIP, "hipMalloc(%llu) failed", (unsigned long long)bytes);
    return nullptr;
  }
  return p;
}
void grdma_device_free(void* p) { if (p) hipFree(p); }
int grdma_host_pin_to_device_node(void) {
  // (no context needed: the calling thread's current HIP device -- device 0 in a process that has not chosen one, or the
  // one grdma_init selected)
  int dev = 0;
  char bus[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) return -1;
  for (char* c = bus; *c; c++) *c = (char)tolower(*c);
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  if (node < 0) return -1;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return -1;
  cpu_set_t set;
  CPU_ZERO(&set);
  int a = 0, b = 0, n = 0;
  while (fscanf(f, "%d", &a) == 1) {  // "0-63,128-191"
    b = a;
    int ch = fgetc(f);
    if (ch == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      ch = fgetc(f);
    }
    for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
      CPU_SET(c, &set);
      n++;
    }
    if (ch != ',') break;
  }
  fclose(f);
  if (n == 0 || sched_setaffinity(0, sizeof(set), &set) != 0) return -1;
  // sched_setaffinity(0) moves the CALLING thread (threads it creates later inherit).  The threads that exist already
  // -- the HIP runtime starts its signal / completion threads with the first HIP call, i.e. a few lines above -- stay
  // where they were started, and a completion thread on the other socket is what a slow run of the vtable leg looked
  // like (one in four or five): move every thread of the process.
  if (DIR* d = opendir("/proc/self/task")) {
    while (struct dirent* e = readdir(d)) {
      const long tid = strtol(e->d_name, nullptr, 10);
      if (tid > 0) sched_setaffinity((pid_t)tid, sizeof(set), &set);  // (a thread that has just exited: ignored)
    }
    closedir(d);
  }
  return node;
}
// The calling thread alone onto the k-th PHYSICAL core of the device's NUMA node (a CPU that is the first of its
// thread_siblings_list; k counts from the END of the node's list, away from where the kernel places new tasks first):
// two busy-polling threads of one process -- the writer and the reader of a streaming endpoint pair -- otherwise share
// a core's two hardware threads every few runs, and each then copies at ~0.7 of its speed.  Returns the CPU or -1.
int grdma_host_pin_thread_to_core(int k) {
  int dev = 0;
  char bus[64] = {0};
  if (k < 0 || hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) return -1;
  for (char* c = bus; *c; c++) *c = (char)tolower(*c);
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  if (node < 0) return -1;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  if (!f) return -1;
  std::vector<int> cores;
  int a = 0, b = 0;
  while (fscanf(f, "%d", &a) == 1) {
    b = a;
    int ch = fgetc(f);
    if (ch == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      ch = fgetc(f);
    }
    for (int c = a; c <= b && c < CPU_SETSIZE; c++) {
      char sp[128];
      snprintf(sp, sizeof(sp), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
      int first = c;
      if (FILE* g = fopen(sp, "r")) {
        if (fscanf(g, "%d", &first) != 1) first = c;
        fclose(g);
      }
      if (first == c) cores.push_back(c);
    }
    if (ch != ',') break;
  }
  fclose(f);
  if ((size_t)k >= cores.size()) return -1;
  const int cpu = cores[cores.size() - 1 - (size_t)k];
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  if (sched_setaffinity(0, sizeof(set), &set) != 0) return -1;
  return cpu;
}
void* grdma_host_alloc_pinned(uint64_t bytes) {
  if (require_ctx()) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return nullptr;
  return p;
}
void grdma_host_free_pinned(void* p) { if (p) hipHostFree(p); }
int grdma_copy_to_device(void* dst, const void* src, uint64_t n) {
  if (int rc = require_ctx()) return rc;
  HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
  return 0;
}
int grdma_copy_to_host(void* dst, const void* src, uint64_t n) {
  if (int rc = require_ctx()) return rc;
  HIP_TRY(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
  return 0;
}
int grdma_device_synchronize(void) {
  if (int rc = require_ctx()) return rc;
  if (int rc = engine_stop()) return rc;  // a resident engine would never let this return
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

}  // extern "C"

// ---- device-resident streaming job -------------------------------------------------
// n independent links (connections) advance in lock step: every launch carries one
// op per link (grid.y = n), so 32 connections with 4 MiB rings fill the chip the way
// one connection with a 128 MiB ring would.
struct grdma_job_link {
  grdma_pair* tx = nullptr;
  grdma_pair* rx = nullptr;
  grdma_sge* d_sges = nullptr;
  uint64_t count = 0;
  grdma_slice_out* d_slices = nullptr;
  uint64_t slices_cap = 0;
  uint8_t* dst = nullptr;
  uint64_t dst_cap = 0;
  // second copies of what two rounds in flight would otherwise share (pipelined mode)
  // index of the slice buffer (k_tx_index / k_tx_fast, grdma_tx_fast.hip): [count + 1] entries each
  uint64_t* d_encpre = nullptr;
  uint64_t* d_lenpre = nullptr;
  uint32_t* d_tilepre = nullptr;
  grdma_plan* d_wireplan2 = nullptr;
  grdma_plan* d_rxplan2 = nullptr;
  uint8_t* d_staging2 = nullptr;
  uint8_t* d_staging_n[2] = {nullptr, nullptr};  // grdma_stream_job_set_sends: both parities' staging for several Sends per plan
  // burst mode (several Sends per round): plans, staging buffers of Sends 1 .. burst-1
  std::vector<grdma_plan*> b_gplan, b_wplan;
  std::vector<uint8_t*> b_staging;
  // persistent link engine (k_link): control block, the three entry tables, extra staging buffers
  lk_ctl* d_lk = nullptr;
  lk_entry* d_tab[3] = {nullptr, nullptr, nullptr};
  std::vector<uint8_t*> d_staging_more;
};

struct grdma_stream_job {
  std::vector<grdma_job_link> links;
  uint64_t rounds = 0;
  // device control block: txop[3][n], rxop[3][n], results, plan pointer arrays.
  // Op set 0 is the first round (resets the cursors), sets 1 / 2 are odd / even rounds:
  // they differ in which of the doubled buffers (wire plan, staging, scatter plan,
  // drain result) they use, so that two rounds can be in flight.
  uint8_t* d_ctl = nullptr;
  grdma_tx_op* d_txop = nullptr;      // [3 * n]
  grdma_rx_op* d_rxop = nullptr;      // [3 * n]
  grdma_tx_result* d_txres = nullptr; // [n]
  grdma_rx_result* d_rxres = nullptr; // [2 * n]
  const grdma_plan** d_plans = nullptr;  // [3 * n]: gather, wire (even), wire (odd)
  grdma_size_hint* d_hints = nullptr; // [3 * n]: the record sizes the Send of a round computed, per op set: what the
                                      // drain of the same round predicts the ring's records from (grdma_rx_op::sizes_in)
  uint64_t* d_limits = nullptr;       // [3 * n]: remote_tail_ after the Send(s) of a round, per op set: what the
                                      // drain of the same round may walk up to (grdma_rx_op::limit_ptr)
  grdma_conn** d_txconns = nullptr;   // [n]: the sending ends, for the arrival report behind the last round
  hipGraphExec_t exec = nullptr;
  // kernel nodes another stage hangs in front of / behind the job inside its graph (grdma_job_set_hooks: the HTTP/2
  // pipe's framing and deframing): a chain in front of the first round, a chain behind k_tx_commit
  std::vector<grdma_job_hook> pre_hooks, post_hooks;
  uint64_t hooks_gen = 0, exec_hooks_gen = 0;
  uint64_t exec_rounds = 0;
  int exec_pipeline = -1;
  int exec_fastkey = -1;              // rx_fast | tx_fast << 1 | deep << 2 the graph was built for
  int slim_after = -1, runs = 0;      // experiment: job kernels only from run `slim_after` on
  int rx_miss = 0, tx_miss = 0;       // consecutive runs whose drains / Sends of link 0 mostly went to the general planner
  uint64_t seen[4] = {0, 0, 0, 0};    // the result blocks' taken / declined counters at the end of the last run
  int pipeline = 0;                   // 1: overlap the send plan / gather / scatter of
                                      // neighbouring rounds on side streams
  int deep = 1;                       // pipelined graph: 1 = the limit-driven schedule (job_build_graph),
                                      // 0 = the schedule of rounds 1-2 (GRDMA_JOB_SCHEDULE=pair)
  hipStream_t s_wire = nullptr, s_rxplan = nullptr, s_apply = nullptr;
  // reserved-CU schedule (job_enqueue_masked): the two planners on streams whose CU mask is a few CUs of
  // their own, the copy kernels on streams masked to the rest, so a one-workgroup planner never waits for a
  // machine-filling copy kernel to retire
  int tx_fast = 1;                    // Sends of one-Send rounds are priced from an index of the slice buffer: k_tx_index at
                                      // the start of a step, k_tx_fast per Send, the general planner behind it for the rest
  grdma_txf_ctl* d_txf = nullptr;     // [n]
  int pair_job = 1;                   // pipelined graph: drain of round t and Send of round t + 1 in one launch (k_plan_pair_job)
  int fuse_ag = 1;                    // paired schedule: the scatter of round t and the gather of round t + 1 in one launch
                                      // (k_rx_apply_gather); GRDMA_JOB_FUSE_AG=0: two launches
  int fuse = 0;                       // GRDMA_JOB_FUSE=1 (experiment, off): the planners ride in the copy kernels' grids
                                      // (k_wire_txplan_job, k_rxplan_gather_job), three launches per round.  Measured: no
                                      // gain -- a planner's dependent loads run ~2.3 x slower beside a copy that saturates
                                      // the memory system (profiles/r03_fused_schedule_experiment.txt)
  int rx_multi = 1;                   // paired schedule: the drain plan laid out by several workgroups (k_plan_pair_mw,
                                      // csrc/grdma_rx_multi.h); GRDMA_RX_MULTI=0: the one-workgroup k_plan_pair_job
  // The index of the slice table (k_tx_index) is a function of the table alone, and the job owns the table: built by
  // the first run, kept for the later ones -- unless something may rewrite the table between steps (kernel nodes hung
  // in front of the job, or a caller that asked where the table lives: the HTTP/2 pipe does both).
  bool index_valid = false, sges_exposed = false;
  int promise = 0;                    // grdma_stream_job_set_promised_credit: the Send of round t + 1 waits, inside the planner
                                      // pair's launch, for the drain plan of round t and is priced with the credit that
                                      // drain's scatter will post (k_plan_pair_mw) -- the paired schedule without its round
                                      // of credit lag
  uint32_t sends = 1;                 // grdma_stream_job_set_sends: consecutive Sends one round's plan holds (paired schedule,
                                      // planners of grdma_tx_multi.h / grdma_rx_multi.h: 16 workgroups per Send's worth of records)
  int fuse_round = 0;                 // GRDMA_JOB_FUSE_ROUND=1: the drain plan of round t, its scatter and the gather of round
                                      // t + 1 in ONE launch (k_round_xag, grdma_rx_plan.hip), the Send of round t + 1 priced
                                      // by a launch of its own in front of it
  int fuse_round_after = -1;          // (tools/ experiment) GRDMA_JOB_FUSE_ROUND_AFTER=k: fuse_round from run k on
  void** d_scratch = nullptr;         // [n] -> global scratch of the general planner inside k_round_xag
  std::vector<void*> scratch_bufs;
  int rx_fast = 1;                    // drains of one-Send rounds go through k_rx_fast first (grdma_rx_fast.hip), the
                                      // general planner behind it only does what that kernel declined
  int cumask_bits = 0;                // planner CUs (low bits of the mask); 0 = off
  hipStream_t m_txplan = nullptr, m_rxplan = nullptr, m_copy = nullptr, m_apply = nullptr;
  std::vector<hipEvent_t> mev;
  std::vector<hipEvent_t> pev;        // dependency events of the pipelined schedule
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> kev;
  std::vector<int> kev_cls;                   // GRDMA_RUN_INSTRUMENTED_SCHEDULE: the class of the launch behind mark i
  hipStream_t stream = nullptr;
  bool direct = false;
  uint64_t max_ring = 0;
  // burst mode: `burst` Sends per round, planned back to back by one k_tx_plan_seq launch, gathered
  // and put on the wire by one k_copy launch each (burst x n plans), drained by ONE receive pass
  uint32_t burst = 1;
  uint8_t* d_bctl = nullptr;
  grdma_tx_op* d_btxop = nullptr;          // [3 sets][burst][n]: the op sets of job_opset() (set 0 resets the cursor)
  const grdma_plan** d_bplans = nullptr;   // [burst * n] gather plans, then [burst * n] wire plans
  // link engine
  lk_ctl** d_lk_ptrs = nullptr;
  uint32_t lk_team = 0;
  uint64_t lk_timeout_ticks = 0;
};

namespace {

inline int job_opset(uint64_t round) { return round == 0 ? 0 : ((round & 1) ? 1 : 2); }
inline bool job_index_needed(const grdma_stream_job* j) { return !j->index_valid || !j->pre_hooks.empty() || j->sges_exposed; }
inline int job_fastkey(const grdma_stream_job* j) {
  return (j->rx_fast ? 1 : 0) | (j->tx_fast ? 2 : 0) | (j->deep ? 4 : 0) | (j->pair_job ? 8 : 0) | (j->fuse ? 16 : 0) |
         (j->fuse_ag ? 32 : 0) | (j->rx_multi ? 64 : 0) | (j->fuse_round ? 128 : 0) | ((int)(j->sends & 7) << 8) | (job_index_needed(j) ? (1 << 12) : 0) | (j->promise ? (1 << 13) : 0) | ((int)j->sends << 16);
}
// copy workgroups (1024 threads: one per CU) next to a planner workgroup in a fused launch: every CU but the planner's
inline uint32_t job_fused_copy_blocks() {
  static const uint32_t v = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2)
      return 255u;
    const char* e = getenv("GRDMA_FUSED_COPY_BLOCKS");
    const long o = e ? atol(e) : 0;
    return o > 0 ? (uint32_t)o : (uint32_t)(cus - 1);
  }();
  return v;
}
inline bool job_exec_stale(const grdma_stream_job* j) {
  return !j->exec || j->exec_rounds != j->rounds || j->exec_pipeline != j->pipeline || j->exec_fastkey != job_fastkey(j) ||
         j->exec_hooks_gen != j->hooks_gen;
}

// the send plan of round t: priced from the index of the slice buffer (built in front of the first round of a
// step), the general planner in the same launch for what that declines
inline bool job_tx_fast(const grdma_stream_job* j) { return j->tx_fast && j->burst == 1; }
// planner workgroups of a round: sixteen per Send's worth of records
// (up to two Sends: priced one after the other, each may carry 4095 records.  More: folded into one cut of the index,
//  a round carries at most sends x max_sge records -- 256 of them per workgroup)
inline uint32_t job_groups(const grdma_stream_job* j, uint32_t per_send) {
  if (j->sends <= grdma_tx_multi_seq_sends()) return per_send * j->sends;
  uint64_t most = 1;
  for (const grdma_job_link& l : j->links) most = std::max<uint64_t>(most, (uint64_t)j->sends * l.tx->max_sge);
  return (uint32_t)std::min<uint64_t>(2 * per_send, (most + 255) / 256);
}
inline uint32_t job_rx_groups(const grdma_stream_job* j) { return job_groups(j, grdma_rx_multi_groups()); }
// (promised credit: every planner workgroup of the launch must be resident at once -- one per CU -- or a Send's
//  workgroups could wait for a drain whose workgroups have no CU yet)
inline bool job_promise(const grdma_stream_job* j);
inline uint32_t job_tx_groups(const grdma_stream_job* j) { return job_groups(j, grdma_tx_multi_groups()); }
inline bool job_mw(const grdma_stream_job* j) { return j->rx_multi && j->pipeline && j->pair_job && !j->fuse && j->rx_fast && j->burst == 1 && j->tx_fast; }
// the sequential schedule (five launches per round, strictly in order) with the small planner workgroups: what carries
// several Sends per plan when the job is not pipelined -- a ring every round fills sees its credit at once here, a round
// late on the paired schedule
inline bool job_mw_seq(const grdma_stream_job* j) { return j->sends > 1 && j->rx_multi && !j->pipeline && j->rx_fast && j->burst == 1 && j->tx_fast; }
hipError_t job_launch_pair_mw(const grdma_rx_op* rxops, const grdma_tx_op* txops, const grdma_txf_ctl* ctls, uint32_t n, uint32_t g_rx,
                              uint32_t g_tx, hipStream_t s) {
  // (an array of GRDMA_JOB_HOOK_ARGS entries, as for every kernel launched by address: the runtime reads as many as the kernel has)
  static uint64_t none = 0;
  void* args[GRDMA_JOB_HOOK_ARGS];
  for (uint32_t a = 0; a < GRDMA_JOB_HOOK_ARGS; a++) args[a] = &none;
  args[0] = (void*)&rxops; args[1] = (void*)&txops; args[2] = (void*)&ctls; args[3] = (void*)&g_rx;
  return hipLaunchKernel(grdma_kernel_fn_plan_pair_mw(), dim3(n, g_rx + g_tx), dim3(grdma_kernel_threads(0)), args, 0, s);
}
inline bool job_promise(const grdma_stream_job* j) {
  // (staged wire only: with a direct wire the gather of round t + 1 writes the ring in the launch of round t's scatter)
  if (!j->promise || !job_mw(j) || j->direct) return false;
  static const int cus = [] {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0;
    return c;
  }();
#ifdef GRDMA_WAVE_EMU
  return true;  // (the emulator runs the workgroups of a launch one after the other, in index order: the drain's first)
#else
  return (uint64_t)j->links.size() * (job_rx_groups(j) + job_tx_groups(j)) <= (uint64_t)cus;
#endif
}
inline uint32_t job_pair_mode(const grdma_stream_job* j) { return job_rx_groups(j) | (job_promise(j) ? (1u << 16) : 0u); }
inline uint32_t job_index_blocks(const grdma_stream_job* j) {  // k_tx_index: 1024 slices per workgroup
  uint64_t most = 1;
  for (const grdma_job_link& l : j->links) most = std::max<uint64_t>(most, l.count);
  return (uint32_t)((most + 1023) / 1024);
}
hipError_t job_launch_tx_plan(grdma_stream_job* j, int k, uint64_t t, uint32_t n, hipStream_t s) {
  if (!job_tx_fast(j)) return grdma_launch_tx_plan(j->d_txop + k * n, n, s);
  hipError_t e = hipSuccess;
  if (t == 0 && job_index_needed(j)) e = grdma_launch_tx_index(j->d_txf, n, job_index_blocks(j), s);
  if (e != hipSuccess) return e;
  // (several Sends per plan: only the planners of grdma_tx_multi.h price those -- also in the eager passes)
  if (j->sends > 1 && (job_mw(j) || job_mw_seq(j))) return job_launch_pair_mw(nullptr, j->d_txop + k * n, j->d_txf, n, 0, job_tx_groups(j), s);
  return grdma_launch_tx_plan_job(j->d_txop + k * n, j->d_txf, n, s);
}

// the receive plan of a round: k_rx_plan_job = the straight-line steady-state body, then the general planner for what it declines
hipError_t job_launch_rx_plan(grdma_stream_job* j, const grdma_rx_op* ops, uint32_t n, hipStream_t s) {
  if (j->sends > 1 && (job_mw(j) || job_mw_seq(j))) return job_launch_pair_mw(ops, nullptr, j->d_txf, n, job_rx_groups(j), 0, s);
  if (j->rx_fast && j->burst == 1) return grdma_launch_rx_plan_job(ops, n, s);
  return grdma_launch_rx_plan(ops, n, s);
}

int job_enqueue(grdma_stream_job* j, hipStream_t s, bool instrument) {
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t tx_blocks = copy_blocks_for(j->max_ring / 2);
  const uint32_t rx_blocks = copy_blocks_for(j->max_ring);
  // keep the grid around 2048 workgroups in total
  const uint32_t grid_cap = copy_blocks_for(~0ull >> 8);
  const uint32_t txb = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / n + 1));
  const uint32_t rxb = std::max<uint32_t>(1, std::min<uint32_t>(rx_blocks, grid_cap / n + 1));
  size_t e = 0;
  auto mark = [&]() -> int {
    if (!instrument) return 0;
    if (e >= j->kev.size()) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      j->kev.push_back(ev);
    }
    HIP_TRY(hipEventRecord(j->kev[e++], s));
    return 0;
  };
  if (int rc = mark()) return rc;
  const uint32_t B = j->burst;
  const uint32_t txb_b = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / (n * B) + 1));
  for (uint64_t r = 0; r < j->rounds; r++) {
    const int k = job_opset(r);
    if (B > 1) {
      HIP_TRY(grdma_launch_tx_plan_seq(j->d_btxop + (size_t)k * B * n, n, B, s));
      if (int rc = mark()) return rc;
      HIP_TRY(grdma_launch_copy(j->d_bplans, B * n, txb_b, s));
      if (int rc = mark()) return rc;
      if (!j->direct) HIP_TRY(grdma_launch_copy(j->d_bplans + (size_t)B * n, B * n, txb_b, s));
      if (int rc = mark()) return rc;
    } else {
    HIP_TRY(job_launch_tx_plan(j, k, r, n, s));
    if (int rc = mark()) return rc;
    HIP_TRY(grdma_launch_copy(j->d_plans, n, txb, s));
    if (int rc = mark()) return rc;
    if (!j->direct) HIP_TRY(grdma_launch_copy(j->d_plans + n * (1 + (r & 1)), n, txb, s));
    if (int rc = mark()) return rc;
    }
    HIP_TRY(job_launch_rx_plan(j, j->d_rxop + k * n, n, s));
    if (int rc = mark()) return rc;
    HIP_TRY(grdma_launch_rx_apply(j->d_rxop + k * n, n, rxb, s));
    if (int rc = mark()) return rc;
  }
  // the drains of the job were told how far to walk by their op (limit_ptr); the connection's own arrival
  // report and the state lines follow once, behind the last round
  HIP_TRY(grdma_launch_tx_commit(j->d_txconns, nullptr, n, s));
  return 0;
}

// GRDMA_RUN_INSTRUMENTED_SCHEDULE: the launches of the DEFAULT schedule of a streaming job -- the chain the graph
// builder below makes of a paired job (planner pair; scatter + next gather; wire) -- one after the other on one
// stream with an event between every two of them.  The graph's order is a chain already, so this is the same
// work in the same order; what the events add is the time of each launch by itself (classes 5 = k_plan_pair_job,
// 6 = k_rx_apply_gather beside the five of the in-order pass).
// k_round_xag's grid: per link G planner, GB gather and SB scatter workgroups (shape = G | links << 8 | SB << 16)
struct job_xag_shape { uint32_t grid, shape; };
inline bool job_round_fused(const grdma_stream_job* j) { return j->fuse_round && j->rx_multi && j->fuse_ag && j->links.size() < 256; }
job_xag_shape job_xag(const grdma_stream_job* j, uint32_t txb, uint32_t rxb, bool gather) {
  const uint32_t n = (uint32_t)j->links.size(), G = job_rx_groups(j);
  static const uint32_t resident = grdma_round_xag_resident_blocks();
  const uint32_t cap = std::max<uint32_t>(1, resident / n);
  const uint32_t GB = gather ? std::max<uint32_t>(1, std::min(txb, cap)) : 0;
  const uint32_t SB = std::max<uint32_t>(1, std::min<uint32_t>(std::min(rxb, cap), 0xFFFFu));
  return {n * (G + GB + SB), G | (n << 8) | (SB << 16)};
}
bool job_is_paired(const grdma_stream_job* j) {
  return j->pipeline && j->burst == 1 && j->rx_fast && job_tx_fast(j) && j->pair_job && !j->fuse && j->rounds >= 1;
}
int job_enqueue_schedule_instrumented(grdma_stream_job* j, hipStream_t s) {
  if (!job_is_paired(j))
    return fail(GRDMA_ERR_INVALID, "GRDMA_RUN_INSTRUMENTED_SCHEDULE times the paired schedule (pipelined job, steady-state planners)");
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t tx_blocks = copy_blocks_for(j->max_ring / 2);
  const uint32_t rx_blocks = copy_blocks_for(j->max_ring);
  const uint32_t grid_cap = copy_blocks_for(~0ull >> 8);
  const uint32_t txb = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / n + 1));
  const uint32_t rxb = std::max<uint32_t>(1, std::min<uint32_t>(rx_blocks, grid_cap / n + 1));
  const uint32_t ct = grdma_kernel_threads(1);
  const uint64_t R = j->rounds;
  size_t e = 0;
  j->kev_cls.clear();
  auto mark = [&](int cls) -> int {
    if (e >= j->kev.size()) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      j->kev.push_back(ev);
    }
    HIP_TRY(hipEventRecord(j->kev[e++], s));
    if (cls >= 0) j->kev_cls.push_back(cls);
    return 0;
  };
  auto launch = [&](const void* fn, dim3 grid, uint32_t threads, const void* a0, const void* a1, const void* a2,
                    uint32_t a3 = 0) -> hipError_t {
    static uint64_t none = 0;
    void* args[GRDMA_JOB_HOOK_ARGS];
    for (uint32_t a = 0; a < GRDMA_JOB_HOOK_ARGS; a++) args[a] = &none;
    args[0] = const_cast<void*>(static_cast<const void*>(&a0));
    args[1] = const_cast<void*>(static_cast<const void*>(&a1));
    args[2] = const_cast<void*>(static_cast<const void*>(&a2));
    args[3] = &a3;
    return hipLaunchKernel(fn, grid, dim3(threads), args, 0, s);
  };
  if (int rc = mark(-1)) return rc;
  for (uint64_t t = 0; t < R; t++) {
    const int k = job_opset(t);
    const void* rxop = j->d_rxop + k * n;
    const void* gplans = j->d_plans;
    const void* wplans = j->d_plans + n * (1 + (t & 1));
    if (t == 0) {
      if (j->rx_multi) {  // (k_tx_index +) the Send priced by k_plan_pair_mw's small workgroups (as the graph does)
        if (job_index_needed(j)) HIP_TRY(grdma_launch_tx_index(j->d_txf, n, job_index_blocks(j), s));
        HIP_TRY(launch(grdma_kernel_fn_plan_pair_mw(), dim3(n, job_tx_groups(j)), grdma_kernel_threads(0), nullptr,
                       j->d_txop + k * n, j->d_txf, 0u));
      } else {
        HIP_TRY(job_launch_tx_plan(j, k, 0, n, s));  // k_tx_index + k_tx_plan_job
      }
      if (int rc = mark(0)) return rc;
    }
    if (t == 0 || !j->fuse_ag) {
      HIP_TRY(launch(grdma_kernel_fn(1), dim3(txb, n), ct, gplans, nullptr, nullptr));
      if (int rc = mark(1)) return rc;
    }
    if (!j->direct) {
      HIP_TRY(launch(grdma_kernel_fn(1), dim3(txb, n), ct, wplans, nullptr, nullptr));
      if (int rc = mark(2)) return rc;
    }
    const bool more = t + 1 < R;
    const void* txop_next = j->d_txop + job_opset(t + 1) * n;
    if (job_round_fused(j)) {  // (class 5 = the Send's planners alone, class 6 = k_round_xag)
      if (more) {
        HIP_TRY(launch(grdma_kernel_fn_plan_pair_mw(), dim3(n, job_tx_groups(j)), grdma_kernel_threads(0), nullptr, txop_next,
                       j->d_txf, 0u));
        if (int rc = mark(5)) return rc;
      }
      const job_xag_shape xs = job_xag(j, txb, rxb, more);
      HIP_TRY(launch(grdma_kernel_fn_round_xag(), dim3(xs.grid), grdma_kernel_threads(0), rxop, gplans, j->d_scratch, xs.shape));
      if (int rc = mark(6)) return rc;
      continue;
    }
    if (j->rx_multi)
      HIP_TRY(launch(grdma_kernel_fn_plan_pair_mw(), dim3(n, job_rx_groups(j) + (more ? job_tx_groups(j) : 0)),
                     grdma_kernel_threads(0), rxop, more ? txop_next : nullptr, j->d_txf, job_pair_mode(j)));
    else
    HIP_TRY(launch(grdma_kernel_fn_plan_pair_job(), dim3(n, more ? 2 : 1), grdma_rx_plan_job_threads(), rxop,
                   more ? txop_next : nullptr, j->d_txf));
    if (int rc = mark(5)) return rc;
    if (more && j->fuse_ag) {
      HIP_TRY(launch(grdma_kernel_fn(8), dim3(std::max(rxb, txb), 2 * n), ct, rxop, gplans, nullptr));
      if (int rc = mark(6)) return rc;
    } else {
      HIP_TRY(launch(grdma_kernel_fn(3), dim3(rxb, n), ct, rxop, nullptr, nullptr));
      if (int rc = mark(4)) return rc;
    }
  }
  HIP_TRY(grdma_launch_tx_commit(j->d_txconns, nullptr, n, s));
  return 0;
}

// The same five kernels per round, scheduled as a software pipeline over four streams.
// What has to stay ordered (t = round):
//   plan_t -> gather_t -> wire_t -> rx_plan_t -> rx_apply_t      the data path of one round
//   wire_{t-2} -> plan_t      two staging buffers (and wire plans) alternate; the one of
//       this parity comes free when the round before last has left it
//   rx_plan_{t-1} -> wire_t   the loop-back wire is a parallel copy: it does not deliver
//       the footer of a record after its payload the way an RC queue pair does, so the
//       receiver must not be walking the chain while new records land behind it
//   rx_apply_{t-2} -> rx_plan_t   the scatter plan and result block of that parity are free
//   rx_apply_{t-2} -> plan_t      the sender sees every credit but (possibly) the last one
// Everything else overlaps: the send plan and gather of round t+1 run while round t is on
// the wire and being walked, and the scatter of round t runs under round t+1.  The
// sender may see the credit of a scatter one round later than in the sequential
// schedule; with rounds of at most ring/6 that never limits a Send.
// Round 3 (j->deep, the default): every drain of a job walks only up to the tail its own Send computed
// (grdma_rx_op::limit_ptr), so the third rule is dropped -- see the graph builder below for the edges.
int job_enqueue_pipelined(grdma_stream_job* j, hipStream_t s) {
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t tx_blocks = copy_blocks_for(j->max_ring / 2);
  const uint32_t rx_blocks = copy_blocks_for(j->max_ring);
  const uint32_t grid_cap = copy_blocks_for(~0ull >> 8);
  const uint32_t txb = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / n + 1));
  const uint32_t rxb = std::max<uint32_t>(1, std::min<uint32_t>(rx_blocks, grid_cap / n + 1));
  const uint64_t R = j->rounds;
  if (!j->s_wire) {
    HIP_TRY(hipStreamCreateWithFlags(&j->s_wire, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&j->s_rxplan, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&j->s_apply, hipStreamNonBlocking));
  }
  while (j->pev.size() < 4 * R + 1) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    j->pev.push_back(ev);
  }
  auto evG = [&](uint64_t t) { return j->pev[4 * t]; };      // gather_t done
  auto evW = [&](uint64_t t) { return j->pev[4 * t + 1]; };  // round t is in the ring
  auto evX = [&](uint64_t t) { return j->pev[4 * t + 2]; };  // rx_plan_t done
  auto evA = [&](uint64_t t) { return j->pev[4 * t + 3]; };  // rx_apply_t done
  hipStream_t sW = j->direct ? s : j->s_wire, sX = j->s_rxplan, sA = j->s_apply;
  hipEvent_t fork = j->pev[4 * R];
  HIP_TRY(hipEventRecord(fork, s));
  if (!j->direct) HIP_TRY(hipStreamWaitEvent(sW, fork, 0));
  HIP_TRY(hipStreamWaitEvent(sX, fork, 0));
  HIP_TRY(hipStreamWaitEvent(sA, fork, 0));
  for (uint64_t t = 0; t < R; t++) {
    const int k = job_opset(t);
    if (j->direct) {
      // the plan itself writes the tags into the peer ring: no part of round t may start
      // before the receiver has finished walking round t-1
      // (limit-driven schedule: the drain walks up to its round's tail, only the credit lag is bounded)
      if (j->deep) {
        if (t >= 2) HIP_TRY(hipStreamWaitEvent(s, evA(t - 2), 0));
      } else if (t >= 1) {
        HIP_TRY(hipStreamWaitEvent(s, evX(t - 1), 0));
      }
      HIP_TRY(job_launch_tx_plan(j, k, t, n, s));
      HIP_TRY(grdma_launch_copy(j->d_plans, n, txb, s));
      HIP_TRY(hipEventRecord(evW(t), s));
    } else {
      if (t >= 2) {
        HIP_TRY(hipStreamWaitEvent(s, evW(t - 2), 0));
        HIP_TRY(hipStreamWaitEvent(s, evA(t - 2), 0));  // bounds the credit lag to one round
      }
      HIP_TRY(job_launch_tx_plan(j, k, t, n, s));
      HIP_TRY(grdma_launch_copy(j->d_plans, n, txb, s));
      HIP_TRY(hipEventRecord(evG(t), s));
      HIP_TRY(hipStreamWaitEvent(sW, evG(t), 0));
      if (t >= 1 && !j->deep) HIP_TRY(hipStreamWaitEvent(sW, evX(t - 1), 0));
      HIP_TRY(grdma_launch_copy(j->d_plans + n * (1 + (t & 1)), n, txb, sW));
      HIP_TRY(hipEventRecord(evW(t), sW));
    }
    HIP_TRY(hipStreamWaitEvent(sX, evW(t), 0));
    if (t >= 2) HIP_TRY(hipStreamWaitEvent(sX, evA(t - 2), 0));
    HIP_TRY(job_launch_rx_plan(j, j->d_rxop + k * n, n, sX));
    HIP_TRY(hipEventRecord(evX(t), sX));
    HIP_TRY(hipStreamWaitEvent(sA, evX(t), 0));
    HIP_TRY(grdma_launch_rx_apply(j->d_rxop + k * n, n, rxb, sA));
    HIP_TRY(hipEventRecord(evA(t), sA));
  }
  // join the side streams back into the launch stream
  if (R > 0) {
    if (!j->direct) HIP_TRY(hipStreamWaitEvent(s, evW(R - 1), 0));
    HIP_TRY(hipStreamWaitEvent(s, evX(R - 1), 0));
    HIP_TRY(hipStreamWaitEvent(s, evA(R - 1), 0));
  }
  HIP_TRY(grdma_launch_tx_commit(j->d_txconns, nullptr, n, s));
  return 0;
}


// The limit-driven schedule on streams with CU masks.  A planner is ONE workgroup that needs most of a
// CU's register file; the copy kernels are grid-strided over every slot of the machine and give none back
// before they end, so a planner that becomes ready while a copy kernel is resident starts only behind it --
// on a shared machine the planners serialise with the copies whatever the dependency edges say (measured:
// the limit-driven graph is SLOWER than the paired one, 1.14 vs 1.04 ms per step).  Here the planners own
// `cumask_bits` CUs (hipExtStreamCreateWithCUMask) and the copies run on the rest:
//   m_txplan: P_t   after W_{t-2}, A_{t-2}, G_{t-1}
//   m_copy:   G_t after P_t, then W_t                       (in stream order; they fill the machine anyway)
//   m_rxplan: X_t   after W_t, A_{t-2}
//   m_apply:  A_t   after X_t
int job_enqueue_masked(grdma_stream_job* j, hipStream_t s) {
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t tx_blocks = copy_blocks_for(j->max_ring / 2);
  const uint32_t rx_blocks = copy_blocks_for(j->max_ring);
  const uint32_t grid_cap = copy_blocks_for(~0ull >> 8);
  const uint32_t txb = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / n + 1));
  const uint32_t rxb = std::max<uint32_t>(1, std::min<uint32_t>(rx_blocks, grid_cap / n + 1));
  const uint64_t R = j->rounds;
  if (!j->m_txplan) {
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int words = (cus + 31) / 32;
    std::vector<uint32_t> mp(words, 0u), mc(words, 0xFFFFFFFFu);
    for (int b = 0; b < j->cumask_bits && b < cus; b++) {
      mp[b / 32] |= 1u << (b % 32);
      mc[b / 32] &= ~(1u << (b % 32));
    }
    HIP_TRY(hipExtStreamCreateWithCUMask(&j->m_txplan, (uint32_t)words, mp.data()));
    HIP_TRY(hipExtStreamCreateWithCUMask(&j->m_rxplan, (uint32_t)words, mp.data()));
    HIP_TRY(hipExtStreamCreateWithCUMask(&j->m_copy, (uint32_t)words, mc.data()));
    HIP_TRY(hipExtStreamCreateWithCUMask(&j->m_apply, (uint32_t)words, mc.data()));
  }
  while (j->mev.size() < 5 * R + 1) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    j->mev.push_back(ev);
  }
  auto evP = [&](uint64_t t) { return j->mev[5 * t]; };
  auto evG = [&](uint64_t t) { return j->mev[5 * t + 1]; };
  auto evW = [&](uint64_t t) { return j->mev[5 * t + 2]; };  // round t is in the ring
  auto evX = [&](uint64_t t) { return j->mev[5 * t + 3]; };
  auto evA = [&](uint64_t t) { return j->mev[5 * t + 4]; };
  hipStream_t sP = j->m_txplan, sC = j->m_copy, sX = j->m_rxplan, sA = j->m_apply;
  hipEvent_t fork = j->mev[5 * R];
  HIP_TRY(hipEventRecord(fork, s));
  for (hipStream_t st : {sP, sC, sX, sA}) HIP_TRY(hipStreamWaitEvent(st, fork, 0));
  for (uint64_t t = 0; t < R; t++) {
    const int k = job_opset(t);
    if (t >= 1) HIP_TRY(hipStreamWaitEvent(sP, evG(t - 1), 0));
    if (t >= 2) {
      HIP_TRY(hipStreamWaitEvent(sP, evW(t - 2), 0));
      HIP_TRY(hipStreamWaitEvent(sP, evA(t - 2), 0));
    }
    HIP_TRY(job_launch_tx_plan(j, k, t, n, sP));
    HIP_TRY(hipEventRecord(evP(t), sP));
    HIP_TRY(hipStreamWaitEvent(sC, evP(t), 0));
    HIP_TRY(grdma_launch_copy(j->d_plans, n, txb, sC));
    HIP_TRY(hipEventRecord(evG(t), sC));
    if (!j->direct) HIP_TRY(grdma_launch_copy(j->d_plans + n * (1 + (t & 1)), n, txb, sC));
    HIP_TRY(hipEventRecord(evW(t), sC));
    HIP_TRY(hipStreamWaitEvent(sX, evW(t), 0));
    if (t >= 2) HIP_TRY(hipStreamWaitEvent(sX, evA(t - 2), 0));
    HIP_TRY(job_launch_rx_plan(j, j->d_rxop + k * n, n, sX));
    HIP_TRY(hipEventRecord(evX(t), sX));
    HIP_TRY(hipStreamWaitEvent(sA, evX(t), 0));
    HIP_TRY(grdma_launch_rx_apply(j->d_rxop + k * n, n, rxb, sA));
    HIP_TRY(hipEventRecord(evA(t), sA));
  }
  if (R > 0) {
    HIP_TRY(hipStreamWaitEvent(s, evW(R - 1), 0));
    HIP_TRY(hipStreamWaitEvent(s, evA(R - 1), 0));
    HIP_TRY(hipStreamWaitEvent(s, evP(R - 1), 0));
  }
  HIP_TRY(grdma_launch_tx_commit(j->d_txconns, nullptr, n, s));
  return 0;
}


// The job as an explicitly built HIP graph: 5 kernel nodes per round, edges exactly as
// listed above (pipelined) or a plain chain (sequential).  Built node by node rather
// than recorded from the streams: the dependency structure is known here, and it keeps
// the replay independent of how a runtime records cross-stream joins.
int job_build_graph(grdma_stream_job* j, hipGraph_t* out) {
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t tx_blocks = copy_blocks_for(j->max_ring / 2);
  const uint32_t rx_blocks = copy_blocks_for(j->max_ring);
  const uint32_t grid_cap = copy_blocks_for(~0ull >> 8);
  const uint32_t txb = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / n + 1));
  const uint32_t rxb = std::max<uint32_t>(1, std::min<uint32_t>(rx_blocks, grid_cap / n + 1));
  const uint64_t R = j->rounds;
  hipGraph_t g;
  HIP_TRY(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> P(R), G(R), W(R), X(R), A(R);
  // hook nodes: a chain of kernels; `after` (may be null) is what the first one waits for, the last one is returned
  hipError_t hook_err = hipSuccess;
  auto add_hooks = [&](std::vector<grdma_job_hook>& hooks, hipGraphNode_t after) -> hipGraphNode_t {
    for (grdma_job_hook& h : hooks) {
      void* args[GRDMA_JOB_HOOK_ARGS];
      for (uint32_t a = 0; a < GRDMA_JOB_HOOK_ARGS; a++) args[a] = &h.args[a];
      hipKernelNodeParams np;
      memset(&np, 0, sizeof(np));
      np.func = const_cast<void*>(h.fn);
      np.gridDim = dim3(h.grid);
      np.blockDim = dim3(h.threads);
      np.kernelParams = args;
      hipGraphNode_t node = nullptr;
      const hipError_t he = hipGraphAddKernelNode(&node, g, after ? &after : nullptr, after ? 1 : 0, &np);
      if (he != hipSuccess) {
        hook_err = he;
        return after;
      }
      after = node;
    }
    return after;
  };
  const hipGraphNode_t pre_last = R > 0 ? add_hooks(j->pre_hooks, nullptr) : nullptr;
  // (every node hands over three pointer-sized parameters; a kernel with fewer ignores the rest)
  auto add3 = [&](hipGraphNode_t* node, const void* fn, dim3 grid, uint32_t threads, const void* arg, const void* arg2,
                  const void* arg3, std::initializer_list<hipGraphNode_t> deps, uint32_t arg4 = 0) -> hipError_t {
    std::vector<hipGraphNode_t> d;
    for (hipGraphNode_t x : deps)
      if (x && std::find(d.begin(), d.end(), x) == d.end()) d.push_back(x);  // (a node twice is an invalid argument)
    if (d.empty() && pre_last) d.push_back(pre_last);  // a root of the job waits for the stage in front of it
    // (an array of GRDMA_JOB_HOOK_ARGS entries for every node -- the runtime reads as many as the kernel has)
    static uint64_t none = 0;
    void* args[GRDMA_JOB_HOOK_ARGS];
    for (uint32_t a = 0; a < GRDMA_JOB_HOOK_ARGS; a++) args[a] = &none;
    args[0] = const_cast<void*>(static_cast<const void*>(&arg));
    args[1] = const_cast<void*>(static_cast<const void*>(&arg2));
    args[2] = const_cast<void*>(static_cast<const void*>(&arg3));
    args[3] = &arg4;  // (a fourth, 4-byte parameter: k_plan_pair_mw's workgroup split)
    hipKernelNodeParams np;
    memset(&np, 0, sizeof(np));
    np.func = const_cast<void*>(fn);
    np.gridDim = grid;
    np.blockDim = dim3(threads);
    np.kernelParams = args;
    return hipGraphAddKernelNode(node, g, d.empty() ? nullptr : d.data(), d.size(), &np);
  };
  auto add2 = [&](hipGraphNode_t* node, const void* fn, dim3 grid, uint32_t threads, const void* arg, const void* arg2,
                  std::initializer_list<hipGraphNode_t> deps) -> hipError_t {
    return add3(node, fn, grid, threads, arg, arg2, nullptr, deps);
  };
  auto add = [&](hipGraphNode_t* node, const void* fn, dim3 grid, uint32_t threads, const void* arg,
                 std::initializer_list<hipGraphNode_t> deps) -> hipError_t {
    return add2(node, fn, grid, threads, arg, nullptr, deps);
  };
  const void* f_pair = grdma_kernel_fn_plan_pair();
  const void* f_txp = grdma_kernel_fn(0);
  const void* f_cpy = grdma_kernel_fn(1);
  const void* f_rxp = grdma_kernel_fn_rx_plan();
  const void* f_rxa = grdma_kernel_fn(3);
  const uint32_t pt = grdma_kernel_threads(0), ct = grdma_kernel_threads(1);
  const bool fast = j->rx_fast && j->burst == 1;
  const bool tfast = job_tx_fast(j);
  const void* f_txi = grdma_kernel_fn_tx_index();
  const void* f_txj = grdma_kernel_fn(6);
  const void* f_rxj = grdma_kernel_fn_rx_plan_job();
  // P[t] = the send plan of round t: (the index of the slice buffer in front of round 0,) then k_tx_plan_job --
  // the Send priced from the index, the general planner behind it in the same launch for what that declines
  auto add_tx = [&](uint64_t t, const void* txop, std::initializer_list<hipGraphNode_t> deps) -> hipError_t {
    if (!tfast) return add(&P[t], f_txp, dim3(n), pt, txop, deps);
    if (t != 0 && job_mw_seq(j)) {  // (several Sends per plan, sequential schedule: the Send's planners alone)
      std::vector<hipGraphNode_t> dq(deps);
      dq.resize(4, nullptr);
      return add3(&P[t], grdma_kernel_fn_plan_pair_mw(), dim3(n, job_tx_groups(j)), grdma_kernel_threads(0), nullptr, txop, j->d_txf,
                  {dq[0], dq[1], dq[2], dq[3]}, 0u);
    }
    if (t != 0) return add2(&P[t], f_txj, dim3(n), grdma_tx_plan_job_threads(), txop, j->d_txf, deps);
    hipGraphNode_t pi = nullptr;
    if (job_index_needed(j)) {  // (the slice table's index: once per job unless the table may change between steps)
      hipError_t e2 = add(&pi, f_txi, dim3(job_index_blocks(j), n), grdma_tx_index_threads(), j->d_txf, deps);
      if (e2 != hipSuccess) return e2;
    }
    std::vector<hipGraphNode_t> dv(deps);
    dv.resize(4, nullptr);  // (round 0's dependencies: at most four, all null today)
    const hipGraphNode_t d0 = pi ? pi : dv[0], d1 = pi ? nullptr : dv[1], d2 = pi ? nullptr : dv[2], d3 = pi ? nullptr : dv[3];
    // (the first Send of a step priced by the small workgroups of the planner pair too: k_plan_pair_mw with no drain)
    if ((j->rx_multi && j->pipeline && j->pair_job && !j->fuse) || job_mw_seq(j))
      return add3(&P[t], grdma_kernel_fn_plan_pair_mw(), dim3(n, job_tx_groups(j)), grdma_kernel_threads(0), nullptr, txop,
                  j->d_txf, {d0, d1, d2, d3}, 0u);
    return add2(&P[t], f_txj, dim3(n), grdma_tx_plan_job_threads(), txop, j->d_txf, {d0, d1, d2, d3});
  };
  // X[t] = the receive plan of round t: k_rx_plan_job -- the steady-state body, the general planner behind it
  auto add_rx = [&](uint64_t t, const void* rxop, std::initializer_list<hipGraphNode_t> deps) -> hipError_t {
    if (job_mw_seq(j)) {  // (the drain's planners alone: rxm_body / rxh_body, the general planner behind them)
      std::vector<hipGraphNode_t> dq(deps);
      dq.resize(4, nullptr);
      return add3(&X[t], grdma_kernel_fn_plan_pair_mw(), dim3(n, job_rx_groups(j)), grdma_kernel_threads(0), rxop, nullptr, j->d_txf,
                  {dq[0], dq[1], dq[2], dq[3]}, job_rx_groups(j));
    }
    return add(&X[t], fast ? f_rxj : f_rxp, dim3(n), fast ? grdma_rx_plan_job_threads() : pt, rxop, deps);
  };
  auto at = [](std::vector<hipGraphNode_t>& v, uint64_t t, uint64_t back) -> hipGraphNode_t {
    return t >= back ? v[t - back] : nullptr;
  };
  hipError_t e = hipSuccess;
  for (uint64_t t = 0; t < R && e == hipSuccess; t++) {
    const int k = job_opset(t);
    const void* txop = j->d_txop + k * n;
    const void* rxop = j->d_rxop + k * n;
    const void* gplans = j->d_plans;
    const void* wplans = j->d_plans + n * (1 + (t & 1));
    if (j->burst > 1) {
      // burst rounds: strictly in order (one connection state, one receive pass per round)
      const uint32_t B = j->burst;
      const uint32_t txb_b = std::max<uint32_t>(1, std::min<uint32_t>(tx_blocks, grid_cap / (n * B) + 1));
      const void* btx = j->d_btxop + (size_t)k * B * n;
      const void* bg = j->d_bplans;
      const void* bw = j->d_bplans + (size_t)B * n;
      hipGraphNode_t prev = at(A, t, 1);
      e = add(&P[t], grdma_kernel_fn(4), dim3(n, B), pt, btx, {prev});
      if (e == hipSuccess) e = add(&G[t], f_cpy, dim3(txb_b, B * n), ct, bg, {P[t]});
      hipGraphNode_t last = G[t];
      W[t] = nullptr;
      if (!j->direct && e == hipSuccess) {
        e = add(&W[t], f_cpy, dim3(txb_b, B * n), ct, bw, {G[t]});
        last = W[t];
      }
      if (e == hipSuccess) e = add(&X[t], f_rxp, dim3(n), pt, rxop, {last});
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t]});
    } else if (!j->pipeline) {
      hipGraphNode_t prev = at(A, t, 1);
      e = add_tx(t, txop, {prev});
      if (e == hipSuccess) e = add(&G[t], f_cpy, dim3(txb, n), ct, gplans, {P[t]});
      hipGraphNode_t last = G[t];
      W[t] = nullptr;
      if (!j->direct && e == hipSuccess) {
        e = add(&W[t], f_cpy, dim3(txb, n), ct, wplans, {G[t]});
        last = W[t];
      }
      if (e == hipSuccess) e = add_rx(t, rxop, {last});
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t]});
    } else if (fast && tfast && j->pair_job && j->fuse && !j->direct) {
      // Fused schedule: the two planners ride in the grids of copy kernels they do not depend on -- the send plan of
      // round t + 1 beside the wire of round t (it needs the credit of round t - 1 and the sender's state, both there),
      // the drain plan of round t beside the gather of round t + 1 (which needs that send plan and a staging buffer
      // the wire of round t - 1 has left).  Three launches per round, and the chip is not idle behind a one-workgroup
      // planner any more; what every plan sees -- credit, cursor, ring -- is what it saw in the paired schedule, so
      // the rounds are the same rounds:
      //   W_t + P_{t+1}: G_t, A_{t-1}      X_t + G_{t+1}: W_t      A_t: X_t
      if (t == 0) {
        e = add_tx(0, txop, {});
        if (e == hipSuccess) e = add(&G[0], f_cpy, dim3(txb, n), ct, gplans, {P[0]});
      }
      const bool more = t + 1 < R;
      const uint32_t fb = 1 + std::max<uint32_t>(1, std::min<uint32_t>((txb + 3) / 4, job_fused_copy_blocks()));
      if (e == hipSuccess) {
        if (more) {
          const void* txop_next = j->d_txop + job_opset(t + 1) * n;
          e = add3(&W[t], grdma_kernel_fn(7), dim3(fb, n), grdma_tx_plan_job_threads(), wplans, txop_next, j->d_txf,
                   {G[t], at(A, t, 1)});
          P[t + 1] = W[t];
        } else {
          e = add(&W[t], f_cpy, dim3(txb, n), ct, wplans, {G[t], at(A, t, 1)});
        }
      }
      if (e == hipSuccess) {
        if (more) {
          e = add2(&X[t], grdma_kernel_fn_rxplan_gather_job(), dim3(fb, n), grdma_rx_plan_job_threads(), rxop, gplans, {W[t]});
          G[t + 1] = X[t];
        } else {
          e = add_rx(t, rxop, {W[t]});
        }
      }
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t]});
    } else if (fast && tfast && j->pair_job) {
      // One launch for the drain of round t and the Send of round t + 1 (k_plan_pair_job): kernels of different
      // branches of a graph do not overlap on this stack (measured: even planner workgroups small enough to sit
      // beside the copy kernels' run behind them), so the round is a chain -- and this one has four links:
      //   G_t: P_t (= X_{t-1})      W_t: G_t      X_t + P_{t+1}: W_t, A_{t-1}      A_t: X_t
      // (j->fuse_ag, default: the scatter of round t and the gather of round t + 1 share a launch -- both are ready
      // behind the planner pair, neither touches the other's bytes: G_{t+1} = A_t, three launches per round)
      if (t == 0) e = add_tx(0, txop, {});
      if (e == hipSuccess && (t == 0 || !j->fuse_ag)) e = add(&G[t], f_cpy, dim3(txb, n), ct, gplans, {P[t], at(A, t, 1)});
      // (a direct wire has no wire kernel: the gather writes the records into the peer ring, two launches per round)
      W[t] = nullptr;
      if (e == hipSuccess && !j->direct) e = add(&W[t], f_cpy, dim3(txb, n), ct, wplans, {G[t]});
      const bool more = t + 1 < R;
      if (e == hipSuccess && job_round_fused(j)) {
        // (GRDMA_JOB_FUSE_ROUND) P_{t+1}: W_t        X_t + A_t + G_{t+1} (k_round_xag): P_{t+1}
        const void* txop_next = j->d_txop + job_opset(t + 1) * n;
        hipGraphNode_t last = j->direct ? G[t] : W[t];
        if (more) {
          e = add3(&P[t + 1], grdma_kernel_fn_plan_pair_mw(), dim3(n, job_tx_groups(j)), grdma_kernel_threads(0), nullptr,
                   txop_next, j->d_txf, {last, at(A, t, 1)}, 0u);
          last = P[t + 1];
        }
        const job_xag_shape xs = job_xag(j, txb, rxb, more);
        if (e == hipSuccess)
          e = add3(&X[t], grdma_kernel_fn_round_xag(), dim3(xs.grid), grdma_kernel_threads(0), rxop, gplans, j->d_scratch,
                   {last, at(A, t, 1)}, xs.shape);
        A[t] = X[t];
        if (more) G[t + 1] = X[t];
        continue;
      }
      if (e == hipSuccess) {
        const void* txop_next = j->d_txop + job_opset(t + 1) * n;
        if (j->rx_multi)
          e = add3(&X[t], grdma_kernel_fn_plan_pair_mw(), dim3(n, job_rx_groups(j) + (more ? job_tx_groups(j) : 0)),
                   grdma_kernel_threads(0), rxop, more ? txop_next : nullptr, j->d_txf, {j->direct ? G[t] : W[t], at(A, t, 1)},
                   job_pair_mode(j));
        else
        e = add3(&X[t], grdma_kernel_fn_plan_pair_job(), dim3(n, more ? 2 : 1), grdma_rx_plan_job_threads(), rxop,
                 more ? txop_next : nullptr, j->d_txf, {j->direct ? G[t] : W[t], at(A, t, 1)});
        if (more) P[t + 1] = X[t];
      }
      if (e == hipSuccess) {
        if (more && j->fuse_ag) {
          e = add2(&A[t], grdma_kernel_fn(8), dim3(std::max(rxb, txb), 2 * n), ct, rxop, gplans, {X[t]});
          G[t + 1] = A[t];
        } else {
          e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t]});
        }
      }
    } else if (j->deep || fast || tfast) {
      // Limit-driven schedule (default): the drain of round t walks exactly up to the tail its Send
      // computed (grdma_rx_op::limit_ptr), so round t + 1 may land in the ring while round t is being
      // walked -- the edge rx_plan_{t-1} -> wire_t of the older schedule is gone and both planners
      // leave the wire's path.  What is left of the ordering:
      //   P_t: P_{t-1} (the sender's state), G_{t-1} (one gather plan), W_{t-2} (staging / wire plan of
      //        this parity), A_{t-2} (credit lag of at most one round; implies X_{t-2}: the limit slot)
      //   G_t: P_t      W_t: G_t      X_t: W_t, X_{t-1} (the reader's state), A_{t-2} (scatter plan / result
      //        of this parity)        A_t: X_t, A_{t-1} (credit reports stay in order)
      // The only cycle that spans rounds is A_{t-2} -> P_t -> G_t -> W_t -> X_t -> A_t: two rounds in
      // flight, (P + G + W + X + A) / 2 per round.
      const hipGraphNode_t wprev2 = j->direct ? at(G, t, 2) : at(W, t, 2);
      e = add_tx(t, txop, {at(P, t, 1), at(G, t, 1), wprev2, at(A, t, 2)});
      if (e == hipSuccess) e = add(&G[t], f_cpy, dim3(txb, n), ct, gplans, {P[t]});
      hipGraphNode_t last = G[t];
      W[t] = nullptr;
      if (!j->direct && e == hipSuccess) {
        e = add(&W[t], f_cpy, dim3(txb, n), ct, wplans, {G[t]});
        last = W[t];
      }
      if (e == hipSuccess) e = add_rx(t, rxop, {last, at(X, t, 1), at(A, t, 2)});
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t], at(A, t, 1)});
    } else if (j->direct) {
      e = add(&P[t], f_txp, dim3(n), pt, txop, {at(G, t, 1), at(X, t, 1)});
      if (e == hipSuccess) e = add(&G[t], f_cpy, dim3(txb, n), ct, gplans, {P[t]});
      W[t] = nullptr;
      if (e == hipSuccess) e = add(&X[t], f_rxp, dim3(n), pt, rxop, {G[t], at(A, t, 2)});
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t], at(A, t, 1)});
    } else {
      // The send plan of round t + 1 shares a launch with the receive plan of round t
      // (k_plan_pair): P[t + 1] and X[t] are the same node.  Its dependencies are the union of
      // both kernels': the wire of round t (which implies gather t and everything of round
      // t - 1 but its scatter) and the scatter of round t - 1 (the credit the next Send may use).
      if (t == 0) e = add(&P[0], f_txp, dim3(n), pt, txop, {});
      if (e == hipSuccess) e = add(&G[t], f_cpy, dim3(txb, n), ct, gplans, {P[t], at(A, t, 2)});
      if (e == hipSuccess) e = add(&W[t], f_cpy, dim3(txb, n), ct, wplans, {G[t], at(X, t, 1)});
      if (e == hipSuccess) {
        const bool more = t + 1 < R;
        const void* txop_next = j->d_txop + job_opset(t + 1) * n;
        e = add2(&X[t], f_pair, dim3(n, more ? 2 : 1), pt, rxop, more ? txop_next : nullptr, {W[t], at(A, t, 1)});
        if (more) P[t + 1] = X[t];
      }
      if (e == hipSuccess) e = add(&A[t], f_rxa, dim3(rxb, n), ct, rxop, {X[t], at(A, t, 1)});
    }
  }
  if (e == hipSuccess && R > 0) {
    // behind the last round: the connection's arrival report and the state lines (k_tx_commit)
    hipGraphNode_t cm = nullptr;
    e = add2(&cm, grdma_kernel_fn(5), dim3(n), 64, j->d_txconns, nullptr, {W[R - 1] ? W[R - 1] : G[R - 1], A[R - 1]});
    if (e == hipSuccess) add_hooks(j->post_hooks, cm);  // (every node of the job reaches k_tx_commit)
  }
  if (e == hipSuccess) e = hook_err;
  if (e != hipSuccess) {
    hipGraphDestroy(g);
    return fail(GRDMA_ERR_HIP, "graph construction failed: %s", hipGetErrorString(e));
  }
  *out = g;
  return 0;
}

// ---- persistent link engine -------------------------------------------------------------------
// One launch of k_link runs the whole job: per link a team of workgroups (sender's leader,
// receiver's leader, gather / wire / scatter worker waves) that stay resident until every slice
// has been delivered.  See grdma_link.h.
int job_engine_prepare(grdma_stream_job* j) {
  if (j->d_lk_ptrs) return 0;
  const uint32_t n = (uint32_t)j->links.size();
  const uint32_t resident = grdma_link_resident_blocks();
  if (resident == 0) return fail(GRDMA_ERR_HIP, "occupancy query for the link engine failed");
  uint32_t cap_blocks = resident;
  if (const char* e = getenv("GRDMA_LINK_BLOCKS")) {  // tuning knob (tools/, bench legs)
    const long v = atol(e);
    if (v > 0 && (uint32_t)v < cap_blocks) cap_blocks = (uint32_t)v;
  }
  uint32_t team = cap_blocks / n;
  if (team < 3) return fail(GRDMA_ERR_CAPACITY, "%u links do not fit the %u resident workgroups of the link engine", n, resident);
  if (team > 1024) team = 1024;
  // worker waves per stage: gather : wire : scatter by the bytes they move per payload byte
  // (2 : 2 : 3; no wire stage when records are built in the peer ring)
  int mix[3] = {2, j->direct ? 0 : 2, j->direct ? 3 : 3};
  if (const char* e = getenv("GRDMA_LINK_MIX")) {
    int a = 0, b = 0, c2 = 0;
    if (sscanf(e, "%d,%d,%d", &a, &b, &c2) == 3 && a > 0 && c2 > 0 && b >= 0) {
      mix[0] = a;
      mix[1] = j->direct ? 0 : (b > 0 ? b : 1);
      mix[2] = c2;
    }
  }
  const uint32_t waves = (team - 2) * (LK_THREADS / 64);
  const uint32_t stages = j->direct ? 2 : 3;
  if (waves < stages) return fail(GRDMA_ERR_CAPACITY, "link engine team of %u workgroups is too small", team);
  const uint32_t msum = (uint32_t)(mix[0] + mix[1] + mix[2]);
  uint32_t nw[3];
  nw[0] = std::max<uint32_t>(1, waves * mix[0] / msum);
  nw[1] = j->direct ? 0 : std::max<uint32_t>(1, waves * mix[1] / msum);
  nw[2] = waves - nw[0] - nw[1];
  if (nw[2] < 1) { nw[2] = 1; if (nw[0] > 1) nw[0]--; }
  int rate_khz = 0;
  if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, g_ctx.device) != hipSuccess || rate_khz <= 0)
    rate_khz = 100000;  // 100 MHz
  long tmo_ms = 4000;
  if (const char* e = getenv("GRDMA_LINK_TIMEOUT_MS")) tmo_ms = std::max<long>(10, atol(e));
  j->lk_timeout_ticks = (uint64_t)rate_khz * (uint64_t)tmo_ms;
  std::vector<lk_ctl*> ptrs(n);
  for (uint32_t i = 0; i < n; i++) {
    grdma_job_link& l = j->links[i];
    const uint64_t ring = l.tx->ring_size;
    if (ring > (256ull << 20))
      return fail(GRDMA_ERR_CAPACITY, "link engine: ring of %llu bytes exceeds 256 MiB", (unsigned long long)ring);
    HIP_TRY(hipMalloc((void**)&l.d_lk, sizeof(lk_ctl)));
    for (int t = 0; t < 3; t++) HIP_TRY(hipMalloc((void**)&l.d_tab[t], sizeof(lk_entry) * LK_TABLE_CAP));
    lk_ctl h;
    memset(&h, 0, sizeof(h));
    h.tx = l.tx->d_conn;
    h.rx = l.rx->d_conn;
    h.slices = l.d_sges;
    h.nslices = l.count;
    {
      std::vector<grdma_sge> tmp(l.count);
      HIP_TRY(hipMemcpy(tmp.data(), l.d_sges, sizeof(grdma_sge) * l.count, hipMemcpyDeviceToHost));
      for (auto& g : tmp) h.total_bytes += g.len;
    }
    h.arena = l.dst;
    h.arena_cap = l.dst_cap;
    h.out_slices = l.d_slices;
    h.slices_cap = l.slices_cap;
    h.direct = j->direct ? 1 : 0;
    // staging buffers: the sender may run this many Sends ahead of the wire (each Send prices
    // its records against a whole staging buffer of ring / 2, pair.cc:104)
    uint32_t nst = 0;
    if (!j->direct) {
      uint64_t want = (384ull << 20) / (ring / 2);  // up to 384 MiB of staging per link: the sender runs ahead of the credit loop
      if (const char* e = getenv("GRDMA_LINK_STAGING")) want = (uint64_t)std::max<long>(1, atol(e));
      want = std::min<uint64_t>(std::max<uint64_t>(want, 2), LK_MAX_STAGING);
      h.staging[nst++] = l.tx->d_staging;
      h.staging[nst++] = l.d_staging2;
      while (nst < want) {
        uint8_t* sb = nullptr;
        HIP_TRY(hipMalloc((void**)&sb, ring / 2 + 64));
        HIP_TRY(hipMemset(sb, 0, ring / 2 + 64));
        l.d_staging_more.push_back(sb);
        h.staging[nst++] = sb;
      }
    }
    h.n_staging = nst ? nst : 1;
    for (int t = 0; t < 3; t++) {
      h.tab[t] = l.d_tab[t];
      h.nwaves[t] = nw[t];
    }
    h.timeout_ms = (uint32_t)tmo_ms;
    HIP_TRY(hipMemcpy(l.d_lk, &h, sizeof(h), hipMemcpyHostToDevice));
    ptrs[i] = l.d_lk;
  }
  HIP_TRY(hipMalloc((void**)&j->d_lk_ptrs, sizeof(lk_ctl*) * n));
  HIP_TRY(hipMemcpy(j->d_lk_ptrs, ptrs.data(), sizeof(lk_ctl*) * n, hipMemcpyHostToDevice));
  j->lk_team = team;
  return 0;
}

int job_engine_enqueue(grdma_stream_job* j, hipStream_t s) {
  if (int rc = job_engine_prepare(j)) return rc;
  // every polled word starts at zero in every launch
  for (auto& l : j->links)
    HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t*>(l.d_lk) + LK_DYNAMIC_OFFSET, 0, sizeof(lk_ctl) - LK_DYNAMIC_OFFSET, s));
  HIP_TRY(grdma_launch_link(j->d_lk_ptrs, (uint32_t)j->links.size(), j->lk_team, j->lk_timeout_ticks, s));
  HIP_TRY(grdma_launch_tx_commit(j->d_txconns, nullptr, (uint32_t)j->links.size(), s));
  return 0;
}

// after a synchronize: did every link finish cleanly?
int job_engine_check(grdma_stream_job* j) {
  for (size_t i = 0; i < j->links.size(); i++) {
    uint64_t ab = 0;
    HIP_TRY(hipMemcpy(&ab, reinterpret_cast<uint8_t*>(j->links[i].d_lk) + offsetof(lk_ctl, abort), sizeof(ab), hipMemcpyDeviceToHost));
    if (ab != 0) {
      static const char* what[] = {"", "a role timed out waiting", "no progress (zero-length slice at the cursor)",
                                   "destination buffer too small", "slice table too small",
                                   "the ring does not hold the records the sender published"};
      uint64_t dbg[8] = {0};
      hipMemcpy(dbg, reinterpret_cast<uint8_t*>(j->links[i].d_lk) + offsetof(lk_ctl, res_dbg), sizeof(dbg), hipMemcpyDeviceToHost);
      return fail(GRDMA_ERR_HIP, "link engine aborted on link %zu: %s (code %llu; site %llu: %llu %llu %llu %llu)", i,
                  ab <= 5 ? what[ab] : "?", (unsigned long long)ab, (unsigned long long)dbg[1], (unsigned long long)dbg[2],
                  (unsigned long long)dbg[3], (unsigned long long)dbg[4], (unsigned long long)dbg[5]);
    }
  }
  return 0;
}

}  // namespace

extern "C" {

grdma_stream_job* grdma_stream_job_create_multi(uint32_t n, grdma_pair* const* tx,
                                                grdma_pair* const* rx, const grdma_slice* slices,
                                                const uint64_t* counts, void* const* rx_dsts,
                                                const uint64_t* rx_dst_caps,
                                                const uint64_t* slices_caps, uint64_t max_rounds) {
  if (require_ctx()) return nullptr;
  if (!n || !tx || !rx || !slices || !counts || !rx_dsts || !rx_dst_caps || !slices_caps) {
    fail(GRDMA_ERR_INVALID, "stream job: null argument");
    return nullptr;
  }
  grdma_stream_job* j = new grdma_stream_job();
  if (const char* e = getenv("GRDMA_JOB_SCHEDULE")) j->deep = strcmp(e, "pair") == 0 ? 0 : 1;
  if (const char* e = getenv("GRDMA_JOB_CUMASK")) j->cumask_bits = atoi(e);
  if (const char* e = getenv("GRDMA_RX_FAST")) j->rx_fast = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_RX_MULTI")) j->rx_multi = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_PAIR_JOB")) j->pair_job = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_JOB_FUSE")) j->fuse = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_JOB_FUSE_AG")) j->fuse_ag = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_JOB_FUSE_ROUND")) j->fuse_round = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_JOB_FUSE_ROUND_AFTER")) j->fuse_round_after = atoi(e);
  if (const char* e = getenv("GRDMA_TX_FAST")) j->tx_fast = atoi(e) != 0;
  if (const char* e = getenv("GRDMA_SLIM_AFTER")) {  // experiment (tools/gpu_slim.sh): see grdma_stream_job_run
    j->slim_after = atoi(e);
    j->rx_fast = j->tx_fast = 0;
  }
  j->rounds = max_rounds;
  j->stream = tx[0]->stream;
  j->direct = (tx[0]->flags & GRDMA_WIRE_DIRECT) != 0;
  j->links.resize(n);
  bool ok = hipEventCreate(&j->ev0) == hipSuccess && hipEventCreate(&j->ev1) == hipSuccess;
  uint64_t off = 0;
  for (uint32_t i = 0; i < n && ok; i++) {
    grdma_job_link& l = j->links[i];
    if (!tx[i] || !rx[i] || tx[i]->peer != rx[i] || !counts[i] || !rx_dsts[i] ||
        tx[i]->stream != j->stream || ((tx[i]->flags & GRDMA_WIRE_DIRECT) != 0) != j->direct) {
      fail(GRDMA_ERR_INVALID, "stream job link %u: needs two connected pairs on the shared stream", i);
      ok = false;
      break;
    }
    l.tx = tx[i];
    l.rx = rx[i];
    l.count = counts[i];
    l.dst = static_cast<uint8_t*>(rx_dsts[i]);
    l.dst_cap = rx_dst_caps[i];
    l.slices_cap = slices_caps[i];
    if (tx[i]->ring_size > j->max_ring) j->max_ring = tx[i]->ring_size;
    ok = hipMalloc((void**)&l.d_sges, sizeof(grdma_sge) * l.count) == hipSuccess &&
         hipMalloc((void**)&l.d_slices, sizeof(grdma_slice_out) * l.slices_cap) == hipSuccess &&
         hipMalloc((void**)&l.d_wireplan2, sizeof(grdma_plan)) == hipSuccess &&
         hipMalloc((void**)&l.d_rxplan2, sizeof(grdma_plan)) == hipSuccess &&
         (j->direct || hipMalloc((void**)&l.d_staging2, tx[i]->ring_size / 2 + 64) == hipSuccess) &&
         hipMalloc((void**)&l.d_encpre, sizeof(uint64_t) * (l.count + 1)) == hipSuccess &&
         hipMalloc((void**)&l.d_lenpre, sizeof(uint64_t) * (l.count + 1)) == hipSuccess &&
         hipMalloc((void**)&l.d_tilepre, sizeof(uint32_t) * (l.count + 1)) == hipSuccess;
    if (!ok) break;
    hipMemset(l.d_wireplan2, 0, sizeof(grdma_plan));
    hipMemset(l.d_rxplan2, 0, sizeof(grdma_plan));
    if (l.d_staging2) hipMemset(l.d_staging2, 0, tx[i]->ring_size / 2 + 64);
    std::vector<grdma_sge> tmp(l.count);
    for (uint64_t q = 0; q < l.count; q++) {
      tmp[q].ptr = static_cast<const uint8_t*>(slices[off + q].ptr);
      tmp[q].len = slices[off + q].len;
    }
    off += l.count;
    ok = hipMemcpy(l.d_sges, tmp.data(), sizeof(grdma_sge) * l.count, hipMemcpyHostToDevice) == hipSuccess;
  }
  // (the size tables of the rounds: GRDMA_JOB_SIZE_HINTS=0 leaves them out -- drains without a period then walk)
  if (ok && !(getenv("GRDMA_JOB_SIZE_HINTS") && atoi(getenv("GRDMA_JOB_SIZE_HINTS")) == 0)) {
    ok = hipMalloc((void**)&j->d_hints, sizeof(grdma_size_hint) * 3 * n) == hipSuccess &&
         hipMemset(j->d_hints, 0, sizeof(grdma_size_hint) * 3 * n) == hipSuccess;
  }
  if (ok && (j->fuse_round || j->fuse_round_after >= 0)) {
    std::vector<void*> ptrs(n, nullptr);
    for (uint32_t i = 0; i < n && ok; i++) {
      ok = hipMalloc(&ptrs[i], grdma_rx_scratch_bytes()) == hipSuccess;
      if (ok) j->scratch_bufs.push_back(ptrs[i]);
    }
    ok = ok && hipMalloc((void**)&j->d_scratch, sizeof(void*) * n) == hipSuccess &&
         hipMemcpy(j->d_scratch, ptrs.data(), sizeof(void*) * n, hipMemcpyHostToDevice) == hipSuccess;
  }
  const size_t sz_tx = sizeof(grdma_tx_op) * 3 * n, sz_rx = sizeof(grdma_rx_op) * 3 * n;
  const size_t sz_txr = sizeof(grdma_tx_result) * n, sz_rxr = sizeof(grdma_rx_result) * 2 * n;
  const size_t sz_pl = sizeof(grdma_plan*) * 3 * n;
  const size_t sz_lim = sizeof(uint64_t) * 3 * n, sz_cn = sizeof(grdma_conn*) * n;
  if (ok) ok = hipMalloc((void**)&j->d_ctl, sz_tx + sz_rx + sz_txr + sz_rxr + sz_pl + sz_lim + sz_cn) == hipSuccess;
  if (!ok) {
    if (g_err.empty()) fail(GRDMA_ERR_HIP, "stream job allocation failed");
    grdma_stream_job_destroy(j);
    return nullptr;
  }
  j->d_txop = reinterpret_cast<grdma_tx_op*>(j->d_ctl);
  j->d_rxop = reinterpret_cast<grdma_rx_op*>(j->d_ctl + sz_tx);
  j->d_txres = reinterpret_cast<grdma_tx_result*>(j->d_ctl + sz_tx + sz_rx);
  j->d_rxres = reinterpret_cast<grdma_rx_result*>(j->d_ctl + sz_tx + sz_rx + sz_txr);
  j->d_plans = reinterpret_cast<const grdma_plan**>(j->d_ctl + sz_tx + sz_rx + sz_txr + sz_rxr);
  j->d_limits = reinterpret_cast<uint64_t*>(j->d_ctl + sz_tx + sz_rx + sz_txr + sz_rxr + sz_pl);
  j->d_txconns = reinterpret_cast<grdma_conn**>(j->d_ctl + sz_tx + sz_rx + sz_txr + sz_rxr + sz_pl + sz_lim);
  std::vector<uint8_t> host(sz_tx + sz_rx + sz_txr + sz_rxr + sz_pl + sz_lim + sz_cn, 0);
  auto* h_tx = reinterpret_cast<grdma_tx_op*>(host.data());
  auto* h_rx = reinterpret_cast<grdma_rx_op*>(host.data() + sz_tx);
  auto** h_pl = reinterpret_cast<const grdma_plan**>(host.data() + sz_tx + sz_rx + sz_txr + sz_rxr);
  for (int k = 0; k < 3; k++)
    for (uint32_t i = 0; i < n; i++) {
      const grdma_job_link& l = j->links[i];
      const bool odd = k == 1;
      grdma_tx_op& t = h_tx[k * n + i];
      t.conn = l.tx->d_conn;
      t.slices = l.d_sges;
      t.nslices = l.count;
      t.plan = l.tx->d_txplan;
      t.wire_plan = odd ? l.d_wireplan2 : l.tx->d_wireplan;
      t.staging_alt = odd ? l.d_staging2 : nullptr;
      t.result = &j->d_txres[i];
      t.use_cursor = k == 0 ? 2 : 1;
      t.tail_out = &j->d_limits[k * n + i];
      t.sizes_out = j->d_hints ? &j->d_hints[k * n + i] : nullptr;
      grdma_rx_op& r = h_rx[k * n + i];
      r.conn = l.rx->d_conn;
      r.plan = odd ? l.d_rxplan2 : l.rx->d_rxplan;
      r.result = &j->d_rxres[(odd ? n : 0) + i];
      r.slices = l.d_slices;
      r.arena = l.dst;
      r.arena_cap = l.dst_cap;
      r.max_reads = GRDMA_MAX_SLICES;
      r.raw_cap = 0;
      r.append = k == 0 ? 2 : 1;
      r.slices_cap = l.slices_cap;
      r.limit_ptr = &j->d_limits[k * n + i];
      r.sizes_in = t.sizes_out;
    }
  auto** h_cn = reinterpret_cast<grdma_conn**>(host.data() + sz_tx + sz_rx + sz_txr + sz_rxr + sz_pl + sz_lim);
  for (uint32_t i = 0; i < n; i++) h_cn[i] = j->links[i].tx->d_conn;
  for (uint32_t i = 0; i < n; i++) {
    h_pl[i] = j->links[i].tx->d_txplan;
    h_pl[n + i] = j->links[i].tx->d_wireplan;   // even rounds
    h_pl[2 * n + i] = j->links[i].d_wireplan2;  // odd rounds
  }
  // the scatter plans are addressed through the rx ops; the plan pointer array is for k_copy
  if (hipMemcpy(j->d_ctl, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
    fail(GRDMA_ERR_HIP, "job control block upload failed");
    grdma_stream_job_destroy(j);
    return nullptr;
  }
  {
    std::vector<grdma_txf_ctl> h_txf(n);
    for (uint32_t i = 0; i < n; i++) {
      const grdma_job_link& l = j->links[i];
      memset(&h_txf[i], 0, sizeof(grdma_txf_ctl));
      h_txf[i].slices = l.d_sges;
      h_txf[i].n = l.count;
      h_txf[i].enc_pre = l.d_encpre;
      h_txf[i].len_pre = l.d_lenpre;
      h_txf[i].tile_pre = l.d_tilepre;
      h_txf[i].tile_shift = GRDMA_PLAN_TILE_SHIFT(l.tx->ring_size);
    }
    if (hipMalloc((void**)&j->d_txf, sizeof(grdma_txf_ctl) * n) != hipSuccess ||
        hipMemcpy(j->d_txf, h_txf.data(), sizeof(grdma_txf_ctl) * n, hipMemcpyHostToDevice) != hipSuccess) {
      fail(GRDMA_ERR_HIP, "job index block allocation failed");
      grdma_stream_job_destroy(j);
      return nullptr;
    }
  }
  return j;
}

grdma_stream_job* grdma_stream_job_create(grdma_pair* tx, grdma_pair* rx,
                                          const grdma_slice* slices, uint64_t count,
                                          void* rx_dst, uint64_t rx_dst_cap,
                                          uint64_t slices_cap, uint64_t max_rounds) {
  return grdma_stream_job_create_multi(1, &tx, &rx, slices, &count, &rx_dst, &rx_dst_cap,
                                       &slices_cap, max_rounds);
}

void grdma_stream_job_destroy(grdma_stream_job* j) {
  if (!j) return;
  if (j->stream) hipStreamSynchronize(j->stream);
  if (j->exec) hipGraphExecDestroy(j->exec);
  for (hipStream_t st : {j->s_wire, j->s_rxplan, j->s_apply, j->m_txplan, j->m_rxplan, j->m_copy, j->m_apply})
    if (st) {
      hipStreamSynchronize(st);
      hipStreamDestroy(st);
    }
  if (j->d_txf) hipFree(j->d_txf);
  if (j->d_hints) hipFree(j->d_hints);
  if (j->d_scratch) hipFree(j->d_scratch);
  for (void* b : j->scratch_bufs) hipFree(b);
  for (grdma_job_link& l : j->links) {
    if (l.d_encpre) hipFree(l.d_encpre);
    if (l.d_lenpre) hipFree(l.d_lenpre);
    if (l.d_tilepre) hipFree(l.d_tilepre);
  }
  for (hipEvent_t e : j->kev) hipEventDestroy(e);
  for (hipEvent_t e : j->pev) hipEventDestroy(e);
  for (hipEvent_t e : j->mev) hipEventDestroy(e);
  if (j->ev0) hipEventDestroy(j->ev0);
  if (j->ev1) hipEventDestroy(j->ev1);
  for (auto& l : j->links) {
    hipFree(l.d_sges);
    hipFree(l.d_slices);
    hipFree(l.d_wireplan2);
    hipFree(l.d_rxplan2);
    hipFree(l.d_staging2);
    for (uint8_t* b : l.d_staging_n)
      if (b) hipFree(b);
    hipFree(l.d_lk);
    for (auto* t : l.d_tab) hipFree(t);
    for (auto* sb : l.d_staging_more) hipFree(sb);
    for (auto* q : l.b_gplan) hipFree(q);
    for (auto* q : l.b_wplan) hipFree(q);
    for (auto* q : l.b_staging) hipFree(q);
  }
  hipFree(j->d_lk_ptrs);
  hipFree(j->d_bctl);
  hipFree(j->d_ctl);
  delete j;
}

int grdma_stream_job_set_rounds(grdma_stream_job* j, uint64_t rounds) {
  if (!j || rounds == 0) return fail(GRDMA_ERR_INVALID, "bad rounds");
  j->rounds = rounds;
  return 0;
}

int grdma_stream_job_set_pipeline(grdma_stream_job* j, int on) {
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  j->pipeline = on ? 1 : 0;
  return 0;
}

// "Promised credit" for the paired schedule of a pipelined job on a staged wire (csrc/grdma_rx_plan.hip, k_plan_pair_mw): the
// Send of round t + 1 is priced with the credit the drain of round t is going to post -- it waits for that drain's plan
// inside the launch they share -- so a ring that every round fills (the reference's default: 4 MiB) carries a full round
// every round, as on the sequential schedule, in three launches instead of five.
int grdma_stream_job_set_promised_credit(grdma_stream_job* j, int on) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  j->promise = on != 0;
  return 0;
}

int grdma_stream_job_set_rebuild_index(grdma_stream_job* j, int on) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  if (j->sges_exposed == (on != 0)) return 0;
  HIP_TRY(hipStreamSynchronize(j->stream));
  j->sges_exposed = on != 0;      // (job_index_needed: the graph is captured with k_tx_index in front of round 0)
  j->index_valid = false;
  if (j->exec) hipGraphExecDestroy(j->exec);
  j->exec = nullptr;
  return 0;
}

// `sends` consecutive Sends per round in ONE plan (1 = the plain schedule): what rdma_flush does while the ring has
// room -- Send, advance the cursor, Send again (rdma_bp_posix.cc:470-524) -- priced by the planners of
// csrc/grdma_tx_multi.h from the index, Send k + 1 from the state Send k leaves; the peer drains once per round.  For
// the paired schedule with the small planner workgroups (the default of a pipelined job); other schedules keep one
// Send per round.  The staging buffers of both parities then hold `sends` x ring / 2 bytes.
int grdma_stream_job_set_sends(grdma_stream_job* j, uint32_t sends) {
  if (int rc = require_ctx()) return rc;
  if (!j || sends == 0 || sends > grdma_tx_multi_max_sends()) return fail(GRDMA_ERR_INVALID, "sends must be 1..%u", grdma_tx_multi_max_sends());
  if (sends == j->sends) return 0;
  const uint32_t n = (uint32_t)j->links.size();
  HIP_TRY(hipStreamSynchronize(j->stream));
  for (uint32_t i = 0; i < n; i++) {
    grdma_job_link& l = j->links[i];
    if (!j->direct && sends > 1) {
      // (the Sends of a round are limited by the free space of the ring: a ring's worth of staging is enough)
      const size_t bytes = (size_t)l.tx->ring_size + 64;
      for (int q = 0; q < 2; q++) {
        if (l.d_staging_n[q]) continue;
        HIP_TRY(hipMalloc((void**)&l.d_staging_n[q], bytes));
        HIP_TRY(hipMemset(l.d_staging_n[q], 0, bytes));
      }
    }
    for (int k = 0; k < 3; k++) {
      uint8_t* alt = (sends > 1 && !j->direct) ? l.d_staging_n[k == 1 ? 1 : 0] : (k == 1 ? l.d_staging2 : nullptr);
      HIP_TRY(hipMemcpy(&j->d_txop[k * n + i].staging_alt, &alt, sizeof(alt), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(&j->d_txf[i].sends, &sends, sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  j->sends = sends;
  j->index_valid = false;  // (built again by the next run's first round)
  return 0;
}

// `burst` Sends per round (1 = the plain schedule).  Every Send of a burst gets its own gather plan,
// wire plan and staging buffer (the reference reuses its one staging buffer after waitDataWrites();
// here the wire of all Sends of a round runs after all of them were planned).
int grdma_stream_job_set_burst(grdma_stream_job* j, uint32_t burst) {
  if (int rc = require_ctx()) return rc;
  if (!j || burst == 0 || burst > 64) return fail(GRDMA_ERR_INVALID, "burst must be 1..64");
  if (burst == j->burst) return 0;
  HIP_TRY(hipStreamSynchronize(j->stream));
  if (j->exec) hipGraphExecDestroy(j->exec);  // the graph is rebuilt for the new schedule
  j->exec = nullptr;
  const uint32_t n = (uint32_t)j->links.size();
  for (grdma_job_link& l : j->links) {
    while (l.b_gplan.size() < burst) {
      // all three or none: a failure leaves the vectors the same length
      grdma_plan *g = nullptr, *w = nullptr;
      uint8_t* st = nullptr;
      const size_t st_bytes = l.tx->ring_size / 2 + 64;
      hipError_t e = hipMalloc((void**)&g, sizeof(grdma_plan));
      if (e == hipSuccess) e = hipMemset(g, 0, sizeof(grdma_plan));
      if (e == hipSuccess) e = hipMalloc((void**)&w, sizeof(grdma_plan));
      if (e == hipSuccess) e = hipMemset(w, 0, sizeof(grdma_plan));
      if (e == hipSuccess && !j->direct) e = hipMalloc((void**)&st, st_bytes);
      if (e == hipSuccess && !j->direct) e = hipMemset(st, 0, st_bytes);
      if (e != hipSuccess) {
        hipFree(g);
        hipFree(w);
        hipFree(st);
        return fail(GRDMA_ERR_HIP, "burst buffers: %s", hipGetErrorString(e));
      }
      l.b_gplan.push_back(g);
      l.b_wplan.push_back(w);
      l.b_staging.push_back(st);
    }
  }
  if (j->d_bctl) hipFree(j->d_bctl);
  j->d_bctl = nullptr;
  j->burst = 1;  // the plain schedule until the burst tables below are in place
  if (burst == 1) return 0;
  const size_t sz_tx = sizeof(grdma_tx_op) * 3 * burst * n, sz_pl = sizeof(grdma_plan*) * 2 * burst * n;
  const size_t sz_res = sizeof(grdma_tx_result) * n;  // scratch results of the Sends before the last
  HIP_TRY(hipMalloc((void**)&j->d_bctl, sz_tx + sz_pl + sz_res));
  j->d_btxop = reinterpret_cast<grdma_tx_op*>(j->d_bctl);
  j->d_bplans = reinterpret_cast<const grdma_plan**>(j->d_bctl + sz_tx);
  grdma_tx_result* scratch = reinterpret_cast<grdma_tx_result*>(j->d_bctl + sz_tx + sz_pl);
  std::vector<uint8_t> host(sz_tx + sz_pl + sz_res, 0);
  auto* h_tx = reinterpret_cast<grdma_tx_op*>(host.data());
  auto** h_pl = reinterpret_cast<const grdma_plan**>(host.data() + sz_tx);
  for (int set = 0; set < 3; set++)
    for (uint32_t k = 0; k < burst; k++)
      for (uint32_t i = 0; i < n; i++) {
        const grdma_job_link& l = j->links[i];
        grdma_tx_op& t = h_tx[((size_t)set * burst + k) * n + i];
        t.conn = l.tx->d_conn;
        t.slices = l.d_sges;
        t.nslices = l.count;
        t.plan = l.b_gplan[k];
        t.wire_plan = l.b_wplan[k];
        t.staging_alt = l.b_staging[k];
        t.result = k + 1 == burst ? &j->d_txres[i] : &scratch[i];
        t.use_cursor = (set == 0 && k == 0) ? 2 : 1;
        t.tail_out = &j->d_limits[set * n + i];  // (the Sends of a burst run in order: the last one's stays)
      }
  for (uint32_t k = 0; k < burst; k++)
    for (uint32_t i = 0; i < n; i++) {
      h_pl[(size_t)k * n + i] = j->links[i].b_gplan[k];
      h_pl[(size_t)(burst + k) * n + i] = j->links[i].b_wplan[k];
    }
  HIP_TRY(hipMemcpy(j->d_bctl, host.data(), host.size(), hipMemcpyHostToDevice));
  j->burst = burst;
  j->index_valid = false;  // (a burst schedule does not keep the slice index up to date)
  return 0;
}

namespace {
// (re)build the executable graph when the job's shape changed: rounds, schedule, which planner kernels it uses
int job_ensure_exec(grdma_stream_job* j) {
  if (!job_exec_stale(j)) return 0;
  if (j->exec) {
    HIP_TRY(hipStreamSynchronize(j->stream));
    hipGraphExecDestroy(j->exec);
  }
  j->exec = nullptr;
  hipGraph_t graph;
  if (int rc = job_build_graph(j, &graph)) return rc;
  HIP_TRY(hipGraphInstantiate(&j->exec, graph, nullptr, nullptr, 0));
  hipGraphDestroy(graph);
  j->exec_rounds = j->rounds;
  j->exec_pipeline = j->pipeline;
  j->exec_fastkey = job_fastkey(j);
  j->exec_hooks_gen = j->hooks_gen;
  return 0;
}
}  // namespace

int grdma_stream_job_run(grdma_stream_job* j, int mode, grdma_stream_result* out) {
  if (int rc = require_ctx()) return rc;
  if (!j || !out) return fail(GRDMA_ERR_INVALID, "null argument");
  hipStream_t s = j->stream;
  const size_t n = j->links.size();
  std::vector<grdma_conn> c0t(n), c0r(n), c1t(n), c1r(n);
  for (size_t i = 0; i < n; i++) {
    if (int rc = fetch_conn(j->links[i].tx, &c0t[i])) return rc;
    if (int rc = fetch_conn(j->links[i].rx, &c0r[i])) return rc;
  }
  memset(out, 0, sizeof(*out));
  if (mode == GRDMA_RUN_GRAPH) {
    if (int rc = job_ensure_exec(j)) return rc;
    HIP_TRY(hipEventRecord(j->ev0, s));
    HIP_TRY(hipGraphLaunch(j->exec, s));
    HIP_TRY(hipEventRecord(j->ev1, s));
  } else if (mode == GRDMA_RUN_ENGINE) {
    if (int rc = job_engine_prepare(j)) return rc;
    HIP_TRY(hipEventRecord(j->ev0, s));
    if (int rc = job_engine_enqueue(j, s)) return rc;
    HIP_TRY(hipEventRecord(j->ev1, s));
  } else {
    HIP_TRY(hipEventRecord(j->ev0, s));
    if (mode == GRDMA_RUN_INSTRUMENTED_SCHEDULE) {
      if (int rc = job_enqueue_schedule_instrumented(j, s)) return rc;
    } else if (j->pipeline && j->burst == 1 && mode == GRDMA_RUN_EAGER && !j->promise) {
      // (a promised-credit job's eager pass runs in order instead: the stream pipeline sees its credit a round late,
      //  the graph of such a job does not)
      if (int rc = (j->cumask_bits > 0 ? job_enqueue_masked(j, s) : job_enqueue_pipelined(j, s))) return rc;
    } else {
      if (int rc = job_enqueue(j, s, mode == GRDMA_RUN_INSTRUMENTED)) return rc;
    }
    HIP_TRY(hipEventRecord(j->ev1, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (mode == GRDMA_RUN_ENGINE)
    if (int rc = job_engine_check(j)) return rc;
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, j->ev0, j->ev1));
  out->ms_total = ms;
  if (mode == GRDMA_RUN_INSTRUMENTED_SCHEDULE) {
    for (size_t e = 0; e < j->kev_cls.size(); e++) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, j->kev[e], j->kev[e + 1]));
      out->ms_class[j->kev_cls[e]] += t;
      out->launches_class[j->kev_cls[e]]++;
    }
  }
  if (mode == GRDMA_RUN_INSTRUMENTED) {
    size_t e = 0;
    for (uint64_t r = 0; r < j->rounds; r++)
      for (int cls = 0; cls < 5; cls++, e++) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, j->kev[e], j->kev[e + 1]));
        if (cls == 2 && j->direct) continue;
        out->ms_class[cls] += t;
        out->launches_class[cls]++;
      }
  }
  out->done = 1;
  for (size_t i = 0; i < n; i++) {
    if (int rc = fetch_conn(j->links[i].tx, &c1t[i])) return rc;
    if (int rc = fetch_conn(j->links[i].rx, &c1r[i])) return rc;
    const uint64_t sent = c1t[i].total_written - c0t[i].total_written;
    const uint64_t deliv = c1r[i].total_read - c0r[i].total_read;
    out->bytes_sent += sent;
    out->bytes_delivered += deliv;
    out->slices_delivered += c1r[i].rx_slice_idx;
    out->tx_rounds = std::max<uint64_t>(out->tx_rounds, c1t[i].tx_rounds - c0t[i].tx_rounds);
    out->rx_rounds = std::max<uint64_t>(out->rx_rounds, c1r[i].rx_rounds - c0r[i].rx_rounds);
    out->tx_records += c1t[i].tx_records - c0t[i].tx_records;
    out->rx_records += c1r[i].rx_records - c0r[i].rx_records;
    if (!(c1t[i].tx_slice_idx >= j->links[i].count && deliv == sent)) out->done = 0;
  }
  // The job kernels try the steady-state bodies first and run the general planners (under a quarter of their
  // register budget) for what those decline.  A job whose last drains / Send keep being declined -- no period in
  // its record sizes, Sends cut by the staging budget, ... -- goes back to the plain planner kernels.
  j->runs++;
  // (round 0 of this run built the index -- only a schedule that prices its Sends from it launches k_tx_index: a run
  //  with a burst, or on the link engine, leaves the index as it was)
  if (mode != GRDMA_RUN_ENGINE && job_tx_fast(j) && j->rounds >= 1) j->index_valid = true;
  if (j->fuse_round_after >= 0 && j->runs >= j->fuse_round_after) j->fuse_round = 1;
  if (j->slim_after >= 0) {
    if (j->runs >= j->slim_after) j->rx_fast = j->tx_fast = 1;
    return 0;
  }
  // (not with several Sends per plan: only the small planner workgroups price those, and what they decline is planned
  //  by the general planners inside the same launch, at their full register budget)
  if (j->burst == 1 && j->sends == 1 && !j->promise && mode != GRDMA_RUN_ENGINE && j->rounds >= 2 && (j->rx_fast || job_tx_fast(j))) {
    uint64_t cnt[3][2];  // {taken, declined with work waiting} of link 0: drains of both parities, Sends
    uint32_t c32[2][2];  // {pad1 = taken, pad0 = declined}
    static_assert(offsetof(grdma_rx_result, pad0) == offsetof(grdma_rx_result, pad1) + 4, "layout");
    HIP_TRY(hipMemcpy(c32[0], &j->d_rxres[0].pad1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c32[1], &j->d_rxres[n].pad1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int q = 0; q < 2; q++) { cnt[q][0] = c32[q][0]; cnt[q][1] = c32[q][1]; }
    HIP_TRY(hipMemcpy(cnt[2], &j->d_txres[0].dbg[10], 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    const uint64_t rx_took = cnt[0][0] + cnt[1][0], rx_decl = cnt[0][1] + cnt[1][1];
    const uint64_t d_rx_took = rx_took - j->seen[0], d_rx_decl = rx_decl - j->seen[1];
    const uint64_t d_tx_took = cnt[2][0] - j->seen[2], d_tx_decl = cnt[2][1] - j->seen[3];
    j->seen[0] = rx_took; j->seen[1] = rx_decl; j->seen[2] = cnt[2][0]; j->seen[3] = cnt[2][1];
    // a miss: the run's drains (Sends) with work waiting went to the general planner, more than a start-up's worth
    if (d_rx_took + d_rx_decl) j->rx_miss = d_rx_decl > 2 + d_rx_took ? j->rx_miss + 1 : 0;
    if (d_tx_took + d_tx_decl) j->tx_miss = d_tx_decl > 2 + d_tx_took ? j->tx_miss + 1 : 0;
    if (j->rx_fast && j->rx_miss >= 2) j->rx_fast = 0;
    if (j->tx_fast && j->tx_miss >= 2) j->tx_fast = 0;
  }
  return 0;
}

int grdma_pair_debug_hist(grdma_pair* p, uint32_t* hist_out, uint64_t* count, uint32_t* period) {
  if (!p) return -1;
  grdma_conn c;
  if (int rc = fetch_conn(p, &c)) return rc;
  hipMemcpy(hist_out, p->d_hist, sizeof(uint32_t) * GRDMA_RX_HIST, hipMemcpyDeviceToHost);
  *count = c.rx_hist_count;
  *period = c.rx_period;
  return 0;
}

int grdma_stream_job_debug(grdma_stream_job* j, uint64_t* tx_dbg, uint64_t* rx_dbg) {
  if (!j) return -1;
  hipStreamSynchronize(j->stream);
  // result blocks alternate between even and odd rounds: GRDMA_DBG_ODD=1 shows the last odd round
  // (with 9 rounds per step the last even round is the short ninth one)
  const size_t slot = getenv("GRDMA_DBG_ODD") ? j->links.size() : 0;
  hipMemcpy(tx_dbg, j->d_txres[slot].dbg, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost);
  hipMemcpy(rx_dbg, j->d_rxres[slot].dbg, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost);
  return 0;
}

int grdma_stream_job_launch(grdma_stream_job* j) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  if (!j->exec) return fail(GRDMA_ERR_INVALID, "run the job once in GRDMA_RUN_GRAPH mode before launching it");
  if (int rc = job_ensure_exec(j)) return rc;  // (the last run may have switched the job to other planner kernels)
  HIP_TRY(hipGraphLaunch(j->exec, j->stream));
  return 0;
}

// What the HTTP/2 pipe (grdma_h2.hip) needs from a job: the device tables of one link and the
// stream the job's launches go to.
extern "C" __attribute__((visibility("hidden"))) int grdma_job_link_view(grdma_stream_job* j, uint32_t link, grdma_sge** d_sges,
                                                                         uint64_t* count, grdma_slice_out** d_slices,
                                                                         uint8_t** dst, hipStream_t* stream) {
  if (!j || link >= j->links.size()) return -1;
  grdma_job_link& l = j->links[link];
  j->sges_exposed = true;  // (the caller may rewrite the table: its index is rebuilt at every step from now on)
  *d_sges = l.d_sges;
  *count = l.count;
  *d_slices = l.d_slices;
  *dst = l.dst;
  *stream = j->stream;
  return 0;
}

// Kernel nodes in front of and behind the job INSIDE its graph (one launch per step, no graph boundary -- ~15-20 us of
// idle device each -- between the stages); null / 0 removes them.  The graph is rebuilt at the next launch.
extern "C" __attribute__((visibility("hidden"))) int grdma_job_set_hooks(grdma_stream_job* j, const grdma_job_hook* pre,
                                                                         uint32_t n_pre, const grdma_job_hook* post,
                                                                         uint32_t n_post) {
  if (!j) return -1;
  j->pre_hooks.assign(pre, pre + (pre ? n_pre : 0));
  j->post_hooks.assign(post, post + (post ? n_post : 0));
  j->hooks_gen++;
  return 0;
}

int grdma_stream_job_launch_engine(grdma_stream_job* j) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  return job_engine_enqueue(j, j->stream);
}

// profiling aid: {Sends, receive chunks, gather / wire / scatter entries, leader wait ticks x 4
// (sender: staging + slots, credit; receiver: data, scatter), abort code, team, waves x 3}
int grdma_stream_job_engine_prof(grdma_stream_job* j, uint32_t link, uint64_t out[12]) {
  if (int rc = require_ctx()) return rc;
  if (!j || !out || link >= j->links.size() || !j->links[link].d_lk) return fail(GRDMA_ERR_INVALID, "bad argument");
  HIP_TRY(hipStreamSynchronize(j->stream));
  static_assert(offsetof(lk_ctl, res_tx_phases) == offsetof(lk_ctl, res_prof) + 8 * sizeof(uint64_t), "layout");
  HIP_TRY(hipMemcpy(out, reinterpret_cast<uint8_t*>(j->links[link].d_lk) + offsetof(lk_ctl, res_prof), sizeof(uint64_t) * 12,
                    hipMemcpyDeviceToHost));
  return 0;
}

// profiling aid: the leaders' event traces (see lk_ctl::trace); out[who][0] = count
int grdma_stream_job_engine_trace(grdma_stream_job* j, uint32_t link, uint64_t out[5][193]) {
  if (int rc = require_ctx()) return rc;
  if (!j || !out || link >= j->links.size() || !j->links[link].d_lk) return fail(GRDMA_ERR_INVALID, "bad argument");
  HIP_TRY(hipStreamSynchronize(j->stream));
  std::vector<uint8_t> h(sizeof(lk_ctl));
  HIP_TRY(hipMemcpy(h.data(), j->links[link].d_lk, sizeof(lk_ctl), hipMemcpyDeviceToHost));
  const lk_ctl* c = reinterpret_cast<const lk_ctl*>(h.data());
  for (int w = 0; w < 5; w++) {
    out[w][0] = c->trace_n[w];
    for (int i = 0; i < 192; i++) out[w][1 + i] = c->trace[w][i];
  }
  return 0;
}

int grdma_stream_job_engine_stats(grdma_stream_job* j, uint32_t link, uint64_t out[16]) {
  if (int rc = require_ctx()) return rc;
  if (!j || !out || link >= j->links.size() || !j->links[link].d_lk) return fail(GRDMA_ERR_INVALID, "bad argument");
  HIP_TRY(hipStreamSynchronize(j->stream));
  lk_ctl h;
  HIP_TRY(hipMemcpy(&h, j->links[link].d_lk, sizeof(h), hipMemcpyDeviceToHost));
  memset(out, 0, sizeof(uint64_t) * 16);
  out[0] = h.res_sends; out[1] = h.res_chunks;
  for (int t = 0; t < 3; t++) out[2 + t] = h.res_entries[t];
  for (int t = 0; t < 4; t++) out[5 + t] = h.res_wait_ticks[t];
  out[9] = h.abort.v; out[10] = j->lk_team;
  for (int t = 0; t < 3; t++) out[11 + t] = h.nwaves[t];
  out[14] = h.n_staging;
  return 0;
}

int grdma_stream_job_launch_streams(grdma_stream_job* j) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  if (j->pipeline && j->cumask_bits > 0 && j->burst == 1) return job_enqueue_masked(j, j->stream);
  return j->pipeline ? job_enqueue_pipelined(j, j->stream) : job_enqueue(j, j->stream, false);
}

int grdma_stream_job_sync(grdma_stream_job* j) {
  if (int rc = require_ctx()) return rc;
  if (!j) return fail(GRDMA_ERR_INVALID, "null job");
  HIP_TRY(hipStreamSynchronize(j->stream));
  if (j->d_lk_ptrs) return job_engine_check(j);
  return 0;
}

// Delivered slices {offset into the link's destination, length} of link `link`.
int grdma_stream_job_slices_of(grdma_stream_job* j, uint32_t link, grdma_read_slice* out, uint64_t cap) {
  if (int rc = require_ctx()) return rc;
  if (!j || !out || link >= j->links.size()) return fail(GRDMA_ERR_INVALID, "bad argument");
  grdma_conn c;
  if (int rc = fetch_conn(j->links[link].rx, &c)) return rc;
  uint64_t n = c.rx_slice_idx < cap ? c.rx_slice_idx : cap;
  static_assert(sizeof(grdma_read_slice) == sizeof(grdma_slice_out), "layout");
  if (n) HIP_TRY(hipMemcpy(out, j->links[link].d_slices, sizeof(grdma_slice_out) * n, hipMemcpyDeviceToHost));
  return (int)n;
}

int grdma_stream_job_slices(grdma_stream_job* j, grdma_read_slice* out, uint64_t cap) {
  return grdma_stream_job_slices_of(j, 0, out, cap);
}

}  // extern "C"
