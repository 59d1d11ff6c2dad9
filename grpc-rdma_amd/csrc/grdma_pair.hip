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

extern "C" {
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
const void* grdma_kernel_fn_plan_pair_mw(void);
uint32_t grdma_rx_multi_groups(void);
hipError_t grdma_launch_rx_plan_mw(const grdma_rx_op*, uint32_t, hipStream_t);
uint32_t grdma_tx_multi_groups(void);
uint32_t grdma_tx_multi_max_sends(void);
uint32_t grdma_tx_multi_seq_sends(void);
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
  grdma_engine_mbox* mbi = nullptr;   // where the host writes its words: == mb, or a copy in device memory written through
                                      // the PCIe BAR (grdma_watch_ctl::inbox; GRDMA_MBOX_BAR=0: always mb)
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

#define GRDMA_HOST_RX_BLOCKS_DEFAULT 64u
#define GRDMA_HOST_TX_BLOCKS_DEFAULT 16u
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

// Workgroups of a copy launch whose bytes cross PCIe (an asynchronous endpoint: the gather reads the endpoint's pinned send
// buffers, the scatter writes the pinned receive window).  A link of ~57 GB/s is kept full by a few dozen waves; the
// grid an HBM copy wants (768 resident workgroups, 48 MiB of stores in flight) queues tens of microseconds of posted
// writes in front of the other direction's read requests: the gather of a Send ran at a quarter of its speed whenever a
// scatter was in flight (profiles/r05_vtable_timeline_before_throttle.txt, tools/pcie_duplex_probe).  dir 0: towards the host (scatter), 1: from
// the host (gather).  GRDMA_HOST_RX_BLOCKS / GRDMA_HOST_TX_BLOCKS: tuning knobs (0 = the HBM grid).
uint32_t host_copy_blocks(int dir, uint32_t hbm_blocks) {
  static const uint32_t cfg[2] = {
      [] { const char* e = getenv("GRDMA_HOST_RX_BLOCKS"); return e ? (uint32_t)atol(e) : GRDMA_HOST_RX_BLOCKS_DEFAULT; }(),
      [] { const char* e = getenv("GRDMA_HOST_TX_BLOCKS"); return e ? (uint32_t)atol(e) : GRDMA_HOST_TX_BLOCKS_DEFAULT; }()};
  const uint32_t v = cfg[dir];
  return v == 0 || v > hbm_blocks ? hbm_blocks : v;
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
  uint8_t* d_zc = nullptr;          // host-writable AND device-readable unless zc_mem == GRDMA_ZC_MEM_DEVICE
  uint64_t zc_cap = 0;
  int zc_mem = GRDMA_ZC_MEM_HOST;
  uint32_t zc_tail = 0;              // zerocopy_buffer_tail_ (std::atomic_uint32_t)
  uint64_t zc_bytes = 0, zc_copy_bytes = 0, zc_last_sges = 0;
  std::mutex zc_mu;
  // standing read order (grdma_pair_arm_read): a watcher workgroup of the engine carries it out when bytes land in this
  // pair's ring, whoever wrote them (k_watch).  armed_reads: max_reads of the order, 0 = none; armed_done: a completion
  // the watcher produced waits in the result block although the order has been taken back (engine stopped, disarmed);
  // then: the slot the order was handed, the sequence word of the next completion, completions taken since the arming
  // (what grdma_engine_mbox::consumed tells the device), completions taken in all
  uint64_t armed_reads = 0;
  std::atomic<bool> armed_done{false};
  grdma_verbs_wire* verbs = nullptr; // != NULL: a NIC writes the peer's ring and mine (csrc/grdma_wire_verbs.cc)
  int watch_slot = -1;
  uint64_t watch_expect = 0, watch_taken = 0, watch_hits = 0;
  uint64_t armed_half = 0;           // which half of the arena the completion `armed_done` stands for lies in (a
                                     // watcher's completion that was left behind when its order was taken back)
  bool watch_parked = false;         // asynchronous endpoint: the last completion was taken while the transport held
                                     // every other window -- the watcher waits for the word that names the next one
  // asynchronous endpoint operations (grdma_endpoint_set_async): the send and the receive direction on streams of
  // their own, one Send and one drain in flight at most, completions in pinned host memory
  // What a peer IN THIS PROCESS that went away first has left in my care: its ring, its connection block and its state
  // line -- my device-side connection block still names them (peer_ring, peer_status / peer_wire, peer_line), and a Send
  // of mine that was under way, or that my owner starts before it has asked get_status(), writes there.  The reference's
  // Send to a peer that has exited fails in the HCA (pair.cc:500-558: a work completion in error); here the memory simply
  // outlives the peer until I go too (grdma_pair_destroy), so that write is harmless instead of a write into freed --
  // or, with the PairPool, into somebody else's -- memory.
  struct orphan { void* ptr; size_t bytes; int kind; };   // kind: pool_free's, 3 = a state line
  std::vector<orphan> orphans;
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


#include "grdma_host_engine.inc"
#include "grdma_host_pair.inc"
#include "grdma_host_poller.inc"
#include "grdma_host_endpoint.inc"
#include "grdma_host_job.inc"
