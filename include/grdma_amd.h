/* grdma_amd.h -- C ABI of the MI355X-native RDMA_BP/BPEV endpoint data plane.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  Each entry point names the reference interface it replaces (paths
 * relative to the pwrliang/grpc-rdma tree).  INTEGRATION.md shows the C++
 * adapter that fills the real grpc_endpoint_vtable from these calls.
 *
 * Memory model: rings, staging buffers, credit words and the per-connection
 * protocol state live in device HBM.  `grdma_slice.ptr` may point to device
 * memory or to host memory that is device-accessible (hipHostMalloc /
 * hipHostRegister); plain pageable host memory is accepted when the call is
 * flagged GRDMA_MEM_HOST (the library then stages it through a pinned bounce
 * buffer).  All functions return <0 (a negated grdma_error) on failure and
 * never fall back to a CPU implementation: without a usable HIP device
 * grdma_init() fails and every other call reports GRDMA_ERR_NO_DEVICE.
 *
 * Threading (the contract of PairPollable, pair.h:64-81): different pairs may be driven from
 * different threads at the same time; on ONE pair at most one thread sends / writes and at most
 * one thread receives / reads at any moment (never two readers or two writers).  The read-only
 * queries (has_message, has_pending_writes, get_status, readable/writable size, poll_pairs) may
 * run beside them from any thread.  The latency engine's mailbox is serialised inside the
 * library.  grdma_last_error() is per thread.
 */
#ifndef GRDMA_AMD_H
#define GRDMA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRDMA_ABI_VERSION 2

enum grdma_error {
  GRDMA_OK = 0,
  GRDMA_ERR_NO_DEVICE = 1,   /* no HIP device / HIP runtime failure at init          */
  GRDMA_ERR_INVALID = 2,     /* bad argument (ring size not a power of two, ...)     */
  GRDMA_ERR_HIP = 3,         /* a HIP call failed; see grdma_last_error()            */
  GRDMA_ERR_NOT_CONNECTED = 4,
  GRDMA_ERR_CAPACITY = 5,    /* slice list / arena larger than the configured caps   */
  GRDMA_ERR_CONFIG = 6,      /* GRPC_PLATFORM_TYPE / GRPC_RDMA_* value rejected      */
  GRDMA_ERR_AGAIN = 7        /* an asynchronous operation has not completed yet       */
};

/* ---- platform selection: src/core/lib/iomgr/iomgr_internal.cc:37-62 --------- */
enum grdma_platform {          /* iomgr_internal.h:45 platform_t */
  GRDMA_IOMGR_TCP = 0,
  GRDMA_IOMGR_RDMA_BP = 1,
  GRDMA_IOMGR_RDMA_BPEV = 2,
  GRDMA_IOMGR_RDMA_EVENT = 3
};
/* Parses a GRPC_PLATFORM_TYPE value: NULL/unset -> TCP, exact strings "TCP",
 * "RDMA_BP", "RDMA_BPEV", "RDMA_EVENT"; anything else -> -GRDMA_ERR_CONFIG
 * (the reference calls exit(1) there, iomgr_internal.cc:57-58). */
int grdma_parse_platform(const char* value);
/* grpc_determine_iomgr_platform(): reads the environment once. */
int grdma_determine_platform(void);

/* ---- Config: src/core/lib/ibverbs/config.cc:45-115 --------------------------- */
typedef struct grdma_config {
  char device_name[64];              /* GRPC_RDMA_DEVICE_NAME   ("" = first)      */
  int32_t port_num;                  /* GRPC_RDMA_PORT_NUM      (1)               */
  int32_t gid_index;                 /* GRPC_RDMA_GID_INDEX     (0)               */
  int32_t poller_thread_num;         /* GRPC_RDMA_POLLER_THREAD_NUM (1)           */
  int32_t busy_polling_timeout_us;   /* GRPC_RDMA_BUSY_POLLING_TIMEOUT_US (500)   */
  int32_t poller_sleep_timeout_ms;   /* GRPC_RDMA_POLLER_SLEEP_TIMEOUT_MS (1000)  */
  uint32_t ring_buffer_size_kb;      /* GRPC_RDMA_RING_BUFFER_SIZE_KB (4096)      */
  uint32_t zerocopy_buffer_size_kb;  /* reads the SAME variable (config.cc:100-106), 32768 */
  uint32_t zerocopy_threshold_kb;    /* GRPC_RDMA_ZEROCOPY_THRESHOLD_KB (UINT32_MAX) */
  int32_t max_sge;                   /* GRPC_RDMA_MAX_SGE: this build's stand-in for
                                        the HCA attribute max_sge (pair.cc:53-60); 30 */
  int32_t hip_device;                /* GRPC_RDMA_HIP_DEVICE (LOCAL_RANK or 0)    */
  int32_t hip_wire_direct;           /* GRPC_RDMA_HIP_WIRE: "direct" (1, default: records are encoded straight into
                                        the peer ring -- every wire of this build is device memory the sender can
                                        address) or "staged" (0: through the ring/2 staging buffer and <= 2 wire
                                        writes, what a NIC posts)                                          */
  uint32_t hip_register_min;         /* GRPC_RDMA_HIP_REGISTER_MIN: host slices of at least this many bytes are read
                                        where they lie (pages registered once), 0 = always copied (default) */
  uint32_t hip_pair_pool_mb;         /* GRPC_RDMA_HIP_PAIR_POOL_MB: how much memory of closed connections the PairPool
                                        keeps for the next ones (default 4096; 0 = none)                    */
} grdma_config;
int grdma_config_from_env(grdma_config* out);

/* ---- library / device ------------------------------------------------------------ */
int grdma_init(int hip_device);      /* Device::Get(), device.cc:45-101           */
int grdma_device_count(void);
const char* grdma_last_error(void);
int grdma_abi_version(void);

/* ---- PairPollable: src/core/lib/ibverbs/pair.h:106-152 ---------------------- */
typedef struct grdma_pair grdma_pair;

typedef struct grdma_slice {         /* GRPC_SLICE_START_PTR / GRPC_SLICE_LENGTH  */
  const void* ptr;
  uint64_t len;
} grdma_slice;

enum grdma_flags {
  GRDMA_MEM_DEVICE = 0,   /* slice pointers are device-accessible                  */
  GRDMA_MEM_HOST = 1,     /* pageable host memory: stage through the bounce buffer */
  GRDMA_WIRE_STAGED = 0,  /* records are built in the staging buffer, then written
                             to the peer ring by <=2 wire writes (what a NIC needs) */
  GRDMA_WIRE_DIRECT = 2,  /* loop-back / xGMI peer: encode straight into the ring  */
  GRDMA_WIRE_ORDERED = 8, /* what writes into THIS pair's ring places the bytes of a write in address order, the
                             footer last -- an RDMA NIC (the reference's wire): header + footer say "complete",
                             as ring_buffer.cc:67-97 assumes.  Without the flag the writer is taken to be a HIP
                             wire (copy kernel, IPC / xGMI peer): a PARALLEL copy, whose sender reports how far
                             its completed writes reach (csrc/grdma_dev.h: grdma_wire_report) and whose receiver
                             never walks past that report */
  GRDMA_RING_FINE_GRAINED = 4  /* allocate the ring (and the connection block holding the 16-byte
                             status report) as fine-grained device memory
                             (hipExtMallocWithFlags, hipDeviceMallocFinegrained): stores are
                             write-through and visible to a peer device / process / NIC once
                             acknowledged, with no cache maintenance at kernel boundaries --
                             what a ring registered for remote writes needs */
};

/* PairPollable() + Init(): allocates the HBM ring (ring_size bytes, power of
 * two > 24, ring_buffer.cc:22-24), the ring/2 staging buffer (pair.cc:104), the
 * status words and a receive arena.  pair.cc:21-76, 85-141. */
grdma_pair* grdma_pair_create(uint64_t ring_size, int max_sge, int flags);
/* Connect(): the loop-back wire replaces the 48-byte address exchange + QP
 * bring-up (rdma_bp_posix.cc:767-771, pair.cc:143-168).  Both ends must use the
 * same ring size (asserted at pair.cc:149). */
int grdma_pair_connect(grdma_pair* a, grdma_pair* b);
/* Bootstrap between processes / GPUs.  The reference swaps a 48-byte Address over the TCP fd
 * (exchange_data, rdma_bp_posix.cc:640-692,767-771; layout address.h:24-31), checks tag and ring
 * size in Connect() (pair.cc:143-149), brings the QP up and swaps the memory regions of ring and
 * status buffer (syncMemoryRegion).  Here the memory region of an HBM ring is a HIP IPC handle
 * (the same allocation exports as a dma-buf for a NIC); both travel in one blob whose first 48
 * bytes are the reference's Address.  export -> send/recv over any byte channel -> connect_remote,
 * or grdma_pair_bootstrap_fd() which does all three over a connected socket like
 * grpc_rdma_bp_create does (rdma_bp_posix.cc:763-784). */
typedef struct grdma_address {          /* address.h:24-31 (48 bytes, natural alignment)          */
  uint32_t lid, qpn, psn, pad0;
  uint8_t gid[16];                      /* union ibv_gid; here: PCI bus id of the HIP device      */
  uint32_t tag, pad1;                   /* 0xa0: peers must agree (pair.cc:72,146)                */
  uint64_t ring_buffer_size;            /* peers must agree (pair.cc:107,147-149)                 */
} grdma_address;
typedef struct grdma_bootstrap_blob {
  grdma_address addr;
  uint32_t magic, version;              /* "GRDM", GRDMA_ABI_VERSION                              */
  int32_t hip_device;
  uint32_t wire_off;                    /* offset of the arrival report (grdma_wire_report) in the conn block */
  uint64_t pid;
  uint64_t status_off;                  /* offset of the 16-byte status_report in the conn block  */
  uint8_t ring_handle[64];              /* hipIpcMemHandle_t of the ring                          */
  uint8_t conn_handle[64];              /* hipIpcMemHandle_t of the connection block              */
} grdma_bootstrap_blob;
int grdma_pair_export_address(grdma_pair* p, grdma_bootstrap_blob* out);
int grdma_pair_connect_remote(grdma_pair* p, const grdma_bootstrap_blob* peer);
int grdma_pair_bootstrap_fd(grdma_pair* p, int fd);
int grdma_pair_disconnect(grdma_pair* p);          /* Disconnect(), pair.cc:325-347 */
void grdma_pair_destroy(grdma_pair* p);

/* ---- PairPool (src/core/lib/ibverbs/pair.h:273-333: Take(id) / Get(id) / Putback over pre-built pairs) --------
 * Here the pool keeps the MEMORY of released pairs (ring, staging, plans, arena, pinned blocks) by size and
 * hands it to the next grdma_pair_create / _take of the same shape, and maps connection ids to pairs. */
int grdma_pair_pool_reserve(uint32_t pairs, uint64_t ring_size, int max_sge, int flags, uint64_t cap_bytes);
grdma_pair* grdma_pair_pool_take(const char* id, uint64_t ring_size, int max_sge, int flags);  /* PairPool::Take  */
grdma_pair* grdma_pair_pool_get(const char* id);                                              /* PairPool::Get   */
void grdma_pair_pool_putback(grdma_pair* p);                                                  /* PairPool::Putback */
int grdma_pair_pool_stats(uint64_t out[5]);  /* blocks held, bytes held, pool hits, runtime allocations, ids registered */
void grdma_pair_pool_trim(void);
/* PairStatus, pair.h:44-51: the values grdma_pair_get_status returns */
#ifndef GRDMA_PAIR_STATUS_DEFINED
#define GRDMA_PAIR_STATUS_DEFINED
enum grdma_pair_status {
  GRDMA_PAIR_UNINITIALIZED = 0,
  GRDMA_PAIR_INITIALIZED = 1,
  GRDMA_PAIR_CONNECTED = 2,
  GRDMA_PAIR_HALF_CLOSED = 3,
  GRDMA_PAIR_DISCONNECTED = 4,
  GRDMA_PAIR_ERROR = 5
};
#endif
int grdma_pair_get_status(grdma_pair* p);          /* get_status(), pair.cc:349-375 */

/* Send(slices, count, byte_idx), pair.cc:645-734.  Returns payload bytes
 * accepted (a prefix of the slice list) or <0. */
int64_t grdma_pair_send(grdma_pair* p, const grdma_slice* slices, uint64_t count,
                        uint64_t byte_idx, int flags);
/* Recv(buf, capacity), pair.cc:264-286: one RingBufferPollable::Read.  dst is
 * device-accessible memory (or host memory with GRDMA_MEM_HOST). */
int64_t grdma_pair_recv(grdma_pair* p, void* dst, uint64_t capacity, int flags);
int grdma_pair_has_message(grdma_pair* p);         /* HasMessage(), ring_buffer.cc:56-65   */
int grdma_pair_has_pending_writes(grdma_pair* p);  /* HasPendingWrites(), pair.cc:303      */
int64_t grdma_pair_readable_size(grdma_pair* p);   /* GetReadableSize(), ring_buffer.cc:67 */
int64_t grdma_pair_writable_size(grdma_pair* p);   /* GetWritableSize(), pair.cc:294-301   */

/* Zero-copy send buffer (pair.h:96,127,140; pair.cc:103-120, 305-323, 793-941).
 * grdma_pair_enable_zerocopy   initSendBuffer(kZeroCopyBuffer): `bytes` owned by the pair (0 =
 *                              GRPC_RDMA_ZEROCOPY_BUFFER_SIZE_KB as the reference's Config reads it);
 *                              allocate_send_buffer enables it on first use.  HOST-WRITABLE and device-readable:
 *                              the reference's caller serialises into it with the CPU (GenericSerialize ->
 *                              SerializeWithCachedSizesToArray, include/grpcpp/impl/codegen/proto_utils.h:68-95, through
 *                              CoreCodegen::grpc_call_allocate_send_buffer, src/cpp/common/core_codegen.cc:122-146).
 *                              _ex names the memory: GRDMA_ZC_MEM_HOST (pinned mapped host memory, the default: the
 *                              gather pulls the payload over PCIe into the peer ring), GRDMA_ZC_MEM_BAR (fine-grained
 *                              device memory the CPU writes through the PCIe BAR), GRDMA_ZC_MEM_DEVICE (device memory
 *                              for serialisers that run on the device; not host-writable).  Environment:
 *                              GRPC_RDMA_HIP_ZEROCOPY_MEM = host | bar | device.
 * grdma_pair_allocate_send_buffer  AllocateSendBuffer(size): a pointer into that buffer, or NULL
 *                              when size == 0, the buffer is not empty (the reference serves one
 *                              allocation at a time: tail != 0 -> nullptr) or the size does not fit.
 *                              The caller serialises its message there (plain CPU stores for HOST / BAR).
 * grdma_pair_send_zerocopy     SendZerocopy(slices, count, byte_idx): a slice that lies inside the
 *                              zero-copy buffer becomes a record whose payload is read where it lies
 *                              (limited by the receiver's credit only; the reference stages 16 +
 *                              padding tag bytes and uses 4 of its max_sge entries for it), any other
 *                              slice is priced as Send prices it.  GRDMA_MEM_DEVICE: every slice is device-
 *                              accessible; GRDMA_MEM_HOST: the slices are the host's (what grpc_endpoint_write hands
 *                              over) -- ranges of the zero-copy buffer are still read in place, the others are
 *                              copied through the pinned bounce buffer as grdma_pair_send does.
 *                              Returns the payload bytes accepted; partial_write, remote_tail and the
 *                              <= 2 work requests (grdma_pair_last_wrs) as the reference leaves them.
 *                              On this wire every record of the call goes straight from its source into
 *                              the peer ring (one gather launch); nothing is written to the staging buffer.
 * grdma_pair_zerocopy_state    out = {zerocopy_buffer_tail_, zerocopy_bytes_, copy_bytes_, scatter-gather
 *                              entries the reference's list would hold after the last call}. */
enum grdma_zc_mem { GRDMA_ZC_MEM_HOST = 0, GRDMA_ZC_MEM_BAR = 1, GRDMA_ZC_MEM_DEVICE = 2 };
int grdma_pair_enable_zerocopy(grdma_pair* p, uint64_t bytes);
int grdma_pair_enable_zerocopy_ex(grdma_pair* p, uint64_t bytes, int mem);
int grdma_pair_zerocopy_mem(grdma_pair* p);   /* the kind of the buffer in use, or -1 */
void* grdma_pair_allocate_send_buffer(grdma_pair* p, uint64_t size);
int64_t grdma_pair_send_zerocopy(grdma_pair* p, const grdma_slice* slices, uint64_t count,
                                 uint64_t byte_idx, int flags);
int grdma_pair_zerocopy_state(grdma_pair* p, uint64_t out[4]);

/* Observability used by the parity tests (debug monitor of pair.h:235-270). */
typedef struct grdma_pair_state {
  uint64_t head, moving_head, remain;              /* ring_buffer.h:203-205 */
  uint64_t remote_tail, remote_head;               /* pair.h:170, 229-233   */
  uint64_t internal_read_size, credit_msgs, partial_write;
  uint64_t total_read, total_written, leftover_cap;
} grdma_pair_state;
int grdma_pair_state_get(grdma_pair* p, grdma_pair_state* out);
/* Copies `len` bytes of the pair's HBM ring / staging buffer to host memory. */
int grdma_pair_peek_ring(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len);
int grdma_pair_peek_staging(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len);
/* Last Send's work requests {remote ring offset, length}; returns the count
 * (GetWriteRequests, ring_buffer.cc:261-330). */
int grdma_pair_last_wrs(grdma_pair* p, uint64_t out[2][2]);
void* grdma_pair_ring_device_ptr(grdma_pair* p);

/* ---- background poller (RDMA_BPEV): src/core/lib/ibverbs/poller.h:16-71, poller.cc:12-106 ----
 * n_threads host threads (GRPC_RDMA_POLLER_THREAD_NUM, must be > 0) share one round-robin cursor over the
 * slot table, as the reference's do (poller.cc:66-69).  What a thread looks at is the pair's host-visible
 * state line -- no device work for a pair whose peer lives in this process.  A pair's wakeup fd (an
 * eventfd; the reference's grpc_wakeup_fd, pair.h:150,187) becomes readable when the pair is connected
 * and readable or writable (a message, a completed drain or Send, credit for a parked write), or when it
 * is half-closed / in error (poller.cc:80-98); it is not signalled again until the consumer has read it
 * (poller.cc:76-78).  The event engine adds the fd to its epoll set exactly as
 * ev_epollex_rdma_bpev_linux.cc does.  sleep_timeout_ms mirrors GRPC_RDMA_POLLER_SLEEP_TIMEOUT_MS. */
typedef struct grdma_poller grdma_poller;
grdma_poller* grdma_poller_create(int n_threads, int sleep_timeout_ms);
void grdma_poller_destroy(grdma_poller* pl);                 /* Poller::Shutdown          */
int grdma_poller_add(grdma_poller* pl, grdma_pair* p);       /* AddPollable; returns the fd */
int grdma_poller_remove(grdma_poller* pl, grdma_pair* p);    /* RemovePollable; afterwards the
                                                              * poller no longer touches p   */
int grdma_poller_stats(grdma_poller* pl, uint64_t* passes, uint64_t* wakeups);
int grdma_poller_threads(grdma_poller* pl);                  /* threads running (GRPC_RDMA_POLLER_THREAD_NUM) */
int grdma_pair_get_wakeup_fd(grdma_pair* p);                 /* PairPollable::get_wakeup_fd */
int grdma_pair_consume_wakeup(grdma_pair* p);                /* grpc_wakeup_fd_consume_wakeup: 1 if one was pending */

/* ---- endpoint read/write on the pair: rdma_bp_posix.cc ----------------------- */
typedef struct grdma_read_slice { uint64_t off, len; } grdma_read_slice;

/* rdma_write()/rdma_flush() (rdma_bp_posix.cc:470-586): sends as much of the
 * slice list as credit allows, continuing from the internal cursor
 * (outgoing_byte_idx).  *done = 1 when the whole buffer went out; otherwise the
 * caller retries when the pair becomes writable (notify_on_write). */
int64_t grdma_endpoint_write_begin(grdma_pair* p, const grdma_slice* slices, uint64_t count,
                                   int flags);
int64_t grdma_endpoint_write_step(grdma_pair* p, int* done);
/* Forget the outstanding write (the error exits of rdma_flush, :505-517, unref the slice
 * buffer; afterwards a new grdma_endpoint_write_begin is accepted). */
int grdma_endpoint_write_abort(grdma_pair* p);

/* rdma_read()/rdma_continue_read()/rdma_do_read() (rdma_bp_posix.cc:180-376):
 * performs up to max_reads endpoint_read completions in ONE device pass; each
 * completion is one slice in the pair's receive arena.  Returns the number of
 * completions; slices[i] = {arena offset, length}.  *would_block = 1 when the
 * last attempt found no complete record (the endpoint re-arms notify_on_read). */
int64_t grdma_endpoint_read(grdma_pair* p, uint64_t max_reads, grdma_read_slice* slices,
                            uint64_t slices_cap, int* would_block);
/* ---- asynchronous endpoint operations ----------------------------------------------------------
 * What the grpc_endpoint_vtable adapter runs on (include/grdma_endpoint_impl.hpp).  grpc_endpoint_write and
 * grpc_endpoint_read return once the device work is enqueued -- a launch chain on the pair's send / receive stream,
 * or one command to the resident latency engine when the pair is in latency mode --; the completion shows up in
 * pinned host memory and the event engine's poll loop finds it with plain loads (grdma_endpoint_readable /
 * _writable are what HasMessage() / HasPendingWrites() mean to ev_epollex_rdma_bp{,ev}_linux.cc for such an
 * endpoint).  Delivered slices lie in receive windows of pinned host memory that the scatter kernel writes over
 * PCIe: the transport gets slices that POINT into the window (no copy on the host), each holding a reference.
 * One Send and one drain in flight per pair; the blocking calls above must not be mixed in while one is.
 *
 * grdma_endpoint_set_async(p, windows, window_bytes)  windows >= 2 (0 = 3); window_bytes 0 = sized from the ring.
 * grdma_endpoint_write_submit(p)    one rdma_flush step (a Send from the cursor of the begun write), not waited for.
 * grdma_endpoint_write_test(p, &done, &sent)   1: the step completed (as grdma_endpoint_write_step reports it),
 *                                   0: still in flight, < 0: error.
 * grdma_endpoint_read_submit(p, max_reads)     0: a drain of up to max_reads endpoint reads is on its way,
 *                                   1: every window still holds slices the transport has not released.
 * grdma_endpoint_read_test(p, slices, cap, &would_block, &window)   >= 0: completions, slices[i].off relative to
 *                                   grdma_window_base(window); the caller owns one reference of the window
 *                                   (grdma_window_unref when the last slice is gone); -GRDMA_ERR_AGAIN: in flight.
 * grdma_pair_arm_read(p, n) on an asynchronous pair in latency mode: a standing read order with a watcher workgroup of
 *                                   the resident engine, carried out when bytes land in the pair's ring (whoever wrote
 *                                   them); to the endpoint it is a drain in flight that completes by itself. */
typedef struct grdma_window grdma_window;
int grdma_endpoint_set_async(grdma_pair* p, int windows, uint64_t window_bytes);
int grdma_endpoint_write_submit(grdma_pair* p);
int grdma_endpoint_write_test(grdma_pair* p, int* done, int64_t* sent);
/* A write submitted BEHIND the burst in flight (asynchronous endpoint, direct wire, a write of more slices than
 * max_sge and at most 16 x max_sge): its chain goes into the send stream now and runs only if the write in front
 * turns out to have gone out whole (gated on the device); slices must be device-visible and stay valid.
 * _queue: 0 = queued, 1 = not possible now.  _adopt, after grdma_endpoint_write_test reported the write in front
 * complete: 1 = the queued write is the outstanding one now, its burst in flight; 0 = it has to be submitted the
 * ordinary way (never queued, or skipped because the write in front came up short). */
int grdma_endpoint_write_queue(grdma_pair* p, const grdma_slice* slices, uint64_t count);
int grdma_endpoint_write_adopt(grdma_pair* p);
int grdma_endpoint_write_queue_stats(grdma_pair* p, uint64_t out[3]);  /* queued, promoted, skipped */
int grdma_endpoint_write_queue_limits(grdma_pair* p, uint64_t out[2]); /* slices, bytes one queued write may hold */
int grdma_endpoint_write_quiesce(grdma_pair* p);  /* nothing of this pair's is left in its send stream when it returns */
int grdma_endpoint_read_submit(grdma_pair* p, uint64_t max_reads);
int64_t grdma_endpoint_read_test(grdma_pair* p, grdma_read_slice* slices, uint64_t slices_cap, int* would_block,
                                 grdma_window** window);
int grdma_endpoint_read_idle(grdma_pair* p);      /* an endpoint read found nothing and submitted no drain: the 256-byte
                                                     slice it allocated stays for the next edge (rdma_bp_posix.cc:283-287) */
int grdma_endpoint_readable(grdma_pair* p);
int grdma_endpoint_writable(grdma_pair* p);
int grdma_endpoint_busy(grdma_pair* p);           /* 1: a Send or a drain of this pair is on the device and has not
                                                     completed yet (an event loop running a connection to quiescence) */
int grdma_endpoint_drain_state(grdma_pair* p);    /* 0: no drain in flight, 1: in flight, 2: completed, not collected
                                                     (a drain may have been posted by the in-process peer's sender
                                                     on behalf of an armed read) */
int grdma_endpoint_free_windows(grdma_pair* p);   /* receive windows no slice of the transport points into */
const void* grdma_window_base(const grdma_window* w);
void grdma_window_ref(grdma_window* w);
void grdma_window_unref(grdma_window* w);
/* GRDMA_MEM_HOST slices of at least this many bytes are read where they lie -- the library registers their pages
 * with the device once (hipHostRegister) and remembers the registration -- instead of being copied into the
 * pinned bounce buffer.  0 (default) = always copy.  The caller guarantees that a registered range stays mapped
 * (memory handed back to the OS must be announced with grdma_forget_host_range first).  Env: GRPC_RDMA_HIP_REGISTER_MIN. */
int grdma_set_host_register_min(uint64_t bytes);
int grdma_forget_host_range(const void* ptr, uint64_t len);

/* Export the ring as a dma-buf file descriptor (hipMemGetHandleForAddressRange,
 * hipMemRangeHandleTypeDmaBufFd): what ibv_reg_dmabuf_mr() takes to register the HBM ring with an
 * RDMA NIC -- the place of ibv_reg_mr() in the reference (rdma_utils.h:108-160, pair.cc:107-119).
 * Returns the fd (owned by the caller: close() it) or < 0. */
int grdma_pair_export_ring_dmabuf(grdma_pair* p);
/* ---- NIC wire (ibverbs): what PairPollable's verbs calls are in the reference --------------------------------
 * grdma_pair_verbs_open     Init(): device, protection domain, completion queue, queue pair; ring and status block
 *                           registered remote-writable -- the ring through its dma-buf where the verbs library has
 *                           ibv_reg_dmabuf_mr -- staging buffer and status_send as local regions (pair.cc:107-119,
 *                           rdma_utils.h:108-160).  The pair must have been created with GRDMA_WIRE_ORDERED (a NIC places
 *                           the bytes of a write in order, footer last) and without GRDMA_WIRE_DIRECT.
 * grdma_pair_verbs_address  what the reference's Address carries for this end (pair.h:53-98): queue pair number, lid,
 *                           gid, ring and status addresses with their rkeys.
 * grdma_pair_verbs_connect  Connect(peer): RESET -> INIT -> RTR -> RTS (pair.cc:143-262); the pair is kConnected and
 *                           from then on a Send's <= 2 chained IBV_WR_RDMA_WRITEs (pair.cc:709-734: what the send planner
 *                           left in grdma_pair_last_wrs) and a drain's 16-byte status write (updateStatus, :624-641) are
 *                           posted through the queue pair and reaped (csrc/grdma_wire_verbs.cc).
 * Libraries built without <infiniband/verbs.h> (grdma_verbs_supported() == 0) report an error from _open. */
typedef struct grdma_verbs_address {
  uint32_t qpn, psn;
  uint16_t lid;
  uint16_t pad0;
  uint32_t ring_rkey;
  uint8_t gid[16];
  uint64_t ring_addr, ring_size;
  uint64_t status_addr;
  uint32_t status_rkey, status_size;
} grdma_verbs_address;
int grdma_verbs_supported(void);
int grdma_pair_verbs_open(grdma_pair* p, const char* device, int port, int gid_index);
int grdma_pair_verbs_address(grdma_pair* p, grdma_verbs_address* out);
int grdma_pair_verbs_connect(grdma_pair* p, const grdma_verbs_address* peer);
int grdma_pair_verbs_counts(grdma_pair* p, uint64_t out[3]);  /* data writes posted, status writes posted, completions reaped */
void* grdma_pair_arena_device_ptr(grdma_pair* p);
uint64_t grdma_pair_arena_size(grdma_pair* p);
int grdma_pair_arena_copy_out(grdma_pair* p, uint64_t off, void* host_dst, uint64_t len);

/* ---- latency path ------------------------------------------------------------------------
 * Latency mode: every blocking call becomes ONE fused kernel launch (the planning
 * workgroup also moves the bytes), the host waits on a sequence word in pinned
 * memory instead of a stream synchronize, and delivered slices are written
 * straight into a pinned host arena. */
int grdma_pair_set_latency_mode(grdma_pair* p, int on);
/* Standing read order: what an outstanding grpc_endpoint_read is to the reference's busy-polling thread
 * (rdma_bp_posix.cc:345-372 arms it, the poller completes it when a record lands, ring_buffer.cc:56-97).
 * "grdma_endpoint_read(max_reads) into the next free window, whenever there is something": the order sits in one
 * of the resident engine's 64 watch slots; a watcher workgroup polls the arrival report of THIS pair's ring and
 * drains when it moves -- whoever wrote the bytes (the in-process peer, another process over IPC, a NIC) --, and the
 * next grdma_endpoint_read (max_reads >= the armed value) finds the completion in pinned memory without a command
 * of its own.  Same bytes, order and state as a read issued at that moment.  Latency mode only; not on a NIC-wire
 * pair.  max_reads = 0 takes the order back.  One reading thread per pair. */
int grdma_pair_arm_read(grdma_pair* p, uint64_t max_reads);
int64_t grdma_pair_watch_hits(const grdma_pair* p);   /* completions a watcher workgroup produced and a read took */
int grdma_engine_watchers(void);                      /* watcher workgroups per engine incarnation (GRDMA_ENGINE_WATCHERS) */
int grdma_pair_armed_ready(const grdma_pair* p);      /* 1: a completion is waiting (host memory only: no device work) */
/* Persistent latency engine: one resident workgroup takes the fused Send / drain
 * commands of latency-mode pairs from a mailbox in pinned host memory (a PCIe
 * doorbell read instead of a kernel launch per call).  It retires by itself after
 * ~1 s without commands; stop it before any device-wide synchronize. */
int grdma_engine_start(void);
int grdma_engine_stop(void);
/* Unary ping-pong (the reference's micro-bench loop, examples/cpp/micro-bench/mb_client.cc):
 * a = client end, b = server end of a connected link.  rtt_ns has `iters` entries. */
int grdma_pingpong(grdma_pair* a, grdma_pair* b, const grdma_slice* req, uint64_t nreq,
                   const grdma_slice* resp, uint64_t nresp, int mem_flags, uint64_t iters,
                   uint64_t warmup, uint64_t* rtt_ns, uint64_t phase_ns[4]);

/* One END of a unary ping-pong whose other end lives in another process (the rings mapped through IPC handles,
 * grdma_pair_bootstrap_fd): is_client = write `out`, then read until in_bytes have arrived; server = the other way
 * round.  With a standing read order (grdma_pair_arm_read) the bytes the OTHER process wrote into this pair's ring are
 * found by a watcher workgroup of this process's engine.  byte_sum = sum of every byte received. */
int grdma_pingpong_end(grdma_pair* p, int is_client, const grdma_slice* out, uint64_t nout, uint64_t in_bytes, int mem_flags,
                       uint64_t iters, uint64_t warmup, uint64_t* rtt_ns, uint64_t* byte_sum);

/* ---- K3 batched message-ready detection ------------------------------------------- */
/* HasMessage()/GetReadableSize() for n pairs in one launch (the busy-poll scan
 * of ev_epollex_rdma_bpev_linux.cc:1105-1149 / poller.cc:84).  readable[i] = bytes,
 * has_message[i] = 0/1. */
int grdma_poll_pairs(grdma_pair* const* pairs, uint32_t n, uint64_t* readable,
                     uint8_t* has_message);

/* ---- device-resident streaming job ------------------------------------------------
 * Pushes a whole slice list (e.g. a batch of framed gRPC messages) through a
 * connected loop-back link without host round trips: each round is one
 * rdma_flush step on `tx` (Send from the device-side cursor), the wire write,
 * and one drain of `rx` (as many endpoint_read completions as are ready),
 * appended to rx_dst.  This is the reference's write loop
 * (rdma_bp_posix.cc:470-557) and read loop (:180-376) with the credit
 * hand-shake of pair.cc:264-301 between them, enqueued back to back. */
typedef struct grdma_stream_job grdma_stream_job;
typedef struct grdma_stream_result {
  uint64_t bytes_sent, bytes_delivered, slices_delivered;
  uint64_t tx_rounds, rx_rounds, tx_records, rx_records;
  uint64_t done;               /* 1: every slice was sent and delivered            */
  double ms_total;             /* HIP-event time of the enqueued rounds            */
  double ms_class[8];          /* instrumented modes: summed kernel time per class */
  uint64_t launches_class[8];  /*   0 tx_plan 1 gather 2 wire 3 rx_plan 4 rx_apply (scatter+zero+credit)
                                *   _INSTRUMENTED_SCHEDULE also: 5 plan_pair (drain plan of round t + send plan of
                                *   round t + 1, one launch) 6 scatter_gather (scatter of round t + gather of t + 1) */
} grdma_stream_result;
enum grdma_stream_mode {
  GRDMA_RUN_EAGER = 0, GRDMA_RUN_GRAPH = 1, GRDMA_RUN_INSTRUMENTED = 2,
  /* (3 was the persistent link engine, k_link: one resident launch per step -- a third of the graph schedule's rate in
   *  rounds 2 - 4, retired in round 5; profiles/r04_bench_engine_*) */
  /* The launches of the default (pipelined, paired) schedule one after the other on one stream with a HIP event
   * between every two: the graph's order is a chain already, so the work and its order are the graph's; the
   * events give the time of every launch by itself.  GRDMA_ERR_INVALID for a job that is not on that schedule. */
  GRDMA_RUN_INSTRUMENTED_SCHEDULE = 4
};

grdma_stream_job* grdma_stream_job_create(grdma_pair* tx, grdma_pair* rx,
                                          const grdma_slice* slices, uint64_t count,
                                          void* rx_dst, uint64_t rx_dst_cap,
                                          uint64_t slices_cap, uint64_t max_rounds);
/* n independent links advancing in lock step: every kernel launch carries one op
 * per link (connections are the data-parallel axis; config 4 =
 * 32 connections per GPU).  slices holds the links' slice lists back to back,
 * counts[i] entries each. */
grdma_stream_job* grdma_stream_job_create_multi(uint32_t n, grdma_pair* const* tx,
                                                grdma_pair* const* rx, const grdma_slice* slices,
                                                const uint64_t* counts, void* const* rx_dsts,
                                                const uint64_t* rx_dst_caps,
                                                const uint64_t* slices_caps, uint64_t max_rounds);
int grdma_stream_job_slices_of(grdma_stream_job* j, uint32_t link, grdma_read_slice* out, uint64_t cap);
void grdma_stream_job_destroy(grdma_stream_job* j);
int grdma_stream_job_run(grdma_stream_job* j, int mode, grdma_stream_result* out);
/* Asynchronous form for timed loops: enqueue one pass (the captured graph) on
 * the link's stream without reading any state back; _sync() waits for it. */
int grdma_stream_job_launch(grdma_stream_job* j);
/* Same work as _launch, issued kernel by kernel on the job's streams (no graph). */
int grdma_stream_job_launch_streams(grdma_stream_job* j);
int grdma_stream_job_sync(grdma_stream_job* j);
int grdma_stream_job_slices(grdma_stream_job* j, grdma_read_slice* out, uint64_t cap);
int grdma_stream_job_set_rounds(grdma_stream_job* j, uint64_t rounds);
/* on != 0: neighbouring rounds share launches, the way two hosts and a NIC work on different rounds at once: the
 * drain plan of round t with the send plan of round t + 1 (k_plan_pair_mw), the scatter of round t with the
 * gather of round t + 1 (k_rx_apply_gather), the wire in between -- three launches per round; two with a direct
 * wire, or with the wire inside the planner pair's launch (grdma_stream_job_set_fused_wire: few links, rings of at
 * most 16 MiB) (DESIGN.md sections 2.5, 2.9).  A drain walks exactly up to the tail its own Send reported; the sender may see a
 * credit one round later than in the sequential schedule.  Same bytes delivered; affects GRDMA_RUN_GRAPH, _EAGER
 * and _INSTRUMENTED_SCHEDULE. */
int grdma_stream_job_set_pipeline(grdma_stream_job* j, int on);
/* `sends` (1..64) consecutive Sends per round in ONE plan -- rdma_flush's loop while the ring has room
 * (rdma_bp_posix.cc:470-524): Send, advance the cursor, Send again -- before the peer drains; paired schedule of a
 * pipelined job only (other schedules keep one Send per round).  Up to two Sends are priced one after the other;
 * more (small max_sge: the reference's default is 30) as one cut of the slice table's index. */
int grdma_stream_job_set_sends(grdma_stream_job* j, uint32_t sends);
/* Paired schedule, staged wire: price the Send of round t + 1 with the credit the drain of round t will post (it waits
 * for that drain's plan inside the launch they share) -- no round of credit lag at a ring every round fills. */
int grdma_stream_job_set_promised_credit(grdma_stream_job* j, int on);
/* Paired schedule, staged wire, a job of few links with rings of at most 16 MiB: the wire of a round rides in the launch
 * of the planner pair (wire workgroups of k_plan_pair_mw; the drain's workgroups wait for them before they look at the
 * ring) instead of a k_copy launch of its own -- two launches per round.  On by default where every workgroup of that
 * launch has a CU at once (GRDMA_JOB_FUSE_WIRE=0 in the environment: off); the getter says how many wire workgroups per
 * link the job's graph carries (0: the wire is a launch of its own). */
int grdma_stream_job_set_fused_wire(grdma_stream_job* j, int on);
uint32_t grdma_stream_job_wire_groups(grdma_stream_job* j);
/* The job's slice tables are rewritten between steps (every grpc_endpoint_write brings a new slice buffer,
 * rdma_bp_posix.cc:559-586): the index its Sends are priced from (k_tx_index: prefix sums over the table) is rebuilt
 * in EVERY step's first round instead of once per job.  bench.py times both. */
int grdma_stream_job_set_rebuild_index(grdma_stream_job* j, int on);

/* ---- diagnostics (profiling aids used by tools/; not needed by an integration) ------------
 * s_memtime stamps / counters the plan kernels leave in their result blocks, the
 * record-size history of a pair, per-command cycle counts of the latency engine. */
int grdma_pair_last_dbg(grdma_pair* p, uint64_t tx_dbg[16], uint64_t rx_dbg[16]);
int grdma_stream_job_debug(grdma_stream_job* j, uint64_t tx_dbg[16], uint64_t rx_dbg[16]);
int grdma_pair_debug_hist(grdma_pair* p, uint32_t* hist_out /* 1024 entries */, uint64_t* count,
                          uint32_t* period);
int grdma_engine_debug(uint64_t out[5]);
uint64_t grdma_express_drains(void);  /* drains served by the single-wave express path so far */
uint64_t grdma_watch_fast_drains(void);      /* diagnostics: drains the watchers' single-wave path took */
int grdma_watch_ticks(uint64_t out[12]);      /* profiling aid: 10 ns ticks of the watchers' drains (arrival found, drain done) */
int grdma_rx_express_ticks(uint64_t out[9]);  /* profiling aid: phase ticks of the express drain (latency engine) */
int grdma_tx_fast_sends(uint64_t out[2]);  /* Sends of streaming jobs planned by k_tx_fast [0], left to the general planner [1] (csrc/grdma_tx_fast.h) */
int grdma_rx_fast_drains(uint64_t out[6]);  /* drains of streaming jobs taken by k_rx_fast [0], declined by reason [1..5] (csrc/grdma_rx_fast.h) */
int grdma_rx_table_cache_stats(uint64_t out[2]);  /* committed drains of the multi-workgroup planner whose read-state tables came out of the connection's table cache [0] / were computed and written back [1] (csrc/grdma_rx_multi.h) */
int grdma_rx_verdict_counts(uint64_t out[2]);  /* drains of the multi-workgroup planner handed to the general planner with a mixed verdict (some workgroups' probes passed, some declined) [0] / with every workgroup declining [1] */
int grdma_debug_set_promise_wait(uint32_t v);  /* test knob: v > 0 makes the promised-credit wait of every other Send workgroup run out after v - 1 polls (0 = the default bound for all) */
uint64_t grdma_wire_wait_runouts(void);  /* fused wire: drain workgroups whose wait for the wire workgroups of their launch ran out (must stay 0) */
int grdma_tx_promise_counts(uint64_t out[4]);  /* promised-credit Sends: priced with it [0], none in the drain [1], an older block [2]; waits that ran out [3] */
int grdma_tx_small_ticks(uint64_t out[8]);  /* profiling aid: phase ticks of the latency engine's small Sends */
/* Scalar ring arithmetic of the host layer (ring_buffer.h:101-143), exported so that the
 * CPU tests can pin it against the oracle without a device. */
uint64_t grdma_host_free_size(uint64_t cap, uint64_t head, uint64_t tail);
uint64_t grdma_host_writable(uint64_t cap, uint64_t head, uint64_t tail);
uint64_t grdma_host_encoded_size(uint64_t payload);
uint64_t grdma_host_calc_writable(uint64_t space);

/* ---- HTTP/2 DATA framing / deframing on the device --------------------------------- */
typedef struct grdma_h2_msg {     /* one gRPC message queued on a stream              */
  const void* payload;            /* serialized message bytes (device-accessible)     */
  uint64_t len;
  uint32_t stream_id;
  uint32_t flags;                 /* 1 = compressed (GRPC_WRITE_INTERNAL_COMPRESS), 2 = END_STREAM */
} grdma_h2_msg;
/* The 5-byte message header (chttp2_transport.cc:1502-1510) + grpc_chttp2_encode_data
 * (frame_data.cc:64-90) for a batch of messages, each sent alone on its stream with
 * open flow-control windows: writes into device memory the slice list
 * grpc_endpoint_write would receive (9-byte frame headers and the 5-byte message
 * header as inlined slices in d_hdr_arena, 32 bytes per slice; payload by
 * reference).  Returns the slice count. */
int64_t grdma_h2_frame_messages(const grdma_h2_msg* msgs, uint64_t n, uint32_t max_frame,
                                grdma_slice* d_slices_out, uint64_t slices_cap,
                                void* d_hdr_arena, uint64_t hdr_cap, uint64_t* wire_bytes);

typedef struct grdma_h2_event {   /* what the deframer saw, in order                  */
  uint32_t kind;                  /* 1 FRAME 2 PAYLOAD 3 MSG_BEGIN 4 MSG_BYTES 5 MSG_END
                                     6 STREAM_OPEN 7 STREAM_CLOSED                       */
  uint32_t a, b, c, d;            /* FRAME: type, flags|status<<8, stream, size         */
                                  /* PAYLOAD: offset in slice, length, is_last          */
                                  /* MSG_BEGIN: compressed, length, stream              */
                                  /* MSG_BYTES: offset in slice, length, stream         */
                                  /* STREAM_OPEN: -, -, stream (accepted from HEADERS)  */
                                  /* STREAM_CLOSED: 1 = left the map / 0 = reads closed, -, stream */
  uint32_t slice;                 /* index of the delivered slice                       */
} grdma_h2_event;
/* connection errors of grpc_chttp2_perform_read (sticky; *h2_error of grdma_h2_deframe) */
enum grdma_h2_error {
  GRDMA_H2_OK = 0,
  GRDMA_H2_ERR_PREFIX = 1,                 /* parsing.cc:91-104  connect string mismatch    */
  GRDMA_H2_ERR_FRAME_TOO_LARGE = 2,        /* parsing.cc:195-205                            */
  GRDMA_H2_ERR_EXPECTED_CONTINUATION = 5,  /* parsing.cc:266-272                            */
  GRDMA_H2_ERR_CONTINUATION_STREAM = 6,    /* parsing.cc:273-281                            */
  GRDMA_H2_ERR_UNEXPECTED_CONTINUATION = 7,/* parsing.cc:287-289                            */
  GRDMA_H2_ERR_FIRST_FRAME = 8,            /* parsing.cc:256-263 first frame must be SETTINGS */
  GRDMA_H2_ERR_MAX_STREAMS = 9,            /* parsing.cc:623-627 (or the stream table is half full) */
  GRDMA_H2_ERR_RST_LENGTH = 10,            /* frame_rst_stream.cc:73-79                     */
  /* malformed control frames (the begin_frame checks of the control-frame parsers): connection errors */
  GRDMA_H2_ERR_SETTINGS_STREAM = 11,       /* parsing.cc:735-739  SETTINGS on a stream      */
  GRDMA_H2_ERR_SETTINGS_ACK_LENGTH = 12,   /* frame_settings.cc:95-101 non-empty ack        */
  GRDMA_H2_ERR_SETTINGS_FLAGS = 13,        /* frame_settings.cc:102-104                     */
  GRDMA_H2_ERR_SETTINGS_LENGTH = 14,       /* frame_settings.cc:105-107 not a multiple of 6 */
  GRDMA_H2_ERR_PING = 15,                  /* frame_ping.cc:58-64 length != 8 or flags      */
  GRDMA_H2_ERR_WINDOW_UPDATE = 16,         /* frame_window_update.cc:56-63                  */
  GRDMA_H2_ERR_GOAWAY = 17,                /* frame_goaway.cc:39-44 shorter than 8 bytes    */
  GRDMA_H2_ERR_TOO_MANY_TRAILERS = 18      /* hpack_parser.cc:1756-1759 third header block without END_HEADERS */
};
enum grdma_h2_parser_flags {
  GRDMA_H2_SERVER = 1,       /* expects the client preface; accepts streams from HEADERS frames
                                (init_header_frame_parser, parsing.cc:596-631)               */
  GRDMA_H2_FIRST_FRAME = 2,  /* fresh connection: the first frame must be SETTINGS           */
  GRDMA_H2_BOUNDARY_STEP = 4,    /* the slice in which a message starts (closing frame of the previous
                                    message + first frame + 5-byte message header, all inside the staged
                                    32 bytes) is parsed in one wave-uniform step instead of byte-wise   */
  GRDMA_H2_NO_BOUNDARY_STEP = 8, /* never; with neither flag the step is on unless the environment
                                    says GRDMA_H2_BOUNDARY_STEP=0                                    */
  GRDMA_H2_BULK_PAIRS = 16,      /* the bulk step gives EVERY lane a frame (lane i: slices s + 2i, s + 2i + 1):
                                    64 frames and 128 slices per step instead of 32 / 64.  On by default since
                                    round 3 (two hardware rounds: 155.7 vs 148.5 and 172.9 vs 155.1 GiB/s in the
                                    with-h2 leg, same events); GRDMA_H2_BULK_PAIRS=0 in the environment or
                                    GRDMA_H2_NO_BULK_PAIRS turns it off                                 */
  GRDMA_H2_NO_BULK_PAIRS = 64,   /* 32 frames per bulk step                                             */
  GRDMA_H2_NO_CHUNKS = 128,      /* always the sequential deframer.  Without it a list of >= 2048 slices is cut at slices in
                                    which a message starts (GRDMA_H2_CHUNKS chunks of >= 128 slices, default 256), the chunks are parsed side
                                    by side and merged when every chunk ended in the state the next one was assumed to
                                    start in -- the sequential deframer does the call otherwise (csrc/grdma_h2_kernels.h) */
  GRDMA_H2_TICKS = 32            /* the parsing wave samples the device clock around its phases (the tick counters
                                    of grdma_h2_pipe_sync / grdma_h2_last_deframe_stats); off by default: a sample
                                    is a scalar memory operation the wave waits for                     */
};
typedef struct grdma_h2_parser grdma_h2_parser;
/* Deframe state of one transport (grpc_chttp2_transport deframe_state & co., internal.h) and
 * its stream map (grpc_chttp2_stream_map) with each stream's grpc_chttp2_data_parser, kept in
 * device memory.  DATA frames are looked up in the map and never create a stream
 * (init_data_frame_parser, parsing.cc:352-374): unknown and read-closed streams are skipped.
 * A server learns its streams from HEADERS frames; a client's streams are opened by the caller
 * when it starts a call.  END_STREAM (on DATA, or on the header block) closes the read side;
 * RST_STREAM closes both; a stream leaves the map once both sides are closed
 * (grpc_chttp2_mark_stream_closed, chttp2_transport.cc:2194-2244) -- the caller reports the write
 * side with grdma_h2_parser_close_writes.  HPACK, SETTINGS, PING, GOAWAY and WINDOW_UPDATE payloads
 * are control plane and are skipped here.
 * grdma_h2_parser_create(prefix, max) = create_ex(prefix ? SERVER | FIRST_FRAME : 0, max, 0xffffffff, 0).
 * table_slots: power of two >= 16 (0 = 4096); at most table_slots / 2 streams are live at once. */
grdma_h2_parser* grdma_h2_parser_create(int expect_client_prefix, uint32_t max_frame_size);
grdma_h2_parser* grdma_h2_parser_create_ex(int flags, uint32_t max_frame_size,
                                           uint32_t max_concurrent_streams, uint32_t table_slots);
void grdma_h2_parser_destroy(grdma_h2_parser* p);
/* Batched stream-map updates from the surface; return the number of ids that failed
 * (duplicate / unknown / table full), or <0. */
int grdma_h2_parser_open_streams(grdma_h2_parser* p, const uint32_t* ids, uint32_t n);
int grdma_h2_parser_close_writes(grdma_h2_parser* p, const uint32_t* ids, uint32_t n);
int64_t grdma_h2_parser_live_streams(grdma_h2_parser* p);
/* Duration of the framing / deframing kernel of the last call (HIP events), microseconds. */
double grdma_h2_last_kernel_us(void);
/* Message starts the last grdma_h2_deframe call parsed with the boundary step. */
uint64_t grdma_h2_last_boundary_steps(void);
/* Counters of the last grdma_h2_deframe call: {bulk steps, frames parsed in bulk steps, boundary steps,
 * then device-clock ticks: waiting for staged windows, in bulk steps, in boundary steps, in the
 * byte-wise path, total}. */
void grdma_h2_last_deframe_stats(uint64_t out[8]);
/* The deframer over chunks (lists of >= 2048 slices, unless GRDMA_H2_NO_CHUNKS): out = {calls planned over chunks,
 * calls whose chunks verified and were merged}; the difference went through the sequential deframer. */
int grdma_h2_parser_chunk_stats(grdma_h2_parser* p, uint64_t out[2]);
/* profiling aid: device-clock phase stamps of the last chunked call, 8 words per chunk + 8 of the merge */
int grdma_h2_parser_chunk_dbg(grdma_h2_parser* p, uint64_t* out, uint64_t cap_words);
/* grpc_chttp2_perform_read (parsing.cc:56-253) + grpc_deframe_unprocessed_incoming_frames
 * (frame_data.cc:92-276) over n delivered slices {offset, length} of d_arena.
 * Returns the number of events; *h2_error = connection error, if any. */
int64_t grdma_h2_deframe(grdma_h2_parser* p, const void* d_arena, const grdma_read_slice* slices,
                         uint64_t n, grdma_h2_event* events_out, uint64_t cap, int* h2_error);

/* HTTP/2 inside the device pipeline: per step, k_h2_frame_index + k_h2_frame_emit rebuild the slice list of the job's
 * link from the message table (grpc_chttp2_encode_data, frame_data.cc:64-90), the streaming job
 * carries it through the connection, and k_h2_deframe parses the slices the job delivered
 * (grpc_chttp2_perform_read + grpc_deframe_unprocessed_incoming_frames) -- three stages on three
 * streams ordered by events, nothing returns to the host in between.  The job must have been run
 * once (graph captured / engine prepared); delivered_slices is what that run delivered.  Two pipes
 * over two jobs of one connection may alternate: framing and deframing then run beside the other
 * job's step.  schedule: 0 = the job's graph (the only one). */
typedef struct grdma_h2_pipe grdma_h2_pipe;
grdma_h2_pipe* grdma_h2_pipe_create(grdma_stream_job* job, uint32_t link, const grdma_h2_msg* msgs, uint64_t nmsgs,
                                    uint32_t max_frame, grdma_h2_parser* parser, uint64_t delivered_slices,
                                    uint64_t events_cap);
int grdma_h2_pipe_enqueue(grdma_h2_pipe* p, int schedule);
/* out = {slices framed, frame overflow, events, deframe overflow, slices parsed, h2 error,
 *        framing kernel us, deframing kernel us (of the last step), bulk steps of the deframer,
 *        frames it parsed in bulk steps, and four profiling tick counts of the deframer: waiting for
 *        staged windows, inside bulk steps, total, inside the byte-wise path} */
int grdma_h2_pipe_sync(grdma_h2_pipe* p, uint64_t out[14], grdma_h2_event* events_out, uint64_t cap);
void grdma_h2_pipe_destroy(grdma_h2_pipe* p);
/* out = {message starts the last synced step parsed with the boundary step, device-clock ticks inside it} */
int grdma_h2_pipe_boundary_stats(grdma_h2_pipe* p, uint64_t out[2]);

/* ---- GRPCProfiler: include/grpcpp/stats_time.h:11-44,111-122, src/core/lib/debug/stats_time.cc ----
 * The reference's scope profiler with its op names in its order: nanoseconds per op per thread slot,
 * opt-in per thread (init(slot) + enable()), the table {Name, Count, Mean, P50, P95, P99, MAX} per slot
 * in the unit GRPC_PROFILING_UNIT names (micro / milli / s).  The host layer records
 * TRANSPORT_{READ,HANDLE_READ,CONTINUE_READ,DO_READ,FLUSH,HANDLE_WRITE,WRITE} in the endpoint mirror
 * and PAIR_SEND / PAIR_RECV around Send / Recv, where rdma_bp_posix.cc and pair.cc place their
 * GRPCProfiler objects.  Host only (no device needed). */
typedef enum grdma_stats_time {
  GRDMA_STATS_TIME_POLLABLE_EPOLL,
  GRDMA_STATS_TIME_POLLSET_WORK,
  GRDMA_STATS_TIME_TRANSPORT_DO_READ,
  GRDMA_STATS_TIME_TRANSPORT_CONTINUE_READ,
  GRDMA_STATS_TIME_TRANSPORT_READ_ALLOCATION_DONE,
  GRDMA_STATS_TIME_TRANSPORT_HANDLE_READ,
  GRDMA_STATS_TIME_TRANSPORT_READ,
  GRDMA_STATS_TIME_TRANSPORT_FLUSH,
  GRDMA_STATS_TIME_TRANSPORT_HANDLE_WRITE,
  GRDMA_STATS_TIME_TRANSPORT_WRITE,
  GRDMA_STATS_TIME_PAIR_SEND,
  GRDMA_STATS_TIME_PAIR_RECV,
  GRDMA_STATS_TIME_CLIENT_PREPARE,
  GRDMA_STATS_TIME_CLIENT_CQ_NEXT,
  GRDMA_STATS_TIME_SERVER_RPC_REQUEST,
  GRDMA_STATS_TIME_SERVER_RPC_FINISH,
  GRDMA_STATS_TIME_SERVER_CQ_NEXT,
  GRDMA_STATS_TIME_BEGIN_WORKER,
  GRDMA_STATS_TIME_ASYNC_NEXT_INTERNAL,
  GRDMA_STATS_TIME_FINALIZE_RESULT,
  GRDMA_STATS_TIME_DESERIALIZE,
  GRDMA_STATS_TIME_ADHOC_1,
  GRDMA_STATS_TIME_ADHOC_2,
  GRDMA_STATS_TIME_ADHOC_3,
  GRDMA_STATS_TIME_ADHOC_4,
  GRDMA_STATS_TIME_ADHOC_5,
  GRDMA_STATS_TIME_ADHOC_6,
  GRDMA_STATS_TIME_ADHOC_7,
  GRDMA_STATS_TIME_ADHOC_8,
  GRDMA_STATS_TIME_ADHOC_9,
  GRDMA_STATS_TIME_ADHOC_10,
  GRDMA_STATS_TIME_MAX_OP_SIZE
} grdma_stats_time;
void grdma_stats_time_init(int slot);          /* grpc_stats_time_init: the calling thread records into `slot` */
void grdma_stats_time_enable(void);
void grdma_stats_time_disable(void);
int grdma_stats_time_enabled(void);            /* enabled AND the calling thread has a slot */
void grdma_stats_time_shutdown(void);
void grdma_stats_time_add(int op, int64_t ns);          /* grpc_stats_time_add */
void grdma_stats_time_add_custom(int op, int64_t val);  /* grpc_stats_time_add_custom: printed unscaled */
const char* grdma_stats_time_op_name(int op);           /* grpc_stats_time_op_to_str */
int64_t grdma_stats_time_now_ns(void);
/* {mean, p50, p95, p99, max} in ns of one op of one slot; returns the count */
uint64_t grdma_stats_time_get(int slot, int op, double out[5]);
/* grpc_stats_time_print into buf (NUL-terminated, truncated to cap); returns the full length */
int64_t grdma_stats_time_print(char* buf, uint64_t cap);

/* ---- device helpers for callers that keep payloads in HBM ---------------------- */
void* grdma_device_alloc(uint64_t bytes);
void grdma_device_free(void* p);
/* Binds every thread of the process (and the threads started later) to the CPUs of the NUMA node the device hangs on,
 * the way `numactl --cpunodebind` would: pinned buffers allocated afterwards and the polling threads sit next to the
 * device's PCIe root.  Returns the node, or -1 when it is unknown (nothing changed then). */
int grdma_host_pin_to_device_node(void);
/* Binds the CALLING thread alone to the k-th physical core of that node, counted from the end of the node's CPU list
 * (`taskset -c`): busy-polling threads of one process then never share a core's two hardware threads.  Returns the CPU,
 * or -1 (nothing changed). */
int grdma_host_pin_thread_to_core(int k);
void* grdma_host_alloc_pinned(uint64_t bytes);
void grdma_host_free_pinned(void* p);
int grdma_copy_to_device(void* dst, const void* src, uint64_t n);
int grdma_copy_to_host(void* dst, const void* src, uint64_t n);
int grdma_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* GRDMA_AMD_H */
