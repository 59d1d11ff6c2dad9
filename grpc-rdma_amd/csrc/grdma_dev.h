// Device-resident data layout of the MI355X ring-buffer endpoint data plane.
//
// Everything a connection needs on the hot path lives in HBM so that the HIP
// kernels own the protocol state (head/tail credit accounting included) and a
// NIC (GPUDirect RDMA) or an xGMI peer can write the ring / status words
// without the host touching them.  Field names follow the reference:
//   RingBufferPollable  src/core/lib/ibverbs/ring_buffer.h:203-208
//   PairPollable        src/core/lib/ibverbs/pair.h:100-103,169-171
//   grpc_rdma           src/core/lib/iomgr/rdma_bp_posix.cc:45-88
#ifndef GRDMA_DEV_H
#define GRDMA_DEV_H

#include <stddef.h>
#include <stdint.h>

#define GRDMA_ALIGN 8ull            // ring_buffer.h:49
#define GRDMA_RESERVED 24ull        // ring_buffer.h:52
#define GRDMA_FOOTER 0xFFFFFFFFFFFFFFFFull  // ring_buffer.h:50

#define GRDMA_MAX_SEGS 16384        // copy segments per plan (<= 2 per record of a 4096-record
                                    // bulk pass, plus wrap splits)
#define GRDMA_MAX_SLICES 8192       // delivered slices per receive plan
#define GRDMA_TX_MAX_RECORDS 4096   // records priced by one send plan
#define GRDMA_HINT_MAX_RECORDS 8192 // records of a round whose sizes its Sends leave for the drain (grdma_size_hint): two Sends' worth
#define GRDMA_RX_HIST 1024          // record sizes remembered per connection
// Behind the history (same allocation): the read-state tables rxm_body derives from a connection's record pattern, kept
// across drains (grdma_rx_multi.h, "table cache").  GRDMA_RX_TAB_SLOTS slots of [hdr 4][key][sss][qpk][qtl][qby][qn],
// each array GRDMA_RX_TAB_STRIDE words (512 pattern positions + the period's total at index 512).
#define GRDMA_RX_TAB_SLOTS 8u
#define GRDMA_RX_TAB_STRIDE 516u
#define GRDMA_RX_TAB_WORDS (4u + 6u * GRDMA_RX_TAB_STRIDE)
#define GRDMA_RX_TAB_MAGIC 0x7AB1E5u
#define GRDMA_RX_HIST_ALLOC_WORDS (GRDMA_RX_HIST + GRDMA_RX_TAB_SLOTS * GRDMA_RX_TAB_WORDS)
#define GRDMA_TILE_BYTES 8192ull    // chunk of the single-wave inline copies (latency paths)
// Bytes one wave copies per tile of a plan: 16 KiB (64 lanes x 16 B x 16 loads in flight) for
// connections with rings of 32 MiB and more -- a 16 KiB payload is then ONE tile and most plans
// hold one tile per segment, which the copy kernels run without any search (measured at the bench
// configuration: k_rx_apply 29.6 -> 23.4 us, gather 18.9 -> 17.9 us) --, 8 KiB below that, where a
// launch moves a few hundred KB to 2 MB and the smaller tile keeps more waves busy.  The planner
// writes its choice into the plan (grdma_plan::tile_bytes); the copy kernels follow it.
#define GRDMA_PLAN_TILE_SHIFT(ring_cap) ((ring_cap) >= (32ull << 20) ? 14u : 13u)
#define GRDMA_MIN_READ_SLICE 256ull // rdma_bp_posix.cc:308

// PairStatus, pair.h:44-51 (also in the public header: what grdma_pair_get_status returns)
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

// status_report, pair.h:100-103: the 16-byte credit message the receiver
// RDMA-writes into the sender's status buffer.
struct grdma_status_report {
  uint64_t remote_head;
  int32_t peer_exit;
  int32_t pad;
};

// Arrival report (this build's addition to the wire: nothing like it exists in the reference, whose NIC
// places the bytes of an RDMA WRITE in order, footer last).  A HIP wire -- the loop-back copy kernel, a peer
// process writing through an IPC mapping, an xGMI peer -- is a PARALLEL copy: a record's header and footer
// can be in place while a tile in the middle is still in flight.  So the sender, once a write has completed
// as a whole, stores the ring offset its writes have reached into the receiver's connection block (the
// reverse direction of the credit report), and the receiver never walks the chain past it.
struct grdma_wire_report {
  uint64_t wire_tail;   // ring offset behind the last record that has landed completely
  uint64_t pad;
};

// Host-visible state line of one connection end: 128 bytes of pinned, coherent host memory that the
// kernels write THROUGH whenever they change what the event engine polls -- HasMessage, HasPendingWrites,
// get_status, GetWritableSize are read-only, lock-free and called from N polling threads in the reference
// (ring_buffer.cc:56-65, pair.cc:294-303,349-375; ev_epollex_rdma_bpev_linux.cc:1103-1145), so here they are
// plain host loads of this line and never a device call.  One writer per word:
//   wire_tail                       the PEER's send commit (in-process peer), or the refresh pass of k_poll
//                                   for a peer in another process (copies conn->wire_recv)
//   rx_head, rx_remain, rx_seq      my own drains
//   remote_head                     the PEER's scatter (credit post), or the refresh pass
//   remote_tail, partial_write, tx_seq   my own sends
//   peer_exit                       the peer's Disconnect (host store), or the refresh pass
struct grdma_hostline {
  uint64_t wire_tail;
  uint64_t rx_head;
  uint64_t rx_remain;
  uint64_t rx_seq;        // drains completed (scatter and credit included)
  uint64_t remote_head;
  uint64_t remote_tail;
  uint64_t partial_write;
  uint64_t tx_seq;        // sends completed (payload has left the caller's slices, wire write done)
  int32_t peer_exit;
  uint32_t pad0;
  uint64_t refresh_seq;   // refresh passes of k_poll that have written this line
  uint64_t pad1[6];
};

// {ptr,len} view of one grpc_slice (GRPC_SLICE_START_PTR / GRPC_SLICE_LENGTH,
// include/grpc/impl/codegen/slice.h:96-101).  ptr must be device-accessible.
struct grdma_sge {
  const uint8_t* ptr;
  uint64_t len;
};

// One connection end.  Lives in HBM; 256-byte aligned so that concurrently
// polled words of different connections never share a cache line pair.
struct grdma_conn {
  // ---- receive side: my ring (RingBufferPollable) -------------------------
  uint8_t* ring;
  uint64_t cap;                 // power of two
  uint64_t head;                // head_
  uint64_t moving_head;         // moving_head_ (what is reported as credit)
  uint64_t remain;              // remain_
  uint64_t internal_read_size;  // pair.h:169
  uint64_t leftover_cap;        // unfilled tail kept in last_read_buffer
  uint64_t credit_msgs;         // status reports sent so far
  uint64_t total_read;          // total_read_size_
  // ---- send side ------------------------------------------------------------
  uint8_t* staging;             // send_buffers_[kDataBuffer], ring/2 bytes
  uint64_t staging_cap;
  uint64_t remote_tail;         // remote_tail_
  uint64_t partial_write;       // partial_write_
  uint64_t total_written;       // total_write_size_
  uint64_t tx_slice_idx;        // rdma_flush cursor: first unsent slice ...
  uint64_t tx_byte_idx;         // ... and outgoing_byte_idx within it
  // ---- wire: where my one-sided writes land ---------------------------------
  uint8_t* peer_ring;                       // remote ring base (same cap)
  struct grdma_status_report* peer_status;  // peer's status_recv
  uint32_t max_sge;             // max_sge_num_
  uint32_t status;              // grdma_pair_status
  uint32_t wire_direct;         // 1: encode straight into peer_ring (no staging)
  uint32_t tx_last_records;     // records the previous Send produced (sizes the next pricing window)
  // ---- credit the peer granted me (written remotely) --------------------------
  struct grdma_status_report status_recv;   // recv_buffers_[kStatusBuffer]
  struct grdma_status_report status_send;   // send_buffers_[kStatusBuffer]
  // ---- streaming-job cursors (append mode of the drain) -----------------------
  uint64_t rx_arena_off;        // next free byte of the caller's destination buffer
  uint64_t rx_slice_idx;        // next free entry of the caller's slice table
  uint64_t tx_rounds;           // Send() calls that produced at least one record
  uint64_t rx_rounds;           // drains that delivered at least one slice
  uint64_t tx_records;          // ring records produced
  uint64_t rx_records;          // ring records consumed
  uint32_t rx_h1;               // encoded sizes of the last / last-but-one record read (0 = none yet):
  uint32_t rx_h2;               // the chain walker's size prediction without a trip to the history ring
  uint64_t tx_remaining;        // bytes of the current slice list not yet accepted
  uint32_t* rx_hist;            // encoded sizes of the last GRDMA_RX_HIST records read
  uint64_t rx_hist_count;       // records read so far (ring index = count % HIST)
  uint32_t rx_period;           // detected period of the record sizes (0 = none)
  uint32_t pad3;
  uint64_t rx_period_retry_at;  // no new period search before this many records were read
  // ---- arrival report and host-visible state (see grdma_wire_report / grdma_hostline) -----------
  struct grdma_wire_report wire_recv;   // written by the peer: how far its completed writes reach in my ring
  uint64_t* peer_wire;                  // the peer's wire_recv.wire_tail (device memory, possibly an IPC mapping)
  struct grdma_hostline* line;          // my state line (pinned host memory), NULL = none
  struct grdma_hostline* peer_line;     // the peer's line when it lives in this process, else NULL
  uint32_t wire_limit;                  // 1: never walk past wire_recv.wire_tail (every HIP wire); 0: the wire
                                        // places bytes in order, footer last (a NIC)
  uint32_t line_remote;                 // 1: nobody pushes wire_tail / remote_head / peer_exit into my line
                                        // (the peer is another process): k_poll's refresh pass copies them
  uint32_t peer_limited;                // 1: the peer's reader is known never to walk past my arrival report (its
                                        // wire_limit is set: a peer in this process) -- the footer of a record I write
                                        // needs no place of its own behind the record's bytes; 0: footers last
  uint32_t pad4;
};

struct grdma_seg {
  uint64_t dst;
  uint64_t src;    // 0 = fill dst with zero bytes
  uint64_t len;
  uint64_t flags;  // GRDMA_SEG_ZERO_SRC: clear the source bytes after copying them
};
#define GRDMA_SEG_ZERO_SRC 1ull
// Record tags ride on the segments instead of being stored one by one by the plan
// workgroup (thousands of scattered 8-byte stores from a single CU were the longest
// phase of both plans): the wave that copies the first / last tile of a record also
// handles its header / its padding + footer.
//   GRDMA_SEG_TAG_HDR   this segment starts a record
//   GRDMA_SEG_TAG_FTR   this segment ends a record
//   GRDMA_SEG_TAG_WRITE sender: header = flags >> 8 (payload bytes) in front of dst, zero
//                       padding and the 0xFF.. footer behind dst + len (AppendHeader /
//                       AppendFooter, ring_buffer.h:84-99); without it, receiver: clear the
//                       same places around src (ring_buffer.cc:146,173-180)
// Tag addresses wrap inside the plan's [tag_base, tag_base + tag_mask] window (the ring);
// a linear staging buffer uses tag_mask = ~0.
#define GRDMA_SEG_TAG_HDR 2ull
#define GRDMA_SEG_TAG_FTR 4ull
#define GRDMA_SEG_TAG_WRITE 8ull
#define GRDMA_SEG_TAG_LEN_SHIFT 8

struct grdma_slice_out {  // one completed endpoint_read: a single slice
  uint64_t off;           // offset into the receive arena
  uint64_t len;
};

// Result block of one send plan (mirrors what PairPollable::Send and
// rdma_flush report back, pair.cc:645-734 / rdma_bp_posix.cc:470-524).
struct grdma_tx_result {
  uint64_t sent;           // payload bytes consumed from the slice list
  uint64_t records;        // ring records produced (== SGEs)
  uint64_t staged;         // encoded bytes (Σ 16 + round_up8(pay))
  uint64_t partial;        // partial_write_
  uint64_t new_remote_tail;
  uint64_t wr_off[2];      // the ≤2 RDMA WRITE work requests:
  uint64_t wr_len[2];      //   remote ring offset / byte count
  uint64_t wr_count;
  uint64_t slice_idx;      // rdma_flush cursor after this send
  uint64_t byte_idx;
  uint64_t done;           // 1 when the whole slice list has been sent
  uint64_t seq;            // bumped last (host polls it)
  uint64_t dbg[16];        // s_memtime stamps of the plan phases (profiling aid)
};

struct grdma_rx_result {
  uint64_t nslices;        // endpoint_read completions emulated
  uint64_t bytes;          // payload bytes delivered
  uint64_t consumed;       // ring bytes consumed (records incl. tags)
  uint64_t records;        // ring records fully or partly consumed
  uint64_t would_block;    // 1: the last read found nothing (re-arm notify_on_read)
  uint64_t credit_sent;    // status reports posted during this call
  uint64_t credit_head;    // remote_head value of the last report
  uint64_t head, moving_head, remain;
  uint64_t arena_used;     // bytes of the arena handed out
  uint64_t zero_off[2];    // ring ranges cleared
  uint64_t zero_len[2];
  uint64_t seq;            // bumped by k_rx_plan
  uint64_t commit_seq;     // bumped by k_rx_commit (copy + zero-fill + credit done)
  uint32_t pad1;           // drains of this block taken by the steady-state body (grdma_rx_fast.h) ...
  uint32_t pad0;           // ... and declined by it with data waiting (what grdma_stream_job_run adapts to)
  uint64_t dbg[16];        // s_memtime stamps of the plan phases (profiling aid)
};

// A list of byte-copy segments plus its decomposition into wave tiles.
struct grdma_plan {
  uint32_t nsegs;
  uint32_t ntiles;
  uint64_t bytes;
  uint64_t tag_base;   // window for the record tags of GRDMA_SEG_TAG_* segments
  uint64_t tag_mask;
  uint32_t tile_bytes; // 8192 or 16384: what tile_prefix counts in (GRDMA_PLAN_TILE_SHIFT)
  uint32_t ready;      // (unused since round 6: the hand-over word of the promised credit is `promise` below)
  struct grdma_seg segs[GRDMA_MAX_SEGS];
  uint32_t tile_prefix[GRDMA_MAX_SEGS + 1];
  // k_rx_apply arrival counter (the last workgroup commits).  It lives in the plan -- always
  // device memory, one per drain in flight -- and not in the result block, which is pinned HOST
  // memory for a stand-alone pair: a thousand workgroups counting in over PCIe took a millisecond.
  // On a line of its own: every workgroup reads the plan header when it starts, and a counter in
  // that line made the early finishers' atomics fight the late starters' loads (2.4x the kernel time).
  uint32_t pad_line0[31];
  uint32_t blocks_done;
  // count-in word of the multi-workgroup planners (grdma_rx_multi.h, grdma_tx_multi.h): workgroups counted in in the low
  // half, workgroups that declined in the high half (drain plan); zero between launches (the committing workgroup --
  // the one dispatched last -- clears it).  On a WIRE plan: wire workgroups of the planner pair's launch that are done.
  uint32_t mw_arrive;
  uint32_t promise_done;  // participants of the promised-credit hand-over that are through with `promise` (promise_leave)
  uint32_t pad_p;
  // promised credit (k_plan_pair_mw): the hand-over word between the drain's committing workgroup and
  // the Send workgroups of the SAME launch, touched by atomics only -- 0 nothing yet; bits 0-1 = 1: KEPT, and the word
  // carries the promise itself (bit 2: the drain posts a credit, bits 3..: its head -- a ring offset, < 2^31); = 2: GIVEN
  // UP (csrc/grdma_devfn.h: promise_keep / promise_wait); zero between launches.  Other kernels neither read nor write it.
  uint64_t promise;
  // workgroups of the multi-workgroup receive planner, but the committing one, whose entries are at the memory side
  // (they count out here when they leave; the committer waits for them before it clears the words: drain_close)
  uint32_t mw_done;
  uint32_t pad_line1[25];
};
static_assert(offsetof(grdma_plan, promise) % 8 == 0, "the hand-over word is a 64-bit atomic");

// A kernel node another stage hangs into a streaming job's graph (grdma_job_set_hooks, csrc/grdma_pair.hip): the
// kernel, its launch shape and its parameters (each at most 8 bytes, one slot per parameter).
#define GRDMA_JOB_HOOK_ARGS 10
struct grdma_job_hook {
  const void* fn;
  uint32_t grid, threads;
  uint64_t args[GRDMA_JOB_HOOK_ARGS];
};

#endif  // GRDMA_DEV_H
