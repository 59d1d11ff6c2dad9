// Operation descriptors consumed by the plan kernels (device-visible memory).
#ifndef GRDMA_OPS_H
#define GRDMA_OPS_H

#include <stdint.h>

#include "grdma_dev.h"

// The payload sizes of the records one Send produced, handed by a streaming job from the Send of a round to the drain
// of the same round (grdma_tx_op::sizes_out -> grdma_rx_op::sizes_in), the way the Send's tail is (tail_out ->
// limit_ptr).  The drain uses it as a PREDICTION and checks every header and footer in the ring against it before it
// consumes anything (csrc/grdma_rx_hint.h); a table that does not start at the reader's head is not used.
struct grdma_size_hint {
  uint64_t start_off;   // ring offset of the first record (the sender's remote_tail_ in front of the Send)
  uint32_t count;       // records; 0 = no table for this Send
  uint32_t pad;
  uint32_t n[GRDMA_HINT_MAX_RECORDS];
};

// One PairPollable::Send / rdma_flush step for one connection.
struct grdma_tx_op {
  struct grdma_conn* conn;
  const struct grdma_sge* slices;  // the grpc_slice_buffer being written
  uint64_t nslices;
  uint64_t byte_idx;               // outgoing_byte_idx when use_cursor == 0
  struct grdma_plan* plan;         // gather plan (slices -> records)
  struct grdma_plan* wire_plan;    // loop-back wire plan (staging -> peer ring)
  struct grdma_tx_result* result;
  uint32_t use_cursor;             // 0: slice 0 + byte_idx; 1: continue from the conn's
                                   // rdma_flush cursor; 2: reset that cursor first; 3: like 2, but only if the
                                   // grdma_tx_result byte_idx points at says done == 1 (a write queued behind another:
                                   // k_tx_plan_seq skips the whole burst otherwise, results done = 2)
  uint32_t inline_copy;            // 1: this workgroup also runs the gather (and wire) tiles --
                                   // one launch per Send for small messages
  uint8_t* staging_alt;            // != NULL: stage this Send here instead of conn->staging
  uint64_t seq_next;               // != 0: the value to publish in result->seq (latency mode: the
                                   // result block is host memory, reading the old value back
                                   // would be a PCIe round trip in front of the release)
  uint64_t* tail_out;              // != NULL: remote_tail_ after this Send is also stored here (a streaming
                                   // job hands it to the drain of the same round: grdma_rx_op::limit_ptr)
  struct grdma_size_hint* sizes_out;  // != NULL: the planners of grdma_tx_multi.h leave the record sizes of this Send
                                   // here (count 0 when the Send was planned by another planner)
};

// Index of the slice buffer a streaming job writes (csrc/grdma_tx_fast.hip): prefix sums over ALL its slices, built
// by k_tx_index at the start of a write; k_tx_fast prices every Send of the write from it.
struct grdma_txf_ctl {
  const struct grdma_sge* slices;  // the buffer the index was built for ...
  uint64_t n;                      // ... and its length
  uint64_t* enc_pre;               // [n + 1] sum of 16 + round_up8(len_j), j < k
  uint64_t* len_pre;               // [n + 1] sum of len_j
  uint32_t* tile_pre;              // [n + 1] sum of ceil(len_j / tile)
  uint32_t tile_shift;             // GRDMA_PLAN_TILE_SHIFT of the connection
  uint32_t valid;                  // k_tx_index: 1 = usable (no empty slice, lengths below 2 GiB)
  uint32_t sends;                  // > 1: a plan of the planners of grdma_tx_multi.h holds this many consecutive Sends
                                   // (grdma_stream_job_set_sends; 0 / 1: one Send per plan)
  uint32_t pad;
};

// One PairPollable::SendZerocopy (pair.cc:793-941) for one connection.
struct grdma_zc_op {
  struct grdma_conn* conn;
  const struct grdma_sge* slices;
  uint64_t nslices;
  uint64_t byte_idx;
  struct grdma_plan* plan;         // gather plan: slices / zero-copy buffer -> records in the peer ring
  struct grdma_tx_result* result;  // dbg[0..4) = zero-copy payload bytes, staged (copied) payload bytes,
                                   // scatter-gather entries after the wrap split, zero-copy records
  const uint8_t* zc_base;          // send_buffers_[kZeroCopyBuffer]
  uint64_t zc_cap;
};

// One drain of a connection's ring: a run of endpoint_read completions.
struct grdma_rx_op {
  struct grdma_conn* conn;
  struct grdma_plan* plan;         // scatter plan (ring -> arena)
  struct grdma_rx_result* result;
  struct grdma_slice_out* slices;  // one entry per completed endpoint_read
  uint8_t* arena;                  // receive arena (HBM or pinned host)
  uint64_t arena_cap;
  uint64_t max_reads;              // stop after this many completions
  uint64_t raw_cap;                // != 0: one PairPollable::Recv(arena, raw_cap) instead
  uint64_t append;                 // 1: continue at conn->rx_arena_off / rx_slice_idx
  uint64_t slices_cap;             // entries in `slices` (append mode)
  uint64_t inline_apply;           // 1: this workgroup also scatters, zero-fills and posts credit
  uint64_t seq_next;               // != 0: the value to publish in result->seq / commit_seq
  const uint64_t* limit_ptr;       // != NULL: the ring offset this drain may walk up to, read when the drain
                                   // starts (a streaming job hands over the tail its Send of the same round
                                   // computed: the graph edge behind that Send's wire write says those bytes
                                   // have landed, and a later round may already be landing behind them);
                                   // NULL: conn->wire_recv.wire_tail when conn->wire_limit is set
  const struct grdma_size_hint* sizes_in;  // != NULL: the sizes the Send of this round computed (see grdma_size_hint)
};

// Mailbox of the persistent latency engine (pinned host memory).
enum { GRDMA_ENGINE_SEND = 1, GRDMA_ENGINE_DRAIN = 2, GRDMA_ENGINE_SEND_INLINE = 3, GRDMA_ENGINE_DRAIN_BLOCK = 4,
       // (5 was round 4's send + chained drain of the in-process peer, with its cut-through of unary-sized records:
       //  a path only two ends inside one engine command could take -- retired for the watcher workgroups below)
       // arm / let go of a watch slot (grdma_watch_cmd): the standing read order of a connection, carried out by a
       // resident watcher workgroup (k_watch) the moment the sender's arrival report -- or, on an ordered wire, a
       // complete record -- shows up in the connection's own ring
       GRDMA_ENGINE_WATCH = 6 };

// Self-contained command block for small messages: the op, its slice table and the
// payload bytes travel in ONE contiguous pinned block that the engine pulls into LDS
// with a single wide read, instead of chasing op -> slices -> payload over PCIe.
#define GRDMA_CMD_MAX_SGES 16
#define GRDMA_CMD_INLINE_BYTES 1024
struct grdma_engine_cmd {
  struct grdma_tx_op tx;
  struct grdma_rx_op rx;
  struct grdma_sge sges[GRDMA_CMD_MAX_SGES];  // ptr = offset into inline_data
  uint8_t inline_data[GRDMA_CMD_INLINE_BYTES];
};
// Fast lane: a small command travels INSIDE the mailbox.  Eight 64-byte lines, each seven
// payload words + the command's sequence number in its last word.  The host fills a line's
// payload before it stamps the line (stores to one cache line become visible in program order),
// and a PCIe read returns a line as a unit: a line that shows the expected stamp holds that
// command's words.  Wave 0 of the engine reads all eight lines with ONE load per poll (lane l =
// word l), so a 64-byte RPC costs one PCIe round trip from doorbell to payload-in-LDS instead of
// three dependent ones (doorbell -> type/op -> command block).
// Payload: [type | nsges << 8 | data bytes << 16] [the op struct] [nsges x {offset, len}] [data].
#define GRDMA_FAST_LINES 8
#define GRDMA_FAST_WORDS (GRDMA_FAST_LINES * 7)  // payload words
// ---- arrival-triggered reads (k_watch) -------------------------------------------------------------------------
// What the reference's busy-polling thread is to an outstanding grpc_endpoint_read (HasMessage() on the connection's
// OWN ring, ring_buffer.cc:56-65; ev_epollex_rdma_bpev_linux.cc:1105-1149, poller.cc:84), a resident watcher
// workgroup is here: it polls the arrival report of every connection it has been handed (one load per connection
// and pass, 64 connections per wave) and, when bytes have landed, runs the standing order -- one drain of up to
// max_reads endpoint reads into the pinned arena -- and bumps the result block's sequence word in pinned host
// memory.  grdma_endpoint_read is then a load from host memory.  It does not matter who wrote the ring: the
// command workgroup of this engine, another process through an IPC mapping, a NIC.
// One completion is outstanding per connection: the next drain waits until the host has taken the last one
// (grdma_engine_mbox::consumed, mirrored into the slot by the command workgroup's doorbell poll).
#define GRDMA_WATCH_SLOTS 64
#define GRDMA_WATCH_MAX_GROUPS 8
#define GRDMA_WATCH_WINDOWS 8
// The word the host publishes per slot (grdma_engine_mbox::consumed, mirrored into grdma_watch_slot::consumed):
// completions taken since the arming in the low 56 bits, and in the high byte the WINDOW the next drain delivers into
// (grdma_watch_slot::win_base: the receive windows of an asynchronous endpoint, or the two halves of a blocking pair's
// arena -- the slices of the completion just taken stay where they are while the next drain runs).
#define GRDMA_WATCH_COUNT_MASK 0x00FFFFFFFFFFFFFFull
struct grdma_watch_slot {         // device memory; one writer per word
  uint64_t gen;                   // command workgroup: != 0 armed (a fresh value per arming), 0 = let go
  uint64_t ack_gen;               // watcher: the generation it has taken over (0: it has let go, nothing in flight)
  uint64_t consumed;              // command workgroup: the host's word (above)
  uint64_t done;                  // watcher: completions produced since the arming
  struct grdma_rx_op op;          // the standing order; op.seq_next = sequence word of the first completion,
                                  // op.arena_cap = bytes of one window (op.arena is not used)
  uint8_t* win_base[GRDMA_WATCH_WINDOWS];
  uint64_t drains_dbg;            // watcher: drains run for this slot (all armings)
  uint64_t pad[32 - 5 - GRDMA_WATCH_WINDOWS - sizeof(struct grdma_rx_op) / 8];
};
struct grdma_engine_mbox;
struct grdma_watch_ctl {          // device memory
  uint64_t quit;                  // command workgroup: the engine incarnation that has been told to leave
  // Where the HOST's words of the mailbox live (cmd_seq / cmd_type / op, exit_flag, the fast lane, consumed[]): NULL = in
  // the mailbox itself (pinned host memory: every poll of the doorbell is a PCIe read, 1.3-2 us there and back);
  // otherwise a second grdma_engine_mbox in fine-grained DEVICE memory that the host writes through the PCIe BAR
  // (write-combined stores + sfence: a posted write, ~0.6 us one way) and the doorbell wave polls locally.  What the
  // engine writes (ack_seq, alive, the profiling totals, watch_alive) stays in pinned host memory, where the host
  // polls it with plain loads.  (round 6: the unary round trip's two mailbox hops)
  struct grdma_engine_mbox* inbox;
  uint64_t pad[30];
  struct grdma_watch_slot slot[GRDMA_WATCH_SLOTS];
};
struct grdma_watch_cmd {          // pinned host memory: GRDMA_ENGINE_WATCH
  uint64_t slot;
  uint64_t gen;                   // 0 = let go
  uint64_t consumed0;             // the host's word at the arming: count 0, the first window
  struct grdma_rx_op op;
  uint8_t* win_base[GRDMA_WATCH_WINDOWS];
};

struct grdma_engine_mbox {
  uint64_t cmd_seq;    // host: bumped last, after cmd_type/op are written
  uint64_t cmd_type;
  const void* op;      // grdma_tx_op* / grdma_rx_op* (device-visible)
  uint64_t pad0[5];
  uint64_t ack_seq;    // engine: last command completed
  uint64_t alive;      // engine: 1 while resident
  uint64_t exit_flag;  // host: ask the engine to leave
  uint64_t pad1[5];
  uint64_t fast[GRDMA_FAST_LINES * 8];
  uint64_t consumed[GRDMA_WATCH_SLOTS];  // host: completions taken per watch slot (read by the doorbell poll, lane = slot)
  uint64_t watch_alive[GRDMA_WATCH_MAX_GROUPS];  // watcher workgroup w: the engine incarnation it belongs to
};

#endif  // GRDMA_OPS_H
