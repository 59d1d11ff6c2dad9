// Control block of the persistent link engine (k_link): one kernel launch pushes a whole slice
// list through a connected loop-back link.  The roles of a connection -- the sender's Send loop,
// its gather waves, the wire, the receiver's ring walk, its scatter waves -- are separate agents
// that run CONCURRENTLY and talk only through memory, the way two hosts and a NIC do:
//
//   TX leader (1 wave)   PairPollable::Send + rdma_flush cursor (pair.cc:645-734,
//                        rdma_bp_posix.cc:470-524): prices the records of one Send per step and
//                        publishes copy entries (gather: slices -> staging records; wire:
//                        staging -> peer ring, the <= 2 work requests of GetWriteRequests)
//   gather / wire /      worker waves: each owns every W-th entry of its stage's table, moves
//   scatter workers      <= 16 KiB per entry with 8 x 16-byte write-through accesses in flight per
//                        lane, then counts the entry done
//   RX leader (1 wave)   GetReadableSize / Read / Recv / rdma_continue_read / rdma_do_read
//                        (ring_buffer.cc:67-191, pair.cc:264-286, rdma_bp_posix.cc:180-326): walks
//                        the record chain of every Send that has fully landed, replays the endpoint
//                        reads, publishes scatter entries, and posts the 16-byte credit report once
//                        the bytes it frees have been copied out and cleared
//
// Nothing is ordered by kernel boundaries here, so every hand-off follows the write-through
// recipe of the CDNA4 guide: payload and table stores are sc1 (write-through) stores, every
// storing wave drains (s_waitcnt vmcnt(0)) before it publishes a counter with a relaxed
// agent-scope atomic, consumers poll relaxed and read with sc1 loads.  All polled words are
// zeroed by the host before every launch; every spin is bounded (abort word + wall-clock timeout).
#ifndef GRDMA_LINK_H
#define GRDMA_LINK_H

#include <stdint.h>

#include "grdma_dev.h"

#define LK_ENTRY_MAX 16384u   // bytes one table entry moves: one wave tile
#define LK_TILE 16384u        // bytes one wave moves per step: 64 lanes x 16 B x 16 in flight
#define LK_TABLE_CAP 32768u   // entries per table (a ring; power of two)
#define LK_SLOTS 16u          // Sends in flight (completion counters per stage)
#define LK_RSLOTS 64u         // receive chunks in flight
#define LK_MAX_STAGING 8u     // staging buffers a sender rotates through
#define LK_THREADS 256
#define LK_REPL 8u            // copies of a word that hundreds of waves poll (one per memory channel group)

enum { LK_GATHER = 0, LK_WIRE = 1, LK_SCATTER = 2 };

// One unit of copy work: at most LK_ENTRY_MAX bytes src -> dst.  flags: GRDMA_SEG_* in the low
// byte (zero the source behind the copy; this entry starts / ends a record: write or clear its
// header / padding + footer), completion-counter slot in bits 8..15.
struct lk_entry {
  uint64_t dst, src;
  uint32_t len, flags;
  uint64_t aux;  // gather: payload length of the record (the header word); wire: gather entries of
                 // the same Send that must be complete before staging may be read
};

// What a published Send looks like to the wire and to the receiver.
struct lk_send_desc {
  uint64_t seq;       // Send number + 1, written last
  uint64_t staged;    // encoded bytes (sum of 16 + round_up8(pay)) = what lands in the ring
  uint32_t n_gather;  // entries in the gather / wire tables
  uint32_t n_wire;
  uint64_t records;
};

struct lk_ctl {
  // ---- configuration, written by the host at job creation -------------------------------
  struct grdma_conn* tx;
  struct grdma_conn* rx;
  const struct grdma_sge* slices;
  uint64_t nslices;
  uint64_t total_bytes;            // sum of the slice lengths
  uint8_t* arena;                  // where delivered slices go
  uint64_t arena_cap;
  struct grdma_slice_out* out_slices;
  uint64_t slices_cap;
  uint8_t* staging[LK_MAX_STAGING];
  uint32_t n_staging;
  uint32_t direct;                 // GRDMA_WIRE_DIRECT: records are built in the peer ring itself
  struct lk_entry* tab[3];
  uint32_t nwaves[3];              // worker waves per stage
  uint32_t timeout_ms;
  uint32_t pad_cfg32[2];
  uint64_t pad_cfg[5];
  // ---- dynamic state, zeroed before every launch --------------------------------------------
  // (each polled word on its own 128-byte line: a poller must not share a line with a word
  //  somebody else is storing to)
  // entries published per stage.  Every idle worker wave polls this word, and a polled word is
  // re-fetched from the memory side each time (sc1): the copies sit 4 KiB + 128 B apart so the polls
  // of a stage spread over several memory channels instead of queueing the copy traffic of one
  struct { uint64_t v; uint64_t pad[527]; } published[3][LK_REPL];
  struct { uint64_t v; uint64_t pad[15]; } closed[3];      // no further entry will be published
  struct { uint64_t v; uint64_t pad[15]; } sends_pub;      // Sends published
  struct { uint64_t v; uint64_t pad[15]; } tx_done;        // the sender has nothing more to send
  struct { uint64_t v; uint64_t pad[15]; } rx_sends_seen;  // Sends whose descriptor the receiver consumed
  struct { uint64_t v; uint64_t pad[15]; } rx_rounds_done; // Sends drained, zero-filled, credits posted
  struct { uint64_t v; uint64_t pad[15]; } abort;          // != 0: error code, everybody leaves
  struct { uint32_t v; uint32_t pad[31]; } done_tx[2][LK_SLOTS];  // entries completed per Send (gather, wire)
  struct { uint32_t v; uint32_t pad[31]; } done_rx[LK_RSLOTS];    // entries completed per receive chunk
  struct lk_send_desc sends[LK_SLOTS];
  // ---- results ---------------------------------------------------------------------------------
  uint64_t res_sends, res_chunks, res_entries[3];
  uint64_t res_wait_ticks[4];      // profiling aid: leader wait time (tx: slots, credit; rx: data, table)
  uint64_t res_prof[8];            // profiling aid: leader busy time (tx: pricing, wire+publish, total; rx: walk, fast steps, scalar steps, total)
  uint64_t res_tx_phases[4];       // profiling aid: sender step phases (slice loads, pricing, counting, emission), ticks
  uint64_t res_dbg[8];             // what the first aborting role saw: {code, site, a, b, c, d}
  // profiling aid: event trace of the two leaders {tag << 56 | wall-clock tick}, tags:
  // 1 Send priced, 2 Send published, 3 ring writes released (arg = Sends released so far),
  // 4 receiver starts waiting for round r, 5 round r has landed, 6 round r walked and planned,
  // 7 credit report posted, 8 round r complete (scatter done)
  // rows 2..4: worker wave 0 of the gather / wire / scatter stage: 9 entry taken (arg = entry),
  // 10 its dependency is met, 11 entry counted done
  uint32_t trace_n[6];
  uint64_t trace[5][192];
};

#define LK_DYNAMIC_OFFSET offsetof(struct lk_ctl, published)

// abort codes
enum {
  LK_ERR_TIMEOUT = 1,
  LK_ERR_NO_PROGRESS = 2,   // a zero-length slice at the cursor: the reference would never get past it either
  LK_ERR_ARENA = 3,         // destination buffer too small
  LK_ERR_SLICES = 4,        // slice table too small
  LK_ERR_CORRUPT = 5        // the ring does not hold the records the Send descriptors promise
};

#endif  // GRDMA_LINK_H
