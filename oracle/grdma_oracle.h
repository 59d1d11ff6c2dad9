/* TEST INFRASTRUCTURE ONLY -- oracle/ is the CPU checker for the HIP data plane.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (grpc-rdma_amd/) never links, imports or
 * falls back to it.
 *
 * A from-scratch plain-C restatement of the reference algorithms on the
 * RDMA_BP/BPEV endpoint hot path.  Every function
 * cites the reference file:line it follows (paths relative to the
 * pwrliang/grpc-rdma tree).
 *
 * PARITY PINNING: the ring/pair half is pinned against the reference's own
 * ring_buffer.cc compiled into oracle/_ref/libref_ring.so
 * (tests/test_oracle_vs_ref.py) and against tests/golden/ring_*.json generated
 * from that build.  The reference tree ships NO unit tests or golden vectors
 * for the ring codec (test/core/ibverbs/ is absent), so that is the strongest
 * pin available.  The HTTP/2 half is pinned against the byte vectors in the
 * reference's test/core/bad_client/tests/{simple_request,head_of_line_blocking}.cc
 * and the constructive framing of test/cpp/microbenchmarks/bm_chttp2_transport.cc:504-560.
 */
#ifndef GRDMA_ORACLE_H
#define GRDMA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ ring -- */
#define ORC_ALIGN 8u            /* ring_buffer.h:49  alignment            */
#define ORC_RESERVED 24u        /* ring_buffer.h:52  reserved_space       */
#define ORC_FOOTER UINT64_MAX   /* ring_buffer.h:50  footer               */

typedef struct orc_ring {
  uint8_t* buf;
  uint64_t cap;
  uint64_t mask;
  uint64_t head;        /* head_        ring_buffer.h:203 */
  uint64_t moving_head; /* moving_head_ ring_buffer.h:204 */
  uint64_t remain;      /* remain_      ring_buffer.h:205 */
} orc_ring;

uint64_t orc_round_up8(uint64_t v);                 /* ring_buffer.h:232-237 */
uint64_t orc_round_down8(uint64_t v);               /* ring_buffer.h:240-245 */
uint64_t orc_encoded_size(uint64_t payload);        /* ring_buffer.h:180-183 */
uint64_t orc_calc_writable(uint64_t space);         /* ring_buffer.h:185-189 */

int orc_ring_init(orc_ring* r, uint8_t* buf, uint64_t cap); /* .cc:20-25,49-54 */
int orc_ring_has_message(const orc_ring* r);                /* .cc:56-65  */
uint64_t orc_ring_readable(const orc_ring* r);              /* .cc:67-97  */
uint64_t orc_ring_free_size(const orc_ring* r, uint64_t head, uint64_t tail); /* .cc:99-104 */
uint64_t orc_ring_writable(const orc_ring* r, uint64_t head, uint64_t tail);  /* .cc:106-116 */
uint64_t orc_ring_read(orc_ring* r, void* dst, uint64_t cap, uint64_t* internal); /* .cc:122-191 */
uint64_t orc_ring_write(orc_ring* r, uint64_t tail, const void* src, uint64_t n); /* .cc:193-226 */

/* ------------------------------------------------------------------ pair -- */
typedef struct orc_slice {
  const uint8_t* ptr;
  uint64_t len;
} orc_slice;

typedef struct orc_status_report { /* pair.h:100-103 */
  uint64_t remote_head;
  int32_t peer_exit;
  int32_t pad;
} orc_status_report;

typedef struct orc_pair {
  orc_ring ring;            /* my receive ring                      */
  uint8_t* staging;         /* kDataBuffer send buffer, ring/2 B    */
  uint64_t staging_cap;
  uint64_t staging_used;    /* bytes staged by the last send        */
  orc_status_report status_recv; /* written by the peer             */
  orc_status_report status_send;
  uint64_t remote_tail;     /* pair.h:170 remote_tail_              */
  uint64_t internal_read_size; /* pair.h:169                        */
  int partial_write;        /* pair.h:171                           */
  int max_sge;              /* pair.h:87 max_sge_num_               */
  uint64_t credit_msgs;     /* number of status reports sent        */
  struct orc_pair* peer;
  /* last send's wire work requests {remote offset, length} */
  uint64_t wr[2][2];
  int wr_count;
  /* endpoint-level state (rdma_bp_posix.cc:45-88) */
  uint64_t leftover_cap;    /* bytes of last_read_buffer retained   */
  /* zero-copy send buffer (pair.h:96,178,195; pair.cc:103,113,120) */
  uint8_t* zc_buf;          /* send_buffers_[kZeroCopyBuffer]       */
  uint64_t zc_cap;
  uint64_t zc_tail;         /* zerocopy_buffer_tail_                */
  uint64_t zc_bytes;        /* zerocopy_bytes_                      */
  uint64_t copy_bytes;      /* copy_bytes_                          */
  uint64_t sge_count;       /* scatter-gather entries of the last SendZerocopy, after the wrap split */
} orc_pair;

/* Plan of one PairPollable::Send (pair.cc:671-707): payload bytes taken from
 * each slice.  Returns the number of records; *sent = total payload bytes. */
uint64_t orc_plan_send(uint64_t ring_cap, uint64_t staging_cap, uint64_t remote_head,
                       uint64_t remote_tail, int max_sge, const uint64_t* lens,
                       uint64_t n, uint64_t byte_idx, uint64_t* pays,
                       uint64_t* sent);

int orc_pair_init(orc_pair* p, uint64_t ring_cap, int max_sge); /* pair.cc:85-141 */
void orc_pair_destroy(orc_pair* p);
void orc_pair_connect(orc_pair* a, orc_pair* b);
uint64_t orc_pair_send(orc_pair* p, const orc_slice* slices, uint64_t n,
                       uint64_t byte_idx);                 /* pair.cc:645-734 */
uint64_t orc_pair_recv(orc_pair* p, void* dst, uint64_t cap); /* pair.cc:264-286 */
/* Zero-copy send buffer.  enable: initSendBuffer(kZeroCopyBuffer, size), pair.cc:103,113 (the
 * reference sizes it from GRPC_RDMA_ZEROCOPY_BUFFER_SIZE_KB ... config.cc:100-106).
 * allocate: PairPollable::AllocateSendBuffer, pair.cc:305-323 -- succeeds only while the buffer is
 * empty (tail == 0) and the size fits; NULL otherwise.
 * send_zerocopy: PairPollable::SendZerocopy, pair.cc:793-941 -- a slice that lies inside the
 * zero-copy buffer goes out as header / payload / padding / footer scatter-gather entries (only the
 * 16 + padding tag bytes are staged; the record is limited by the receiver's credit, not by the
 * staging buffer; it needs 4 free entries and 24 staging bytes), any other slice is encoded into the
 * staging buffer as Send does; the wire image is the concatenation of the entries at remote_tail,
 * wrapping once (GetWriteRequests(sg_list), ring_buffer.cc:261-330). */
int orc_pair_enable_zerocopy(orc_pair* p, uint64_t zc_cap);
uint8_t* orc_pair_allocate_send_buffer(orc_pair* p, uint64_t size);
uint64_t orc_pair_send_zerocopy(orc_pair* p, const orc_slice* slices, uint64_t n,
                                uint64_t byte_idx);
uint64_t orc_pair_writable(const orc_pair* p);              /* pair.cc:294-301 */

/* One grpc_endpoint_read completion as rdma_bp_posix.cc performs it
 * (:343-376 rdma_read, :306-326 rdma_continue_read, :180-291 rdma_do_read).
 * dst must hold max(256, readable, leftover) bytes.  Returns bytes delivered in
 * the single slice handed to the read callback; 0 = nothing ready (the
 * endpoint would re-arm notify_on_read).  *alloc = size of the slice that was
 * used as the read target. */
uint64_t orc_endpoint_read(orc_pair* p, uint8_t* dst, uint64_t* alloc);

/* ------------------------------------------------------------ HTTP/2 TX -- */
#define ORC_H2_FRAME_DATA 0          /* frame.h:31 */
#define ORC_H2_FLAG_END_STREAM 1     /* frame.h:39 */
#define ORC_SLICE_INLINED_SIZE 23    /* slice.h:47-48 on LP64 */

/* chttp2_transport.cc:1502-1510 */
void orc_grpc_msg_header(uint8_t out[5], int compressed, uint32_t len);
/* frame_data.cc:73-82 */
void orc_h2_data_header(uint8_t out[9], uint32_t len, int end_stream, uint32_t stream_id);

/* The slice list chttp2 hands to grpc_endpoint_write for ONE gRPC message of
 * msg_len bytes sent alone on stream_id: 5-byte message header via
 * grpc_slice_buffer_tiny_add (chttp2_transport.cc:1502), then per DATA frame
 * grpc_chttp2_encode_data (frame_data.cc:64-90) with the inlined-slice merge
 * rule of grpc_slice_buffer_add (slice_buffer.cc:136-171) and the split rule
 * of grpc_slice_buffer_move_first_no_ref (slice_buffer.cc:283-313).
 * wire receives the concatenated bytes (cap wire_cap); lens[] the slice
 * lengths (cap lens_cap).  Returns the slice count, or -1 on overflow. */
int64_t orc_h2_frame_message(const uint8_t* msg, uint64_t msg_len, int compressed,
                             uint32_t stream_id, uint32_t max_frame, int end_stream,
                             uint8_t* wire, uint64_t wire_cap, uint64_t* wire_len,
                             uint64_t* lens, uint64_t lens_cap);

/* Same for n messages queued back to back on one outbuf (flags: 1 compressed,
 * 2 END_STREAM). */
int64_t orc_h2_frame_batch(const uint8_t* const* msgs, const uint64_t* msg_lens,
                           const uint32_t* stream_ids, const uint32_t* flags, uint64_t n,
                           uint32_t max_frame, uint8_t* wire, uint64_t wire_cap,
                           uint64_t* wire_len, uint64_t* lens, uint64_t lens_cap);

/* ------------------------------------------------------------ HTTP/2 RX -- */
enum {
  ORC_EV_FRAME = 1,     /* a = type, b = flags | status << 8, c = stream id, d = frame size */
  ORC_EV_PAYLOAD = 2,   /* a = offset in the fed chunk, b = length, c = is_last    */
  ORC_EV_MSG_BEGIN = 3, /* a = compressed flag, b = message length, c = stream id  */
  ORC_EV_MSG_BYTES = 4, /* a = offset in the fed chunk, b = length, c = stream id  */
  ORC_EV_MSG_END = 5,   /* c = stream id                                            */
  ORC_EV_STREAM_OPEN = 6,   /* c = stream id: accepted from a HEADERS frame (server) */
  ORC_EV_STREAM_CLOSED = 7  /* a = 1 if the stream left the map (read and write side closed),
                               0 if only the read side closed; c = stream id */
};
enum {
  ORC_H2_OK = 0,
  ORC_H2_ERR_PREFIX = 1,        /* parsing.cc:91-104 connect string mismatch */
  ORC_H2_ERR_FRAME_TOO_LARGE = 2, /* parsing.cc:195-205 */
  ORC_H2_ERR_DATA_FLAGS = 3,    /* frame_data.cc:47-52 (stream error, reported in EV_FRAME) */
  ORC_H2_ERR_GRPC_FRAME_TYPE = 4, /* frame_data.cc:123-140 (stream error) */
  ORC_H2_ERR_EXPECTED_CONTINUATION = 5, /* parsing.cc:266-272 */
  ORC_H2_ERR_CONTINUATION_STREAM = 6,   /* parsing.cc:273-281 */
  ORC_H2_ERR_UNEXPECTED_CONTINUATION = 7, /* parsing.cc:287-289 */
  ORC_H2_ERR_FIRST_FRAME = 8,   /* parsing.cc:256-263 */
  ORC_H2_ERR_MAX_STREAMS = 9,   /* parsing.cc:623-627 */
  ORC_H2_ERR_RST_LENGTH = 10,   /* frame_rst_stream.cc:73-79 */
  /* malformed control frames: the begin_frame checks of the control-frame parsers, all connection errors */
  ORC_H2_ERR_SETTINGS_STREAM = 11,      /* parsing.cc:735-739 "Settings frame received for grpc_chttp2_stream" */
  ORC_H2_ERR_SETTINGS_ACK_LENGTH = 12,  /* frame_settings.cc:95-101 "non-empty settings ack frame received" */
  ORC_H2_ERR_SETTINGS_FLAGS = 13,       /* frame_settings.cc:102-104 "invalid flags on settings frame" */
  ORC_H2_ERR_SETTINGS_LENGTH = 14,      /* frame_settings.cc:105-107 "settings frames must be a multiple of six bytes" */
  ORC_H2_ERR_PING = 15,                 /* frame_ping.cc:58-64 "invalid ping: length, flags" */
  ORC_H2_ERR_WINDOW_UPDATE = 16,        /* frame_window_update.cc:56-63 "invalid window update: length, flags" */
  ORC_H2_ERR_GOAWAY = 17,               /* frame_goaway.cc:39-44 "goaway frame too short" */
  ORC_H2_ERR_TOO_MANY_TRAILERS = 18,    /* hpack_parser.cc:1756-1759: a third header block on one stream, met by the
                                           skipping header parser with is_boundary set (parsing.cc:667-669, 318-327) */
  ORC_H2_ERR_EVENT_OVERFLOW = 100
};
enum { ORC_H2_SERVER = 1, ORC_H2_FIRST_FRAME = 2 };

typedef struct orc_h2_event {
  uint32_t kind;
  uint32_t a, b, c, d;
} orc_h2_event;

/* One entry of the transport's stream map (grpc_chttp2_stream_map, internal.h) with the
 * fields the deframe path reads: the per-stream grpc_chttp2_data_parser (frame_data.h),
 * read_closed / write_closed (chttp2_transport.cc:2194-2244), header_frames_received. */
typedef struct orc_h2_stream {
  uint32_t stream_id;
  int state;            /* data parser: 0..4 = FH_0..FH_4, 5 = FRAME, 6 = ERROR */
  uint32_t frame_size;  /* remaining bytes of the current message  */
  int compressed;
  int read_closed, write_closed, header_frames_received;
} orc_h2_stream;

typedef struct orc_h2_parser { /* internal.h grpc_chttp2_transport deframe fields */
  int state;            /* 0..23 prefix, 24..32 FH_0..FH_8, 33 FRAME */
  uint32_t incoming_frame_size;
  uint8_t incoming_frame_type;
  uint8_t incoming_frame_flags;
  uint32_t incoming_stream_id;
  uint32_t max_frame_size;
  int check_frame_size;
  int cur_parser;       /* which payload parser the frame in flight uses: 0 skip 1 data 2 header 3 rst 4 third header block */
  int is_server, is_first_frame;
  uint32_t expect_continuation_stream_id;
  int header_eof, header_boundary, received_last_frame;
  uint32_t last_new_stream_id, max_concurrent_streams;
  orc_h2_stream* streams; /* the stream map: unordered, grows on demand */
  uint64_t nstreams, streams_cap;
} orc_h2_parser;

void orc_h2_parser_init(orc_h2_parser* p, int expect_client_prefix, uint32_t max_frame_size);
void orc_h2_parser_init_ex(orc_h2_parser* p, int flags, uint32_t max_frame_size,
                           uint32_t max_concurrent_streams);
void orc_h2_parser_free(orc_h2_parser* p);
/* What the surface does outside the read path: a client starts a call on stream `id`
 * (grpc_chttp2_stream_map_add, chttp2_transport.cc maybe_start_some_streams); the write side
 * of a stream closes (grpc_chttp2_mark_stream_closed(close_writes)); a stream already read-closed
 * then leaves the map.  Return 0, or -1 for an unknown / duplicate id. */
int orc_h2_parser_open_stream(orc_h2_parser* p, uint32_t id);
int orc_h2_parser_close_writes(orc_h2_parser* p, uint32_t id);
uint64_t orc_h2_parser_live_streams(const orc_h2_parser* p);
/* Feed one slice (parsing.cc:56-253 grpc_chttp2_perform_read).  DATA frame
 * payload is additionally run through the per-stream gRPC message deframer
 * (frame_data.cc:92-276).  Events are appended to ev[*nev..cap). */
int orc_h2_parser_feed(orc_h2_parser* p, const uint8_t* data, uint64_t len,
                       orc_h2_event* ev, uint64_t cap, uint64_t* nev);

/* --------------------------------------------------------- cpu baseline -- */
/* Streams n_msgs messages (each already framed into slices) a->b through the
 * full pair protocol, draining with orc_endpoint_read, single thread.
 * Returns payload bytes delivered; *seconds = CLOCK_MONOTONIC wall time. */
uint64_t orc_stream_baseline(uint64_t ring_cap, int max_sge, const uint8_t* wire,
                             const uint64_t* lens, uint64_t nslices, uint64_t n_msgs,
                             double* seconds, uint64_t* checksum);

/* The sequential schedule (one Send from the cursor, endpoint reads until one would block),
 * `passes` times over one link; see the definition for what is reported. */
int orc_stream_rounds(uint64_t ring_cap, int max_sge, const uint8_t* wire, const uint64_t* lens,
                      uint64_t nslices, int passes, uint64_t* out_lens, uint64_t out_cap,
                      uint64_t* n_out, uint64_t* first_rounds, uint64_t state[9], int* stream_ok,
                      int* ring_zero);
int orc_stream_rounds_burst(uint64_t ring_cap, int max_sge, int burst, const uint8_t* wire, const uint64_t* lens,
                            uint64_t nslices, int passes, uint64_t* out_lens, uint64_t out_cap,
                            uint64_t* n_out, uint64_t* first_rounds, uint64_t state[9], int* stream_ok,
                            int* ring_zero);

#ifdef __cplusplus
}
#endif
#endif /* GRDMA_ORACLE_H */
