// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// ref_h2_perform_read_trace: the reference's OWN grpc_chttp2_perform_read (K8) -- parsing.cc compiled unmodified:
// the client-preface / nine-byte frame-header state machine (parsing.cc:56-253), init_frame_parser and the per-type
// init_*_parser functions with their stream-map rules (:255-564, :566-750), parse_frame_slice (:752-776) -- together
// with the reference's own stream map (stream_map.cc), DATA parser and message deframer (frame_data.cc), RST_STREAM
// parser (frame_rst_stream.cc) and slice layer (slice.cc, slice_buffer.cc), all unmodified.  What the oracle's
// orc_h2_parser_feed is pinned to (tests/test_oracle_vs_ref.py).
//
// What is NOT the reference's code here, and says so: the rest of the transport, which parsing.cc calls into and
// which cannot be built in this image (chttp2_transport.cc, hpack_parser.cc, flow_control.cc and the SETTINGS / PING /
// GOAWAY / WINDOW_UPDATE payload parsers need the whole of gRPC core).  The stand-ins below keep what the frame
// parser's control flow depends on and nothing else:
//   * grpc_chttp2_mark_stream_closed     chttp2_transport.cc:2194-2244: read_closed / write_closed, the stream leaves the
//                                        map when both are set (remove_stream -> grpc_chttp2_stream_map_delete)
//   * grpc_chttp2_parsing_accept_stream  :767-797: a new stream object enters the map (the surface's accept callback)
//   * grpc_chttp2_maybe_complete_recv_message  :1871-1960 with a receive always pending: frame_storage is swapped into
//                                        unprocessed_incoming_frames_buffer and the reference's deframer is called until it
//                                        is empty; the byte stream it feeds is the stub of ref_h2_deframe_trace.cc
//   * grpc_chttp2_header_parser_parse    hpack_parser.cc:1746-1790 WITHOUT the HPACK decoding: at the last piece of a
//                                        header block, header_frames_received++ on a boundary and reads closed on eof
//   * SETTINGS / PING / GOAWAY / WINDOW_UPDATE begin_frame + parse: accept and skip (the test sends well-formed ones)
//   * flow control: the reference's own TransportFlowControlDisabled / StreamFlowControlDisabled (flow_control.h, inline);
//     with flow control disabled parsing.cc:195-205 does not check the frame size, so that error stays pinned by vectors
//   * errors: an object that keeps the description and whether GRPC_ERROR_INT_STREAM_ID was set (what parsing.cc asks)
//
// stdin (binary): u32 is_client, u32 is_first_frame, u32 max_concurrent_streams, u32 next_stream_id, then operations
//     'o' u32 id            the application opened stream id (a client's call: the stream is in the map)
//     'w' u32 id            the write side of stream id closed (mark_stream_closed(close_writes))
//     'f' u32 len, bytes    one slice handed to grpc_chttp2_perform_read
// stdout: O id | C id gone | B id flags len | Y id n | E id | G id (bad gRPC message flag byte, once) | X description (connection error: the trace ends) and,
// behind every slice, P deframe_state incoming_frame_size; at the end  S rst_streams_queued live_streams.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <grpc/slice.h>
#include <grpc/slice_buffer.h>
#include <grpc/support/alloc.h>
#include <grpc/support/log.h>

#include "src/core/ext/transport/chttp2/transport/frame_data.h"
#include "src/core/ext/transport/chttp2/transport/internal.h"
#include "src/core/lib/gprpp/fork.h"
#include "src/core/lib/iomgr/exec_ctx.h"
#include "src/core/lib/slice/slice_internal.h"
#include "src/core/lib/transport/static_metadata.h"

extern "C" void gpr_log(const char* file, int line, gpr_log_severity sev, const char* fmt, ...) {
  if (sev != GPR_LOG_SEVERITY_ERROR) return;  // (GPR_ASSERT reports through here before it aborts)
  va_list ap;
  va_start(ap, fmt);
  fprintf(stderr, "%s:%d: ", file, line);
  vfprintf(stderr, fmt, ap);
  fputc('\n', stderr);
  va_end(ap);
}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }
extern "C" void* gpr_malloc(size_t n) { return malloc(n ? n : 1); }
extern "C" void* gpr_zalloc(size_t n) { return calloc(n ? n : 1, 1); }
extern "C" void* gpr_realloc(void* p, size_t n) { return realloc(p, n ? n : 1); }
extern "C" void gpr_free(void* p) { free(p); }

// ---- errors: description + "has a stream id" -------------------------------------------------------------------
struct trace_error {
  std::string desc;
  bool has_stream_id = false;
};
static trace_error* as_te(grpc_error_handle e) { return reinterpret_cast<trace_error*>(e); }
grpc_error_handle grpc_error_create(const char*, int, const grpc_slice& desc, grpc_error_handle*, size_t) {
  trace_error* e = new trace_error();
  e->desc.assign(reinterpret_cast<const char*>(GRPC_SLICE_START_PTR(desc)), GRPC_SLICE_LENGTH(desc));
  return reinterpret_cast<grpc_error_handle>(e);
}
grpc_error_handle grpc_error_do_ref(grpc_error_handle e) { return e; }
void grpc_error_do_unref(grpc_error_handle) {}
grpc_error_handle grpc_error_set_int(grpc_error_handle e, grpc_error_ints which, intptr_t) {
  if (which == GRPC_ERROR_INT_STREAM_ID) as_te(e)->has_stream_id = true;
  return e;
}
bool grpc_error_get_int(grpc_error_handle e, grpc_error_ints which, intptr_t* out) {
  if (which == GRPC_ERROR_INT_STREAM_ID && as_te(e)->has_stream_id) {
    *out = 0;
    return true;
  }
  return false;
}
grpc_error_handle grpc_error_set_str(grpc_error_handle e, grpc_error_strs, const grpc_slice&) { return e; }
std::string grpc_error_std_string(grpc_error_handle e) { return as_te(e)->desc; }
char* grpc_dump_slice(const grpc_slice&, uint32_t) { return strdup(""); }
namespace grpc_core {
TraceFlag::TraceFlag(bool, const char* name) : name_(name), value_(false) {}
const StaticMetadataSlice* g_static_metadata_slice_table = nullptr;  // (only the header callbacks look at it: never called)
void ExecCtx::Run(const DebugLocation&, grpc_closure*, grpc_error_handle) { abort(); }  // (on_next: no byte stream is ever pending)
}  // namespace grpc_core
grpc_core::TraceFlag grpc_http_trace(false, "http");

// ---- the parts of the transport the frame parser calls into ------------------------------------------------------
static std::vector<grpc_chttp2_stream*> g_all_streams;

static grpc_chttp2_stream* new_stream(grpc_chttp2_transport* t, uint32_t id) {
  grpc_chttp2_stream* s = static_cast<grpc_chttp2_stream*>(calloc(1, sizeof(grpc_chttp2_stream)));
  *const_cast<grpc_chttp2_transport**>(&s->t) = t;
  *const_cast<uint32_t*>(&s->id) = id;
  s->flow_control.Init<grpc_core::chttp2::StreamFlowControlDisabled>();
  grpc_slice_buffer_init(&s->frame_storage);
  grpc_slice_buffer_init(&s->unprocessed_incoming_frames_buffer);
  new (&s->data_parser) grpc_chttp2_data_parser();
  grpc_chttp2_stream_map_add(&t->stream_map, id, s);
  g_all_streams.push_back(s);
  return s;
}

grpc_chttp2_stream* grpc_chttp2_parsing_accept_stream(grpc_chttp2_transport* t, uint32_t id) {
  printf("O %u\n", id);
  return new_stream(t, id);
}

void grpc_chttp2_mark_stream_closed(grpc_chttp2_transport* t, grpc_chttp2_stream* s, int close_reads, int close_writes,
                                    grpc_error_handle) {
  if (s->read_closed && s->write_closed) return;  // (already closed, chttp2_transport.cc:2197-2202)
  if (close_reads && !s->read_closed) s->read_closed = true;
  if (close_writes && !s->write_closed) s->write_closed = true;
  const bool gone = s->read_closed && s->write_closed;
  if (gone) grpc_chttp2_stream_map_delete(&t->stream_map, s->id);  // (remove_stream)
  printf("C %u %d\n", s->id, gone ? 1 : 0);
}

void grpc_chttp2_maybe_complete_recv_message(grpc_chttp2_transport*, grpc_chttp2_stream* s) {
  while (s->unprocessed_incoming_frames_buffer.length > 0 || s->frame_storage.length > 0) {
    if (s->unprocessed_incoming_frames_buffer.length == 0) {
      grpc_slice_buffer_swap(&s->unprocessed_incoming_frames_buffer, &s->frame_storage);
    }
    grpc_core::OrphanablePtr<grpc_core::ByteStream> stream_out;
    // (the transport passes no slice_out and leaves the payload to the byte stream's Pull, chttp2_transport.cc:3058-3111,
    //  which calls the deframer with one; here the message is pulled at once, as in ref_h2_deframe_trace.cc)
    grpc_slice out = grpc_empty_slice();
    grpc_error_handle e = grpc_deframe_unprocessed_incoming_frames(&s->data_parser, s,
                                                                  &s->unprocessed_incoming_frames_buffer, &out, &stream_out);
    if (stream_out != nullptr) stream_out.release();  // (the surface's reference)
    if (e != GRPC_ERROR_NONE) {
      if (!s->seen_error) printf("G %u\n", s->id);  // (a message flag byte > 1; the parser stays in its error state)
      s->seen_error = true;
      grpc_slice_buffer_reset_and_unref_internal(&s->frame_storage);
      grpc_slice_buffer_reset_and_unref_internal(&s->unprocessed_incoming_frames_buffer);
      break;
    }
  }
}

grpc_error_handle grpc_chttp2_header_parser_parse(void* hpack_parser, grpc_chttp2_transport* t, grpc_chttp2_stream* s,
                                                  const grpc_slice&, int is_last) {
  grpc_chttp2_hpack_parser* parser = static_cast<grpc_chttp2_hpack_parser*>(hpack_parser);
  if (is_last) {
    if (s != nullptr) {
      if (parser->is_boundary) {
        if (s->header_frames_received == GPR_ARRAY_SIZE(s->metadata_buffer)) {
          return GRPC_ERROR_CREATE_FROM_STATIC_STRING("Too many trailer frames");
        }
        s->header_frames_received++;
      }
      if (parser->is_eof) grpc_chttp2_mark_stream_closed(t, s, true, false, GRPC_ERROR_NONE);
    }
    parser->is_boundary = 0xde;
    parser->is_eof = 0xde;
  }
  return GRPC_ERROR_NONE;
}
void grpc_chttp2_hpack_parser_set_has_priority(grpc_chttp2_hpack_parser*) {}
void grpc_chttp2_hptbl_set_max_bytes(grpc_chttp2_hptbl*, uint32_t) {}

#define SKIPPING_PARSER(name)                                                                                       \
  grpc_error_handle grpc_chttp2_##name##_parser_parse(void*, grpc_chttp2_transport*, grpc_chttp2_stream*,          \
                                                       const grpc_slice&, int) {                                    \
    return GRPC_ERROR_NONE;                                                                                         \
  }
SKIPPING_PARSER(ping)
SKIPPING_PARSER(goaway)
SKIPPING_PARSER(settings)
SKIPPING_PARSER(window_update)
// (the begin_frame functions of these four parsers -- the length / flag / stream-id checks of a control frame's header --
//  are the REFERENCE's: frame_ping.cc, frame_goaway.cc, frame_settings.cc, frame_window_update.cc are compiled
//  unmodified into this binary; the *_parser_parse stand-ins above take precedence over theirs at link time
//  (oracle/Makefile: this file first, -Wl,--allow-multiple-definition; their own parse functions, which need the
//  rest of the transport, are discarded unreferenced))
void grpc_chttp2_act_on_flowctl_action(const grpc_core::chttp2::FlowControlAction&, grpc_chttp2_transport*,
                                       grpc_chttp2_stream*) {}
void schedule_bdp_ping_locked(grpc_chttp2_transport*) {}
void grpc_chttp2_cancel_stream(grpc_chttp2_transport*, grpc_chttp2_stream*, grpc_error_handle) {}
// (header callbacks of parsing.cc -- on_initial_header / on_trailing_header -- are never called: no HPACK decoding here)
bool grpc_http2_decode_timeout(const grpc_slice&, grpc_millis*) { return false; }
void* grpc_mdelem_get_user_data(grpc_mdelem, void (*)(void*)) { return nullptr; }
void* grpc_mdelem_set_user_data(grpc_mdelem, void (*)(void*), void* p) { return p; }
void grpc_mdelem_on_final_unref(grpc_mdelem_data_storage, void*, uint32_t) {}
grpc_error_handle grpc_chttp2_incoming_metadata_buffer_add(grpc_chttp2_incoming_metadata_buffer*, grpc_mdelem) {
  return GRPC_ERROR_NONE;
}
void grpc_chttp2_incoming_metadata_buffer_set_deadline(grpc_chttp2_incoming_metadata_buffer*, grpc_millis) {}

namespace grpc_core {
GPR_TLS_CLASS_DEF(ExecCtx::exec_ctx_);
Atomic<bool> Fork::support_enabled_(false);
void Fork::DoIncExecCtxCount() {}
void Fork::DoDecExecCtxCount() {}
bool ExecCtx::Flush() { return false; }
grpc_millis ExecCtx::Now() { return 0; }
namespace channelz {
void SocketNode::RecordMessageReceived() {}
void SocketNode::RecordStreamStartedFromRemote() {}
}  // namespace channelz
namespace chttp2 {
TransportFlowControlDisabled::TransportFlowControlDisabled(grpc_chttp2_transport*) {}  // (flow_control.cc:151-171 sets windows nobody reads here)
}  // namespace chttp2

// the byte stream the deframer feeds (as in ref_h2_deframe_trace.cc: the reference's remaining-bytes arithmetic)
Chttp2IncomingByteStream::Chttp2IncomingByteStream(grpc_chttp2_transport* transport, grpc_chttp2_stream* stream,
                                                   uint32_t frame_size, uint32_t flags)
    : ByteStream(frame_size, flags), transport_(transport), stream_(stream), refs_(2), remaining_bytes_(frame_size) {
  printf("B %u %u %u\n", stream->id, flags, frame_size);
}
void Chttp2IncomingByteStream::Orphan() { Unref(); }
bool Chttp2IncomingByteStream::Next(size_t, grpc_closure*) { abort(); }
grpc_error_handle Chttp2IncomingByteStream::Pull(grpc_slice*) { abort(); }
void Chttp2IncomingByteStream::Shutdown(grpc_error_handle) { abort(); }
grpc_error_handle Chttp2IncomingByteStream::Push(const grpc_slice& slice, grpc_slice* slice_out) {
  if (remaining_bytes_ < GRPC_SLICE_LENGTH(slice)) {
    printf("T %u\n", stream_->id);
    grpc_slice_unref_internal(slice);
    return GRPC_ERROR_CREATE_FROM_STATIC_STRING("Too many bytes in stream");
  }
  remaining_bytes_ -= static_cast<uint32_t> GRPC_SLICE_LENGTH(slice);
  printf("Y %u %zu\n", stream_->id, (size_t)GRPC_SLICE_LENGTH(slice));
  if (slice_out != nullptr) *slice_out = slice;
  return GRPC_ERROR_NONE;
}
grpc_error_handle Chttp2IncomingByteStream::Finished(grpc_error_handle error, bool) {
  if (error == GRPC_ERROR_NONE && remaining_bytes_ != 0) {
    printf("U %u\n", stream_->id);
    error = GRPC_ERROR_CREATE_FROM_STATIC_STRING("Truncated message");
  } else if (error == GRPC_ERROR_NONE) {
    printf("E %u\n", stream_->id);
  }
  stream_->pending_byte_stream = false;  // (the surface is done with the message: reset_byte_stream / Orphan, chttp2_transport.cc:2939, 2974)
  Unref();
  return error;
}
}  // namespace grpc_core

// parsing.cc:195-205 checks a frame's size against the acknowledged MAX_FRAME_SIZE only when flow control is enabled.
// The reference's enabled class (flow_control.cc) cannot be built here; the check needs nothing of it but the answer
// to flow_control_enabled(): everything else behaves like TransportFlowControlDisabled.
namespace {
class FlowControlSaysEnabled final : public grpc_core::chttp2::TransportFlowControlBase {
 public:
  bool flow_control_enabled() const override { return true; }
  uint32_t MaybeSendUpdate(bool) override { return 0; }
  grpc_core::chttp2::FlowControlAction MakeAction() override { return grpc_core::chttp2::FlowControlAction(); }
  grpc_core::chttp2::FlowControlAction PeriodicUpdate() override { return grpc_core::chttp2::FlowControlAction(); }
  void StreamSentData(int64_t) override {}
  grpc_error_handle RecvData(int64_t) override { return GRPC_ERROR_NONE; }
  void RecvUpdate(uint32_t) override {}
};
}  // namespace

static bool rd32(uint32_t* v) { return fread(v, 4, 1, stdin) == 1; }

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 1 << 16);
  uint32_t is_client = 0, first_frame = 0, max_streams = 0, next_stream_id = 0, max_frame = 0;
  if (!rd32(&is_client) || !rd32(&first_frame) || !rd32(&max_streams) || !rd32(&next_stream_id) || !rd32(&max_frame)) return 3;
  grpc_chttp2_transport* t = static_cast<grpc_chttp2_transport*>(calloc(1, sizeof(grpc_chttp2_transport)));
  t->is_client = is_client != 0;
  t->is_first_frame = first_frame != 0;
  t->deframe_state = is_client ? GRPC_DTS_FH_0 : GRPC_DTS_CLIENT_PREFIX_0;  // (chttp2_transport.cc: the constructor)
  t->next_stream_id = next_stream_id;
  // (what this side announced: a SETTINGS ack copies SENT over ACKED, parsing.cc:734-741)
  t->settings[GRPC_LOCAL_SETTINGS][GRPC_CHTTP2_SETTINGS_MAX_CONCURRENT_STREAMS] = max_streams;
  t->settings[GRPC_SENT_SETTINGS][GRPC_CHTTP2_SETTINGS_MAX_CONCURRENT_STREAMS] = max_streams;
  t->settings[GRPC_ACKED_SETTINGS][GRPC_CHTTP2_SETTINGS_MAX_CONCURRENT_STREAMS] = max_streams;
  t->flow_control.Init<grpc_core::chttp2::TransportFlowControlDisabled>(t);
  if (max_frame != 0) {  // the frame-size check on: what this side announced and the peer acknowledged
    static_assert(sizeof(FlowControlSaysEnabled) <= sizeof(grpc_core::chttp2::TransportFlowControlDisabled), "fits the member's storage");
    t->flow_control->~TransportFlowControlBase();
    new (t->flow_control.get()) FlowControlSaysEnabled();
    for (int which : {GRPC_LOCAL_SETTINGS, GRPC_SENT_SETTINGS, GRPC_ACKED_SETTINGS})
      t->settings[which][GRPC_CHTTP2_SETTINGS_MAX_FRAME_SIZE] = max_frame;
  }
  grpc_chttp2_stream_map_init(&t->stream_map, 8);
  grpc_slice_buffer_init(&t->qbuf);
  bool dead = false;
  int kind;
  while ((kind = getchar()) != EOF) {
    uint32_t a = 0;
    if (!rd32(&a)) return 3;
    if (kind == 'o') {
      if (!dead) new_stream(t, a);
    } else if (kind == 'w') {
      grpc_chttp2_stream* s = static_cast<grpc_chttp2_stream*>(grpc_chttp2_stream_map_find(&t->stream_map, a));
      if (!dead && s != nullptr) grpc_chttp2_mark_stream_closed(t, s, false, true, GRPC_ERROR_NONE);
    } else if (kind == 'f') {
      grpc_slice sl = a ? grpc_slice_malloc_large(a) : grpc_empty_slice();
      if (a && fread(GRPC_SLICE_START_PTR(sl), 1, a, stdin) != a) return 3;
      if (dead) continue;
      grpc_error_handle e = grpc_chttp2_perform_read(t, sl);
      // The application pulls: while a message is being delivered (pending_byte_stream, frame_data.cc:201) the DATA
      // parser only stores the payload (frame_data.cc:283-298) until the surface asks for more -- here it asks after
      // every slice.  WHEN it asks is the application's business, so the order of message bytes against stream
      // events is not part of the comparison (the test compares the B / Y / E sequence per stream).
      for (grpc_chttp2_stream* s : g_all_streams) grpc_chttp2_maybe_complete_recv_message(t, s);
      if (e != GRPC_ERROR_NONE) {
        std::string d = as_te(e)->desc;
        for (char& ch : d)
          if (static_cast<unsigned char>(ch) < 32) ch = '?';  // (the preface error quotes the offending byte)
        printf("X %s\n", d.c_str());
        dead = true;  // (read_action_locked: the transport closes, chttp2_transport.cc:2533-2553)
        continue;
      }
      printf("P %d %u\n", (int)t->deframe_state, t->deframe_state == GRPC_DTS_FRAME ? t->incoming_frame_size : 0u);
    } else {
      return 4;
    }
  }
  printf("S %u %zu\n", (unsigned)t->num_pending_induced_frames, grpc_chttp2_stream_map_size(&t->stream_map));
  fflush(stdout);
  _exit(0);
}
