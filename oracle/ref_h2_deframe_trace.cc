// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// ref_h2_deframe_trace: the reference's OWN gRPC message deframer, grpc_deframe_unprocessed_incoming_frames
// (src/core/ext/transport/chttp2/transport/frame_data.cc:92-276: the FH_0 .. FH_4 / FRAME state machine over a slice
// buffer of DATA payload pieces), compiled unmodified over its own slice layer (slice.cc, slice_buffer.cc).  It is fed
// the payload pieces of a stream one by one, the way grpc_chttp2_data_parser_parse stores them (frame_data.cc:278-294),
// and called until the buffer is empty, the way the incoming byte stream pulls (chttp2_transport.cc:3031-3111).  What
// the deframer hands to grpc_core::Chttp2IncomingByteStream is the trace: the stub of that class below keeps the
// reference's own arithmetic of Push / Finished (chttp2_transport.cc:3113-3155: remaining_bytes_) and prints
//     B <flags> <message length>     a message begins (the constructor)
//     Y <bytes>                      a piece of its payload (Push)
//     E                              the message is complete (Finished)
//     X                              the deframer returned an error (bad gRPC frame type byte)
// The stream and transport objects are zeroed storage: the deframer reads s->t, t->channelz_socket (null), s->id and
// counts in s->stats.  stdin (binary): u32 pieces, then per piece u32 length + bytes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

extern "C" void gpr_log(const char*, int, gpr_log_severity, const char*, ...) {}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }
extern "C" void* gpr_malloc(size_t n) { return malloc(n ? n : 1); }
extern "C" void* gpr_zalloc(size_t n) { return calloc(n ? n : 1, 1); }
extern "C" void* gpr_realloc(void* p, size_t n) { return realloc(p, n ? n : 1); }
extern "C" void gpr_free(void* p) { free(p); }
// the error path (a bad frame type byte): an opaque non-null handle is all the trace needs
static char g_error_object;
grpc_error_handle grpc_error_create(const char*, int, const grpc_slice&, grpc_error_handle*, size_t) {
  return reinterpret_cast<grpc_error_handle>(&g_error_object);
}
grpc_error_handle grpc_error_do_ref(grpc_error_handle e) { return e; }
void grpc_error_do_unref(grpc_error_handle) {}
grpc_error_handle grpc_error_set_int(grpc_error_handle e, grpc_error_ints, intptr_t) { return e; }
grpc_error_handle grpc_error_set_str(grpc_error_handle e, grpc_error_strs, const grpc_slice&) { return e; }
char* grpc_dump_slice(const grpc_slice&, uint32_t) { return strdup(""); }
namespace grpc_core {
GPR_TLS_CLASS_DEF(ExecCtx::exec_ctx_);
Atomic<bool> Fork::support_enabled_(false);
void Fork::DoIncExecCtxCount() {}
void Fork::DoDecExecCtxCount() {}
bool ExecCtx::Flush() { return false; }
namespace channelz {
void SocketNode::RecordMessageReceived() {}
}  // namespace channelz

// The byte stream the deframer feeds: the reference's bookkeeping (chttp2_transport.cc:2956-2966, 3113-3155), the
// transport side of it (closures on the combiner, flow control) left out.
Chttp2IncomingByteStream::Chttp2IncomingByteStream(grpc_chttp2_transport* transport, grpc_chttp2_stream* stream,
                                                   uint32_t frame_size, uint32_t flags)
    : ByteStream(frame_size, flags), transport_(transport), stream_(stream), refs_(2), remaining_bytes_(frame_size) {
  printf("B %u %u\n", flags, frame_size);
}
void Chttp2IncomingByteStream::Orphan() { Unref(); }
bool Chttp2IncomingByteStream::Next(size_t, grpc_closure*) { abort(); }
grpc_error_handle Chttp2IncomingByteStream::Pull(grpc_slice*) { abort(); }
void Chttp2IncomingByteStream::Shutdown(grpc_error_handle) { abort(); }
grpc_error_handle Chttp2IncomingByteStream::Push(const grpc_slice& slice, grpc_slice* slice_out) {
  if (remaining_bytes_ < GRPC_SLICE_LENGTH(slice)) {
    printf("T\n");  // "Too many bytes in stream": the deframer never pushes more than frame_size
    grpc_slice_unref_internal(slice);
    return reinterpret_cast<grpc_error_handle>(&g_error_object);
  }
  remaining_bytes_ -= static_cast<uint32_t> GRPC_SLICE_LENGTH(slice);
  printf("Y %zu\n", (size_t)GRPC_SLICE_LENGTH(slice));
  if (slice_out != nullptr) *slice_out = slice;
  return GRPC_ERROR_NONE;
}
grpc_error_handle Chttp2IncomingByteStream::Finished(grpc_error_handle error, bool) {
  if (error == GRPC_ERROR_NONE && remaining_bytes_ != 0) {
    printf("U\n");  // "Truncated message"
    error = reinterpret_cast<grpc_error_handle>(&g_error_object);
  } else if (error == GRPC_ERROR_NONE) {
    printf("E\n");
  }
  Unref();
  return error;
}
}  // namespace grpc_core

int main() {
  void* t_mem = calloc(1, sizeof(grpc_chttp2_transport));
  void* s_mem = calloc(1, sizeof(grpc_chttp2_stream));
  grpc_chttp2_transport* t = static_cast<grpc_chttp2_transport*>(t_mem);
  grpc_chttp2_stream* s = static_cast<grpc_chttp2_stream*>(s_mem);
  *const_cast<grpc_chttp2_transport**>(&s->t) = t;
  *const_cast<uint32_t*>(&s->id) = 1;
  grpc_chttp2_data_parser* p = new grpc_chttp2_data_parser();  // (default member initialisers: FH_0, nothing in flight)
  grpc_slice_buffer buf;
  grpc_slice_buffer_init(&buf);
  uint32_t n = 0;
  if (fread(&n, 4, 1, stdin) != 1) return 3;
  bool dead = false;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t len = 0;
    if (fread(&len, 4, 1, stdin) != 1) return 3;
    grpc_slice sl = grpc_slice_malloc_large(len ? len : 1);
    if (len && fread(GRPC_SLICE_START_PTR(sl), 1, len, stdin) != len) return 3;
    if (!len) sl = grpc_empty_slice();
    if (dead) continue;
    grpc_slice_buffer_add(&buf, sl);
    while (buf.count > 0) {
      grpc_slice out = grpc_empty_slice();
      grpc_core::OrphanablePtr<grpc_core::ByteStream> stream_out;
      grpc_error_handle e = grpc_deframe_unprocessed_incoming_frames(p, s, &buf, &out, &stream_out);
      if (stream_out != nullptr) stream_out.release();  // (the transport would hand it to the surface; its reference is dropped with the process)
      if (e != GRPC_ERROR_NONE) {
        printf("X\n");
        dead = true;  // (the transport closes the stream)
        break;
      }
    }
  }
  printf("S %llu %llu\n", (unsigned long long)s->stats.incoming.framing_bytes, (unsigned long long)s->stats.incoming.data_bytes);
  fflush(stdout);
  _exit(0);
}
