// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// ref_h2_trace: the reference's OWN grpc_chttp2_encode_data (src/core/ext/transport/chttp2/transport/frame_data.cc:63-90)
// over its OWN slice layer (src/core/lib/slice/slice.cc, slice_buffer.cc: grpc_slice_buffer_tiny_add, _add with its
// inlined-slice merge rule, _move_first_no_ref with its splits), compiled unmodified.  What is written here is only the
// two call sites around them, in the reference's order:
//   * the 5-byte gRPC message header in front of a message, chttp2_transport.cc:1502-1510
//     (grpc_slice_buffer_tiny_add(&s->flow_controlled_buffer, 5), then the message slice is added);
//   * the loop that cuts the flow-controlled buffer into DATA frames of at most max_frame_size bytes,
//     DataSendContext::FlushUncompressedBytes, writing.cc:344-355 (flow-control windows open).
// The parser half of frame_data.cc is not reached and is dropped by the linker (--gc-sections).
// tests/test_oracle_vs_ref.py feeds the same messages to the oracle's orc_h2_frame_batch and compares the slice list
// (lengths, in order) and the bytes.
//
//   ops:  M <stream_id> <compressed> <end_stream> <max_frame> <len> <seed>   queue one message on the outbuf
//         F                                                                   print and reset the outbuf
//   out:  F <n slices> <crc32 of the bytes> <framing bytes> <data bytes> <len_1> ... <len_n>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <grpc/slice.h>
#include <grpc/slice_buffer.h>
#include <grpc/support/alloc.h>
#include <grpc/support/log.h>

#include "src/core/ext/transport/chttp2/transport/frame_data.h"
#include "src/core/lib/gprpp/fork.h"
#include "src/core/lib/iomgr/exec_ctx.h"
#include "src/core/lib/slice/slice_internal.h"

// ---- the few symbols of the gRPC core these three files reference -------------------------------------------------
extern "C" void gpr_log(const char*, int, gpr_log_severity, const char*, ...) {}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }
extern "C" void* gpr_malloc(size_t n) { return malloc(n ? n : 1); }
extern "C" void* gpr_zalloc(size_t n) { return calloc(n ? n : 1, 1); }
extern "C" void* gpr_realloc(void* p, size_t n) { return realloc(p, n ? n : 1); }
extern "C" void gpr_free(void* p) { free(p); }
namespace grpc_core {
GPR_TLS_CLASS_DEF(ExecCtx::exec_ctx_);
Atomic<bool> Fork::support_enabled_(false);
void Fork::DoIncExecCtxCount() {}
void Fork::DoDecExecCtxCount() {}
bool ExecCtx::Flush() { return false; }
}  // namespace grpc_core

namespace {
uint32_t crc32_of(const uint8_t* p, uint64_t n, uint32_t c) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t v = i;
      for (int k = 0; k < 8; k++) v = (v & 1) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
      table[i] = v;
    }
    init = true;
  }
  for (uint64_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c;
}
inline uint8_t pat(uint64_t seed, uint64_t j) { return (uint8_t)(seed * 131 + j * 7 + (j >> 8)); }
}  // namespace

int main() {
  grpc_slice_buffer outbuf, fcb;
  grpc_slice_buffer_init(&outbuf);
  grpc_slice_buffer_init(&fcb);
  grpc_transport_one_way_stats stats;
  memset(&stats, 0, sizeof(stats));
  char op;
  while (scanf(" %c", &op) == 1) {
    if (op == 'M') {
      unsigned id, compressed, end_stream, max_frame;
      unsigned long long len, seed;
      if (scanf("%u %u %u %u %llu %llu", &id, &compressed, &end_stream, &max_frame, &len, &seed) != 6) return 3;
      // chttp2_transport.cc:1502-1510
      uint8_t* frame_hdr = grpc_slice_buffer_tiny_add(&fcb, 5);
      frame_hdr[0] = compressed != 0;
      frame_hdr[1] = static_cast<uint8_t>(len >> 24);
      frame_hdr[2] = static_cast<uint8_t>(len >> 16);
      frame_hdr[3] = static_cast<uint8_t>(len >> 8);
      frame_hdr[4] = static_cast<uint8_t>(len);
      if (len > 0) {  // the serialized message: one reference-counted slice, as a byte stream hands it over
        grpc_slice m = grpc_slice_malloc_large(len);
        uint8_t* q = GRPC_SLICE_START_PTR(m);
        for (unsigned long long j = 0; j < len; j++) q[j] = pat(seed, j);
        grpc_slice_buffer_add(&fcb, m);
      }
      // writing.cc:344-355
      while (fcb.length > 0) {
        const uint32_t send_bytes = fcb.length < max_frame ? (uint32_t)fcb.length : max_frame;
        const int is_last = end_stream && send_bytes == fcb.length;
        grpc_chttp2_encode_data(id, &fcb, send_bytes, is_last, &stats, &outbuf);
      }
    } else if (op == 'F') {
      uint32_t c = 0xFFFFFFFFu;
      for (size_t i = 0; i < outbuf.count; i++)
        c = crc32_of(GRPC_SLICE_START_PTR(outbuf.slices[i]), GRPC_SLICE_LENGTH(outbuf.slices[i]), c);
      printf("F %zu %u %llu %llu", outbuf.count, c ^ 0xFFFFFFFFu, (unsigned long long)stats.framing_bytes,
             (unsigned long long)stats.data_bytes);
      for (size_t i = 0; i < outbuf.count; i++) printf(" %zu", (size_t)GRPC_SLICE_LENGTH(outbuf.slices[i]));
      printf("\n");
      grpc_slice_buffer_reset_and_unref(&outbuf);
      memset(&stats, 0, sizeof(stats));
    } else {
      return 3;
    }
  }
  fflush(stdout);
  _exit(0);
}
