// HTTP/2 DATA framing (K6/K7) and deframing (K8/K9) on the device.
//
// TX  k_h2_frame_index + k_h2_frame_emit   build, in HBM, the slice list chttp2 hands to
//                    grpc_endpoint_write for a batch of gRPC messages:
//                    5-byte message header (chttp2_transport.cc:1502-1510), 9-byte
//                    DATA frame headers (grpc_chttp2_encode_data, frame_data.cc:64-90),
//                    payload sub-slices by reference, with the inlined-slice merge
//                    rule of grpc_slice_buffer_add (slice_buffer.cc:136-171) and the
//                    split rule of move_first_no_ref (slice_buffer.cc:270-313).
//                    No payload byte is copied: K1 (k_copy) gathers straight from
//                    the message buffers.
//                    One wave per message (closed-form layout; sequential only around
//                    empty messages, whose inlined slices merge across message boundaries).
// RX  k_h2_deframe   the resumable frame-header state machine of
//                    grpc_chttp2_perform_read (parsing.cc:56-253) and the gRPC
//                    message deframer (frame_data.cc:92-276) over the slices an
//                    endpoint_read delivered.  Frame headers sit at data-dependent
//                    offsets (a linked list again); one wave stages the first 32
//                    bytes of the next 64 slices in registers so the automaton
//                    never waits on memory for a header that starts a slice -- the
//                    common case, because the ring preserves slice boundaries.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "grdma_h2_kernels.h"

// --------------------------------------------------------------------- host API
// Everything here runs on one non-blocking stream of its own and waits with
// hipStreamSynchronize: a device-wide synchronize would sit behind the resident latency
// engine until that idles out.  Scratch buffers and events are kept per parser / per process.
struct grdma_h2_parser {
  grdma_h2_parser_dev* d = nullptr;
  grdma_h2_stream_dev* d_tab = nullptr;
  grdma_slice_out* d_sl = nullptr;
  uint64_t sl_cap = 0;
  grdma_h2_event* d_ev = nullptr;
  uint64_t ev_cap = 0;
  grdma_h2_deframe_result* d_res = nullptr;
  grdma_h2_table_op* d_ops = nullptr;
  uint32_t ops_cap = 0;
  // the deframer over chunks (grdma_h2_kernels.h: grdma_h2_chunks): control block, private stream maps, event segments
  uint32_t slots = 0;
  int chunks_want = 0;                 // 0 = off
  grdma_h2_chunks* d_chunks = nullptr;
  grdma_h2_stream_dev* d_tabs = nullptr;
  grdma_h2_event* d_ev_tmp = nullptr;
  uint64_t ev_tmp_cap = 0;             // events the segments hold in total
  // the last deframing a pipe enqueued for this parser: the next one is ordered behind it (same stream, or this event)
  hipStream_t last_stream = nullptr;
  hipEvent_t last_deframed = nullptr;
};

// How many chunks a parser created without saying so cuts a long list into: GRDMA_H2_CHUNKS (default and at most 256, 0 or 1 = the
// sequential deframer only).
static int h2_chunks_default() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GRDMA_H2_CHUNKS");
    v = e ? atoi(e) : H2_KMAX;
    if (v < 2) v = 0;
    if (v > H2_KMAX) v = H2_KMAX;
  }
  return v;
}

// (re)size the chunk buffers of a parser for calls that may produce ev_cap events
static bool h2_chunks_prepare(grdma_h2_parser* p, uint64_t ev_cap, hipStream_t st) {
  if (p->chunks_want < 2) return false;
  const uint64_t need = 4 * (ev_cap ? ev_cap : 1) + H2_KMAX * 64;
  if (p->d_chunks && p->ev_tmp_cap >= need) return true;
  if (!p->d_chunks) {
    if (hipMalloc((void**)&p->d_chunks, sizeof(grdma_h2_chunks)) != hipSuccess ||
        hipMalloc((void**)&p->d_tabs, sizeof(grdma_h2_stream_dev) * (size_t)H2_KMAX * p->slots) != hipSuccess)
      return false;
    if (hipMemsetAsync(p->d_chunks, 0, sizeof(grdma_h2_chunks), st) != hipSuccess) return false;
  }
  // The larger segment first, then the old one goes -- behind every deframing call that may still read it (another
  // pipe of this parser may run on another stream: the device is drained, this is a resize, not a hot path).  A
  // failure leaves the parser WITHOUT chunk buffers (capacity 0, chunks off) rather than with a control block that
  // points at freed memory.
  grdma_h2_event* bigger = nullptr;
  if (hipMalloc((void**)&bigger, sizeof(grdma_h2_event) * need) != hipSuccess) {
    (void)hipGetLastError();
    p->ev_tmp_cap = 0;
    p->chunks_want = 0;
    return false;
  }
  if (hipStreamSynchronize(st) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    hipFree(bigger);
    p->ev_tmp_cap = 0;
    p->chunks_want = 0;
    return false;
  }
  if (p->d_ev_tmp) hipFree(p->d_ev_tmp);
  p->d_ev_tmp = bigger;
  p->ev_tmp_cap = need;
  // the host-owned words of the control block (the counters stay)
  struct { grdma_h2_stream_dev* tabs; grdma_h2_event* ev_tmp; uint64_t ev_total; uint32_t slots, pad; } tail =
      {p->d_tabs, p->d_ev_tmp, need, p->slots, 0};
  static_assert(offsetof(grdma_h2_chunks, pad) + sizeof(uint32_t) - offsetof(grdma_h2_chunks, tabs) == sizeof(tail), "layout");
  const uint32_t want = (uint32_t)p->chunks_want;
  const bool ok = hipMemcpyAsync(reinterpret_cast<uint8_t*>(p->d_chunks) + offsetof(grdma_h2_chunks, tabs), &tail, sizeof(tail),
                                 hipMemcpyHostToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(reinterpret_cast<uint8_t*>(p->d_chunks) + offsetof(grdma_h2_chunks, want), &want, sizeof(want),
                                 hipMemcpyHostToDevice, st) == hipSuccess &&
                  hipStreamSynchronize(st) == hipSuccess;
  if (!ok) {  // (the control block may not know the new segment: no chunked deframing with it)
    p->ev_tmp_cap = 0;
    p->chunks_want = 0;
  }
  return ok;
}

// The deframing of one list of delivered slices, enqueued on st: over chunks when the parser has the buffers (plan,
// the chunks side by side, then the merge -- or the sequential deframer over the whole list when the chain did not hold), else the sequential deframer alone.
static void h2_enqueue_deframe(grdma_h2_parser* p, const uint8_t* arena, const grdma_slice_out* d_slices, uint64_t n,
                               grdma_h2_event* d_ev, uint64_t ev_cap, grdma_h2_deframe_result* d_res, hipStream_t st,
                               bool chunked) {
  if (!chunked || n < H2_CHUNK_MIN_SLICES) {
    hipLaunchKernelGGL(k_h2_deframe, dim3(1), dim3(H2_DEFRAME_THREADS), 0, st, p->d, arena, d_slices, n, d_ev, ev_cap, d_res);
    return;
  }
  hipLaunchKernelGGL(k_h2_deframe_chunks, dim3((unsigned)p->chunks_want), dim3(H2_DEFRAME_THREADS), 0, st, p->d, p->d_chunks,
                     arena, d_slices, n);
  hipLaunchKernelGGL(k_h2_merge_or_deframe, dim3(H2_MERGE_GRID), dim3(H2_DEFRAME_THREADS), 0, st, p->d, p->d_chunks, arena,
                     d_slices, n, d_ev, ev_cap, d_res);
}

// The framing of one message table, enqueued on st: sizes and positions, then one wave per message (grdma_h2_kernels.h).
static void h2_enqueue_frame(const grdma_h2_msg_dev* d_msgs, uint64_t n, uint32_t max_frame, grdma_sge* out, uint64_t cap,
                             uint8_t* hdr, uint64_t hdr_cap, grdma_h2_msg_pos* d_pos, grdma_h2_frame_result* d_res,
                             hipStream_t st) {
  const uint64_t per = H2_EMIT_THREADS / 64;
  if (n <= H2_FRAME_ONE_MAX) {  // one launch: every workgroup sums what lies in front of its messages itself
    hipLaunchKernelGGL(k_h2_frame_one, dim3((unsigned)((n + per - 1) / per)), dim3(H2_EMIT_THREADS), 0, st, d_msgs, n, max_frame,
                       out, cap, hdr, hdr_cap, d_res);
    return;
  }
  hipLaunchKernelGGL(k_h2_frame_index, dim3(1), dim3(256), 0, st, d_msgs, n, max_frame, cap, hdr_cap, d_pos, d_res);
  hipLaunchKernelGGL(k_h2_frame_emit, dim3((unsigned)((n + per - 1) / per)), dim3(H2_EMIT_THREADS), 0, st, d_msgs, n, max_frame,
                     out, cap, hdr, hdr_cap, (const grdma_h2_msg_pos*)d_pos);
}

static double g_h2_last_kernel_us = 0;
static uint64_t g_h2_last_boundary_steps = 0;
static uint64_t g_h2_last_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// 64 frames per bulk step (GRDMA_H2_BULK_PAIRS): on unless GRDMA_H2_NO_BULK_PAIRS or GRDMA_H2_BULK_PAIRS=0 says otherwise
static int h2_bulk_pairs_default() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GRDMA_H2_BULK_PAIRS");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}
// What a parser created without either flag does: the boundary step is on unless the environment
// says GRDMA_H2_BOUNDARY_STEP=0.
static int h2_boundary_default() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GRDMA_H2_BOUNDARY_STEP");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}

namespace {
struct h2_host_ctx {
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
h2_host_ctx* h2_ctx() {
  static h2_host_ctx c;
  if (!c.stream) {
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreate(&c.e0) != hipSuccess || hipEventCreate(&c.e1) != hipSuccess) return nullptr;
  }
  return &c;
}
template <typename T>
bool h2_grow(T** buf, uint64_t* cap, uint64_t need) {
  if (need <= *cap && *buf) return true;
  if (*buf) hipFree(*buf);
  *buf = nullptr;
  uint64_t n = *cap ? *cap : 64;
  while (n < need) n *= 2;
  if (hipMalloc((void**)buf, sizeof(T) * n) != hipSuccess) { *cap = 0; return false; }
  *cap = n;
  return true;
}
}  // namespace

extern "C" {

const char* grdma_last_error(void);
// duration of the framing / deframing kernel of the last call (HIP events), microseconds
double grdma_h2_last_kernel_us(void) { return g_h2_last_kernel_us; }
// message starts the last grdma_h2_deframe call took through the boundary step
uint64_t grdma_h2_last_boundary_steps(void) { return g_h2_last_boundary_steps; }
// counters of the last grdma_h2_deframe call: {bulk steps, frames parsed in bulk steps, boundary steps,
// then device-clock ticks: waiting for staged windows, in bulk steps, in boundary steps, in the
// byte-wise path, total}
void grdma_h2_last_deframe_stats(uint64_t out[8]) {
  for (int i = 0; i < 8; i++) out[i] = g_h2_last_stats[i];
}

int64_t grdma_h2_frame_messages(const grdma_h2_msg* msgs, uint64_t n, uint32_t max_frame,
                                grdma_slice* d_slices_out, uint64_t slices_cap,
                                void* d_hdr_arena, uint64_t hdr_cap, uint64_t* wire_bytes) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!msgs || !n || !d_slices_out || !d_hdr_arena || max_frame == 0 || max_frame >= (1u << 24))
    return -GRDMA_ERR_INVALID;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  std::vector<grdma_h2_msg_dev> tmp(n);
  for (uint64_t i = 0; i < n; i++) {
    tmp[i].payload = static_cast<const uint8_t*>(msgs[i].payload);
    tmp[i].len = msgs[i].len;
    tmp[i].stream_id = msgs[i].stream_id;
    tmp[i].flags = msgs[i].flags;
    if (msgs[i].len >= (1ull << 32)) return -GRDMA_ERR_INVALID;  // 32-bit message length field
  }
  static grdma_h2_msg_dev* d_msgs = nullptr;
  static uint64_t msgs_cap = 0;
  static grdma_h2_frame_result* d_res = nullptr;
  static grdma_h2_msg_pos* d_pos = nullptr;
  static uint64_t pos_cap = 0;
  grdma_h2_frame_result h_res;
  if (!h2_grow(&d_msgs, &msgs_cap, n) || !h2_grow(&d_pos, &pos_cap, n)) return -GRDMA_ERR_HIP;
  if (!d_res && hipMalloc((void**)&d_res, sizeof(grdma_h2_frame_result)) != hipSuccess) return -GRDMA_ERR_HIP;
  hipStream_t st = hc->stream;
  if (hipMemcpyAsync(d_msgs, tmp.data(), sizeof(grdma_h2_msg_dev) * n, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(d_res, 0, sizeof(grdma_h2_frame_result), st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  hipEventRecord(hc->e0, st);
  h2_enqueue_frame(d_msgs, n, max_frame, reinterpret_cast<grdma_sge*>(d_slices_out), slices_cap,
                   static_cast<uint8_t*>(d_hdr_arena), hdr_cap, d_pos, d_res, st);
  hipEventRecord(hc->e1, st);
  if (hipMemcpyAsync(&h_res, d_res, sizeof(h_res), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  float ms = 0;
  if (hipEventElapsedTime(&ms, hc->e0, hc->e1) == hipSuccess) g_h2_last_kernel_us = 1e3 * ms;
  if (h_res.overflow) return -GRDMA_ERR_CAPACITY;
  if (wire_bytes) *wire_bytes = h_res.wire_bytes;
  return (int64_t)h_res.nslices;
}

grdma_h2_parser* grdma_h2_parser_create_ex(int flags, uint32_t max_frame_size,
                                           uint32_t max_concurrent_streams, uint32_t table_slots) {
  if (grdma_device_count() <= 0) return nullptr;
  if (table_slots == 0) table_slots = 4096;
  if (table_slots < 16 || (table_slots & (table_slots - 1)) != 0) return nullptr;
  grdma_h2_parser* p = new grdma_h2_parser();
  grdma_h2_parser_dev init;
  memset(&init, 0, sizeof(init));
  init.is_server = (flags & GRDMA_H2_SERVER) ? 1 : 0;
  init.is_first_frame = (flags & GRDMA_H2_FIRST_FRAME) ? 1 : 0;  // chttp2_transport.cc: t->is_first_frame
  init.state = init.is_server ? 0 : 24;        // a server starts at GRPC_DTS_CLIENT_PREFIX_0
  init.max_frame_size = max_frame_size;        // http2_settings.cc:56 default 16384
  init.max_concurrent = max_concurrent_streams;  // http2_settings.cc:46 default 0xffffffff
  init.tab_mask = table_slots - 1;
  init.boundary_step = (flags & GRDMA_H2_BOUNDARY_STEP) ? 1 : (flags & GRDMA_H2_NO_BOUNDARY_STEP) ? 0 : h2_boundary_default();
  init.bulk_pairs = (flags & GRDMA_H2_BULK_PAIRS) ? 1 : (flags & GRDMA_H2_NO_BULK_PAIRS) ? 0 : h2_bulk_pairs_default();
  init.ticks = (flags & GRDMA_H2_TICKS) ? 1 : 0;
  p->slots = table_slots;
  p->chunks_want = (flags & GRDMA_H2_NO_CHUNKS) ? 0 : h2_chunks_default();
  if (hipMalloc((void**)&p->d, sizeof(init)) != hipSuccess ||
      hipMalloc((void**)&p->d_tab, sizeof(grdma_h2_stream_dev) * table_slots) != hipSuccess ||
      hipMalloc((void**)&p->d_res, sizeof(grdma_h2_deframe_result)) != hipSuccess ||
      hipMemset(p->d_tab, 0, sizeof(grdma_h2_stream_dev) * table_slots) != hipSuccess) {
    grdma_h2_parser_destroy(p);
    return nullptr;
  }
  init.tab = p->d_tab;
  if (hipMemcpy(p->d, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
    grdma_h2_parser_destroy(p);
    return nullptr;
  }
  return p;
}

grdma_h2_parser* grdma_h2_parser_create(int expect_client_prefix, uint32_t max_frame_size) {
  return grdma_h2_parser_create_ex(expect_client_prefix ? (GRDMA_H2_SERVER | GRDMA_H2_FIRST_FRAME) : 0,
                                   max_frame_size, 0xffffffffu, 0);
}

void grdma_h2_parser_destroy(grdma_h2_parser* p) {
  if (!p) return;
  hipFree(p->d);
  hipFree(p->d_tab);
  hipFree(p->d_sl);
  hipFree(p->d_ev);
  hipFree(p->d_res);
  hipFree(p->d_ops);
  hipFree(p->d_chunks);
  hipFree(p->d_tabs);
  hipFree(p->d_ev_tmp);
  delete p;
}

static int h2_table_ops(grdma_h2_parser* p, uint32_t op, const uint32_t* ids, uint32_t n) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p || (!ids && n)) return -GRDMA_ERR_INVALID;
  if (n == 0) return 0;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  uint64_t cap = p->ops_cap;
  if (!h2_grow(&p->d_ops, &cap, n)) return -GRDMA_ERR_HIP;
  p->ops_cap = (uint32_t)cap;
  std::vector<grdma_h2_table_op> h(n);
  for (uint32_t i = 0; i < n; i++) h[i] = {op, ids[i], 0, 0};
  hipStream_t st = hc->stream;
  if (hipMemcpyAsync(p->d_ops, h.data(), sizeof(grdma_h2_table_op) * n, hipMemcpyHostToDevice, st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  hipLaunchKernelGGL(k_h2_table_ops, dim3(1), dim3(1), 0, st, p->d, p->d_ops, n);
  if (hipMemcpyAsync(h.data(), p->d_ops, sizeof(grdma_h2_table_op) * n, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  int failed = 0;
  for (uint32_t i = 0; i < n; i++) failed += h[i].rc != 0;
  return failed;
}

int grdma_h2_parser_open_streams(grdma_h2_parser* p, const uint32_t* ids, uint32_t n) {
  return h2_table_ops(p, 1, ids, n);
}
int grdma_h2_parser_close_writes(grdma_h2_parser* p, const uint32_t* ids, uint32_t n) {
  return h2_table_ops(p, 2, ids, n);
}
// {calls the chunked deframer planned, calls whose chunks verified and were merged} since the parser was created
int grdma_h2_parser_chunk_stats(grdma_h2_parser* p, uint64_t out[2]) {
  if (!p || !out) return -GRDMA_ERR_INVALID;
  out[0] = out[1] = 0;
  if (!p->d_chunks) return 0;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  uint64_t v[2];
  if (hipMemcpyAsync(v, reinterpret_cast<uint8_t*>(p->d_chunks) + offsetof(grdma_h2_chunks, n_planned), sizeof(v),
                     hipMemcpyDeviceToHost, hc->stream) != hipSuccess ||
      hipStreamSynchronize(hc->stream) != hipSuccess)
    return -GRDMA_ERR_HIP;
  out[0] = v[0];
  out[1] = v[1];
  return 0;
}
// profiling aid: the phase stamps of the last chunked call, (H2_KMAX + 1) rows of 8 (csrc/grdma_h2_kernels.h)
int grdma_h2_parser_chunk_dbg(grdma_h2_parser* p, uint64_t* out, uint64_t cap_words) {
  if (!p || !out || !p->d_chunks) return -GRDMA_ERR_INVALID;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  const uint64_t words = std::min<uint64_t>(cap_words, (H2_KMAX + 1) * 8);
  if (hipMemcpyAsync(out, reinterpret_cast<uint8_t*>(p->d_chunks) + offsetof(grdma_h2_chunks, dbg), words * 8,
                     hipMemcpyDeviceToHost, hc->stream) != hipSuccess ||
      hipStreamSynchronize(hc->stream) != hipSuccess)
    return -GRDMA_ERR_HIP;
  return (int)words;
}
int64_t grdma_h2_parser_live_streams(grdma_h2_parser* p) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p) return -GRDMA_ERR_INVALID;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  grdma_h2_parser_dev h;
  if (hipMemcpyAsync(&h, p->d, sizeof(h), hipMemcpyDeviceToHost, hc->stream) != hipSuccess ||
      hipStreamSynchronize(hc->stream) != hipSuccess)
    return -GRDMA_ERR_HIP;
  return (int64_t)h.live_streams;
}

int64_t grdma_h2_deframe(grdma_h2_parser* p, const void* d_arena, const grdma_read_slice* slices,
                         uint64_t n, grdma_h2_event* events_out, uint64_t cap, int* h2_error) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p || !d_arena || (!slices && n) || !events_out) return -GRDMA_ERR_INVALID;
  h2_host_ctx* hc = h2_ctx();
  if (!hc) return -GRDMA_ERR_HIP;
  grdma_h2_deframe_result h_res;
  memset(&h_res, 0, sizeof(h_res));
  static_assert(sizeof(grdma_read_slice) == sizeof(grdma_slice_out), "layout");
  if (!h2_grow(&p->d_sl, &p->sl_cap, n ? n : 1) || !h2_grow(&p->d_ev, &p->ev_cap, cap ? cap : 1))
    return -GRDMA_ERR_HIP;
  hipStream_t st = hc->stream;
  if (n && hipMemcpyAsync(p->d_sl, slices, sizeof(grdma_slice_out) * n, hipMemcpyHostToDevice, st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  const bool chunked = n >= H2_CHUNK_MIN_SLICES && h2_chunks_prepare(p, cap, st);
  hipEventRecord(hc->e0, st);
  h2_enqueue_deframe(p, static_cast<const uint8_t*>(d_arena), p->d_sl, n, p->d_ev, cap, p->d_res, st, chunked);
  hipEventRecord(hc->e1, st);
  if (hipMemcpyAsync(&h_res, p->d_res, sizeof(h_res), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return -GRDMA_ERR_HIP;
  float ms = 0;
  if (hipEventElapsedTime(&ms, hc->e0, hc->e1) == hipSuccess) g_h2_last_kernel_us = 1e3 * ms;
  g_h2_last_boundary_steps = h_res.boundary_steps;
  {
    const uint64_t st[8] = {h_res.bulk_steps, h_res.bulk_frames, h_res.boundary_steps, h_res.t_wait,
                            h_res.t_bulk, h_res.t_boundary, h_res.t_serial, h_res.t_total};
    for (int i = 0; i < 8; i++) g_h2_last_stats[i] = st[i];
  }
  const uint64_t m = h_res.nevents < cap ? h_res.nevents : cap;
  if (m && (hipMemcpyAsync(events_out, p->d_ev, sizeof(grdma_h2_event) * m, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess))
    return -GRDMA_ERR_HIP;
  if (h2_error) *h2_error = (int)h_res.error;
  return h_res.overflow ? -GRDMA_ERR_CAPACITY : (int64_t)m;
}

// ---- HTTP/2 inside the device pipeline ------------------------------------------------------
// frame (k_h2_frame_index + k_h2_frame_emit rebuild the job's slice list from the message table) -> the streaming job
// -> deframe (the deframer over the slices the job delivered), all enqueued.  By default the two stages are kernel
// nodes of the job's own graph (one launch per step); with GRDMA_H2_PIPE_FUSED=0, and under the engine schedule, they
// are enqueued around the job's launch, ordered by events, with per-stage event timing.
struct grdma_stream_job;
int grdma_job_link_view(grdma_stream_job* j, uint32_t link, grdma_sge** d_sges, uint64_t* count,
                        grdma_slice_out** d_slices, uint8_t** dst, hipStream_t* stream);
int grdma_stream_job_launch(grdma_stream_job* j);
extern "C" int grdma_job_set_hooks(grdma_stream_job* j, const grdma_job_hook* pre, uint32_t n_pre, const grdma_job_hook* post,
                                   uint32_t n_post);

struct grdma_h2_pipe {
  grdma_stream_job* job = nullptr;
  grdma_h2_parser* parser = nullptr;
  grdma_sge* d_sges = nullptr;
  uint64_t count = 0;
  grdma_slice_out* d_slices = nullptr;
  uint8_t* dst = nullptr;
  hipStream_t job_stream = nullptr, frame_stream = nullptr, deframe_stream = nullptr;
  hipEvent_t framed = nullptr, job_done = nullptr, deframed = nullptr;
  hipEvent_t t_f0 = nullptr, t_d0 = nullptr, t_f1 = nullptr, t_d1 = nullptr;  // kernel start / end stamps of the last step
  grdma_h2_msg_dev* d_msgs = nullptr;
  grdma_h2_msg_pos* d_pos = nullptr;
  uint64_t nmsgs = 0;
  uint32_t max_frame = 16384;
  uint8_t* d_hdr = nullptr;
  uint64_t hdr_cap = 0;
  grdma_h2_frame_result* d_fres = nullptr;
  grdma_h2_deframe_result* d_dres = nullptr;
  grdma_h2_event* d_ev = nullptr;
  uint64_t ev_cap = 0, delivered = 0;
  uint64_t boundary_steps = 0, t_boundary = 0;  // of the last synced step
  bool launched = false;
  bool chunked = false;  // the parser has chunk buffers for this pipe's event capacity
  bool fused = false;    // framing and deframing are nodes of the job's graph (one launch per step)
  bool timed = false;    // the last step recorded the per-stage timing events
};

namespace {
hipStream_t g_pipe_frame_stream = nullptr, g_pipe_deframe_stream = nullptr;
}

grdma_h2_pipe* grdma_h2_pipe_create(grdma_stream_job* job, uint32_t link, const grdma_h2_msg* msgs, uint64_t nmsgs,
                                    uint32_t max_frame, grdma_h2_parser* parser, uint64_t delivered_slices,
                                    uint64_t events_cap) {
  if (grdma_device_count() <= 0 || !job || !msgs || !nmsgs || !parser || max_frame == 0 || max_frame >= (1u << 24))
    return nullptr;
  if (!g_pipe_frame_stream &&
      (hipStreamCreateWithFlags(&g_pipe_frame_stream, hipStreamNonBlocking) != hipSuccess ||
       hipStreamCreateWithFlags(&g_pipe_deframe_stream, hipStreamNonBlocking) != hipSuccess))
    return nullptr;
  grdma_h2_pipe* p = new grdma_h2_pipe();
  p->job = job;
  p->parser = parser;
  p->nmsgs = nmsgs;
  p->max_frame = max_frame;
  p->delivered = delivered_slices;
  p->ev_cap = events_cap;
  p->frame_stream = g_pipe_frame_stream;      // shared by all pipes: framings are ordered among themselves
  p->deframe_stream = g_pipe_deframe_stream;  // shared: the parser state is handed from one deframing to the next
  std::vector<grdma_h2_msg_dev> tmp(nmsgs);
  for (uint64_t i = 0; i < nmsgs; i++) {
    tmp[i].payload = static_cast<const uint8_t*>(msgs[i].payload);
    tmp[i].len = msgs[i].len;
    tmp[i].stream_id = msgs[i].stream_id;
    tmp[i].flags = msgs[i].flags;
  }
  bool ok = grdma_job_link_view(job, link, &p->d_sges, &p->count, &p->d_slices, &p->dst, &p->job_stream) == 0;
  // The deframing goes behind the job on the JOB's stream unless GRDMA_H2_DEFRAME_STREAM=1 asks for a stream of its
  // own: the next job does not start before the deframing has ended either way (measured: kernels of the two streams
  // do not run side by side), and a hand-over between streams costs ~20 us of idle device on each side of it.
  static const bool own_stream = [] { const char* e = getenv("GRDMA_H2_DEFRAME_STREAM"); return e && atoi(e) != 0; }();
  if (ok && !own_stream) p->deframe_stream = p->job_stream;
  p->hdr_cap = 32 * (p->count + 64);
  ok = ok && hipMalloc((void**)&p->d_msgs, sizeof(grdma_h2_msg_dev) * nmsgs) == hipSuccess &&
       hipMalloc((void**)&p->d_hdr, p->hdr_cap) == hipSuccess &&
       hipMalloc((void**)&p->d_pos, sizeof(grdma_h2_msg_pos) * nmsgs) == hipSuccess &&
       hipMalloc((void**)&p->d_fres, sizeof(grdma_h2_frame_result)) == hipSuccess &&
       hipMalloc((void**)&p->d_dres, sizeof(grdma_h2_deframe_result)) == hipSuccess &&
       hipMalloc((void**)&p->d_ev, sizeof(grdma_h2_event) * (events_cap ? events_cap : 1)) == hipSuccess &&
       hipMemcpy(p->d_msgs, tmp.data(), sizeof(grdma_h2_msg_dev) * nmsgs, hipMemcpyHostToDevice) == hipSuccess &&
       hipEventCreateWithFlags(&p->framed, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&p->job_done, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&p->deframed, hipEventDisableTiming) == hipSuccess &&
       hipEventCreate(&p->t_f0) == hipSuccess && hipEventCreate(&p->t_f1) == hipSuccess &&
       hipEventCreate(&p->t_d0) == hipSuccess && hipEventCreate(&p->t_d1) == hipSuccess;
  if (!ok) {
    grdma_h2_pipe_destroy(p);
    return nullptr;
  }
  p->chunked = delivered_slices >= H2_CHUNK_MIN_SLICES && h2_chunks_prepare(parser, events_cap, p->deframe_stream);
  // Framing and deframing as nodes of the job's own graph (default; GRDMA_H2_PIPE_FUSED=0: stages enqueued around the
  // graph launch, with per-stage event timing): a graph boundary costs ~15-20 us of idle device on each side.
  const char* fe = getenv("GRDMA_H2_PIPE_FUSED");
  if (!fe || atoi(fe) != 0) {
    auto arg = [](uint64_t v) { return v; };
    auto ptr = [](const void* q) { return (uint64_t)(uintptr_t)q; };
    grdma_job_hook pre[2], post[2];
    memset(pre, 0, sizeof(pre));
    memset(post, 0, sizeof(post));
    const uint64_t per = H2_EMIT_THREADS / 64;
    uint32_t n_pre = 2;
    if (p->nmsgs <= H2_FRAME_ONE_MAX) {
      pre[0] = grdma_job_hook{(const void*)k_h2_frame_one, (uint32_t)((p->nmsgs + per - 1) / per), H2_EMIT_THREADS,
                              {ptr(p->d_msgs), arg(p->nmsgs), arg(p->max_frame), ptr(p->d_sges), arg(p->count), ptr(p->d_hdr),
                               arg(p->hdr_cap), ptr(p->d_fres)}};
      n_pre = 1;
    } else {
      pre[0] = grdma_job_hook{(const void*)k_h2_frame_index, 1, 256,
                              {ptr(p->d_msgs), arg(p->nmsgs), arg(p->max_frame), arg(p->count), arg(p->hdr_cap), ptr(p->d_pos),
                               ptr(p->d_fres)}};
      pre[1] = grdma_job_hook{(const void*)k_h2_frame_emit, (uint32_t)((p->nmsgs + per - 1) / per), H2_EMIT_THREADS,
                              {ptr(p->d_msgs), arg(p->nmsgs), arg(p->max_frame), ptr(p->d_sges), arg(p->count), ptr(p->d_hdr),
                               arg(p->hdr_cap), ptr(p->d_pos)}};
    }
    uint32_t n_post;
    if (p->chunked) {
      post[0] = grdma_job_hook{(const void*)k_h2_deframe_chunks, (uint32_t)parser->chunks_want, H2_DEFRAME_THREADS,
                               {ptr(parser->d), ptr(parser->d_chunks), ptr(p->dst), ptr(p->d_slices), arg(p->delivered)}};
      post[1] = grdma_job_hook{(const void*)k_h2_merge_or_deframe, H2_MERGE_GRID, H2_DEFRAME_THREADS,
                               {ptr(parser->d), ptr(parser->d_chunks), ptr(p->dst), ptr(p->d_slices), arg(p->delivered),
                                ptr(p->d_ev), arg(p->ev_cap), ptr(p->d_dres)}};
      n_post = 2;
    } else {
      post[0] = grdma_job_hook{(const void*)k_h2_deframe, 1, H2_DEFRAME_THREADS,
                               {ptr(parser->d), ptr(p->dst), ptr(p->d_slices), arg(p->delivered), ptr(p->d_ev), arg(p->ev_cap),
                                ptr(p->d_dres)}};
      n_post = 1;
    }
    if (grdma_job_set_hooks(job, pre, n_pre, post, n_post) != 0) {
      grdma_h2_pipe_destroy(p);
      return nullptr;
    }
    p->fused = true;
    p->deframe_stream = p->job_stream;
  }
  return p;
}

void grdma_h2_pipe_destroy(grdma_h2_pipe* p) {
  if (!p) return;
  if (p->launched) {
    hipStreamSynchronize(p->frame_stream);
    hipStreamSynchronize(p->job_stream);
    hipStreamSynchronize(p->deframe_stream);
  }
  if (p->parser && p->parser->last_deframed == p->deframed) p->parser->last_deframed = nullptr;  // (synchronised above)
  if (p->fused && p->job) grdma_job_set_hooks(p->job, nullptr, 0, nullptr, 0);
  hipFree(p->d_msgs);
  hipFree(p->d_pos);
  hipFree(p->d_hdr);
  hipFree(p->d_fres);
  hipFree(p->d_dres);
  hipFree(p->d_ev);
  if (p->framed) hipEventDestroy(p->framed);
  if (p->job_done) hipEventDestroy(p->job_done);
  if (p->deframed) hipEventDestroy(p->deframed);
  for (hipEvent_t e : {p->t_f0, p->t_f1, p->t_d0, p->t_d1})
    if (e) hipEventDestroy(e);
  delete p;
}

// One step on the job's captured graph (schedule 0: the only schedule since the link engine was retired).
int grdma_h2_pipe_enqueue(grdma_h2_pipe* p, int schedule) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p || schedule != 0) return -GRDMA_ERR_INVALID;
  if (p->fused && schedule == 0) {
    // one graph launch: [k_h2_frame_index, k_h2_frame_emit] -> the job's rounds -> [the deframer]; steps and pipes
    // of one connection are ordered by the job's stream (a parser last used on another stream: by its event)
    if (p->parser->last_deframed && p->parser->last_stream != p->job_stream &&
        hipStreamWaitEvent(p->job_stream, p->parser->last_deframed, 0) != hipSuccess)
      return -GRDMA_ERR_HIP;
    const int rc = grdma_stream_job_launch(p->job);
    if (rc < 0) return rc;
    if (hipEventRecord(p->deframed, p->job_stream) != hipSuccess) return -GRDMA_ERR_HIP;
    p->parser->last_stream = p->job_stream;
    p->parser->last_deframed = p->deframed;
    p->launched = true;
    p->timed = false;
    return 0;
  }
  p->timed = true;
  // framing overwrites the slice table the job's previous step read
  if (p->launched && hipStreamWaitEvent(p->frame_stream, p->job_done, 0) != hipSuccess) return -GRDMA_ERR_HIP;
  if (hipMemsetAsync(p->d_fres, 0, sizeof(grdma_h2_frame_result), p->frame_stream) != hipSuccess) return -GRDMA_ERR_HIP;
  hipEventRecord(p->t_f0, p->frame_stream);
  h2_enqueue_frame(p->d_msgs, p->nmsgs, p->max_frame, p->d_sges, p->count, p->d_hdr, p->hdr_cap, p->d_pos, p->d_fres,
                   p->frame_stream);
  hipEventRecord(p->t_f1, p->frame_stream);
  if (hipEventRecord(p->framed, p->frame_stream) != hipSuccess) return -GRDMA_ERR_HIP;
  // the job reads the slice table and overwrites what the previous deframing parsed
  if (hipStreamWaitEvent(p->job_stream, p->framed, 0) != hipSuccess) return -GRDMA_ERR_HIP;
  if (p->launched && p->deframe_stream != p->job_stream && hipStreamWaitEvent(p->job_stream, p->deframed, 0) != hipSuccess)
    return -GRDMA_ERR_HIP;
  const int rc = grdma_stream_job_launch(p->job);
  if (rc < 0) return rc;
  if (hipEventRecord(p->job_done, p->job_stream) != hipSuccess) return -GRDMA_ERR_HIP;
  if (p->deframe_stream != p->job_stream && hipStreamWaitEvent(p->deframe_stream, p->job_done, 0) != hipSuccess)
    return -GRDMA_ERR_HIP;
  if (p->parser->last_deframed && p->parser->last_stream != p->deframe_stream &&
      hipStreamWaitEvent(p->deframe_stream, p->parser->last_deframed, 0) != hipSuccess)
    return -GRDMA_ERR_HIP;  // (the parser state is handed from one deframing to the next)
  hipEventRecord(p->t_d0, p->deframe_stream);
  h2_enqueue_deframe(p->parser, p->dst, p->d_slices, p->delivered, p->d_ev, p->ev_cap, p->d_dres, p->deframe_stream, p->chunked);
  hipEventRecord(p->t_d1, p->deframe_stream);
  if (hipEventRecord(p->deframed, p->deframe_stream) != hipSuccess) return -GRDMA_ERR_HIP;
  p->parser->last_stream = p->deframe_stream;
  p->parser->last_deframed = p->deframed;
  p->launched = true;
  return 0;
}

// Wait for the last step and report it: out = {slices framed, frame overflow, events, deframe
// overflow, slices parsed, h2 error, framing kernel us, deframing kernel us, bulk steps, frames
// parsed by bulk steps, then the deframer's device-clock ticks: waiting for the look-ahead ring,
// in bulk steps, total, in the byte-wise path}; events_out (may be NULL) receives up to cap events.
int grdma_h2_pipe_sync(grdma_h2_pipe* p, uint64_t out[14], grdma_h2_event* events_out, uint64_t cap) {
  if (grdma_device_count() <= 0) return -GRDMA_ERR_NO_DEVICE;
  if (!p || !out) return -GRDMA_ERR_INVALID;
  if (hipStreamSynchronize(p->frame_stream) != hipSuccess || hipStreamSynchronize(p->job_stream) != hipSuccess ||
      hipStreamSynchronize(p->deframe_stream) != hipSuccess)
    return -GRDMA_ERR_HIP;
  grdma_h2_frame_result fr;
  grdma_h2_deframe_result dr;
  if (hipMemcpy(&fr, p->d_fres, sizeof(fr), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(&dr, p->d_dres, sizeof(dr), hipMemcpyDeviceToHost) != hipSuccess)
    return -GRDMA_ERR_HIP;
  out[0] = fr.nslices;
  out[1] = fr.overflow;
  out[2] = dr.nevents;
  out[3] = dr.overflow;
  out[4] = dr.slices_done;
  out[5] = (uint64_t)dr.error;
  out[8] = dr.bulk_steps;
  out[9] = dr.bulk_frames;
  out[10] = dr.t_wait;
  out[11] = dr.t_bulk;
  out[12] = dr.t_total;
  out[13] = dr.t_serial;
  p->boundary_steps = dr.boundary_steps;
  p->t_boundary = dr.t_boundary;
  float fms = 0, dms = 0;
  out[6] = out[7] = 0;
  if (p->launched && p->timed && hipEventElapsedTime(&fms, p->t_f0, p->t_f1) == hipSuccess) out[6] = (uint64_t)(fms * 1e3f);
  if (p->launched && p->timed && hipEventElapsedTime(&dms, p->t_d0, p->t_d1) == hipSuccess) out[7] = (uint64_t)(dms * 1e3f);
  const uint64_t m = std::min<uint64_t>(std::min<uint64_t>(dr.nevents, cap), p->ev_cap);
  if (events_out && m && hipMemcpy(events_out, p->d_ev, sizeof(grdma_h2_event) * m, hipMemcpyDeviceToHost) != hipSuccess)
    return -GRDMA_ERR_HIP;
  return 0;
}


// {message starts taken by the boundary step, device-clock ticks inside it} of the last synced step
int grdma_h2_pipe_boundary_stats(grdma_h2_pipe* p, uint64_t out[2]) {
  if (!p || !out) return -GRDMA_ERR_INVALID;
  out[0] = p->boundary_steps;
  out[1] = p->t_boundary;
  return 0;
}

}  // extern "C"
