// Unary round trips THROUGH THE ENDPOINT VTABLE (grpc_endpoint_write / grpc_endpoint_read,
// include/grdma_endpoint.hpp), host slices in and out, the way chttp2 drives rdma_bp_posix.cc: the client writes
// one request -- [9-byte DATA frame header + 5-byte message header][payload], two slices --, the server's
// outstanding read completes, the server answers the same way, the client's outstanding read completes.  Both ends
// keep a read armed (re-armed from the read callback through the closure queue, as read_action_locked does) and a
// miniature pollset_work loop polls the two endpoints.  This is what a gRPC process would see per unary RPC at the
// transport boundary, PCIe both ways included; bench.py reports it next to the pair-level RTT, never instead of it.
//
// usage: endpoint_pingpong <iters> <payload_bytes> <mode>      prints one JSON line
//   mode 0: launch chains (one Send / one drain enqueued per write / read, completion seen in pinned memory)
//   mode 1: both pairs in latency mode: every write / read is one command posted to the resident engine
//   mode 2: mode 1 + standing reads (grdma_pair_arm_read): a watcher workgroup of the engine drains when the bytes land
//           and the poll sees the completion in host memory, no command between the arrival and the read callback
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "grdma_endpoint.hpp"

using namespace grdma_core;

#define CHECK(x)                                                             \
  do {                                                                       \
    if (!(x)) {                                                              \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); \
      exit(2);                                                               \
    }                                                                        \
  } while (0)

static std::deque<grpc_closure*> g_queue;  // a miniature ExecCtx

struct side {
  grpc_endpoint* ep = nullptr;
  grpc_slice_buffer outgoing, incoming;
  grpc_closure done_write, done_read, next_read;
  size_t got = 0;        // bytes of the message being received
  uint64_t sum = 0;      // byte sum of everything received
  bool message = false;  // a whole message has arrived
  bool failed = false;
  bool writing = false;
  size_t want = 0;
};

static void on_write(void* p, grpc_error_handle e) {
  auto* s = static_cast<side*>(p);
  if (e != GRPC_ERROR_NONE) s->failed = true;
  grpc_slice_buffer_reset_and_unref(&s->outgoing);
  s->writing = false;
}
static void do_read(void* p, grpc_error_handle) {
  auto* s = static_cast<side*>(p);
  grpc_endpoint_read(s->ep, &s->incoming, &s->done_read, false);
}
static void on_read(void* p, grpc_error_handle e) {
  auto* s = static_cast<side*>(p);
  if (e != GRPC_ERROR_NONE) { s->failed = true; return; }
  for (size_t i = 0; i < s->incoming.count; i++) {
    const grpc_slice& sl = s->incoming.slices[i];
    const uint8_t* b = GRPC_SLICE_START_PTR(sl);
    const size_t n = GRPC_SLICE_LENGTH(sl);
    for (size_t k = 0; k < n; k++) s->sum += b[k];
    s->got += n;
  }
  if (s->got >= s->want) {
    s->got -= s->want;
    s->message = true;
  }
  g_queue.push_back(&s->next_read);  // the transport re-arms its read (chttp2_transport.cc:2582-2590)
}

static void write_message(side* s, const std::vector<grpc_slice>& msg) {
  for (const grpc_slice& m : msg) {
    grpc_slice c = m;
    if (c.refcount) c.refcount->refs.fetch_add(1);
    grpc_slice_buffer_add_indexed(&s->outgoing, c);
  }
  s->writing = true;
  grpc_endpoint_write(s->ep, &s->outgoing, &s->done_write, nullptr);
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <iters> <payload_bytes> <mode 0|1|2>\n", argv[0]);
    return 1;
  }
  const size_t iters = strtoull(argv[1], nullptr, 10), payload = strtoull(argv[2], nullptr, 10);
  const int mode = atoi(argv[3]);
  const size_t warmup = iters / 10 + 5;
  setenv("GRPC_PLATFORM_TYPE", "RDMA_BP", 0);
  // (like `numactl --cpunodebind` on the GPU's node: the mailbox and the result lines then sit next to its PCIe root)
  if (!getenv("GRDMA_NO_NUMA_PIN")) grdma_host_pin_to_device_node();
  side cl, sv;
  cl.ep = grpc_endpoint_create(3, "ipv4:127.0.0.1:1", false);
  sv.ep = grpc_endpoint_create(4, "ipv4:127.0.0.1:2", true);
  CHECK(cl.ep && sv.ep && grpc_rdma_bp_connect_loopback(cl.ep, sv.ep));
  if (mode >= 1) {
    CHECK(grdma_endpoint_set_latency_mode(cl.ep, true, mode >= 2 ? 1024 : 0));
    CHECK(grdma_endpoint_set_latency_mode(sv.ep, true, mode >= 2 ? 1024 : 0));
    CHECK(grdma_engine_start() == 0);
  }

  // one message: [frame header 9 B | message header 5 B] (one inlined slice, as chttp2 merges them) + payload
  std::vector<grpc_slice> msg;
  uint64_t sum_per_msg = 0;
  {
    const size_t L = payload, F = 5 + L;
    const uint8_t head[14] = {(uint8_t)(F >> 16), (uint8_t)(F >> 8), (uint8_t)F, 0, 0, 0, 0, 0, 1,
                              0, (uint8_t)(L >> 24), (uint8_t)(L >> 16), (uint8_t)(L >> 8), (uint8_t)L};
    msg.push_back(grpc_slice_from_copied_buffer(reinterpret_cast<const char*>(head), 14));
    if (L) {
      grpc_slice s = grpc_slice_malloc(L);
      for (size_t i = 0; i < L; i++) GRPC_SLICE_START_PTR(s)[i] = (uint8_t)((i * 7 + 3) % 251);
      msg.push_back(s);
    }
    for (grpc_slice& s : msg)
      for (size_t i = 0; i < GRPC_SLICE_LENGTH(s); i++) sum_per_msg += GRPC_SLICE_START_PTR(s)[i];
  }
  const size_t msg_bytes = 14 + payload;
  for (side* s : {&cl, &sv}) {
    s->want = msg_bytes;
    grpc_slice_buffer_init(&s->outgoing);
    grpc_slice_buffer_init(&s->incoming);
    GRPC_CLOSURE_INIT(&s->done_write, on_write, s, nullptr);
    GRPC_CLOSURE_INIT(&s->done_read, on_read, s, nullptr);
    GRPC_CLOSURE_INIT(&s->next_read, do_read, s, nullptr);
    do_read(s, GRPC_ERROR_NONE);
  }

  std::vector<uint64_t> rtt;
  rtt.reserve(iters);
  auto pump = [&](bool* flag) {  // the pollset_work loop until *flag
    long idle = 0;
    while (!*flag) {
      int ran = grdma_endpoint_poll(sv.ep) + grdma_endpoint_poll(cl.ep);
      while (!g_queue.empty()) {
        grpc_closure* c = g_queue.front();
        g_queue.pop_front();
        c->cb(c->cb_arg, GRPC_ERROR_NONE);
        ran++;
      }
      CHECK(!cl.failed && !sv.failed);
      if (ran) idle = 0;
      else if (++idle > 20000000) CHECK(!"endpoint made no progress");
    }
  };
  const auto T0 = std::chrono::steady_clock::now();
  for (size_t it = 0; it < warmup + iters; it++) {
    const auto t0 = std::chrono::steady_clock::now();
    write_message(&cl, msg);
    pump(&sv.message);
    sv.message = false;
    write_message(&sv, msg);
    pump(&cl.message);
    cl.message = false;
    const auto t1 = std::chrono::steady_clock::now();
    if (it >= warmup) rtt.push_back((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count());
    while (cl.writing || sv.writing) {  // (a write callback still outstanding: let the loop run it)
      grdma_endpoint_poll(sv.ep);
      grdma_endpoint_poll(cl.ep);
      while (!g_queue.empty()) {
        grpc_closure* c = g_queue.front();
        g_queue.pop_front();
        c->cb(c->cb_arg, GRPC_ERROR_NONE);
      }
    }
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
  CHECK(cl.sum == sum_per_msg * (warmup + iters) && sv.sum == sum_per_msg * (warmup + iters));
  std::sort(rtt.begin(), rtt.end());
  // (completions the engine's watcher workgroups produced -- arrival-triggered drains, k_watch)
  const int64_t whits = mode >= 2 ? grdma_pair_watch_hits(grdma_endpoint_pair(cl.ep)) + grdma_pair_watch_hits(grdma_endpoint_pair(sv.ep)) : 0;
  printf("{\"iters\": %zu, \"payload\": %zu, \"mode\": %d, \"p50_us\": %.2f, \"p95_us\": %.2f, \"p99_us\": %.2f, "
         "\"seconds\": %.3f, \"watch_hits\": %lld, \"checked\": true}\n",
         iters, payload, mode, rtt[iters / 2] / 1e3, rtt[(size_t)(iters * 0.95)] / 1e3, rtt[(size_t)(iters * 0.99)] / 1e3,
         sec, (long long)whits);
  if (mode >= 1) grdma_engine_stop();
  grpc_endpoint_shutdown(cl.ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("done"));
  grpc_endpoint_shutdown(sv.ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("done"));
  while (!g_queue.empty()) g_queue.pop_front();
  grpc_endpoint_destroy(cl.ep);
  grpc_endpoint_destroy(sv.ep);
  for (grpc_slice& s : msg) grpc_slice_unref(s);
  return 0;
}
