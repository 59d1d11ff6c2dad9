// Endpoint conformance harness for the HIP RDMA_BP endpoint, in the shape of the
// reference's pluggable endpoint test (test/core/iomgr/endpoint_tests.cc:341-355:
// multiple_shutdown_test, read_and_write_test with a byte pattern i % 256 checked by
// the reader, the write = slice = i sweep).  Sizes are scaled to what the blocking
// C ABI moves in seconds; every case is also byte-checked.
//
// Usage: endpoint_conformance <num_bytes> <write_size> <slice_size> <shutdown 0|1>
//        endpoint_conformance sweep <lo> <hi>
//        endpoint_conformance multiple_shutdown
//        endpoint_conformance pollset <connections> <rounds> <bpev 0|1>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <chrono>
#include <string>
#include <mutex>
#include <atomic>
#include <thread>
#include <vector>

#include "grdma_endpoint.hpp"

using namespace grdma_core;

#define CHECK(x)                                                          \
  do {                                                                    \
    if (!(x)) {                                                           \
      fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); \
      exit(2);                                                            \
    }                                                                     \
  } while (0)

// A miniature ExecCtx: closures scheduled from handlers run from the main loop, so
// long read/write chains do not recurse (endpoint_tests.cc:147-150).
static std::deque<std::pair<grpc_closure*, grpc_error_handle>> g_exec_queue;
static void exec_ctx_run(grpc_closure* c, grpc_error_handle e) { g_exec_queue.emplace_back(c, e); }
static bool exec_ctx_flush() {
  bool did = false;
  while (!g_exec_queue.empty()) {
    auto item = g_exec_queue.front();
    g_exec_queue.pop_front();
    item.first->cb(item.first->cb_arg, item.second);
    did = true;
  }
  return did;
}

struct fixture {
  grpc_endpoint* client_ep;
  grpc_endpoint* server_ep;
};

static fixture create_fixture() {
  setenv("GRPC_PLATFORM_TYPE", "RDMA_BP", 1);
  fixture f;
  f.client_ep = grpc_endpoint_create(3, "ipv4:127.0.0.1:1", false);
  f.server_ep = grpc_endpoint_create(4, "ipv4:127.0.0.1:2", true);
  CHECK(f.client_ep != nullptr && f.server_ep != nullptr);
  CHECK(grpc_rdma_bp_connect_loopback(f.client_ep, f.server_ep));
  return f;
}

// bytes 0,1,2,...,255,0,... across all slices of the stream
static size_t count_slices(grpc_slice* slices, size_t nslices, int* current_data) {
  size_t num_bytes = 0;
  for (size_t i = 0; i < nslices; ++i) {
    unsigned char* buf = GRPC_SLICE_START_PTR(slices[i]);
    for (size_t j = 0; j < GRPC_SLICE_LENGTH(slices[i]); ++j) {
      CHECK(buf[j] == *current_data);
      *current_data = (*current_data + 1) % 256;
    }
    num_bytes += GRPC_SLICE_LENGTH(slices[i]);
  }
  return num_bytes;
}

static void fill_buffer(grpc_slice_buffer* out, size_t num_bytes, size_t slice_size, uint8_t* current) {
  grpc_slice_buffer_reset_and_unref(out);
  while (num_bytes > 0) {
    const size_t n = num_bytes < slice_size ? num_bytes : slice_size;
    grpc_slice s = grpc_slice_malloc(n);
    unsigned char* buf = GRPC_SLICE_START_PTR(s);
    for (size_t j = 0; j < n; ++j) buf[j] = (*current)++;
    grpc_slice_buffer_add_indexed(out, s);
    num_bytes -= n;
  }
}

struct read_and_write_test_state {
  grpc_endpoint* read_ep;
  grpc_endpoint* write_ep;
  size_t target_bytes, bytes_read, current_write_size, bytes_written, slice_size;
  int current_read_data;
  uint8_t current_write_data;
  int read_done, write_done;
  grpc_slice_buffer incoming, outgoing;
  grpc_closure done_read, done_write, read_scheduler, write_scheduler;
};

static void read_scheduler(void* data, grpc_error_handle) {
  auto* st = static_cast<read_and_write_test_state*>(data);
  grpc_endpoint_read(st->read_ep, &st->incoming, &st->done_read, /*urgent=*/false);
}

static void read_handler(void* data, grpc_error_handle error) {
  auto* st = static_cast<read_and_write_test_state*>(data);
  st->bytes_read += count_slices(st->incoming.slices, st->incoming.count, &st->current_read_data);
  if (st->bytes_read == st->target_bytes || error != GRPC_ERROR_NONE) {
    st->read_done = 1 + (error == GRPC_ERROR_NONE);
  } else {
    exec_ctx_run(&st->read_scheduler, GRPC_ERROR_NONE);
  }
}

static void write_scheduler(void* data, grpc_error_handle) {
  auto* st = static_cast<read_and_write_test_state*>(data);
  grpc_endpoint_write(st->write_ep, &st->outgoing, &st->done_write, nullptr);
}

static void write_handler(void* data, grpc_error_handle error) {
  auto* st = static_cast<read_and_write_test_state*>(data);
  if (error == GRPC_ERROR_NONE) {
    st->bytes_written += st->current_write_size;
    if (st->target_bytes - st->bytes_written < st->current_write_size)
      st->current_write_size = st->target_bytes - st->bytes_written;
    if (st->current_write_size != 0) {
      fill_buffer(&st->outgoing, st->current_write_size, st->slice_size, &st->current_write_data);
      exec_ctx_run(&st->write_scheduler, GRPC_ERROR_NONE);
      return;
    }
  }
  st->write_done = 1 + (error == GRPC_ERROR_NONE);
}

static void read_and_write_test(size_t num_bytes, size_t write_size, size_t slice_size, bool shutdown) {
  fixture f = create_fixture();
  read_and_write_test_state st;
  st.read_ep = f.client_ep;
  st.write_ep = f.server_ep;
  st.target_bytes = num_bytes;
  st.bytes_read = 0;
  st.current_write_size = write_size;
  st.bytes_written = 0;
  st.slice_size = slice_size;
  st.read_done = st.write_done = 0;
  st.current_read_data = 0;
  st.current_write_data = 0;
  GRPC_CLOSURE_INIT(&st.read_scheduler, read_scheduler, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.done_read, read_handler, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.write_scheduler, write_scheduler, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.done_write, write_handler, &st, nullptr);
  grpc_slice_buffer_init(&st.outgoing);
  grpc_slice_buffer_init(&st.incoming);

  // start by pretending an initial write completed (same handler for every iteration)
  st.bytes_written -= st.current_write_size;
  write_handler(&st, GRPC_ERROR_NONE);
  exec_ctx_flush();
  grpc_endpoint_read(st.read_ep, &st.incoming, &st.done_read, /*urgent=*/false);
  if (shutdown) {
    grpc_endpoint_shutdown(st.read_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("Test Shutdown"));
    grpc_endpoint_shutdown(st.write_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("Test Shutdown"));
  }
  exec_ctx_flush();
  long spins = 0;
  while (!st.read_done || !st.write_done) {  // the pollset_work loop
    int ran = grdma_endpoint_poll(st.read_ep) + grdma_endpoint_poll(st.write_ep);
    if (exec_ctx_flush()) ran++;
    if (!ran && ++spins > 2000000) CHECK(!"endpoint made no progress");
    if (ran) spins = 0;
  }
  if (!shutdown) {
    CHECK(st.read_done == 2 && st.write_done == 2);
    CHECK(st.bytes_read == num_bytes && st.bytes_written == num_bytes);
  } else {
    CHECK(st.read_done >= 1 && st.write_done >= 1);
  }
  grpc_endpoint_shutdown(st.read_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("test done"));
  grpc_endpoint_shutdown(st.write_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("test done"));
  grpc_endpoint_destroy(st.read_ep);
  grpc_endpoint_destroy(st.write_ep);
  grpc_slice_buffer_destroy(&st.outgoing);
  grpc_slice_buffer_destroy(&st.incoming);
  printf("read_and_write_test num_bytes=%zu write_size=%zu slice_size=%zu shutdown=%d: ok\n", num_bytes,
         write_size, slice_size, (int)shutdown);
}

static int g_fail_count = 0;
static void inc_on_failure(void*, grpc_error_handle error) { g_fail_count += (error != GRPC_ERROR_NONE); }

// endpoint_tests.cc multiple_shutdown_test: a pending read fails once on shutdown;
// reads and writes after shutdown fail immediately; shutting down twice is harmless.
static void multiple_shutdown_test() {
  fixture f = create_fixture();
  grpc_slice_buffer slice_buffer;
  grpc_slice_buffer_init(&slice_buffer);
  grpc_closure cb;
  GRPC_CLOSURE_INIT(&cb, inc_on_failure, nullptr, nullptr);
  grpc_endpoint_read(f.client_ep, &slice_buffer, &cb, false);
  CHECK(g_fail_count == 0);
  grpc_endpoint_shutdown(f.client_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("Test Shutdown"));
  CHECK(g_fail_count == 1);
  grpc_endpoint_read(f.client_ep, &slice_buffer, &cb, false);
  CHECK(g_fail_count == 2);
  grpc_slice_buffer_add(&slice_buffer, grpc_slice_from_copied_buffer("a", 1));
  grpc_endpoint_write(f.client_ep, &slice_buffer, &cb, nullptr);
  CHECK(g_fail_count == 3);
  grpc_endpoint_shutdown(f.client_ep, GRPC_ERROR_CREATE_FROM_STATIC_STRING("Test Shutdown"));
  CHECK(g_fail_count == 3);
  grpc_slice_buffer_destroy(&slice_buffer);
  grpc_endpoint_destroy(f.client_ep);
  grpc_endpoint_destroy(f.server_ep);
  printf("multiple_shutdown_test: ok\n");
}

// half-close: the peer disconnects, a pending read completes with "Pair closed" /
// UNAVAILABLE (rdma_bp_posix.cc:220-228, 86-96)
static int g_status = -1;
static std::string g_desc;
static void record_error(void*, grpc_error_handle e) {
  if (e) { g_status = e->grpc_status; g_desc = e->description; }
}
// polls the endpoint until `cond` holds (asynchronous Sends / drains complete on the device's clock), 5 s at most
template <class F>
static int poll_until(grpc_endpoint* ep, F cond) {
  int ran = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (!cond() && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) ran += grdma_endpoint_poll(ep);
  return ran;
}
static void half_close_test() {
  fixture f = create_fixture();
  grpc_slice_buffer in;
  grpc_slice_buffer_init(&in);
  grpc_closure cb;
  GRPC_CLOSURE_INIT(&cb, record_error, nullptr, nullptr);
  grpc_endpoint_read(f.client_ep, &in, &cb, false);
  CHECK(grdma_endpoint_poll(f.client_ep) == 0);
  grpc_endpoint_destroy(f.server_ep);  // Disconnect(): peer_exit = 1 in my status buffer
  g_status = -1;
  g_desc.clear();
  const int ran = poll_until(f.client_ep, [&] { return g_status != -1; });
  CHECK(ran == 1 && g_status == GRPC_STATUS_UNAVAILABLE && g_desc == "Pair closed");
  grpc_slice_buffer_destroy(&in);
  grpc_endpoint_destroy(f.client_ep);
  printf("half_close_test: ok\n");
}

// A write that cannot complete because the peer exited fails with "Peer has been exited"
// (rdma_bp_posix.cc:505-510), the endpoint forgets the slices it unreffed, and the NEXT write
// is taken normally and fails the same way instead of touching freed memory or spinning.
static int g_write_cbs = 0;
static void record_write(void* p, grpc_error_handle e) {
  g_write_cbs++;
  record_error(p, e);
}
static void write_after_peer_exit_test() {
  setenv("GRPC_RDMA_RING_BUFFER_SIZE_KB", "64", 1);
  setenv("GRPC_RDMA_HIP_SEND_BUFFER_KB", "0", 1);  // the reference's flow: a write is outstanding until its last Send
  fixture f = create_fixture();
  unsetenv("GRPC_RDMA_RING_BUFFER_SIZE_KB");
  unsetenv("GRPC_RDMA_HIP_SEND_BUFFER_KB");
  grpc_slice_buffer out;
  grpc_slice_buffer_init(&out);
  uint8_t cur = 0;
  fill_buffer(&out, 300000, 8192, &cur);  // more than the 64 KiB ring takes: the write parks
  grpc_closure cb;
  GRPC_CLOSURE_INIT(&cb, record_write, nullptr, nullptr);
  g_write_cbs = 0; g_status = -1; g_desc.clear();
  grpc_endpoint_write(f.client_ep, &out, &cb, nullptr);
  CHECK(g_write_cbs == 0);
  grpc_endpoint_destroy(f.server_ep);  // peer_exit = 1: the client end is half closed
  int ran = 0;
  ran += poll_until(f.client_ep, [&] { return g_write_cbs != 0; });
  CHECK(g_write_cbs == 1 && g_status == GRPC_STATUS_UNAVAILABLE && g_desc == "Peer has been exited");
  CHECK(out.count == 0);  // reset_and_unref
  // a second write: accepted, and reported through its own closure
  fill_buffer(&out, 300000, 8192, &cur);
  g_status = -1; g_desc.clear();
  grpc_endpoint_write(f.client_ep, &out, &cb, nullptr);
  poll_until(f.client_ep, [&] { return g_write_cbs >= 2; });
  CHECK(g_write_cbs == 2 && g_status == GRPC_STATUS_UNAVAILABLE && g_desc == "Peer has been exited");
  grpc_slice_buffer_destroy(&out);
  grpc_endpoint_destroy(f.client_ep);
  printf("write_after_peer_exit_test: ok\n");
}

// The same with the endpoint's send buffers (the default): a write that fits one completes as soon as its bytes have
// been copied -- a socket's behaviour -- and goes out behind the callback; when the peer exits before it has, the
// write AFTER the failure is the one that reports it, with the reference's wording, and nothing spins or leaks.
static void buffered_write_after_peer_exit_test() {
  setenv("GRPC_RDMA_RING_BUFFER_SIZE_KB", "64", 1);
  fixture f = create_fixture();
  unsetenv("GRPC_RDMA_RING_BUFFER_SIZE_KB");
  grpc_slice_buffer out;
  grpc_slice_buffer_init(&out);
  uint8_t cur = 0;
  fill_buffer(&out, 300000, 8192, &cur);  // more than the 64 KiB ring takes, less than a send buffer
  grpc_closure cb;
  GRPC_CLOSURE_INIT(&cb, record_write, nullptr, nullptr);
  g_write_cbs = 0; g_status = -1; g_desc.clear();
  grpc_endpoint_write(f.client_ep, &out, &cb, nullptr);
  CHECK(g_write_cbs == 1 && g_status == -1);  // completed into the send buffer, no error
  CHECK(out.count == 0);
  // a second one takes the other buffer and waits behind the first
  fill_buffer(&out, 100000, 8192, &cur);
  grpc_endpoint_write(f.client_ep, &out, &cb, nullptr);
  CHECK(g_write_cbs == 2 && g_status == -1);
  grpc_endpoint_destroy(f.server_ep);  // peer_exit = 1: the client end is half closed
  // the parked Send finds out from the writable edge
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200)) grdma_endpoint_poll(f.client_ep);
  fill_buffer(&out, 20000, 8192, &cur);
  g_status = -1; g_desc.clear();
  grpc_endpoint_write(f.client_ep, &out, &cb, nullptr);
  CHECK(g_write_cbs == 3 && g_status == GRPC_STATUS_UNAVAILABLE && g_desc == "Peer has been exited");
  CHECK(out.count == 0);
  grpc_slice_buffer_destroy(&out);
  grpc_endpoint_destroy(f.client_ep);
  printf("buffered_write_after_peer_exit_test: ok\n");
}

// ---- many endpoints in ONE pollset (the shape of a server's pollset: every accepted
// connection's fd lands in the same pollable, ev_epollex_rdma_bpev_linux.cc:705-745) --------
// n connections, each client writes `rounds` messages of a seeded size, each server echoes what it
// reads; everything is driven by grdma_pollset_work() alone.  bpev = 1: RDMA_BPEV -- the
// endpoints register with the background poller and the pollset falls back to epoll_wait on
// their wakeup fds after the busy-polling budget.
struct echo_conn {
  grpc_endpoint *client, *server;
  grpc_slice_buffer c_out, c_in, s_out, s_in;
  grpc_closure c_wrote, c_read, s_wrote, s_read;
  size_t rounds_left, msg_len, c_got, s_pending;
  uint32_t seed;
  int pattern_w, pattern_r;
  bool c_writing, s_writing, done, failed;
  // (several threads may be inside pollset_work: the client's write callback and its read callback can run on
  // different threads, and the next message may only be written once the callback of the last write has run)
  std::mutex mu;
  bool send_when_written = false;
};
static std::atomic<size_t> g_conns_done{0};

static void echo_client_send(echo_conn* c) {
  c->seed = c->seed * 1664525u + 1013904223u;
  c->msg_len = 1 + (c->seed >> 8) % 40000;
  uint8_t cur = (uint8_t)c->pattern_w;
  fill_buffer(&c->c_out, c->msg_len, 8192, &cur);
  c->pattern_w = cur;
  c->c_got = 0;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->c_writing = true;
  }
  grpc_endpoint_write(c->client, &c->c_out, &c->c_wrote, nullptr);
}
static void echo_c_wrote(void* p, grpc_error_handle e) {
  auto* c = static_cast<echo_conn*>(p);
  if (e != GRPC_ERROR_NONE) c->failed = true;
  bool send;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->c_writing = false;
    send = c->send_when_written;
    c->send_when_written = false;
  }
  if (send && !c->failed) echo_client_send(c);
}
static void echo_c_read(void* p, grpc_error_handle e) {
  auto* c = static_cast<echo_conn*>(p);
  if (e != GRPC_ERROR_NONE) { c->failed = true; return; }
  c->c_got += count_slices(c->c_in.slices, c->c_in.count, &c->pattern_r);
  if (c->c_got >= c->msg_len) {
    CHECK(c->c_got == c->msg_len);
    if (--c->rounds_left == 0) {
      c->done = true;
      g_conns_done++;
      return;
    }
    bool send;
    {
      std::lock_guard<std::mutex> lk(c->mu);
      send = !c->c_writing;
      if (!send) c->send_when_written = true;
    }
    if (send) echo_client_send(c);
  }
  grpc_endpoint_read(c->client, &c->c_in, &c->c_read, false);
}
static void echo_s_wrote(void* p, grpc_error_handle e) {
  auto* c = static_cast<echo_conn*>(p);
  c->s_writing = false;
  if (e != GRPC_ERROR_NONE) { c->failed = true; return; }
  grpc_endpoint_read(c->server, &c->s_in, &c->s_read, false);
}
static void echo_s_read(void* p, grpc_error_handle e) {
  auto* c = static_cast<echo_conn*>(p);
  if (e != GRPC_ERROR_NONE) { c->failed = true; return; }
  // echo: the read slices become the write buffer (one write outstanding, then read again)
  grpc_slice_buffer_swap(&c->s_in, &c->s_out);
  c->s_writing = true;
  grpc_endpoint_write(c->server, &c->s_out, &c->s_wrote, nullptr);
}

static void pollset_echo_test(size_t n_conns, size_t rounds, bool bpev, int threads) {
  setenv("GRPC_PLATFORM_TYPE", bpev ? "RDMA_BPEV" : "RDMA_BP", 1);
  grpc_pollset* ps = grdma_pollset_create(bpev, /*busy_polling_timeout_us=*/200);
  CHECK(ps != nullptr);
  std::vector<echo_conn*> conns;
  for (size_t i = 0; i < n_conns; i++) {
    auto* c = new echo_conn();
    c->client = grpc_endpoint_create(100 + 2 * (int)i, "ipv4:127.0.0.1:1", false);
    c->server = grpc_endpoint_create(101 + 2 * (int)i, "ipv4:127.0.0.1:2", true);
    CHECK(c->client && c->server && grpc_rdma_bp_connect_loopback(c->client, c->server));
    grpc_slice_buffer_init(&c->c_out); grpc_slice_buffer_init(&c->c_in);
    grpc_slice_buffer_init(&c->s_out); grpc_slice_buffer_init(&c->s_in);
    GRPC_CLOSURE_INIT(&c->c_wrote, echo_c_wrote, c, nullptr);
    GRPC_CLOSURE_INIT(&c->c_read, echo_c_read, c, nullptr);
    GRPC_CLOSURE_INIT(&c->s_wrote, echo_s_wrote, c, nullptr);
    GRPC_CLOSURE_INIT(&c->s_read, echo_s_read, c, nullptr);
    c->rounds_left = rounds;
    c->seed = 12345u + 977u * (uint32_t)i;
    c->pattern_w = c->pattern_r = 0;
    c->client->vtable->add_to_pollset(c->client, ps);
    c->server->vtable->add_to_pollset(c->server, ps);
    conns.push_back(c);
  }
  CHECK(grdma_pollset_size(ps) == 2 * n_conns);
  g_conns_done = 0;
  for (auto* c : conns) {
    grpc_endpoint_read(c->server, &c->s_in, &c->s_read, false);  // first read: arms notify_on_read
    grpc_endpoint_read(c->client, &c->c_in, &c->c_read, false);
  }
  size_t first = 0;
  if (bpev) {
    // nothing to do yet: the pass burns its busy-polling budget, then sleeps in epoll_wait
    CHECK(grdma_pollset_work(ps, 30) == 0);
    grdma_pollset_stats st0;
    grdma_pollset_get_stats(ps, &st0);
    CHECK(st0.epoll_waits == 1);
    // connection 0 starts from ANOTHER thread while this one sleeps in epoll_wait: the background
    // poller sees the message and signals the pair's wakeup fd (poller.cc:84-100), the pollset
    // wakes up, consumes the wakeup and delivers the read
    std::thread late([&]() {
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
      echo_client_send(conns[0]);
    });
    const auto t0 = std::chrono::steady_clock::now();
    int ran = 0;
    while (ran == 0 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) ran = grdma_pollset_work(ps, 2000);
    late.join();
    CHECK(ran > 0);
    grdma_pollset_get_stats(ps, &st0);
    CHECK(st0.wakeups_consumed >= 1);
    first = 1;
  }
  for (size_t i = first; i < conns.size(); i++) echo_client_send(conns[i]);
  // `threads` workers inside pollset_work at the same time (the reference serialises their passes over the fds with
  // p->rdma_mu and runs the closures outside, ev_epollex_rdma_bpev_linux.cc:1103-1145)
  auto worker = [&]() {
    long idle = 0;
    while (g_conns_done < n_conns) {
      const int ran = grdma_pollset_work(ps, /*timeout_ms=*/20);
      CHECK(ran >= 0);
      for (auto* c : conns) CHECK(!c->failed);
      if (ran) idle = 0;
      else if (++idle > 3000) CHECK(!"pollset made no progress");
    }
  };
  std::vector<std::thread> extra;
  for (int t = 1; t < threads; t++) extra.emplace_back(worker);
  worker();
  for (std::thread& t : extra) t.join();
  grdma_pollset_stats st;
  grdma_pollset_get_stats(ps, &st);
  if (bpev) CHECK(st.epoll_waits > 0 && st.wakeups_consumed > 0);
  for (auto* c : conns) {
    grpc_endpoint_shutdown(c->client, GRPC_ERROR_CREATE_FROM_STATIC_STRING("test done"));
    grpc_endpoint_shutdown(c->server, GRPC_ERROR_CREATE_FROM_STATIC_STRING("test done"));
    grpc_endpoint_destroy(c->client);
    grpc_endpoint_destroy(c->server);
    grpc_slice_buffer_destroy(&c->c_out); grpc_slice_buffer_destroy(&c->c_in);
    grpc_slice_buffer_destroy(&c->s_out); grpc_slice_buffer_destroy(&c->s_in);
    delete c;
  }
  CHECK(grdma_pollset_size(ps) == 0);  // grpc_fd_orphan took every fd out of the set
  grdma_pollset_destroy(ps);
  printf("pollset_echo_test conns=%zu rounds=%zu bpev=%d threads=%d: ok (passes %llu, device polls %llu, epoll waits %llu, "
         "wakeups %llu, closures %llu)\n",
         n_conns, rounds, (int)bpev, threads, (unsigned long long)st.passes, (unsigned long long)st.device_polls,
         (unsigned long long)st.epoll_waits, (unsigned long long)st.wakeups_consumed,
         (unsigned long long)st.closures_run);
}

// GRDMA_PROFILE=1: the run records into profiler slot 0 and prints the reference's table at exit
// (grpc_stats_time_init / _enable / _print, the way examples/cpp/helloworld.benchmark does)
static void print_profile() {
  static char buf[16384];
  grdma_stats_time_print(buf, sizeof(buf));
  fputs(buf, stdout);
}

int main(int argc, char** argv) {
  if (getenv("GRDMA_PROFILE") && getenv("GRDMA_PROFILE")[0] == '1') {
    grdma_stats_time_init(0);
    grdma_stats_time_enable();
    atexit(print_profile);
  }
  if (argc >= 5 && !strcmp(argv[1], "pollset")) {
    pollset_echo_test((size_t)atol(argv[2]), (size_t)atol(argv[3]), atoi(argv[4]) != 0, argc >= 6 ? atoi(argv[5]) : 1);
    return 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "multiple_shutdown")) {
    multiple_shutdown_test();
    half_close_test();
    write_after_peer_exit_test();
    buffered_write_after_peer_exit_test();
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "sweep")) {  // endpoint_tests.cc:350-352
    for (size_t i = (size_t)atol(argv[2]); i < (size_t)atol(argv[3]); i = (i < 5 ? i + 1 : i * 5 / 4))
      read_and_write_test(40320, i, i, false);
    return 0;
  }
  if (argc >= 5) {
    read_and_write_test((size_t)atol(argv[1]), (size_t)atol(argv[2]), (size_t)atol(argv[3]), atoi(argv[4]) != 0);
    return 0;
  }
  fprintf(stderr, "usage: see the file header\n");
  return 64;
}
