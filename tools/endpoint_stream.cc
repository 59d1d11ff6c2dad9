// Client-streaming throughput THROUGH THE ENDPOINT VTABLE (grpc_endpoint_write / grpc_endpoint_read,
// include/grdma_endpoint.hpp), the way chttp2 drives rdma_bp_posix.cc: every message of
// <payload> bytes is one grpc_endpoint_write of HTTP/2 DATA frames -- a 9-byte frame header slice
// (inlined) in front of every <= 16384-byte payload slice (refcounted, host memory), the first
// payload prefixed by the 5-byte gRPC message header -- and the reader re-arms grpc_endpoint_read
// from its callback.  Host slices in, host slices out: this is the PCIe-inclusive rate of the
// boundary, reported next to the device-resident figure of bench.py (never instead of it).
//
// usage: endpoint_stream <n_msgs> <payload_bytes> [check 0|1] [latency 0|1] [threads 1|2]     prints one JSON line
//        latency 1: both pairs in latency mode, commands through the resident engine (no kernel launch per
//        write / read)
//        threads 2 (default): the writing side and the reading side each run their own poll loop on a thread of
//        their own, the way a gRPC process has the two directions of a connection on different threads
//        (pair.h:64-81: one writer, one reader per pair); 1: one loop polls both endpoints
// env:   GRPC_RDMA_RING_BUFFER_SIZE_KB, GRPC_RDMA_MAX_SGE ... as the reference reads them
//        ENDPOINT_STREAM_SEED=<s != 0>: randomised -- every write takes a random number of the message's DATA frames
//        (2 .. all of its slices: writes below and above max_sge, queued chains of every length), and the reader stalls
//        for a random 50 - 1500 us after one read in four, so the ring fills and drains at random moments (queued chains
//        promoted AND skipped in one run); bytes and byte sum of what was delivered still equal what was written
#include <emmintrin.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
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

static thread_local std::deque<grpc_closure*> g_queue;  // a miniature ExecCtx (one per polling thread)

// Sum of the bytes of a slice (the check of the delivered stream; independent of where the slices are cut).  The check
// sits inside the timed loop, on the reading thread, and should not be what is measured: psadbw sums sixteen bytes per
// instruction (SSE2, the x86-64 baseline), four accumulators -- ~18 GB/s from memory where the word-wise loop it
// replaces did 12, i.e. 57 instead of 87 us of the reader's time per MiB.
__attribute__((target("avx2"))) static uint64_t sum_bytes_avx2(const uint8_t* b, size_t n) {
  // (32 bytes per psadbw where the CPU has AVX2 -- every x86-64 server since 2015: the reader's check is then bound by
  // what one core streams out of memory, not by its instruction count)
  __m256i a0 = _mm256_setzero_si256(), a1 = a0, a2 = a0, a3 = a0;
  const __m256i z = _mm256_setzero_si256();
  size_t k = 0;
  for (; k + 128 <= n; k += 128) {
    _mm_prefetch(reinterpret_cast<const char*>(b + k + 1024), _MM_HINT_T0);
    _mm_prefetch(reinterpret_cast<const char*>(b + k + 1088), _MM_HINT_T0);
    a0 = _mm256_add_epi64(a0, _mm256_sad_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + k)), z));
    a1 = _mm256_add_epi64(a1, _mm256_sad_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + k + 32)), z));
    a2 = _mm256_add_epi64(a2, _mm256_sad_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + k + 64)), z));
    a3 = _mm256_add_epi64(a3, _mm256_sad_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(b + k + 96)), z));
  }
  a0 = _mm256_add_epi64(_mm256_add_epi64(a0, a1), _mm256_add_epi64(a2, a3));
  uint64_t lanes[4];
  _mm256_storeu_si256(reinterpret_cast<__m256i*>(lanes), a0);
  uint64_t total = lanes[0] + lanes[1] + lanes[2] + lanes[3];
  for (; k < n; k++) total += b[k];
  return total;
}
static uint64_t sum_bytes(const uint8_t* b, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("ENDPOINT_STREAM_SSE2");
  if (avx2 && n >= 256) return sum_bytes_avx2(b, n);
  __m128i a0 = _mm_setzero_si128(), a1 = _mm_setzero_si128(), a2 = _mm_setzero_si128(), a3 = _mm_setzero_si128();
  const __m128i z = _mm_setzero_si128();
  size_t k = 0;
  for (; k + 64 <= n; k += 64) {
    a0 = _mm_add_epi64(a0, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + k)), z));
    a1 = _mm_add_epi64(a1, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + k + 16)), z));
    a2 = _mm_add_epi64(a2, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + k + 32)), z));
    a3 = _mm_add_epi64(a3, _mm_sad_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(b + k + 48)), z));
  }
  a0 = _mm_add_epi64(_mm_add_epi64(a0, a1), _mm_add_epi64(a2, a3));
  uint64_t total = (uint64_t)_mm_cvtsi128_si64(a0) + (uint64_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(a0, a0));
  for (; k < n; k++) total += b[k];
  return total;
}

struct state {
  grpc_endpoint *tx, *rx;
  grpc_slice_buffer outgoing, incoming;
  grpc_closure done_write, done_read, next_write, next_read;
  std::vector<grpc_slice> frames;  // the slices of one message, re-referenced for every write
  size_t msgs_target, msgs_written, bytes_per_msg, bytes_read, bytes_target;
  bool check, failed;
  volatile bool write_done, read_done;
  uint64_t sum_read, sum_per_msg;
  // randomised mode (ENDPOINT_STREAM_SEED): frame pairs of write k, xorshift state of the reading side
  std::vector<uint32_t> pairs_of_write;
  uint64_t sum_target = 0, rng_read = 0;
};
static inline uint64_t xorshift(uint64_t* s) {
  uint64_t x = *s;
  x ^= x << 13; x ^= x >> 7; x ^= x << 17;
  return *s = x;
}

static void do_write(void* p, grpc_error_handle) {
  auto* st = static_cast<state*>(p);
  const size_t take = st->pairs_of_write.empty() ? st->frames.size() : 2 * (size_t)st->pairs_of_write[st->msgs_written];
  for (size_t i = 0; i < take; i++) {
    grpc_slice& s = st->frames[i];
    if (s.refcount) s.refcount->refs.fetch_add(1);
    grpc_slice_buffer_add_indexed(&st->outgoing, s);  // (indexed: no merging, like chttp2's frame slices)
  }
  grpc_endpoint_write(st->tx, &st->outgoing, &st->done_write, nullptr);
}
static void on_write(void* p, grpc_error_handle e) {
  auto* st = static_cast<state*>(p);
  if (e != GRPC_ERROR_NONE) { st->failed = st->write_done = true; return; }
  grpc_slice_buffer_reset_and_unref(&st->outgoing);  // (chttp2 resets its outbuf in write_action_end)
  if (++st->msgs_written == st->msgs_target) { st->write_done = true; return; }
  g_queue.push_back(&st->next_write);
}
static void do_read(void* p, grpc_error_handle) {
  auto* st = static_cast<state*>(p);
  grpc_endpoint_read(st->rx, &st->incoming, &st->done_read, false);
}
static void on_read(void* p, grpc_error_handle e) {
  auto* st = static_cast<state*>(p);
  if (e != GRPC_ERROR_NONE) { st->failed = st->read_done = true; return; }
  for (size_t i = 0; i < st->incoming.count; i++) {
    const grpc_slice& s = st->incoming.slices[i];
    const size_t n = GRPC_SLICE_LENGTH(s);
    if (st->check) {
      st->sum_read += sum_bytes(GRPC_SLICE_START_PTR(s), n);
    }
    st->bytes_read += n;
  }
  if (st->bytes_read >= st->bytes_target) { st->read_done = true; return; }
  if (st->rng_read && (xorshift(&st->rng_read) & 3) == 0)  // (randomised: the application is slow to read now and then)
    std::this_thread::sleep_for(std::chrono::microseconds(50 + xorshift(&st->rng_read) % 1450));
  g_queue.push_back(&st->next_read);
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <n_msgs> <payload_bytes> [check]\n", argv[0]);
    return 1;
  }
  state st{};
  st.msgs_target = strtoull(argv[1], nullptr, 10);
  const size_t payload = strtoull(argv[2], nullptr, 10);
  st.check = argc > 3 && atoi(argv[3]) != 0;
  setenv("GRPC_PLATFORM_TYPE", "RDMA_BP", 0);
  // Run on the CPUs of the GPU's NUMA node (what `numactl --cpunodebind` does for a deployed process): threads and
  // pinned buffers then sit next to the PCIe root the GPU hangs on -- 15.6 against 10.4 GiB/s on a two-socket box.
  const int numa_node = getenv("GRDMA_NO_NUMA_PIN") ? -1 : grdma_host_pin_to_device_node();
  st.tx = grpc_endpoint_create(3, "ipv4:127.0.0.1:1", false);
  st.rx = grpc_endpoint_create(4, "ipv4:127.0.0.1:2", true);
  CHECK(st.tx && st.rx && grpc_rdma_bp_connect_loopback(st.tx, st.rx));
  const bool latency = argc > 4 && atoi(argv[4]) != 0;
  if (latency) {
    CHECK(grdma_endpoint_set_latency_mode(st.tx, true, 0));
    CHECK(grdma_endpoint_set_latency_mode(st.rx, true, 0));
    CHECK(grdma_engine_start() == 0);
  }

  // one serialized message: [5-byte gRPC header][0x0a varint(len)][payload], cut into DATA frames
  std::vector<uint8_t> msg;
  {
    std::vector<uint8_t> body;
    body.push_back(0x0a);
    for (size_t n = payload;;) {
      uint8_t b = n & 0x7f;
      n >>= 7;
      body.push_back(b | (n ? 0x80 : 0));
      if (!n) break;
    }
    for (size_t i = 0; i < payload; i++) body.push_back((uint8_t)((i * 7 + 3) % 251));
    const uint32_t L = (uint32_t)body.size();
    msg = {0, (uint8_t)(L >> 24), (uint8_t)(L >> 16), (uint8_t)(L >> 8), (uint8_t)L};
    msg.insert(msg.end(), body.begin(), body.end());
  }
  for (size_t off = 0; off < msg.size();) {
    const size_t n = msg.size() - off < 16384 ? msg.size() - off : 16384;
    const uint8_t hdr[9] = {(uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, 0, 0, 0, 0, 0, 1};
    st.frames.push_back(grpc_slice_from_copied_buffer(reinterpret_cast<const char*>(hdr), 9));  // inlined (< 24 B)
    grpc_slice s = grpc_slice_malloc(n);
    memcpy(GRPC_SLICE_START_PTR(s), msg.data() + off, n);
    st.frames.push_back(s);
    off += n;
  }
  st.bytes_per_msg = 0;
  for (grpc_slice& s : st.frames) {
    st.bytes_per_msg += GRPC_SLICE_LENGTH(s);
    st.sum_per_msg += sum_bytes(GRPC_SLICE_START_PTR(s), GRPC_SLICE_LENGTH(s));
  }
  st.bytes_target = st.bytes_per_msg * st.msgs_target;
  st.sum_target = st.sum_per_msg * st.msgs_target;
  if (const char* e = getenv("ENDPOINT_STREAM_SEED")) {
    uint64_t seed = strtoull(e, nullptr, 10);
    if (seed) {
      uint64_t rs = seed * 0x9E3779B97F4A7C15ull + 1;
      st.rng_read = seed * 0xD1B54A32D192ED03ull + 7;
      std::vector<uint64_t> pre_b(1, 0), pre_s(1, 0);  // bytes / byte sum of the first k frame pairs
      for (size_t i = 0; i + 1 < st.frames.size(); i += 2) {
        uint64_t b = 0, sm = 0;
        for (size_t q = i; q < i + 2; q++) {
          b += GRPC_SLICE_LENGTH(st.frames[q]);
          sm += sum_bytes(GRPC_SLICE_START_PTR(st.frames[q]), GRPC_SLICE_LENGTH(st.frames[q]));
        }
        pre_b.push_back(pre_b.back() + b);
        pre_s.push_back(pre_s.back() + sm);
      }
      const size_t npairs = st.frames.size() / 2;
      st.bytes_target = st.sum_target = 0;
      for (size_t k = 0; k < st.msgs_target; k++) {
        const uint32_t pairs = 1 + (uint32_t)(xorshift(&rs) % npairs);
        st.pairs_of_write.push_back(pairs);
        st.bytes_target += pre_b[pairs];
        st.sum_target += pre_s[pairs];
      }
    }
  }
  grpc_slice_buffer_init(&st.outgoing);
  grpc_slice_buffer_init(&st.incoming);
  GRPC_CLOSURE_INIT(&st.done_write, on_write, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.done_read, on_read, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.next_write, do_write, &st, nullptr);
  GRPC_CLOSURE_INIT(&st.next_read, do_read, &st, nullptr);

  const int threads = argc > 5 ? atoi(argv[5]) : 2;
  // the pollset_work loop of one thread: polls `ep` (and `ep2`, if any) until *flag (and *flag2)
  // ENDPOINT_STREAM_PROFILE=1: where each polling thread's time goes -- inside the closures it runs (the writer's:
  // grpc_endpoint_write with its copy into the send buffer; the reader's: the byte sum and grpc_endpoint_read), inside
  // polls that made progress, inside polls that found nothing (waiting for the device)
  const bool profile = getenv("ENDPOINT_STREAM_PROFILE") && atoi(getenv("ENDPOINT_STREAM_PROFILE")) != 0;
  struct prof_t { double closures = 0, poll_busy = 0, poll_idle = 0; uint64_t n_closures = 0, n_busy = 0, n_idle = 0; };
  static prof_t prof[2];
  auto loop = [&](grpc_endpoint* ep, grpc_endpoint* ep2, volatile bool* flag, volatile bool* flag2, int who = 0) {
    const auto tl = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    prof_t& pr = prof[who];
    while (!*flag || (flag2 && !*flag2)) {
      std::chrono::steady_clock::time_point a, b;
      if (profile) a = std::chrono::steady_clock::now();
      int ran = grdma_endpoint_poll(ep) + (ep2 ? grdma_endpoint_poll(ep2) : 0);
      if (profile) {
        b = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double>(b - a).count();
        if (ran) { pr.poll_busy += d; pr.n_busy++; } else { pr.poll_idle += d; pr.n_idle++; }
      }
      while (!g_queue.empty()) {
        grpc_closure* c = g_queue.front();
        g_queue.pop_front();
        if (profile) a = std::chrono::steady_clock::now();
        c->cb(c->cb_arg, GRPC_ERROR_NONE);
        if (profile) { pr.closures += std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); pr.n_closures++; }
        ran++;
      }
      if (!ran && (++spins & 0xFFFFF) == 0)
        CHECK(std::chrono::steady_clock::now() - tl < std::chrono::seconds(100) && "endpoint made no progress");
    }
  };
  // Writer and reader each on a physical core of their own, both under ONE L3 (neighbouring cores of the device's node;
  // `taskset -c` per thread): measured on the 2 x 64-core box of the GPU pool, 26 runs of 26 at 14.5-15.3 GiB/s this way,
  // against 15.4-15.6 in two runs of three and 11.5 in the third when the two threads float over the node, and the slow
  // mode again (4 of 16) with the threads pinned to cores of different L3s -- the two threads share the library's state
  // lines and locks.  GRDMA_PIN_CORES="w,r" = indices into the node's physical cores, counted from the end of its list;
  // "none" lets the threads float.
  int core_w = 1, core_r = 0;
  if (const char* e = getenv("GRDMA_PIN_CORES")) {
    if (sscanf(e, "%d,%d", &core_w, &core_r) != 2) core_w = core_r = -1;
  }
  const bool pin_cores = numa_node >= 0 && core_w >= 0 && core_r >= 0;
  std::atomic<int> reader_ready{0};
  std::atomic<bool> go{false};
  auto t0 = std::chrono::steady_clock::now();
  if (threads >= 2) {
    if (pin_cores) grdma_host_pin_thread_to_core(core_w);
    std::thread reader([&] {
      if (pin_cores) grdma_host_pin_thread_to_core(core_r);
      reader_ready.store(1, std::memory_order_release);  // (the sysfs walk of the pin stays outside the timed region)
      while (!go.load(std::memory_order_acquire)) {}
      do_read(&st, GRPC_ERROR_NONE);
      loop(st.rx, nullptr, &st.read_done, nullptr, 1);
    });
    while (!reader_ready.load(std::memory_order_acquire)) {}
    t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    do_write(&st, GRPC_ERROR_NONE);
    // (a write completes when its bytes sit in the endpoint's send buffer: like a gRPC poller, this thread keeps
    // polling the sending endpoint until the stream has arrived, not just until the last write callback)
    loop(st.tx, nullptr, &st.write_done, &st.read_done);
    reader.join();
  } else {
    do_read(&st, GRPC_ERROR_NONE);
    do_write(&st, GRPC_ERROR_NONE);
    loop(st.rx, st.tx, &st.read_done, &st.write_done);
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CHECK(!st.failed && st.bytes_read == st.bytes_target);
  if (st.check) CHECK(st.sum_read == st.sum_target);
  uint64_t wq[3] = {0, 0, 0};  // writes queued behind the Send in flight / promoted / skipped on the device
  grdma_endpoint_write_queue_stats(grdma_endpoint_pair(st.tx), wq);
  printf("{\"msgs\": %zu, \"payload\": %zu, \"slices_per_write\": %zu, \"endpoint_bytes\": %zu, \"seconds\": %.6f, "
         "\"GiBps\": %.4f, \"checked\": %s, \"latency_mode\": %s, \"threads\": %d, \"ring_kib\": %s, \"max_sge\": %s, "
         "\"wire\": \"%s\", \"register_min\": %s, \"writes_queued\": [%llu, %llu, %llu], \"numa_pinned\": %s, "
         "\"thread_cores\": [%d, %d]}\n",
         st.msgs_target, payload, st.frames.size(), st.bytes_target, sec,
         (double)(payload * st.msgs_target) / sec / (double)(1ull << 30), st.check ? "true" : "false",
         latency ? "true" : "false", threads >= 2 ? 2 : 1,
         getenv("GRPC_RDMA_RING_BUFFER_SIZE_KB") ? getenv("GRPC_RDMA_RING_BUFFER_SIZE_KB") : "4096",
         getenv("GRPC_RDMA_MAX_SGE") ? getenv("GRPC_RDMA_MAX_SGE") : "30",
         getenv("GRPC_RDMA_HIP_WIRE") ? getenv("GRPC_RDMA_HIP_WIRE") : "direct",
         getenv("GRPC_RDMA_HIP_REGISTER_MIN") ? getenv("GRPC_RDMA_HIP_REGISTER_MIN") : "0",
         (unsigned long long)wq[0], (unsigned long long)wq[1], (unsigned long long)wq[2],
         numa_node >= 0 ? "true" : "false", pin_cores && threads >= 2 ? core_w : -1, pin_cores && threads >= 2 ? core_r : -1);
  if (profile)
    for (int w = 0; w < 2; w++)
      fprintf(stderr, "profile %s: closures %.1f ms (%llu), polls that ran something %.1f ms (%llu), idle polls %.1f ms (%llu) of %.1f ms\n",
              w == 0 ? "writer" : "reader", prof[w].closures * 1e3, (unsigned long long)prof[w].n_closures, prof[w].poll_busy * 1e3,
              (unsigned long long)prof[w].n_busy, prof[w].poll_idle * 1e3, (unsigned long long)prof[w].n_idle, sec * 1e3);
  if (profile) {
    uint64_t fd[6] = {0, 0, 0, 0, 0, 0};
    grdma_rx_fast_drains(fd);
    fprintf(stderr, "profile drains: %llu predicted by the steady-state bodies; declined: %llu state, %llu pattern / count, %llu header, "
                    "%llu small-record run, %llu room\n", (unsigned long long)fd[0], (unsigned long long)fd[1], (unsigned long long)fd[2],
            (unsigned long long)fd[3], (unsigned long long)fd[4], (unsigned long long)fd[5]);
  }
  if (latency) grdma_engine_stop();
  grpc_endpoint_shutdown(st.tx, GRPC_ERROR_CREATE_FROM_STATIC_STRING("done"));
  grpc_endpoint_shutdown(st.rx, GRPC_ERROR_CREATE_FROM_STATIC_STRING("done"));
  grpc_endpoint_destroy(st.tx);
  grpc_endpoint_destroy(st.rx);
  for (grpc_slice& s : st.frames) grpc_slice_unref(s);
  grpc_slice_buffer_destroy(&st.outgoing);
  grpc_slice_buffer_destroy(&st.incoming);
  return 0;
}
