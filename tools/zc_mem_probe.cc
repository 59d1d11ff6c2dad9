// Where should the zero-copy send buffer live?  (csrc/grdma_host_poller.inc: GRDMA_ZC_MEM_*)
// The reference's caller serialises a message into the buffer with the CPU and the HCA reads it by DMA.  Here the reader
// is the gather kernel, and the buffer can be pinned host memory (the gather crosses PCIe) or fine-grained device memory
// written by the CPU through the PCIe BAR (the serialisation crosses PCIe).  For each kind, in a child process (a BAR
// that is not mapped for the host ends the child, not the probe):
//   fill     GiB/s of the CPU copying 1 MiB messages into the buffer (what protobuf's serialisation is bounded by)
//   send     GiB/s of AllocateSendBuffer -> memcpy -> SendZerocopy([14-byte header in host memory][the message]) ->
//            the peer's endpoint read, one message at a time over a loop-back link with a 64 MiB ring, bytes checked
// and beside them the same messages through grdma_pair_send from pageable host memory (copied into the pinned bounce
// buffer by the library, then gathered): what the zero-copy buffer saves.
//   build: g++ -O2 -std=c++17 -Iinclude tools/zc_mem_probe.cc -o tools/zc_mem_probe -Lgrpc-rdma_amd -lgrdma_amd -Wl,-rpath,$PWD/grpc-rdma_amd
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "grdma_amd.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int run_kind(int kind, uint64_t msg, int iters) {
  if (grdma_init(0) < 0) { printf("{\"error\": \"%s\"}\n", grdma_last_error()); return 2; }
  const uint64_t R = 64ull << 20;
  grdma_pair* a = grdma_pair_create(R, 30, GRDMA_WIRE_DIRECT);
  grdma_pair* b = grdma_pair_create(R, 30, GRDMA_WIRE_DIRECT);
  if (!a || !b || grdma_pair_connect(a, b) < 0) { printf("{\"error\": \"%s\"}\n", grdma_last_error()); return 2; }
  std::vector<uint8_t> src(msg), hdr(14, 7);
  for (uint64_t i = 0; i < msg; i++) src[i] = (uint8_t)(i * 131 + (i >> 9));
  std::vector<grdma_read_slice> rs(64);
  std::vector<uint8_t> out(msg + 64);
  double fill = 0, send = 0;
  if (kind >= 0) {
    if (grdma_pair_enable_zerocopy_ex(a, 4 * msg, kind) < 0) { printf("{\"error\": \"%s\"}\n", grdma_last_error()); return 2; }
    if (kind != GRDMA_ZC_MEM_DEVICE) {
      uint8_t* p = (uint8_t*)grdma_pair_allocate_send_buffer(a, msg);
      if (!p) return 2;
      memcpy(p, src.data(), msg);  // (first touch: a BAR the host cannot write ends the process here)
      const double t0 = now_s();
      for (int i = 0; i < iters; i++) memcpy(p, src.data(), msg);
      fill = (double)msg * iters / (now_s() - t0) / (1 << 30);
      grdma_slice one{p, msg};
      if (grdma_pair_send_zerocopy(a, &one, 1, 0, GRDMA_MEM_HOST) != (int64_t)msg) { printf("{\"error\": \"%s\"}\n", grdma_last_error()); return 2; }
      int wb = 0;
      while (grdma_endpoint_read(b, 64, rs.data(), 64, &wb) > 0) {}
    }
  }
  uint64_t bad = 0;
  std::vector<uint8_t> tail;
  const double t0 = now_s();
  for (int i = 0; i < iters; i++) {
    src[0] = (uint8_t)i;
    int64_t sent;
    if (kind >= 0 && kind != GRDMA_ZC_MEM_DEVICE) {
      uint8_t* p = (uint8_t*)grdma_pair_allocate_send_buffer(a, msg);
      if (!p) { printf("{\"error\": \"no buffer at message %d\"}\n", i); return 2; }
      memcpy(p, src.data(), msg);
      grdma_slice sl[2] = {{hdr.data(), 14}, {p, msg}};
      sent = grdma_pair_send_zerocopy(a, sl, 2, 0, GRDMA_MEM_HOST);
    } else {
      grdma_slice sl[2] = {{hdr.data(), 14}, {src.data(), msg}};
      sent = grdma_pair_send(a, sl, 2, 0, GRDMA_MEM_HOST);
    }
    if (sent != (int64_t)(msg + 14)) { printf("{\"error\": \"sent %lld: %s\"}\n", (long long)sent, grdma_last_error()); return 2; }
    int wb = 0;
    uint64_t got = 0;
    for (;;) {
      const int64_t n = grdma_endpoint_read(b, 64, rs.data(), 64, &wb);
      if (n <= 0) break;
      for (int64_t k = 0; k < n; k++) got += rs[k].len;
      if (i == iters - 1)  // (the last message: bytes checked; the endpoint reads cut the stream where they like)
        for (int64_t k = 0; k < n; k++) {
          const size_t at = tail.size();
          tail.resize(at + rs[k].len);
          grdma_pair_arena_copy_out(b, rs[k].off, tail.data() + at, rs[k].len);
        }
    }
    if (got != msg + 14) bad++;
    if (i == iters - 1 && (tail.size() != msg + 14 || memcmp(tail.data(), hdr.data(), 14) != 0 || memcmp(tail.data() + 14, src.data(), msg) != 0)) bad++;
  }
  send = (double)msg * iters / (now_s() - t0) / (1 << 30);
  printf("{\"kind\": \"%s\", \"msg\": %llu, \"iters\": %d, \"fill_GiBps\": %.2f, \"send_GiBps\": %.2f, \"checked\": %s}\n",
         kind < 0 ? "no zero-copy buffer (grdma_pair_send, bounce copy)" : kind == 0 ? "host (pinned, mapped)" : kind == 1 ? "bar (fine-grained device memory)" : "device",
         (unsigned long long)msg, iters, fill, send, bad ? "false" : "true");
  grdma_pair_destroy(a);
  grdma_pair_destroy(b);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  const uint64_t msg = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 20);
  const int iters = argc > 2 ? atoi(argv[2]) : 400;
  for (int kind : {-1, 0, 1}) {
    fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {
      const int rc = run_kind(kind, msg, iters);
      fflush(stdout);
      _exit(rc);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) printf("{\"kind\": %d, \"error\": \"the child was ended by signal %d: this memory is not host-writable here\"}\n", kind, WTERMSIG(st));
    else if (WEXITSTATUS(st) != 0) printf("{\"kind\": %d, \"rc\": %d}\n", kind, WEXITSTATUS(st));
  }
  return 0;
}
