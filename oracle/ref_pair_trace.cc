// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// ref_pair_trace: the reference's OWN src/core/lib/ibverbs/pair.cc (PairPollable::Send / Recv / GetWritableSize / the
// credit rule / partial_write_), compiled unmodified together with ring_buffer.cc, device.cc, memory_region.cc,
// buffer.cc, address.cc and config.cc over the software verbs of oracle/fakeverbs, replays a list of operations on two
// pairs connected to each other in this process and prints what every operation returned.  tests/test_oracle_vs_ref.py
// feeds the same list to the plain-C oracle (oracle/grdma_oracle.c) and compares line by line: the oracle -- and through
// it the HIP path -- is pinned to pair.cc itself, not to a transcription of it.
//
//   usage: ref_pair_trace < ops        (ring size: GRPC_RDMA_RING_BUFFER_SIZE_KB, max_sge: FAKEVERBS_MAX_SGE)
//   ops:   S <side> <byte_idx> <seed> <n> <len_1> ... <len_n>     Send(slices, n, byte_idx); slice i byte j = pat(seed,i,j)
//          R <side> <cap>                                         Recv(buf, cap)
//          A <side> <size>                                        AllocateSendBuffer(size)
//          W <side> <off> <seed> <len>                            fill the zero-copy buffer at off
//          Z <side> <byte_idx> <seed> <n> {<len> <zc offset | -1>} x n    SendZerocopy
//          Q                                                      state of both sides
//   out:   S <sent> <writable> <partial_write> <pending_writes>
//          R <n> <crc32 of the bytes> <readable> <has_message> <writable of the peer>
//          A <offset | -1>      Z <sent> <writable> <partial_write> <zerocopy_buffer_tail>
//          Q <side> <remote_tail> <internal_read_size> <remote_head seen> <get_head() = moving_head_> <crc32 of the ring memory>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <grpc/support/log.h>

// This translation unit reads state PairPollable keeps private (remote_tail_, partial_write_, the receive buffer).  Every
// header pair.h includes is included first, so the two defines below touch pair.h's own text only; neither changes a
// layout or a mangled name.
#include <array>
#include <atomic>
#include <memory>
#include <queue>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unistd.h>
#include <grpc/slice.h>
#include "src/core/lib/ibverbs/address.h"
#include "src/core/lib/ibverbs/buffer.h"
#include "src/core/lib/ibverbs/device.h"
#include "src/core/lib/ibverbs/memory_region.h"
#include "src/core/lib/ibverbs/ring_buffer.h"
#include "src/core/lib/iomgr/wakeup_fd_posix.h"
#define private public
#define class struct
#include "src/core/lib/ibverbs/pair.h"
#undef class
#undef private
#include "grpcpp/stats_time.h"

// ---- the few symbols of the gRPC core the seven files reference ---------------------------------------------------
extern "C" void gpr_log(const char*, int, gpr_log_severity, const char*, ...) {}
extern "C" int gpr_should_log(gpr_log_severity) { return 0; }
extern "C" void gpr_free(void* p) { free(p); }
char* gpr_getenv(const char* name) {
  const char* v = getenv(name);
  return v ? strdup(v) : nullptr;
}
grpc_error_handle grpc_wakeup_fd_init(grpc_wakeup_fd* fd) {
  fd->read_fd = fd->write_fd = -1;
  return GRPC_ERROR_NONE;
}
void grpc_wakeup_fd_destroy(grpc_wakeup_fd*) {}
GRPCProfiler::GRPCProfiler(grpc_stats_time op) : op_(op) {}
GRPCProfiler::~GRPCProfiler() {}

namespace {
using grpc_core::ibverbs::PairPollable;

// CRC-32 (IEEE, reflected: what zlib.crc32 computes on the Python side)
uint64_t fnv(const uint8_t* p, uint64_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (uint64_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
inline uint8_t pat(uint64_t seed, uint64_t i, uint64_t j) { return (uint8_t)(seed * 131 + i * 17 + j * 7 + (j >> 8)); }
grpc_slice_refcount* const kFakeRefcount = reinterpret_cast<grpc_slice_refcount*>(uintptr_t{0x10});
}  // namespace

int main() {
  PairPollable* p[2] = {new PairPollable(), new PairPollable()};
  p[0]->Init();
  p[1]->Init();
  const std::vector<char> a0 = p[0]->get_self_address().bytes(), a1 = p[1]->get_self_address().bytes();
  bool ok[2] = {false, false};
  std::thread t([&] { ok[1] = p[1]->Connect(a0); });  // (each side waits for the other's memory regions)
  ok[0] = p[0]->Connect(a1);
  t.join();
  if (!ok[0] || !ok[1]) {
    fprintf(stderr, "connect failed: %s / %s\n", p[0]->get_error().c_str(), p[1]->get_error().c_str());
    return 2;
  }
  char op;
  std::vector<uint8_t> buf;
  while (scanf(" %c", &op) == 1) {
    if (op == 'S') {
      int side;
      unsigned long long byte_idx, seed, n;
      if (scanf("%d %llu %llu %llu", &side, &byte_idx, &seed, &n) != 4) return 3;
      std::vector<std::vector<uint8_t>> mem(n);
      std::vector<grpc_slice> sl(n);
      for (unsigned long long i = 0; i < n; i++) {
        unsigned long long len;
        if (scanf("%llu", &len) != 1) return 3;
        mem[i].resize(len ? len : 1);
        for (unsigned long long j = 0; j < len; j++) mem[i][j] = pat(seed, i, j);
        memset(&sl[i], 0, sizeof(grpc_slice));
        sl[i].refcount = kFakeRefcount;
        sl[i].data.refcounted.length = len;
        sl[i].data.refcounted.bytes = mem[i].data();
      }
      const uint64_t sent = p[side]->Send(sl.data(), n, byte_idx);
      printf("S %llu %llu %d %d\n", (unsigned long long)sent, (unsigned long long)p[side]->GetWritableSize(),
             p[side]->partial_write_.load() ? 1 : 0, p[side]->HasPendingWrites() ? 1 : 0);
    } else if (op == 'R') {
      int side;
      unsigned long long cap;
      if (scanf("%d %llu", &side, &cap) != 2) return 3;
      buf.assign(cap ? cap : 1, 0);
      const uint64_t n = p[side]->Recv(buf.data(), cap);
      printf("R %llu %llu %llu %d %llu\n", (unsigned long long)n, (unsigned long long)fnv(buf.data(), n),
             (unsigned long long)p[side]->GetReadableSize(), p[side]->HasMessage() ? 1 : 0,
             (unsigned long long)p[1 - side]->GetWritableSize());
    } else if (op == 'A') {  // AllocateSendBuffer(size) -> offset into the zero-copy buffer, or -1
      int side;
      unsigned long long size;
      if (scanf("%d %llu", &side, &size) != 2) return 3;
      uint8_t* q = p[side]->AllocateSendBuffer(size);
      uint8_t* base = p[side]->send_buffers_[PairPollable::kZeroCopyBuffer]->data();
      printf("A %lld\n", q ? (long long)(q - base) : -1ll);
    } else if (op == 'W') {  // bytes pat(seed, 0, j) into the zero-copy buffer at off
      int side;
      unsigned long long off, seed, len;
      if (scanf("%d %llu %llu %llu", &side, &off, &seed, &len) != 4) return 3;
      uint8_t* base = p[side]->send_buffers_[PairPollable::kZeroCopyBuffer]->data();
      for (unsigned long long j = 0; j < len; j++) base[off + j] = pat(seed, 0, j);
    } else if (op == 'Z') {  // SendZerocopy: every slice is {len, offset into the zero-copy buffer | -1 = a plain slice}
      int side;
      unsigned long long byte_idx, seed, n;
      if (scanf("%d %llu %llu %llu", &side, &byte_idx, &seed, &n) != 4) return 3;
      uint8_t* base = p[side]->send_buffers_[PairPollable::kZeroCopyBuffer]->data();
      std::vector<std::vector<uint8_t>> mem(n);
      std::vector<grpc_slice> sl(n);
      for (unsigned long long i = 0; i < n; i++) {
        unsigned long long len;
        long long zoff;
        if (scanf("%llu %lld", &len, &zoff) != 2) return 3;
        memset(&sl[i], 0, sizeof(grpc_slice));
        sl[i].refcount = kFakeRefcount;
        sl[i].data.refcounted.length = len;
        if (zoff >= 0) {
          sl[i].data.refcounted.bytes = base + zoff;
        } else {
          mem[i].resize(len ? len : 1);
          for (unsigned long long j = 0; j < len; j++) mem[i][j] = pat(seed, i, j);
          sl[i].data.refcounted.bytes = mem[i].data();
        }
      }
      const uint64_t sent = p[side]->SendZerocopy(sl.data(), n, byte_idx);
      printf("Z %llu %llu %d %u\n", (unsigned long long)sent, (unsigned long long)p[side]->GetWritableSize(),
             p[side]->partial_write_.load() ? 1 : 0, (unsigned)p[side]->zerocopy_buffer_tail_.load());
    } else if (op == 'Q') {
      for (int s = 0; s < 2; s++) {
        auto* rb = p[s]->recv_buffers_[PairPollable::kDataBuffer].get();
        printf("Q %d %llu %llu %llu %llu %llu\n", s, (unsigned long long)p[s]->remote_tail_,
               (unsigned long long)p[s]->internal_read_size_, (unsigned long long)p[s]->get_remote_head(),
               (unsigned long long)p[s]->ring_buf_.get_head(), (unsigned long long)fnv(rb->data(), rb->size()));
      }
    } else {
      return 3;
    }
  }
  fflush(stdout);
  _exit(0);  // (no teardown: the pairs' destructors would tear the shared fake device down in an order nobody tests)
}
