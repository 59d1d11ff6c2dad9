// src/core/lib/ibverbs/pair.h AS THE EVENT ENGINES SEE IT, over libgrdma_amd.so.
//
// ev_epollex_rdma_bp_linux.cc and ev_epollex_rdma_bpev_linux.cc keep the pointer the endpoint hands them through
// grpc_fd_set_arg() as a grpc_core::ibverbs::PairPollable* (ev_epollex_rdma_bpev_linux.cc:478) and call, from every
// polling thread on every pass, get_status() / HasMessage() / HasPendingWrites() (:1017-1031, :1111-1116) and
// get_wakeup_fd()->read_fd (:526, :729).  With this directory FIRST on the include path of the gRPC tree those two
// files compile UNMODIFIED against the class below: the same names, and behind them plain loads of the pair's
// host-visible state (grdma_endpoint_readable / _writable / grdma_pair_get_status: no device call, no lock).
// integration/rdma_hip_posix.cc creates one PairPollable per endpoint and passes IT to grpc_fd_set_arg.
//
// Same include guard as the reference's header on purpose: whichever of the two is found first wins.
#ifndef GRPC_SRC_CORE_LIB_IBVERBS_PAIR_H
#define GRPC_SRC_CORE_LIB_IBVERBS_PAIR_H
#ifdef GRPC_USE_IBVERBS
#include <grpc/slice.h>

#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "src/core/lib/ibverbs/config.h"
#include "src/core/lib/iomgr/wakeup_fd_posix.h"

#include "grdma_amd.h"

#define IBVERBS_PAIR_TAG_POLLABLE (0xa0)

namespace grpc_core {
namespace ibverbs {

enum class PairStatus {  // pair.h:44-51; the values grdma_pair_get_status returns
  kUninitialized,
  kInitialized,
  kConnected,
  kHalfClosed,
  kDisconnected,
  kError
};

class PairPollable {
 public:
  explicit PairPollable(grdma_pair* pair) : pair_(pair) {
    // grpc_wakeup_fd of the pair (pair.h:150,187): an eventfd; the engines register read_fd with epoll and
    // consume it with grpc_wakeup_fd_consume_wakeup (an eventfd read)
    wakeup_fd_.read_fd = grdma_pair_get_wakeup_fd(pair);
    wakeup_fd_.write_fd = -1;
  }
  PairPollable(const PairPollable&) = delete;
  PairPollable& operator=(const PairPollable&) = delete;

  // read-only, lock-free, any number of threads (ring_buffer.cc:56-65, pair.cc:303, 349-375)
  bool HasMessage() const { return grdma_endpoint_readable(pair_) > 0; }
  bool HasPendingWrites() const { return grdma_endpoint_writable(pair_) > 0; }
  PairStatus get_status() { return static_cast<PairStatus>(grdma_pair_get_status(pair_)); }
  uint64_t GetReadableSize() const {
    const int64_t n = grdma_pair_readable_size(pair_);
    return n > 0 ? static_cast<uint64_t>(n) : 0;
  }
  uint64_t GetWritableSize() const {
    const int64_t n = grdma_pair_writable_size(pair_);
    return n > 0 ? static_cast<uint64_t>(n) : 0;
  }
  grpc_wakeup_fd* get_wakeup_fd() { return &wakeup_fd_; }
  const std::string& get_error() const {
    error_ = grdma_last_error();
    return error_;
  }
  void Disconnect() { grdma_pair_disconnect(pair_); }
  grdma_pair* hip_pair() const { return pair_; }

  // ---- zero-copy send buffer (pair.h:127,140; pair.cc:305-323, 793-941) -------------------------------------------
  // AllocateSendBuffer: the pointer CoreCodegen::grpc_call_allocate_send_buffer hands to GenericSerialize
  // (src/cpp/common/core_codegen.cc:122-146, include/grpcpp/impl/codegen/proto_utils.h:68-95), which lets protobuf
  // serialise into it with the CPU: pinned host memory the gather kernel reads in place (GRDMA_ZC_MEM_HOST).  One
  // allocation at a time, nullptr otherwise -- the reference's rule.
  uint8_t* AllocateSendBuffer(size_t size) {
    return static_cast<uint8_t*>(grdma_pair_allocate_send_buffer(pair_, static_cast<uint64_t>(size)));
  }
  // SendZerocopy(slices, count, byte_idx): the slices as grpc_endpoint_write holds them; the one that lies in the
  // zero-copy buffer leaves from where it is, the rest (frame and message headers) as Send sends them
  uint64_t SendZerocopy(grpc_slice* slices, size_t slice_count, size_t byte_idx) {
    std::vector<grdma_slice> v(slice_count);
    for (size_t i = 0; i < slice_count; i++) {
      v[i].ptr = GRPC_SLICE_START_PTR(slices[i]);
      v[i].len = GRPC_SLICE_LENGTH(slices[i]);
    }
    const int64_t n = grdma_pair_send_zerocopy(pair_, v.data(), slice_count, byte_idx, GRDMA_MEM_HOST);
    return n > 0 ? static_cast<uint64_t>(n) : 0;
  }

 private:
  grdma_pair* pair_;
  grpc_wakeup_fd wakeup_fd_;
  mutable std::string error_;
};

// PairPool (pair.h:273-333): Take(id) / Get(id) / Putback over the library's pool of pair MEMORY
// (grdma_pair_pool_take / _putback), keeping the id -> PairPollable* table the zero-copy hook looks a call's pair up in
// (core_codegen.cc:130-139: PairPool::Get().Get(peer) with the string grpc_call_get_peer_id returns, surface/call.cc:663-672).
// Take() shapes the pair from Config, as the reference's PairPollable() does; the endpoint (integration/
// rdma_hip_posix.cc) takes its pair here and puts it back in rdma_free.
class PairPool {
  PairPool() {}

 public:
  PairPool(const PairPool&) = delete;
  PairPool& operator=(const PairPool&) = delete;

  static PairPool& Get() {
    static PairPool pool;
    return pool;
  }

  PairPollable* Take(const std::string& id) {
    grdma_config cfg;
    if (grdma_config_from_env(&cfg) < 0 || grdma_init(cfg.hip_device) < 0) return nullptr;
    static const int pool_on = grdma_pair_pool_reserve(0, 0, 0, 0, static_cast<uint64_t>(cfg.hip_pair_pool_mb) << 20);
    (void)pool_on;
    // (fine-grained: the peer -- another process -- writes this pair's ring and status block through an IPC mapping)
    grdma_pair* pair = grdma_pair_pool_take(id.c_str(), static_cast<uint64_t>(cfg.ring_buffer_size_kb) * 1024, cfg.max_sge,
                                            (cfg.hip_wire_direct ? GRDMA_WIRE_DIRECT : GRDMA_WIRE_STAGED) | GRDMA_RING_FINE_GRAINED);
    if (pair == nullptr) return nullptr;
    PairPollable* pollable = new PairPollable(pair);
    std::unique_lock<std::shared_timed_mutex> lock(mu_);
    id_pair_[id] = pollable;
    pair_id_[pollable] = id;  // (the reference never fills this table, so its Putback never erases: SURVEY.md A.10)
    return pollable;
  }

  void Putback(PairPollable* pollable) {
    if (pollable == nullptr) return;
    {
      std::unique_lock<std::shared_timed_mutex> lock(mu_);
      auto it = pair_id_.find(pollable);
      if (it != pair_id_.end()) {
        auto by_id = id_pair_.find(it->second);
        if (by_id != id_pair_.end() && by_id->second == pollable) id_pair_.erase(by_id);
        pair_id_.erase(it);
      }
    }
    grdma_pair* pair = pollable->hip_pair();
    delete pollable;
    grdma_pair_pool_putback(pair);
  }

  PairPollable* Get(const std::string& id) {
    std::shared_lock<std::shared_timed_mutex> lock(mu_);
    auto it = id_pair_.find(id);
    return it != id_pair_.end() ? it->second : nullptr;
  }

 private:
  std::shared_timed_mutex mu_;
  std::unordered_map<std::string, PairPollable*> id_pair_;
  std::unordered_map<PairPollable*, std::string> pair_id_;
};

}  // namespace ibverbs
}  // namespace grpc_core
#endif
#endif  // GRPC_SRC_CORE_LIB_IBVERBS_PAIR_H
