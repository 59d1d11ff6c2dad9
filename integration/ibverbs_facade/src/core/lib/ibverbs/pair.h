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
#include <string>

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

 private:
  grdma_pair* pair_;
  grpc_wakeup_fd wakeup_fd_;
  mutable std::string error_;
};

}  // namespace ibverbs
}  // namespace grpc_core
#endif
#endif  // GRPC_SRC_CORE_LIB_IBVERBS_PAIR_H
