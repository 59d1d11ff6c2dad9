// src/core/lib/ibverbs/poller.h over libgrdma_amd.so: Poller::Get() with AddPollable / RemovePollable / Shutdown
// (poller.h:16-71), backed by grdma_poller -- GRPC_RDMA_POLLER_THREAD_NUM host threads sharing one round-robin cursor
// over the slot table, kicking a pair's wakeup fd when the pair is readable, writable, half-closed or in error
// (poller.cc:52-106).  ev_epollex_rdma_bpev_linux.cc:1794 calls Poller::Get().Shutdown(); the endpoint calls
// AddPollable / RemovePollable (rdma_bp_posix.cc:119-121, 789-791).
#ifndef GRPC_SRC_CORE_LIB_IBVERBS_POLLER_H
#define GRPC_SRC_CORE_LIB_IBVERBS_POLLER_H
#ifdef GRPC_USE_IBVERBS
#include "src/core/lib/ibverbs/config.h"
#include "src/core/lib/ibverbs/pair.h"

#define GRPC_IBVERBS_POLLER_CAPACITY (4096)
namespace grpc_core {
namespace ibverbs {

class Poller {
  Poller() {
    poller_ = grdma_poller_create(Config::Get().get_poller_thread_num(), Config::Get().get_poller_sleep_timeout_ms());
  }
  ~Poller() { Shutdown(); }

 public:
  static Poller& Get() {
    static Poller poller;
    return poller;
  }
  void Shutdown() {
    grdma_poller* p = poller_;
    poller_ = nullptr;
    if (p != nullptr) grdma_poller_destroy(p);
  }
  void AddPollable(PairPollable* pollable) {
    if (poller_ != nullptr) grdma_poller_add(poller_, pollable->hip_pair());
  }
  void RemovePollable(PairPollable* pollable) {
    if (poller_ != nullptr) grdma_poller_remove(poller_, pollable->hip_pair());
  }

 private:
  grdma_poller* poller_;
};

}  // namespace ibverbs
}  // namespace grpc_core
#endif
#endif  // GRPC_SRC_CORE_LIB_IBVERBS_POLLER_H
