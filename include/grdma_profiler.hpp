// GRPCProfiler of the reference (include/grpcpp/stats_time.h:111-122) over the C ABI's
// grdma_stats_time_* (include/grdma_amd.h): an object records its lifetime under `op`.
#ifndef GRDMA_PROFILER_HPP
#define GRDMA_PROFILER_HPP
#include "grdma_amd.h"

class grdma_profiler {
 public:
  explicit grdma_profiler(int op) : op_(op), begin_(grdma_stats_time_enabled() ? grdma_stats_time_now_ns() : -1) {}
  ~grdma_profiler() {
    if (begin_ >= 0) grdma_stats_time_add(op_, grdma_stats_time_now_ns() - begin_);
  }
  grdma_profiler(const grdma_profiler&) = delete;
  grdma_profiler& operator=(const grdma_profiler&) = delete;

 private:
  int op_;
  int64_t begin_;
};
#endif  // GRDMA_PROFILER_HPP
