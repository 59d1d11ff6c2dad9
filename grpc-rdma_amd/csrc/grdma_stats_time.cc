// Scope profiler with the reference's op names (include/grpcpp/stats_time.h:11-44,
// src/core/lib/debug/stats_time.cc): nanoseconds per op into a histogram per thread slot, opt-in per
// thread (grdma_stats_time_init(slot) + grdma_stats_time_enable()), printed as the table
// {Name, Count, Mean, P50, P95, P99, MAX} per slot in the unit GRPC_PROFILING_UNIT names, so that a
// stage breakdown taken here lines up with one taken on the reference build.  Host only: no device
// is touched.  The hooks sit where the reference's GRPCProfiler objects sit: the endpoint's
// rdma_read / rdma_handle_read / rdma_continue_read / rdma_do_read / rdma_flush /
// rdma_handle_write / rdma_write (csrc/grdma_endpoint.cc) and PairPollable::Send / Recv
// (csrc/grdma_pair.hip).
//
// The histogram keeps three significant digits like the reference's HdrHistogram (hdr_init(1,
// 6e13, 3)): values below 2048 exactly, above that 1024 linear sub-buckets per power of two,
// allocated per op on first use.
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/grdma_amd.h"

namespace {

constexpr int kSubBits = 11;                    // 2048 values per half-open power-of-two range
constexpr int kSub = 1 << kSubBits;
constexpr int64_t kMaxValue = 60LL * 1000 * 1000 * 1000 * 1000;  // GRPC_PROFILING_MAX_VALUE
constexpr int kRanges = 64 - kSubBits;

struct histogram {
  std::vector<uint32_t> counts;  // kSub exact values, then kSub / 2 buckets per further bit
  uint64_t total = 0;
  long double sum = 0;
  int64_t max = 0;

  static size_t index_of(int64_t v) {
    if (v < kSub) return (size_t)v;
    const int msb = 63 - __builtin_clzll((uint64_t)v);
    const int shift = msb - (kSubBits - 1);              // v >> shift lies in [kSub / 2, kSub)
    return (size_t)kSub + (size_t)(shift - 1) * (kSub / 2) + (size_t)((v >> shift) - kSub / 2);
  }
  // the highest value that falls into bucket i (what hdr_value_at_percentile reports)
  static int64_t value_of(size_t i) {
    if (i < (size_t)kSub) return (int64_t)i;
    const size_t r = (i - kSub) / (kSub / 2), k = (i - kSub) % (kSub / 2);
    const int shift = (int)r + 1;
    return (int64_t)(((uint64_t)(k + kSub / 2 + 1) << shift) - 1);
  }
  void add(int64_t v) {
    if (v < 0) v = 0;
    if (counts.empty()) counts.assign((size_t)kSub + (size_t)kRanges * (kSub / 2), 0);
    const size_t i = index_of(v);
    if (i < counts.size()) counts[i]++;
    total++;
    sum += (long double)v;
    if (v > max) max = v;
  }
  void clear() {
    counts.clear();
    total = 0;
    sum = 0;
    max = 0;
  }
  double mean() const { return total ? (double)(sum / (long double)total) : 0.0; }
  int64_t percentile(double q) const {  // q in [0, 1]
    if (!total) return 0;
    uint64_t want = (uint64_t)(q * (double)total + 0.5);
    if (want < 1) want = 1;
    if (want > total) want = total;
    uint64_t seen = 0;
    for (size_t i = 0; i < counts.size(); i++) {
      seen += counts[i];
      if (seen >= want) {
        const int64_t v = value_of(i);
        return v < max ? v : max;
      }
    }
    return max;
  }
};

struct entry {
  histogram h;
  bool scale = true;  // false: a custom quantity (grdma_stats_time_add_custom), printed unscaled
};

struct slot_data {
  int slot = 0;
  entry per_op[GRDMA_STATS_TIME_MAX_OP_SIZE];
  std::mutex mu;  // a slot belongs to one thread; print / get may come from another
};

std::atomic<bool> g_enabled{false};
thread_local int t_slot = -1;
std::mutex g_mu;
std::vector<std::shared_ptr<slot_data>> g_slots;  // index = slot number; users hold a reference of their own while they
                                                  // work on a slot (shutdown / re-init on another thread only drop the table's)

const char* const kNames[GRDMA_STATS_TIME_MAX_OP_SIZE] = {
    "POLLABLE_EPOLL", "POLLSET_WORK", "TRANSPORT_DO_READ", "TRANSPORT_CONTINUE_READ",
    "TRANSPORT_READ_ALLOCATION_DONE", "TRANSPORT_HANDLE_READ", "TRANSPORT_READ", "TRANSPORT_FLUSH",
    "TRANSPORT_HANDLE_WRITE", "TRANSPORT_WRITE", "PAIR_SEND", "PAIR_RECV", "CLIENT_PREPARE",
    "CLIENT_CQ_NEXT", "SERVER_RPC_REQUEST", "SERVER_RPC_FINISH", "SERVER_CQ_NEXT", "BEGIN_WORKER",
    "ASYNC_NEXT_INTERNAL", "FINALIZE_RESULT", "DESERIALIZE", "ADHOC_1", "ADHOC_2", "ADHOC_3", "ADHOC_4",
    "ADHOC_5", "ADHOC_6", "ADHOC_7", "ADHOC_8", "ADHOC_9", "ADHOC_10"};

std::shared_ptr<slot_data> my_slot() {
  if (t_slot < 0) return nullptr;
  std::lock_guard<std::mutex> lg(g_mu);
  return (size_t)t_slot < g_slots.size() ? g_slots[(size_t)t_slot] : nullptr;
}

void add(int op, int64_t val, bool scale) {
  if (!g_enabled.load(std::memory_order_relaxed) || op < 0 || op >= GRDMA_STATS_TIME_MAX_OP_SIZE) return;
  const std::shared_ptr<slot_data> s = my_slot();
  if (!s) return;
  if (val >= kMaxValue) val = kMaxValue - 1;  // (the reference asserts)
  std::lock_guard<std::mutex> lg(s->mu);
  s->per_op[op].h.add(val);
  s->per_op[op].scale = scale;
}

// GRPC_PROFILING_UNIT: micro (default) / milli / s, stats_time.cc:25-45
int unit_scale(const char** name) {
  const char* u = getenv("GRPC_PROFILING_UNIT");
  *name = "us";
  if (!u || strcmp(u, "micro") == 0) return 1;
  if (strcmp(u, "milli") == 0) { *name = "ms"; return 1000; }
  if (strcmp(u, "s") == 0) { *name = "s"; return 1000 * 1000; }
  return 1;
}

}  // namespace

extern "C" {

const char* grdma_stats_time_op_name(int op) {
  return (op >= 0 && op < GRDMA_STATS_TIME_MAX_OP_SIZE) ? kNames[op] : "";
}

void grdma_stats_time_init(int slot) {  // stats_time.cc:47-58
  if (slot < 0) return;
  t_slot = slot;
  std::lock_guard<std::mutex> lg(g_mu);
  if (g_slots.size() <= (size_t)slot) g_slots.resize((size_t)slot + 1);
  g_slots[(size_t)slot].reset(new slot_data());
  g_slots[(size_t)slot]->slot = slot;
}

void grdma_stats_time_enable(void) { g_enabled = true; }
void grdma_stats_time_disable(void) { g_enabled = false; }
int grdma_stats_time_enabled(void) { return (g_enabled.load(std::memory_order_relaxed) && t_slot >= 0) ? 1 : 0; }

void grdma_stats_time_shutdown(void) {  // stats_time.cc:60-64
  std::lock_guard<std::mutex> lg(g_mu);
  g_enabled = false;
  g_slots.clear();
}

void grdma_stats_time_add(int op, int64_t ns) { add(op, ns, true); }
void grdma_stats_time_add_custom(int op, int64_t val) { add(op, val, false); }

int64_t grdma_stats_time_now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}

// {mean, p50, p95, p99, max} of one op of one slot in nanoseconds (unscaled); returns the count
uint64_t grdma_stats_time_get(int slot, int op, double out[5]) {
  if (slot < 0 || op < 0 || op >= GRDMA_STATS_TIME_MAX_OP_SIZE) return 0;
  std::shared_ptr<slot_data> s;
  {
    std::lock_guard<std::mutex> lg(g_mu);
    if ((size_t)slot < g_slots.size()) s = g_slots[(size_t)slot];
  }
  if (!s) return 0;
  std::lock_guard<std::mutex> lg(s->mu);
  const histogram& h = s->per_op[op].h;
  if (out) {
    out[0] = h.mean();
    out[1] = (double)h.percentile(0.50);
    out[2] = (double)h.percentile(0.95);
    out[3] = (double)h.percentile(0.99);
    out[4] = (double)h.max;
  }
  return h.total;
}

// The table of grpc_stats_time_print (stats_time.cc:161-244) into buf; returns the number of bytes the
// whole text needs (like snprintf).  Mean and percentiles are divided by 1000 and by the unit's scale,
// MAX by the scale only -- as the reference prints them.
int64_t grdma_stats_time_print(char* buf, uint64_t cap) {
  const char* unit;
  const int scale = unit_scale(&unit);
  std::string out;
  char line[256];
  out += "=================================Profiling Result=================================\n";
  snprintf(line, sizeof(line), "Unit %s\n", unit);
  out += line;
  std::vector<std::shared_ptr<slot_data>> slots;
  {
    std::lock_guard<std::mutex> lg(g_mu);
    for (auto& s : g_slots)
      if (s) slots.push_back(s);
  }
  for (const std::shared_ptr<slot_data>& s : slots) {
    std::lock_guard<std::mutex> lg(s->mu);
    std::string rows;
    for (int op = 0; op < GRDMA_STATS_TIME_MAX_OP_SIZE; op++) {
      const entry& e = s->per_op[op];
      if (!e.h.total) continue;
      double mean = e.h.mean(), p50 = (double)e.h.percentile(0.5), p95 = (double)e.h.percentile(0.95),
             p99 = (double)e.h.percentile(0.99), mx = (double)e.h.max;
      std::string name = kNames[op];
      if (e.scale) {
        mean = mean / 1000 / scale; p50 = p50 / 1000 / scale; p95 = p95 / 1000 / scale; p99 = p99 / 1000 / scale;
        mx = mx / scale;
      } else {
        name += " (custom)";
      }
      snprintf(line, sizeof(line), "| %-40s | %12llu | %14.2f | %14.2f | %14.2f | %14.2f | %16.2f |\n", name.c_str(),
               (unsigned long long)e.h.total, mean, p50, p95, p99, mx);
      rows += line;
    }
    if (rows.empty()) continue;
    snprintf(line, sizeof(line), "Slot: %d\n", s->slot);
    out += line;
    snprintf(line, sizeof(line), "| %-40s | %12s | %14s | %14s | %14s | %14s | %16s |\n", "Name", "Count", "Mean", "P50",
             "P95", "P99", "MAX");
    out += line;
    out += rows;
  }
  out += "==================================================================================\n";
  if (buf && cap) {
    const size_t n = out.size() < cap - 1 ? out.size() : (size_t)cap - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size();
}

}  // extern "C"
