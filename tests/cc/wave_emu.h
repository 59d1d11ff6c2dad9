// TEST INFRASTRUCTURE.  A small host emulator for wave-level HIP kernels: enough of the CDNA execution
// model to run the latency-bound planner / parser kernels of this repository on a CPU, lane for lane, and
// compare their results with the oracle when no GPU is at hand (hipcc cross-compiles here, but nothing runs).
//
// Model.  A workgroup is emulated one at a time.  Every wavefront is one OS thread; its 64 lanes are
// coroutines (ucontext) that the wave thread resumes round-robin.  A cross-lane operation (__shfl*,
// __ballot, readfirstlane, DPP) completes when every lane of the wave that is still running has arrived at
// it -- the lanes of a wave must reach cross-lane operations in the same order, which is what wave-uniform
// control flow means on the hardware too; a lane that arrives early yields until the last one is there.
// __syncthreads() is the same inside a wave plus a barrier between the wave threads.  __shared__ objects are
// statics (one workgroup at a time); global memory is host memory; atomics map to the compiler's __atomic
// builtins (scopes collapse: everything is coherent); s_sleep yields to the other lanes and to the OS.
// What this checks is the LOGIC of a kernel (index arithmetic, ballots, prefix sums, hand-offs through LDS
// flags).  It says nothing about timing, cache behaviour or weaker-than-x86 memory ordering.
//
// Use: include this header FIRST (it defines __device__, __global__, threadIdx, the intrinsics, and empties
// <hip/hip_runtime.h>), then the kernel's source, then  emu::launch(grid, block, [&] { kernel(args...); }).
#pragma once
#define HIP_INCLUDE_HIP_HIP_RUNTIME_H  // (guards of the real headers, in case an include path finds them)
#define GRDMA_WAVE_EMU 1

#include <execinfo.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
#define address_space(x)  // __attribute__((address_space(1))) -> __attribute__(())
#define amdgpu_waves_per_eu(...)
// "every vector-memory operation of this wave is acknowledged": host stores are already there
#define GRDMA_WAIT_VMEM() asm volatile("" ::: "memory")
#define GRDMA_WAIT_LOADS() asm volatile("" ::: "memory")
#define GRDMA_WAVE_CONVERGE() emu::exchange(0)
#define GRDMA_WAVE_LOAD_LINES(base, lane) emu::wave_load_lines(reinterpret_cast<const uint64_t*>(base))

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 1024 * 1024;  // per lane; mapped lazily, with a guard page below it

struct lane_ctx {
  ucontext_t ctx;
  char* stack = nullptr;  // mmap'ed: kGuard bytes PROT_NONE, then kStack bytes
  dim3 tid;
  bool done = false;
};
constexpr size_t kGuard = 4096;
struct stack_pool {
  std::mutex mu;
  std::vector<char*> free_list;
};
inline stack_pool& stacks() {
  static stack_pool p;
  return p;
}
inline char* stack_alloc() {
  {
    std::lock_guard<std::mutex> lk(stacks().mu);
    if (!stacks().free_list.empty()) {
      char* q = stacks().free_list.back();
      stacks().free_list.pop_back();
      return q;
    }
  }
  void* p = mmap(nullptr, kGuard + kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) abort();
  mprotect(p, kGuard, PROT_NONE);  // a lane that overruns its stack faults here instead of in a neighbour
  return static_cast<char*>(p);
}
inline void stack_free(char* p) {  // back to the pool: the next launch reuses the mapping (and its touched pages)
  if (!p) return;
  std::lock_guard<std::mutex> lk(stacks().mu);
  stacks().free_list.push_back(p);
}

struct block_sync {  // barrier between the wave threads of a workgroup
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0, nwaves = 0;
  uint64_t gen = 0;
  void arrive_and_wait() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++waiting == nwaves) {
      waiting = 0;
      gen++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
  void drop() {  // a wave finished: it no longer takes part
    std::unique_lock<std::mutex> lk(mu);
    nwaves--;
    if (nwaves > 0 && waiting == nwaves) {
      waiting = 0;
      gen++;
      cv.notify_all();
    }
  }
};

struct wave_ctx {
  ucontext_t sched;
  lane_ctx lanes[kWave];
  int nlanes = 0;        // lanes this wave has (the last wave of a block may be partial)
  int active = 0;        // lanes still running
  int cur = 0;
  // cross-lane exchange
  uint64_t slot[kWave];
  uint64_t res[kWave];
  bool res_valid[kWave];
  int arrived = 0;
  uint64_t gen = 0;
  bool in_slot[kWave];
  void* site = nullptr;  // call site of the operation the lanes are gathering at
  int site_lane = 0;
  bool restart = false;  // an operation completed: resume the lanes from lane 0, in ascending order
  uint64_t lines[kWave];      // snapshot of a wave-wide block load (wave_load_lines)
  uint64_t lines_gen = ~0ull; // the exchange generation it belongs to
  block_sync* bs = nullptr;
  const std::function<void()>* body = nullptr;
  dim3 bid, bdim, gdim;
};

inline thread_local wave_ctx* t_wave = nullptr;

// Debugging aid: a lane appends (tag, value) without any effect on scheduling; when the wave finishes the
// logs of its lanes are compared and the first entry where a lane differs from lane 0 is reported -- the
// place where supposedly wave-uniform state stopped being uniform.
struct trace_entry { const char* tag; uint64_t v; };
inline thread_local std::vector<trace_entry>* t_traces = nullptr;  // [lane]
inline void trace(const char* tag, uint64_t v);

inline void trace_report(int nlanes) {
  if (!t_traces) return;
  const std::vector<trace_entry>* traces = t_traces;
  for (int i = 1; i < nlanes; i++) {
    const auto &a = traces[0], &b = traces[i];
    size_t k = 0;
    while (k < a.size() && k < b.size() && a[k].v == b[k].v && a[k].tag == b[k].tag) k++;
    if ((k < a.size() && k < b.size())) {
      fprintf(stderr, "wave_emu trace: lane %d differs from lane 0 at entry %zu: lane0 %s=%llu, lane%d %s=%llu (previous: %s=%llu)\n", i, k,
              a[k].tag, (unsigned long long)a[k].v, i, b[k].tag, (unsigned long long)b[k].v,
              k ? a[k - 1].tag : "-", k ? (unsigned long long)a[k - 1].v : 0ull);
      return;
    }
  }
  fprintf(stderr, "wave_emu trace: the lanes agree on every entry they share\n");
  // (one lane ran ahead of the others: the tails say how far each got)
  for (int i = 0; i < nlanes && i < 2; i++) {
    const auto& a = traces[i];
    fprintf(stderr, "wave_emu trace: lane %d has %zu entries, the last ones:", i, a.size());
    for (size_t k = a.size() > 6 ? a.size() - 6 : 0; k < a.size(); k++) fprintf(stderr, " %s=%llu", a[k].tag, (unsigned long long)a[k].v);
    fprintf(stderr, "\n");
  }
}
inline lane_ctx* cur_lane() { return &t_wave->lanes[t_wave->cur]; }
inline void trace(const char* tag, uint64_t v) {
  if (t_traces) t_traces[t_wave->cur].push_back({tag, v});
}
inline void yield_lane() {
  wave_ctx* w = t_wave;
  swapcontext(&w->lanes[w->cur].ctx, &w->sched);
}

// All running lanes arrive with a 64-bit value; afterwards res[] holds every lane's value (res_valid marks
// the lanes that took part).
// The lanes of a wave must meet at the SAME operation: a lane that arrives from a different call site means
// the kernel's control flow around a cross-lane operation is not wave-uniform (or depends on lanes running in
// lockstep between two such operations, which this emulator does not model).
__attribute__((noinline)) inline void exchange(uint64_t v) {
  wave_ctx* w = t_wave;
  const int me = w->cur;
  void* const here = __builtin_return_address(0);
  if (w->arrived == 0) {
    w->site = here;
    w->site_lane = me;
  } else if (w->site != here) {
    fprintf(stderr, "wave_emu: lane %d is at the cross-lane operation called from %p, lane %d at the one from %p\n", me,
            here, w->site_lane, w->site);
    void* bt[24];
    backtrace_symbols_fd(bt, backtrace(bt, 24), 2);
    trace_report(w->nlanes);
    abort();
  }
  w->slot[me] = v;
  w->in_slot[me] = true;
  const uint64_t g = w->gen;
  if (++w->arrived == w->active) {
    for (int i = 0; i < kWave; i++) {
      w->res[i] = w->slot[i];
      w->res_valid[i] = w->in_slot[i];
      w->in_slot[i] = false;
    }
    w->arrived = 0;
    w->gen++;
    // the stretch up to the next cross-lane operation runs lane 0 first, then 1, 2, ...: a value lane 0 stores
    // is there for the others, as it is when the lanes execute together
    w->restart = true;
    yield_lane();
  } else {
    while (w->gen == g) yield_lane();
  }
}

// Lane l gets word l of the 64-word block at base.  The first lane to run after the wave has met takes the
// snapshot for everyone, reading each 64-byte line's last word (its stamp) BEFORE the line's other words: a writer
// that fills a line and stamps it last is then never seen with a new stamp over old words -- what one request per
// line gives on the GPU.
inline uint64_t wave_load_lines(const uint64_t* base) {
  exchange(0);
  wave_ctx* w = t_wave;
  if (w->lines_gen != w->gen) {
    for (int line = 0; line < kWave / 8; line++) {
      w->lines[8 * line + 7] = __atomic_load_n(base + 8 * line + 7, __ATOMIC_ACQUIRE);
      for (int k = 0; k < 7; k++) w->lines[8 * line + k] = __atomic_load_n(base + 8 * line + k, __ATOMIC_RELAXED);
    }
    w->lines_gen = w->gen;
  }
  return w->lines[w->cur];
}

// a lane left the kernel: operations the others are waiting in may now be complete
inline void lane_exit() {
  wave_ctx* w = t_wave;
  w->lanes[w->cur].done = true;
  w->active--;
  if (w->active > 0 && w->arrived == w->active) {
    for (int i = 0; i < kWave; i++) {
      w->res[i] = w->slot[i];
      w->res_valid[i] = w->in_slot[i];
      w->in_slot[i] = false;
    }
    w->arrived = 0;
    w->gen++;
  }
}

inline void lane_entry() {
  (*t_wave->body)();
  lane_exit();
  yield_lane();  // never resumed
}

inline void run_wave(wave_ctx* w) {
  t_wave = w;
  std::vector<trace_entry> traces[kWave];
  t_traces = getenv("EMU_TRACE") ? traces : nullptr;
  w->active = w->nlanes;
  for (int i = 0; i < w->nlanes; i++) {
    lane_ctx& L = w->lanes[i];
    L.stack = stack_alloc();
    getcontext(&L.ctx);
    L.ctx.uc_stack.ss_sp = L.stack + kGuard;
    L.ctx.uc_stack.ss_size = kStack;
    L.ctx.uc_link = &w->sched;
    makecontext(&L.ctx, (void (*)())lane_entry, 0);
  }
  while (w->active > 0) {
    for (int i = 0; i < w->nlanes; i++) {
      if (w->lanes[i].done) continue;
      w->cur = i;
      swapcontext(&w->sched, &w->lanes[i].ctx);
      if (w->restart) {
        w->restart = false;
        break;
      }
    }
  }
  for (int i = 0; i < w->nlanes; i++) {
    stack_free(w->lanes[i].stack);
    w->lanes[i].stack = nullptr;
  }
  trace_report(w->nlanes);
  t_traces = nullptr;
  w->bs->drop();
  t_wave = nullptr;
}

// Runs grid.x * grid.y workgroups of block.x threads, one workgroup after the other.
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
      const int nw = (int)((block.x + kWave - 1) / kWave);
      block_sync bs;
      bs.nwaves = nw;
      std::vector<wave_ctx*> waves;
      for (int wv = 0; wv < nw; wv++) {
        wave_ctx* w = new wave_ctx();
        w->bs = &bs;
        w->body = &body;
        w->bid = dim3(bx, by, 0);
        w->bdim = block;
        w->gdim = grid;
        w->nlanes = (int)std::min<unsigned>(kWave, block.x - (unsigned)wv * kWave);
        for (int i = 0; i < w->nlanes; i++) w->lanes[i].tid = dim3((unsigned)(wv * kWave + i), 0, 0);
        memset(w->in_slot, 0, sizeof(w->in_slot));
        waves.push_back(w);
      }
      std::vector<std::thread> th;
      for (wave_ctx* w : waves) th.emplace_back(run_wave, w);
      for (auto& t : th) t.join();
      for (wave_ctx* w : waves) delete w;
    }
}

struct tid_proxy {
  struct field {
    int which;
    operator unsigned() const {
      const dim3& d = cur_lane()->tid;
      return which == 0 ? d.x : which == 1 ? d.y : d.z;
    }
  };
  field x{0}, y{1}, z{2};
};
struct wave_dim_proxy {
  int sel;  // 0 blockIdx, 1 blockDim, 2 gridDim
  struct field {
    int sel, which;
    operator unsigned() const {
      const wave_ctx* w = t_wave;
      const dim3& d = sel == 0 ? w->bid : sel == 1 ? w->bdim : w->gdim;
      return which == 0 ? d.x : which == 1 ? d.y : d.z;
    }
  };
  field x, y, z;
  explicit wave_dim_proxy(int s) : sel(s), x{s, 0}, y{s, 1}, z{s, 2} {}
};

template <typename T>
inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T>
inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}

}  // namespace emu

static emu::tid_proxy threadIdx;
static emu::wave_dim_proxy blockIdx(0), blockDim(1), gridDim(2);

// ---- cross-lane intrinsics ----------------------------------------------------------------------------
template <typename T>
__attribute__((always_inline)) inline T __shfl(T v, int src, int /*width*/ = 64) {
  emu::exchange(emu::to_bits(v));
  const emu::wave_ctx* w = emu::t_wave;
  src &= 63;
  return w->res_valid[src] ? emu::from_bits<T>(w->res[src]) : v;
}
template <typename T>
__attribute__((always_inline)) inline T __shfl_up(T v, unsigned d, int /*width*/ = 64) {
  emu::exchange(emu::to_bits(v));
  const emu::wave_ctx* w = emu::t_wave;
  const int me = w->cur, src = me - (int)d;
  return (src >= 0 && w->res_valid[src]) ? emu::from_bits<T>(w->res[src]) : v;
}
template <typename T>
__attribute__((always_inline)) inline T __shfl_xor(T v, int mask, int /*width*/ = 64) {
  emu::exchange(emu::to_bits(v));
  const emu::wave_ctx* w = emu::t_wave;
  const int src = (w->cur ^ mask) & 63;
  return w->res_valid[src] ? emu::from_bits<T>(w->res[src]) : v;
}
__attribute__((always_inline)) inline uint64_t __ballot(int pred) {
  emu::exchange(pred ? 1u : 0u);
  const emu::wave_ctx* w = emu::t_wave;
  uint64_t m = 0;
  for (int i = 0; i < 64; i++)
    if (w->res_valid[i] && w->res[i]) m |= 1ull << i;
  return m;
}
__attribute__((always_inline)) inline int __builtin_amdgcn_readfirstlane(int v) {
  emu::exchange((uint64_t)(uint32_t)v);
  const emu::wave_ctx* w = emu::t_wave;
  for (int i = 0; i < 64; i++)
    if (w->res_valid[i]) return (int)(uint32_t)w->res[i];
  return v;
}
__attribute__((always_inline)) inline int __builtin_amdgcn_readlane(int v, int src) {
  emu::exchange((uint64_t)(uint32_t)v);
  const emu::wave_ctx* w = emu::t_wave;
  src &= 63;
  return w->res_valid[src] ? (int)(uint32_t)w->res[src] : v;
}
// DPP: the controls this repository uses (row_shr:1/2/4/8, row_bcast:15, row_bcast:31, wave_shl:1, wave_rol:1)
__attribute__((always_inline)) inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  emu::exchange((uint64_t)(uint32_t)src);
  const emu::wave_ctx* w = emu::t_wave;
  const int me = w->cur, row = me >> 4, bank = (me & 15) >> 2;
  if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
  int from = -1;
  if (ctrl >= 0x111 && ctrl <= 0x11F) {  // row_shr:n
    const int n = ctrl - 0x110;
    if ((me & 15) - n >= 0) from = me - n;
  } else if (ctrl == 0x142) {  // row_bcast:15
    if (row >= 1) from = (row - 1) * 16 + 15;
  } else if (ctrl == 0x143) {  // row_bcast:31
    if (row >= 2) from = 31;
  } else if (ctrl == 0x130) {  // wave_shl:1
    if (me + 1 < 64) from = me + 1;
  } else if (ctrl == 0x134) {  // wave_rol:1
    from = (me + 1) & 63;
  } else {
    __builtin_trap();
  }
  if (from < 0 || !w->res_valid[from]) return bound_ctrl ? 0 : old;
  return (int)(uint32_t)w->res[from];
}

__attribute__((always_inline)) inline int __all(int pred) { return __ballot(!pred) == 0; }
__attribute__((always_inline)) inline int __any(int pred) { return __ballot(pred) != 0; }

inline void __syncthreads() {
  emu::wave_ctx* w = emu::t_wave;
  const uint64_t g = w->gen;
  w->in_slot[w->cur] = true;
  if (++w->arrived == w->active) {
    w->bs->arrive_and_wait();  // the last lane of the wave waits for the other waves
    for (int i = 0; i < 64; i++) w->in_slot[i] = false;
    w->arrived = 0;
    w->gen++;
    w->restart = true;
    emu::yield_lane();
  } else {
    while (w->gen == g) emu::yield_lane();
  }
}

// ---- scalar intrinsics ----------------------------------------------------------------------------------
inline uint64_t __builtin_amdgcn_s_memtime() {
  static std::atomic<uint64_t> t{0};
  return t.fetch_add(1, std::memory_order_relaxed);
}
inline uint64_t __builtin_amdgcn_s_memrealtime() { return __builtin_amdgcn_s_memtime(); }
inline void __builtin_amdgcn_s_sleep(int) {
  emu::yield_lane();
  sched_yield();
}
inline void __builtin_amdgcn_s_setprio(int) {}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_compare_exchange_strong(p, expected, desired, so, fo, scope) __atomic_compare_exchange_n((p), (expected), (desired), false, (so), (fo))
template <typename T>
inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T>
inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <typename T>
inline T atomicMax(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T>
inline T atomicMin(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

// ---- buffer instructions: bounds-checked accesses through a {base, bytes} descriptor --------------------
struct __amdgpu_buffer_rsrc_t {
  uint8_t* base;
  uint32_t bytes;
};
typedef uint32_t emu_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t emu_u32x2 __attribute__((ext_vector_type(2)));
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int bytes, int) {
  return {static_cast<uint8_t*>(p), (uint32_t)bytes};
}
inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int off, int, int) {
  emu_u32x4 v = {0, 0, 0, 0};
  if (off >= 0 && (uint64_t)(uint32_t)off + 16 <= r.bytes) memcpy(&v, r.base + off, 16);
  return v;
}
inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int off, int, int) {
  emu_u32x2 v = {0, 0};
  if (off >= 0 && (uint64_t)(uint32_t)off + 8 <= r.bytes) memcpy(&v, r.base + off, 8);
  return v;
}
inline uint8_t __builtin_amdgcn_raw_buffer_load_b8(__amdgpu_buffer_rsrc_t r, int off, int, int) {
  return (off >= 0 && (uint32_t)off < r.bytes) ? r.base[off] : 0;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int off, int, int) {
  if (off >= 0 && (uint64_t)(uint32_t)off + 16 <= r.bytes) memcpy(r.base + off, &v, 16);
}
inline void __builtin_amdgcn_raw_buffer_store_b8(uint8_t v, __amdgpu_buffer_rsrc_t r, int off, int, int) {
  if (off >= 0 && (uint32_t)off < r.bytes) r.base[off] = v;
}

#include "hip_api_emu.h"
