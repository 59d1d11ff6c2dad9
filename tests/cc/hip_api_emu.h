// TEST INFRASTRUCTURE.  The slice of the HIP host API this repository's host layer uses, over host memory and
// the wave emulator (tests/cc/wave_emu.h): "device" memory is malloc'ed, copies are memcpy, streams and events
// are synchronous no-ops, a kernel launch runs the kernel under the emulator before it returns.  Together with
// wave_emu.h this lets the PRODUCT sources (csrc/*.hip, *.cc) be compiled for the CPU into
// oracle/_build/libgrdma_emu.so, which the Python parity tests can load instead of libgrdma_amd.so
// (GRDMA_LIB_PATH) when no GPU is at hand -- see tests/cc/build_emu.sh and tests/test_emu_pair.py.
// Graphs of kernel nodes run their nodes in the order they were added; the resident latency engine runs in a
// thread of its own.  Not emulated: IPC handles, dma-buf export (they report an error).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <unistd.h>

#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

typedef int hipError_t;
enum {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotReady = 600,
  hipErrorNotSupported = 801,
};
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
struct emu_graph_node;
struct emu_graph;
struct emu_graph_exec;
typedef emu_graph* hipGraph_t;
typedef emu_graph_exec* hipGraphExec_t;
typedef emu_graph_node* hipGraphNode_t;
typedef void* hipDeviceptr_t;
struct hipIpcMemHandle_t {
  char reserved[64];
};
struct hipKernelNodeParams {
  dim3 blockDim;
  void** extra;
  void* func;
  dim3 gridDim;
  void** kernelParams;
  unsigned int sharedMemBytes;
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocCoherent = 0x40000000, hipHostMallocMapped = 2, hipHostMallocNonCoherent = 0x80000000u,
       hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1, hipMemRangeHandleTypeDmaBufFd = 1 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1, hipDeviceAttributeWallClockRate = 2, hipDeviceAttributeIsLargeBar = 3 };

namespace emu {
inline std::recursive_mutex& launch_mutex() {
  static std::recursive_mutex m;
  return m;
}
// Device memory.  With EMU_GUARD_ALLOC=1 every allocation ends right in front of an inaccessible page (the
// start is 16-byte aligned, so an overrun of 16 bytes or more -- a vector load past the end of a ring, an arena,
// a plan -- faults at once, with EMU_SEGV_TRACE=1 naming the kernel line); the GPU's coarse page mapping lets
// such accesses through most of the time.  Guarded blocks are never unmapped before exit (a stale pointer
// faults too).
inline bool guard_alloc() {
  static const bool on = [] { const char* e = getenv("EMU_GUARD_ALLOC"); return e && e[0] == '1'; }();
  return on;
}
struct guard_hdr { size_t map_len; };
inline void* dev_alloc(size_t n) {
  if (guard_alloc()) {
    const size_t page = 4096, need = ((n ? n : 1) + 15) & ~(size_t)15;
    const size_t body = (need + page - 1) / page * page;
    char* m = static_cast<char*>(mmap(nullptr, body + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (m == MAP_FAILED) return nullptr;
    mprotect(m + body, page, PROT_NONE);
    char* p = m + body - need;
    memset(p, 0xA5, need);
    return p;
  }
  void* p = nullptr;
  if (posix_memalign(&p, 256, n ? n : 1) != 0) return nullptr;
  memset(p, 0xA5, n);  // device memory does not come zeroed
  return p;
}
// Fine-grained device memory (hipExtMallocWithFlags: the rings and connection blocks another PROCESS may map through a
// HIP IPC handle) lives in an anonymous shared-memory file (memfd): the handle names {pid, fd, length}, the peer process
// opens /proc/<pid>/fd/<fd> and maps the same pages -- two emulated processes then share a ring the way two processes
// on one GPU do (tests/test_two_process_emu.py).  Nothing is left behind in /dev/shm; the pages go with the processes.
struct shared_block { int fd; size_t len; };
inline std::map<void*, shared_block>& shared_blocks() {
  static std::map<void*, shared_block> m;
  return m;
}
inline std::mutex& shared_mu() {
  static std::mutex m;
  return m;
}
inline void* dev_alloc_shared(size_t n) {
  const size_t len = ((n ? n : 1) + 4095) & ~(size_t)4095;
  const int fd = memfd_create("grdma_emu_finegrained", 0);
  if (fd < 0 || ftruncate(fd, (off_t)len) != 0) return nullptr;
  void* p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  if (p == MAP_FAILED) {
    close(fd);
    return nullptr;
  }
  memset(p, 0xA5, len);
  std::lock_guard<std::mutex> lk(shared_mu());
  shared_blocks()[p] = shared_block{fd, len};
  return p;
}
inline void dev_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(shared_mu());
    auto it = shared_blocks().find(p);
    if (it != shared_blocks().end()) {
      munmap(p, it->second.len);
      close(it->second.fd);
      shared_blocks().erase(it);
      return;
    }
  }
  if (guard_alloc()) return;  // (kept mapped: see above)
  free(p);
}
struct ipc_handle { uint32_t magic; int32_t pid, fd; uint32_t pad; uint64_t len; };
}  // namespace emu

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 2 : a == hipDeviceAttributeIsLargeBar ? 1 : 100000;  // (device memory IS host memory here)
  return hipSuccess;
}
inline hipError_t hipDeviceGetPCIBusId(char* s, int n, int) {
  strncpy(s, "0000:00:00.0", (size_t)n);
  return hipSuccess;
}
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }

template <typename T>
inline hipError_t hipMalloc(T** p, size_t n) { *p = static_cast<T*>(emu::dev_alloc(n)); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T>
inline hipError_t hipExtMallocWithFlags(T** p, size_t n, unsigned) {
  *p = static_cast<T*>(emu::dev_alloc_shared(n));
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T>
inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
enum { hipHostRegisterMapped = 2 };
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }  // host memory IS device memory here
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipFree(void* p) { emu::dev_free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { emu::dev_free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
template <typename T>
inline hipError_t hipMemcpyFromSymbol(void* d, const T& sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
  memcpy(d, reinterpret_cast<const char*>(&sym) + off, n);
  return hipSuccess;
}
template <typename T>
inline hipError_t hipMemcpyToSymbol(T& sym, const void* s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
  memcpy(reinterpret_cast<char*>(&sym) + off, s, n);
  return hipSuccess;
}
#define HIP_SYMBOL(x) x

inline hipError_t hipStreamCreate(hipStream_t* s) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s);  // (defined below: waits for a resident kernel on the stream)
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }  // (launches have completed when they return)
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
  std::lock_guard<std::mutex> lk(emu::shared_mu());
  auto it = emu::shared_blocks().find(p);
  if (it == emu::shared_blocks().end()) return hipErrorNotSupported;  // (only fine-grained allocations are exportable here)
  emu::ipc_handle v{0x454d5549u, (int32_t)getpid(), it->second.fd, 0, it->second.len};
  static_assert(sizeof(v) <= sizeof(h->reserved), "handle fits");
  memset(h, 0, sizeof(*h));
  memcpy(h->reserved, &v, sizeof(v));
  return hipSuccess;
}
inline std::map<void*, size_t>& emu_ipc_mappings() {
  static std::map<void*, size_t> m;
  return m;
}
inline hipError_t hipIpcOpenMemHandle(void** out, hipIpcMemHandle_t h, unsigned) {
  emu::ipc_handle v;
  memcpy(&v, h.reserved, sizeof(v));
  if (v.magic != 0x454d5549u || v.pid == (int32_t)getpid()) return hipErrorNotSupported;  // (as on hardware: not where it was made)
  char path[64];
  snprintf(path, sizeof(path), "/proc/%d/fd/%d", v.pid, v.fd);
  const int fd = open(path, O_RDWR);
  if (fd < 0) return hipErrorNotSupported;
  void* p = mmap(nullptr, v.len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(emu::shared_mu());
  emu_ipc_mappings()[p] = v.len;
  *out = p;
  return hipSuccess;
}
inline hipError_t hipIpcCloseMemHandle(void* p) {
  std::lock_guard<std::mutex> lk(emu::shared_mu());
  auto it = emu_ipc_mappings().find(p);
  if (it != emu_ipc_mappings().end()) {
    munmap(p, it->second);
    emu_ipc_mappings().erase(it);
  }
  return hipSuccess;
}
inline hipError_t hipMemGetHandleForAddressRange(void*, hipDeviceptr_t, size_t, int, unsigned long long) { return hipErrorNotSupported; }

// HIP graphs of kernel nodes (what csrc/grdma_pair.hip builds for a streaming job: every node carries two
// pointer-sized parameters).  The nodes run in the order they were added, which is a valid order of the graph:
// a node's dependencies are always nodes added before it.
#define EMU_GRAPH_NODE_ARGS 10  // (= GRDMA_JOB_HOOK_ARGS: the product hands over that many parameter slots per node)
struct emu_graph_node {
  void* func;
  dim3 grid, block;
  uint64_t a[EMU_GRAPH_NODE_ARGS];
};
struct emu_graph {
  std::vector<emu_graph_node*> nodes;
};
struct emu_graph_exec {
  std::vector<emu_graph_node> nodes;
};
inline hipError_t hipGraphCreate(hipGraph_t* g, unsigned) { *g = new emu_graph(); return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t g) {
  if (g) {
    for (emu_graph_node* n : g->nodes) delete n;
    delete g;
  }
  return hipSuccess;
}
inline hipError_t hipGraphAddKernelNode(hipGraphNode_t* node, hipGraph_t g, const hipGraphNode_t*, size_t,
                                        const hipKernelNodeParams* p) {
  // (the product always hands over EMU_GRAPH_NODE_ARGS parameter slots of 8 bytes; kernels with fewer ignore the rest)
  emu_graph_node* n = new emu_graph_node{p->func, p->gridDim, p->blockDim, {}};
  for (int i = 0; i < EMU_GRAPH_NODE_ARGS; i++) n->a[i] = *static_cast<uint64_t*>(p->kernelParams[i]);
  g->nodes.push_back(n);
  *node = n;
  return hipSuccess;
}
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
  emu_graph_exec* x = new emu_graph_exec();
  for (emu_graph_node* n : g->nodes) x->nodes.push_back(*n);
  *e = x;
  return hipSuccess;
}
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  std::lock_guard<std::recursive_mutex> lk(emu::launch_mutex());
  for (const emu_graph_node& n : e->nodes) {
    // (integer / pointer parameters only: a kernel with fewer parameters ignores the registers and stack slots behind its own)
    typedef void (*fn_t)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
    fn_t f = reinterpret_cast<fn_t>(n.func);
    const emu_graph_node c = n;
    emu::launch(n.grid, n.block, [=] { f(c.a[0], c.a[1], c.a[2], c.a[3], c.a[4], c.a[5], c.a[6], c.a[7], c.a[8], c.a[9]); });
  }
  return hipSuccess;
}
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
// hipLaunchKernel by function pointer (the instrumented schedule of a streaming job): same convention as a graph
// node -- EMU_GRAPH_NODE_ARGS parameter slots of 8 bytes, a kernel with fewer ignores the rest
inline hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t, hipStream_t) {
  std::lock_guard<std::recursive_mutex> lk(emu::launch_mutex());
  typedef void (*fn_t)(uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t);
  fn_t f = reinterpret_cast<fn_t>(const_cast<void*>(func));
  uint64_t a[EMU_GRAPH_NODE_ARGS];
  for (int i = 0; i < EMU_GRAPH_NODE_ARGS; i++) a[i] = *static_cast<uint64_t*>(args[i]);
  emu::launch(grid, block, [=] { f(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9]); });
  return hipSuccess;
}

// A launch runs the whole grid under the emulator and returns when it is done.  One launch at a time: the
// __shared__ objects of a kernel are statics.  The exception is a RESIDENT kernel (the latency engine k_engine,
// which serves a mailbox until the host tells it to leave): it runs in a thread of its own, outside the launch
// lock, and hipStreamSynchronize on its stream waits for it.
namespace emu {
inline bool resident_kernel(const char* name) { return strcmp(name, "k_engine") == 0 || strcmp(name, "k_watch") == 0; }
struct async_table {
  std::mutex mu;
  std::map<hipStream_t, std::thread> running;
};
inline async_table& asyncs() {
  static async_table t;
  return t;
}
inline void launch_async(hipStream_t s, dim3 grid, dim3 block, std::function<void()> body) {
  std::lock_guard<std::mutex> lk(asyncs().mu);
  auto it = asyncs().running.find(s);
  if (it != asyncs().running.end()) {
    if (it->second.joinable()) it->second.join();
    asyncs().running.erase(it);
  }
  asyncs().running.emplace(s, std::thread([=] { emu::launch(grid, block, body); }));
}
inline void stream_wait(hipStream_t s) {
  std::thread t;
  {
    std::lock_guard<std::mutex> lk(asyncs().mu);
    auto it = asyncs().running.find(s);
    if (it == asyncs().running.end()) return;
    t = std::move(it->second);
    asyncs().running.erase(it);
  }
  if (t.joinable()) t.join();
}
}  // namespace emu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                          \
  do {                                                                                                       \
    if (emu::resident_kernel(#kernel)) {                                                                     \
      emu::launch_async(stream, dim3(grid), dim3(block), [=] { kernel(__VA_ARGS__); });                      \
    } else {                                                                                                 \
      std::lock_guard<std::recursive_mutex> emu_lk_(emu::launch_mutex());                                    \
      emu::launch(dim3(grid), dim3(block), [=] { kernel(__VA_ARGS__); });                                    \
    }                                                                                                        \
  } while (0)

inline hipError_t hipStreamSynchronize(hipStream_t s) {
  emu::stream_wait(s);
  return hipSuccess;
}
