// host_probe: measurements that size the host boundary of the endpoint.
//   1. kernel launch cost on the host, and launch -> flag-visible-in-pinned-memory latency
//   2. hipStreamWriteValue64 into pinned host memory (does it work, what does it cost)
//   3. a full-grid copy kernel reading pinned / registered host memory (the gather of host slices)
//      and writing pinned host memory (the scatter into a host arena): GB/s over PCIe
//   4. hipHostRegister / hipHostUnregister cost per size
//   5. can the host dereference fine-grained device memory (checked in a child process)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/host_probe tools/host_probe.hip
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("{\"error\": \"%s -> %s (line %d)\"}\n", #x, hipGetErrorString(e_), __LINE__);   \
      fflush(stdout);                                                                          \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_flag(volatile uint64_t* flag, uint64_t v) {
  __hip_atomic_store(const_cast<uint64_t*>(flag), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_nop() {}
__global__ __launch_bounds__(256) void k_copy16(u32x4* dst, const u32x4* src, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = __builtin_nontemporal_load(src + i);
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double copy_rate(u32x4* dst, const u32x4* src, size_t bytes, int blocks, hipStream_t s, int reps) {
  hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s, dst, src, bytes / 16);
  CK(hipStreamSynchronize(s));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, s, dst, src, bytes / 16);
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint64_t* flag;
  CK(hipHostMalloc((void**)&flag, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  memset(flag, 0, 4096);
  printf("{");
  // ---- 1. launch cost / flag latency
  {
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);
    CK(hipStreamSynchronize(s));
    const int N = 2000;
    double t0 = now_us();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);
    double t1 = now_us();
    CK(hipStreamSynchronize(s));
    double t2 = now_us();
    printf("\"launch_host_us\": %.2f, \"nop_kernel_back_to_back_us\": %.2f, ", (t1 - t0) / N, (t2 - t0) / N);
    std::vector<double> lat;
    for (int i = 1; i <= 500; i++) {
      double a = now_us();
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, flag, (uint64_t)i);
      while (*(volatile uint64_t*)flag != (uint64_t)i) {}
      lat.push_back(now_us() - a);
    }
    std::sort(lat.begin(), lat.end());
    printf("\"launch_to_flag_p50_us\": %.2f, \"launch_to_flag_p95_us\": %.2f, ", lat[250], lat[475]);
    // sync cost
    std::vector<double> sy;
    for (int i = 0; i < 300; i++) {
      double a = now_us();
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);
      CK(hipStreamSynchronize(s));
      sy.push_back(now_us() - a);
    }
    std::sort(sy.begin(), sy.end());
    printf("\"launch_plus_sync_p50_us\": %.2f, ", sy[150]);
  }
  // ---- 2. stream write value
  {
    flag[8] = 0;
    hipError_t e = hipStreamWriteValue64(s, flag + 8, 0x1234, 0);
    if (e == hipSuccess) {
      e = hipStreamSynchronize(s);
      printf("\"write_value64\": \"%s value=%llx\", ", hipGetErrorString(e), (unsigned long long)flag[8]);
      if (e == hipSuccess && flag[8] == 0x1234) {
        std::vector<double> lat;
        for (int i = 1; i <= 300; i++) {
          double a = now_us();
          hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s);
          (void)hipStreamWriteValue64(s, flag + 8, 0x10000 + i, 0);
          while (*(volatile uint64_t*)(flag + 8) != (uint64_t)(0x10000 + i)) {}
          lat.push_back(now_us() - a);
        }
        std::sort(lat.begin(), lat.end());
        printf("\"nop_then_write_value_p50_us\": %.2f, ", lat[150]);
      }
    } else {
      printf("\"write_value64\": \"%s\", ", hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  // ---- 3. PCIe rates with a compute kernel
  {
    const size_t big = 64u << 20;
    u32x4 *d, *hp;
    CK(hipMalloc((void**)&d, big));
    CK(hipHostMalloc((void**)&hp, big, hipHostMallocCoherent | hipHostMallocMapped));
    memset(hp, 1, big);
    void* raw = nullptr;
    if (posix_memalign(&raw, 4096, big) != 0) return 1;
    memset(raw, 2, big);
    double r0 = now_us();
    CK(hipHostRegister(raw, big, hipHostRegisterMapped));
    double r1 = now_us();
    void* rawd = nullptr;
    CK(hipHostGetDevicePointer(&rawd, raw, 0));
    printf("\"register_64MiB_us\": %.1f, ", r1 - r0);
    for (size_t sz : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
      for (int blocks : {64, 256, 1024}) {
        printf("\"h2d_pinned_%zuMiB_b%d_GBps\": %.1f, ", sz >> 20, blocks, copy_rate(d, hp, sz, blocks, s, 20));
      }
      printf("\"h2d_registered_%zuMiB_b256_GBps\": %.1f, ", sz >> 20, copy_rate(d, (u32x4*)rawd, sz, 256, s, 20));
      printf("\"d2h_pinned_%zuMiB_b256_GBps\": %.1f, ", sz >> 20, copy_rate(hp, d, sz, 256, s, 20));
      printf("\"d2h_pinned_%zuMiB_b1024_GBps\": %.1f, ", sz >> 20, copy_rate(hp, d, sz, 1024, s, 20));
    }
    // single launch latency for 1 MiB h2d and d2h (launch -> complete)
    {
      std::vector<double> a1, a2;
      for (int i = 0; i < 100; i++) {
        double a = now_us();
        hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s, d, hp, (size_t)(1 << 20) / 16);
        CK(hipStreamSynchronize(s));
        a1.push_back(now_us() - a);
        a = now_us();
        hipLaunchKernelGGL(k_copy16, dim3(256), dim3(256), 0, s, hp, d, (size_t)(1 << 20) / 16);
        CK(hipStreamSynchronize(s));
        a2.push_back(now_us() - a);
      }
      std::sort(a1.begin(), a1.end());
      std::sort(a2.begin(), a2.end());
      printf("\"h2d_1MiB_launch_sync_p50_us\": %.1f, \"d2h_1MiB_launch_sync_p50_us\": %.1f, ", a1[50], a2[50]);
      // hipMemcpyAsync both ways
      std::vector<double> m1, m2;
      for (int i = 0; i < 100; i++) {
        double a = now_us();
        CK(hipMemcpyAsync(d, hp, 1 << 20, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        m1.push_back(now_us() - a);
        a = now_us();
        CK(hipMemcpyAsync(hp, d, 1 << 20, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        m2.push_back(now_us() - a);
      }
      std::sort(m1.begin(), m1.end());
      std::sort(m2.begin(), m2.end());
      printf("\"memcpyasync_h2d_1MiB_p50_us\": %.1f, \"memcpyasync_d2h_1MiB_p50_us\": %.1f, ", m1[50], m2[50]);
    }
    CK(hipHostUnregister(raw));
    // ---- 4. registration cost per size
    for (size_t sz : {(size_t)16 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
      std::vector<double> reg, unreg;
      for (int i = 0; i < 30; i++) {
        double a = now_us();
        CK(hipHostRegister(raw, sz, hipHostRegisterMapped));
        double b = now_us();
        CK(hipHostUnregister(raw));
        double c = now_us();
        reg.push_back(b - a);
        unreg.push_back(c - b);
      }
      std::sort(reg.begin(), reg.end());
      std::sort(unreg.begin(), unreg.end());
      printf("\"register_%zuKiB_p50_us\": %.1f, \"unregister_%zuKiB_p50_us\": %.1f, ", sz >> 10, reg[15], sz >> 10, unreg[15]);
    }
    // host memcpy rate for comparison (1 MiB, cold-ish)
    {
      std::vector<uint8_t> a(64u << 20), b(64u << 20);
      memset(a.data(), 3, a.size());
      double t0 = now_us();
      for (int i = 0; i < 64; i++) memcpy(b.data() + ((size_t)i << 20), a.data() + ((size_t)i << 20), 1 << 20);
      double t1 = now_us();
      printf("\"host_memcpy_GBps\": %.1f, ", 64.0 * (1 << 20) / ((t1 - t0) * 1e-6) / 1e9);
      double t2 = now_us();
      for (int i = 0; i < 64; i++) memcpy(reinterpret_cast<uint8_t*>(hp) + ((size_t)i << 20), a.data() + ((size_t)i << 20), 1 << 20);
      double t3 = now_us();
      printf("\"host_memcpy_to_pinned_GBps\": %.1f, ", 64.0 * (1 << 20) / ((t3 - t2) * 1e-6) / 1e9);
      double t4 = now_us();
      uint64_t sum = 0;
      for (size_t i = 0; i < (64u << 20) / 8; i++) sum += reinterpret_cast<uint64_t*>(hp)[i];
      double t5 = now_us();
      printf("\"host_read_pinned_GBps\": %.1f, \"sum\": %llu, ", 64.0 * (1 << 20) / ((t5 - t4) * 1e-6) / 1e9, (unsigned long long)(sum & 0xff));
    }
  }
  // ---- 5. host access to fine-grained device memory
  {
    uint64_t* fg = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      printf("\"finegrained_alloc\": \"%s\", ", hipGetErrorString(e));
    } else {
      CK(hipMemset(fg, 0x5a, 4096));
      CK(hipDeviceSynchronize());
      fflush(stdout);
      pid_t pid = fork();
      if (pid == 0) {
        volatile uint64_t v = *(volatile uint64_t*)fg;
        _exit(v == 0x5a5a5a5a5a5a5a5aull ? 0 : 3);
      }
      int st = 0;
      waitpid(pid, &st, 0);
      printf("\"host_deref_finegrained\": \"%s\", ", WIFEXITED(st) ? (WEXITSTATUS(st) == 0 ? "readable" : "wrong value") : "fault");
    }
  }
  printf("\"done\": true}\n");
  return 0;
}
