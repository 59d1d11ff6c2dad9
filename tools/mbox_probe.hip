// Where should a resident kernel's doorbell live?  One wave polls a 64-bit word and echoes it into a second word; the
// host writes the first and spins on the second: round trip per placement --
//   host/host:     both words in pinned host memory (what the latency engine's mailbox is today)
//   device/host:   the doorbell in fine-grained DEVICE memory written by the host through the BAR, the echo in host memory
//   device/device: both in device memory (the host reads the echo over the BAR)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mbox_probe tools/mbox_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

__global__ void k_echo(volatile uint64_t* bell, volatile uint64_t* echo, volatile uint64_t* quit) {
  uint64_t last = 0;
  for (uint64_t spins = 0; spins < (1ull << 34); spins++) {
    const uint64_t v = __hip_atomic_load((uint64_t*)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != last) {
      last = v;
      __hip_atomic_store((uint64_t*)echo, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v == ~0ull) return;
    }
    if ((spins & 1023) == 1023 && __hip_atomic_load((uint64_t*)quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
  }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static double run(const char* name, volatile uint64_t* bell_h, uint64_t* bell_d, volatile uint64_t* echo_h, uint64_t* echo_d,
                  volatile uint64_t* quit_h, uint64_t* quit_d, int iters) {
  hipStream_t s;
  hipStreamCreate(&s);
  *quit_h = 0;
  hipLaunchKernelGGL(k_echo, dim3(1), dim3(64), 0, s, (volatile uint64_t*)bell_d, (volatile uint64_t*)echo_d, (volatile uint64_t*)quit_d);
  std::vector<double> us;
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  if (sigsetjmp(g_jmp, 1)) {
    printf("%-14s host access to that memory faults\n", name);
    *quit_h = 1;
    hipStreamSynchronize(s);
    return -1;
  }
  for (int i = 1; i <= iters + 200; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    *bell_h = (uint64_t)i;
    uint64_t spins = 0;
    while (*echo_h != (uint64_t)i)
      if (++spins > (1ull << 28)) { printf("%-14s no echo\n", name); *quit_h = 1; hipStreamSynchronize(s); return -1; }
    const auto t1 = std::chrono::steady_clock::now();
    if (i > 200) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  *bell_h = ~0ull;
  hipStreamSynchronize(s);
  std::sort(us.begin(), us.end());
  printf("%-14s p50 %.2f us  p95 %.2f us\n", name, us[us.size() / 2], us[us.size() * 95 / 100]);
  hipStreamDestroy(s);
  return us[us.size() / 2];
}

int main() {
  uint64_t *hb, *he, *hq, *db = nullptr, *de = nullptr;
  hipHostMalloc((void**)&hb, 4096, hipHostMallocCoherent | hipHostMallocMapped);
  hipHostMalloc((void**)&he, 4096, hipHostMallocCoherent | hipHostMallocMapped);
  hipHostMalloc((void**)&hq, 4096, hipHostMallocCoherent | hipHostMallocMapped);
  *hb = *he = *hq = 0;
  run("host/host", hb, hb, he, he, hq, hq, 20000);
  if (hipExtMallocWithFlags((void**)&db, 4096, hipDeviceMallocFinegrained) == hipSuccess &&
      hipExtMallocWithFlags((void**)&de, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
    hipMemset(db, 0, 4096);
    hipMemset(de, 0, 4096);
    hipDeviceSynchronize();
    *he = 0;
    run("device/host", db, db, he, he, hq, hq, 20000);
    run("device/device", db, db, de, de, hq, hq, 20000);
  } else {
    printf("no fine-grained device memory\n");
  }
  return 0;
}
