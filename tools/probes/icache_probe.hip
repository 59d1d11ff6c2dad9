// Cold vs warm instruction fetch: a straight-line body of N distinct instructions run twice inside one launch by one
// wave (tools/probes: measurements behind DESIGN.md's planner notes, not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))
template <int KB>
__global__ void k(uint64_t* out, float a, float b, int passes) {
  float v = a + threadIdx.x;
  uint64_t t[5];
  t[0] = __builtin_amdgcn_s_memtime();
  for (int p = 0; p < passes; p++) {
    // each statement: one v_fma_f32 (8 bytes with a literal?  4-8 bytes) -- 256 x KB / 2 statements ~ KB KiB of code
#pragma unroll
    for (int r = 0; r < KB / 2; r++) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(b));) }
    asm volatile("s_nop 0" ::: "memory");
    t[p + 1] = __builtin_amdgcn_s_memtime() + ((uint64_t)(v == 12345.f));
  }
  if (threadIdx.x == 0) { for (int i = 0; i <= passes; i++) out[i] = t[i]; out[7] = (uint64_t)v; }
}
template <int KB> void run(uint64_t* d) {
  uint64_t h[8];
  for (int it = 0; it < 3; it++) {
    hipLaunchKernelGGL(k<KB>, dim3(1), dim3(64), 0, 0, d, 1.0f, 0.5f, 4);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%3d KiB body, launch %d: pass ticks %llu %llu %llu %llu\n", KB, it, (unsigned long long)(h[1] - h[0]), (unsigned long long)(h[2] - h[1]),
           (unsigned long long)(h[3] - h[2]), (unsigned long long)(h[4] - h[3]));
  }
}
int main() {
  uint64_t* d; hipMalloc(&d, 64);
  run<4>(d); run<16>(d); run<48>(d);
  // the s_memtime rate: ticks over a timed sleep
  return 0;
}
