// cumask_probe: where do the workgroups of a CU-masked stream run?  For a handful of masks the kernel
// records (XCC_ID, SE, SH, CU) of every workgroup; the host prints the distinct places per mask, and
// whether a one-workgroup kernel on a masked stream starts while a machine-filling kernel on the
// complementary mask is resident (the reservation the streaming job's planners need).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x)                                                                                \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      printf("error: %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);            \
      exit(1);                                                                               \
    }                                                                                        \
  } while (0)

__global__ __launch_bounds__(256) void k_where(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < (int64_t)spin) __builtin_amdgcn_s_sleep(8);
}

// fills the machine for `ticks` (holding ~128 VGPRs like the copy kernels would not matter here: wave slots do)
__global__ __launch_bounds__(256) void k_hog(uint64_t ticks, uint64_t* sink) {
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  uint64_t x = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) x += __builtin_amdgcn_s_memtime();
  if (x == 1) *sink = x;
}
__global__ void k_stamp(uint64_t* out) {
  if (threadIdx.x == 0) *out = __builtin_amdgcn_s_memtime();
}

static void places(const char* tag, const std::vector<uint32_t>& mask, int blocks) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) {
    printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", tag, hipGetErrorString(e));
    return;
  }
  uint32_t* d;
  CK(hipMalloc(&d, blocks * 4));
  CK(hipMemset(d, 0xFF, blocks * 4));
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 20000);
  CK(hipStreamSynchronize(s));
  std::vector<uint32_t> h(blocks);
  CK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
  std::set<uint32_t> cus;
  std::set<uint32_t> xccs;
  for (uint32_t v : h) {
    const uint32_t xcc = v >> 16, cu = (v >> 8) & 0xF, sh = (v >> 12) & 1, se = (v >> 13) & 7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
    xccs.insert(xcc);
  }
  printf("%s: %zu distinct CUs on %zu XCCs:", tag, cus.size(), xccs.size());
  int n = 0;
  for (uint32_t c : cus) {
    if (n++ < 24) printf(" x%u.se%u.sh%u.cu%u", c >> 12, (c >> 8) & 7, (c >> 4) & 1, c & 0xF);
  }
  printf("\n");
  CK(hipFree(d));
  CK(hipStreamDestroy(s));
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  const int words = (p.multiProcessorCount + 31) / 32;
  auto full = [&] { return std::vector<uint32_t>(words, 0xFFFFFFFFu); };
  auto none = [&] { return std::vector<uint32_t>(words, 0u); };
  {
    auto m = full();
    places("all", m, 2048);
  }
  {
    auto m = none();
    m[0] = 0xFF;
    places("bits 0-7", m, 512);
  }
  {
    auto m = none();
    m[0] = 0x1;
    places("bit 0", m, 64);
  }
  {
    auto m = none();
    m[0] = 0x101;
    places("bits 0,8", m, 64);
  }
  {
    auto m = none();
    m[0] = 0xFFFF;
    places("bits 0-15", m, 512);
  }
  {
    auto m = full();
    m[0] = 0xFFFF0000u;
    places("all but bits 0-15", m, 4096);
  }
  // reservation: hog on the complement, then a one-workgroup kernel on the reserved CUs
  {
    auto mc = full();
    mc[0] = 0xFFFF0000u;
    auto mp = none();
    mp[0] = 0xFFFF;
    hipStream_t sc, sp, sn;
    CK(hipExtStreamCreateWithCUMask(&sc, (uint32_t)mc.size(), mc.data()));
    CK(hipExtStreamCreateWithCUMask(&sp, (uint32_t)mp.size(), mp.data()));
    CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    uint64_t *d, *h;
    CK(hipMalloc(&d, 64));
    CK(hipHostMalloc(&h, 64));
    for (int variant = 0; variant < 2; variant++) {
      hipStream_t probe = variant == 0 ? sp : sn;
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_hog, dim3(8192), dim3(256), 0, sc, 400000ull /* ~ms at 100 MHz.. or shader clock */, d);
      // give the hog time to become resident
      while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 300) {
      }
      const auto t1 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, probe, h);
      CK(hipStreamSynchronize(probe));
      const auto t2 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(sc));
      const auto t3 = std::chrono::steady_clock::now();
      printf("%s stream: one-workgroup kernel done %.1f us after its launch; the hog ran %.1f us in total\n",
             variant == 0 ? "masked (reserved CUs)" : "unmasked", std::chrono::duration<double, std::micro>(t2 - t1).count(),
             std::chrono::duration<double, std::micro>(t3 - t0).count());
    }
  }
  return 0;
}
