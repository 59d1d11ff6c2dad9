// copy_probe: what one MI355X sustains for the byte-moving shapes of the data plane, measured
// with HIP events on a dedicated stream.  Printed next to the 8 TB/s HBM3E spec in bench.py /
// DESIGN.md (SURVEY.md section 8d asks for a measured DtoD figure beside the datasheet one).
//
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.hip
//   run:   tools/copy_probe [MiB per buffer, default 1024] [json]
//
// Shapes: hipMemcpyDtoD; a grid-strided 16-byte copy (plain / nontemporal / sc1 write-through
// loads+stores through buffer descriptors), 4 or 8 units in flight per lane; the K4 shape
// (copy out + zero the source behind); the same into fine-grained (uncached) destination
// memory, which is what a ring registered for a NIC would be.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { M_PLAIN = 0, M_NT = 1, M_SC1 = 2 };

// each wave moves TILE = 64 lanes x 16 B x U bytes per step; waves walk tiles grid-strided
template <int U, int MODE, bool ZERO_SRC>
__global__ __launch_bounds__(256) void k_copy(uint8_t* dst, uint8_t* src, uint64_t bytes) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  constexpr uint64_t TILE = 64ull * 16 * U;
  const uint64_t ntiles = bytes / TILE;
  for (uint64_t t = wave; t < ntiles; t += nwaves) {
    // wave-uniform tile base (the compiler cannot prove it: readfirstlane both halves)
    const uint64_t off = t * TILE;
    const uint64_t sb = (uint64_t)src + off, db = (uint64_t)dst + off;
    u32x4 v[U];
    if (MODE == M_SC1) {
      // (readfirstlane returns int: cast each half to uint32_t before widening, or a low half with
      // bit 31 set sign-extends over the high half)
      const uint64_t sbu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb);
      const uint64_t dbu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(db >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)db);
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)sbu, 0, (uint32_t)TILE, 0x00020000);
      auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dbu, 0, (uint32_t)TILE, 0x00020000);
#pragma unroll
      for (int i = 0; i < U; i++) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 1024, 0, 16);
#pragma unroll
      for (int i = 0; i < U; i++) __builtin_amdgcn_raw_buffer_store_b128(v[i], rd, lane * 16 + i * 1024, 0, 16);
      if (ZERO_SRC) {
#pragma unroll
        for (int i = 0; i < U; i++) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0, 0, 0, 0}, rs, lane * 16 + i * 1024, 0, 16);
      }
    } else {
      const u32x4* s = reinterpret_cast<const u32x4*>(sb);
      u32x4* d = reinterpret_cast<u32x4*>(db);
#pragma unroll
      for (int i = 0; i < U; i++) v[i] = MODE == M_NT ? __builtin_nontemporal_load(s + lane + i * 64) : s[lane + i * 64];
#pragma unroll
      for (int i = 0; i < U; i++) {
        if (MODE == M_NT) __builtin_nontemporal_store(v[i], d + lane + i * 64); else d[lane + i * 64] = v[i];
      }
      if (ZERO_SRC) {
        u32x4* z = reinterpret_cast<u32x4*>(sb);
#pragma unroll
        for (int i = 0; i < U; i++) z[lane + i * 64] = u32x4{0, 0, 0, 0};
      }
    }
  }
}

struct result { std::string name; double gbps_payload, gbps_traffic, us; };

int main(int argc, char** argv) {
  const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1024;
  const bool json = argc > 2 && !strcmp(argv[2], "json");
  const uint64_t bytes = mib << 20;
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint8_t *a, *b, *fg = nullptr;
  CK(hipMalloc((void**)&a, bytes));
  CK(hipMalloc((void**)&b, bytes));
  const bool have_fg = hipExtMallocWithFlags((void**)&fg, bytes, hipDeviceMallocFinegrained) == hipSuccess;
  if (!have_fg) { (void)hipGetLastError(); fg = nullptr; }
  CK(hipMemsetAsync(a, 0x5a, bytes, st));
  CK(hipMemsetAsync(b, 0, bytes, st));
  if (fg) CK(hipMemsetAsync(fg, 0, bytes, st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  if (!json) {
    printf("%llu MiB per buffer, %d CUs, fine-grained alloc %s\n", (unsigned long long)mib, cus, fg ? "ok" : "unavailable");
    fflush(stdout);
  }
  std::vector<result> out;
  std::vector<uint8_t> probe(1 << 16);
  auto timeit = [&](const char* name, double traffic_factor, auto&& fn) {
    CK(hipMemsetAsync(a, 0x5a, bytes, st));
    CK(hipMemsetAsync(b, 0, bytes, st));
    fn();  // warm, and checked: the tail of the destination must hold the pattern
    CK(hipStreamSynchronize(st));
    if (strncmp(name, "hipMemset", 9) != 0 && !strstr(name, "finegrained")) {
      CK(hipMemcpy(probe.data(), b + bytes - probe.size(), probe.size(), hipMemcpyDeviceToHost));
      for (uint8_t v : probe) if (v != 0x5a) { printf("%s: WRONG DATA in the destination\n", name); fflush(stdout); break; }
    }
    CK(hipMemsetAsync(a, 0x5a, bytes, st));
    CK(hipStreamSynchronize(st));
    const int reps = 5;
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) fn();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    out.push_back({name, bytes / us / 1e3, traffic_factor * bytes / us / 1e3, us});
    if (!json) {  // progressively: a fault in a later variant must not lose the earlier rows
      printf("%-40s %9.1f us  payload %8.1f GB/s  traffic %8.1f GB/s\n", name, us, bytes / us / 1e3, traffic_factor * bytes / us / 1e3);
      fflush(stdout);
    }
  };
  timeit("hipMemcpyDtoD", 2, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st)); });
  timeit("hipMemsetD8", 1, [&] { CK(hipMemsetAsync(b, 0, bytes, st)); });
  const int grids[] = {cus * 4, cus * 8};
  for (int g : grids) {
    char nm[128];
#define RUN(U, MODE, Z, label, dstp, tf)                                                        \
    snprintf(nm, sizeof nm, "%s u%d grid%d", label, U, g);                                        \
    timeit(nm, tf, [&] { hipLaunchKernelGGL((k_copy<U, MODE, Z>), dim3(g), dim3(256), 0, st, dstp, a, bytes); });
    RUN(4, M_PLAIN, false, "copy plain", b, 2)
    RUN(8, M_PLAIN, false, "copy plain", b, 2)
    RUN(4, M_NT, false, "copy nt", b, 2)
    RUN(8, M_NT, false, "copy nt", b, 2)
    RUN(4, M_SC1, false, "copy sc1", b, 2)
    RUN(8, M_SC1, false, "copy sc1", b, 2)
    RUN(4, M_NT, true, "copy+zero-src nt", b, 3)
    RUN(8, M_NT, true, "copy+zero-src nt", b, 3)
    RUN(8, M_SC1, true, "copy+zero-src sc1", b, 3)
    if (fg) {
      RUN(8, M_NT, false, "copy nt -> finegrained", fg, 2)
      RUN(8, M_SC1, false, "copy sc1 -> finegrained", fg, 2)
    }
  }
  if (json) {
    printf("{\"MiB\": %llu, \"cus\": %d, \"finegrained\": %s, \"rows\": [", (unsigned long long)mib, cus, fg ? "true" : "false");
    for (size_t i = 0; i < out.size(); i++)
      printf("%s{\"name\": \"%s\", \"payload_GBps\": %.1f, \"traffic_GBps\": %.1f, \"us\": %.1f}", i ? ", " : "",
             out[i].name.c_str(), out[i].gbps_payload, out[i].gbps_traffic, out[i].us);
    printf("]}\n");
  } else {
    printf("done\n");
  }
  return 0;
}
