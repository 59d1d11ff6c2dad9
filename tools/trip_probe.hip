// trip_probe: what one link of a streaming round's launch chain costs on this stack -- an empty kernel between two HIP
// events, and a one-workgroup kernel with K DEPENDENT memory round trips behind a producer kernel that has just
// rewritten the words it reads (the planner's situation: every word it needs was written by the launch before, on
// other CUs).  Also: the same K trips when the first address comes from the kernel arguments instead of a load, a
// 4096-record probe from 1 / 4 / 16 workgroups, and 400 KB of plan stores from 1 / 4 / 16 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/trip_probe tools/trip_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty(int) {}
__global__ __launch_bounds__(1024) void k_empty1024(int) {}

// producer: rewrites the chain (value i -> next index) and the probe area from the whole chip
__global__ void k_produce(uint64_t* chain, uint32_t n, uint32_t stride_words, uint64_t* ring, uint64_t ring_words, uint32_t salt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) chain[(uint64_t)i * stride_words] = ((uint64_t)(i + 1) % n) * stride_words + (uint64_t)salt * 0;  // next
  for (uint64_t w = i; w < ring_words; w += (uint64_t)gridDim.x * blockDim.x) ring[w] = w ^ salt;
}

// K dependent loads by every thread of one 1024-thread workgroup (all threads chase the same chain: one request per step)
template <bool SC1>
__global__ __launch_bounds__(1024) void k_chase(const uint64_t* chain, uint64_t start, int K, uint64_t* out) {
  uint64_t p = start;
  for (int k = 0; k < K; k++) {
    if (SC1) p = __hip_atomic_load(&chain[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p = chain[p];
  }
  if (threadIdx.x == 0) out[0] = p;
}

// probe: V records, one 16-byte load each at stride `step` bytes, from G workgroups of 1024 threads; reduce; last lane stores
__global__ __launch_bounds__(1024) void k_probe(const uint64_t* ring, uint32_t V, uint32_t step_words, uint64_t* out) {
  const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
  uint64_t acc = 0;
  for (uint32_t r = i; r < V; r += gridDim.x * 1024) {
    const uint64_t a = __hip_atomic_load(&ring[(uint64_t)r * step_words], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t b = __hip_atomic_load(&ring[(uint64_t)r * step_words + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += a ^ b;
  }
  __shared__ uint64_t s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  if (acc == 0x1234567) atomicAdd((unsigned long long*)&s, 1ull);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// emit: `bytes` of 32-byte descriptors from G workgroups
__global__ __launch_bounds__(1024) void k_emit(uint64_t* plan, uint32_t nseg) {
  for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < nseg; i += gridDim.x * 1024) {
    plan[4ull * i] = i; plan[4ull * i + 1] = i + 1; plan[4ull * i + 2] = 16384; plan[4ull * i + 3] = 7;
  }
}


// straight-line code executed ONCE against the same instruction count as a loop over a short body: eight independent
// accumulators, x = (x ^ (x >> 3)) + literal (nothing folds, nothing waits on the instruction before): what a COLD
// instruction stream costs per byte (the planners are a few thousand instructions executed once per launch)
#define STEP8(i) \
  a0 = (a0 ^ (a0 >> 3)) + (0x9E3779B1u + 8u * (i)); a1 = (a1 ^ (a1 >> 5)) + (0x85EBCA77u + 8u * (i)); \
  a2 = (a2 ^ (a2 >> 7)) + (0xC2B2AE3Du + 8u * (i)); a3 = (a3 ^ (a3 >> 9)) + (0x27D4EB2Fu + 8u * (i)); \
  a4 = (a4 ^ (a4 >> 11)) + (0x165667B1u + 8u * (i)); a5 = (a5 ^ (a5 >> 13)) + (0xD3A2646Cu + 8u * (i)); \
  a6 = (a6 ^ (a6 >> 15)) + (0xFD7046C5u + 8u * (i)); a7 = (a7 ^ (a7 >> 17)) + (0xB55A4F09u + 8u * (i));
template <int N>  // N x 8 steps of 3 instructions, all unrolled
__global__ __launch_bounds__(1024) void k_code(uint32_t seed, uint32_t* out) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#pragma unroll
  for (int i = 0; i < N; i++) { STEP8(i) }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = a0;
}
template <int BODY>  // the same work as a loop whose body is BODY x 8 steps
__global__ __launch_bounds__(1024) void k_loop(uint32_t seed, uint32_t iters, uint32_t* out) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#pragma unroll 1
  for (uint32_t r = 0; r < iters; r++) {
#pragma unroll
    for (int i = 0; i < BODY; i++) { STEP8(i) }
  }
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = a0;
}

int main() {
  const uint32_t n = 64, stride_words = 4096 / 8 * 5;  // chain entries 20 KB apart
  uint64_t *chain, *ring, *out, *plan;
  const uint64_t ring_words = (64ull << 20) / 8;
  CK(hipMalloc((void**)&chain, (uint64_t)n * stride_words * 8));
  CK(hipMalloc((void**)&ring, ring_words * 8));
  CK(hipMalloc((void**)&out, 4096));
  CK(hipMalloc((void**)&plan, 16384ull * 32));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto produce = [&](uint32_t salt) { hipLaunchKernelGGL(k_produce, dim3(2048), dim3(256), 0, 0, chain, n, stride_words, ring, ring_words, salt); };
  // every configuration as a CHAIN of 40 launches on the stream: [small producer][consumer] x 20, total / 20, minus the
  // producer-only chain -- a single launch between two events has a ~6 us floor that hides the first microseconds of a kernel
  auto small_produce = [&](uint32_t salt) { hipLaunchKernelGGL(k_produce, dim3(64), dim3(256), 0, 0, chain, n, stride_words, ring, (uint64_t)(4096ull * 1024), salt); };
  auto chain_time = [&](auto&& launch) {
    std::vector<float> v;
    for (int it = 0; it < 12; it++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < 20; r++) { small_produce(r); launch(); }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) v.push_back(ms * 1e3f / 20);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  const float base = chain_time([&] {});
  printf("producer-only chain: %.2f us per launch\n", base);
  auto timeit = [&](const char* name, auto&& launch) {
    const float t = chain_time(launch);
    printf("%-44s %6.2f us per launch in a chain (pair %.2f)\n", name, t - base, t);
  };
  timeit("empty kernel 1 x 64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, 0); });
  timeit("empty kernel 1 x 1024", [&] { hipLaunchKernelGGL(k_empty1024, dim3(1), dim3(1024), 0, 0, 0); });
  timeit("empty kernel 16 x 1024", [&] { hipLaunchKernelGGL(k_empty1024, dim3(16), dim3(1024), 0, 0, 0); });
  timeit("two empty kernels", [&] { hipLaunchKernelGGL(k_empty1024, dim3(1), dim3(1024), 0, 0, 0); hipLaunchKernelGGL(k_empty1024, dim3(1), dim3(1024), 0, 0, 0); });
  timeit("straight-line 16 x 8 steps (~3 KB)", [&] { hipLaunchKernelGGL(k_code<16>, dim3(1), dim3(1024), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 64 x 8 steps (~12 KB)", [&] { hipLaunchKernelGGL(k_code<64>, dim3(1), dim3(1024), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 128 x 8 steps (~25 KB)", [&] { hipLaunchKernelGGL(k_code<128>, dim3(1), dim3(1024), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 256 x 8 steps (~50 KB)", [&] { hipLaunchKernelGGL(k_code<256>, dim3(1), dim3(1024), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 256 x 8, 256 threads", [&] { hipLaunchKernelGGL(k_code<256>, dim3(1), dim3(256), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 256 x 8, 64 threads", [&] { hipLaunchKernelGGL(k_code<256>, dim3(1), dim3(64), 0, 0, 7u, (uint32_t*)out); });
  timeit("straight-line 256 x 8, 4 WG x 1024", [&] { hipLaunchKernelGGL(k_code<256>, dim3(4), dim3(1024), 0, 0, 7u, (uint32_t*)out); });
  timeit("loop 16 x (16 x 8 steps): same work as 256", [&] { hipLaunchKernelGGL(k_loop<16>, dim3(1), dim3(1024), 0, 0, 7u, 16u, (uint32_t*)out); });
  timeit("loop 16 x (16 x 8 steps), 256 threads", [&] { hipLaunchKernelGGL(k_loop<16>, dim3(1), dim3(256), 0, 0, 7u, 16u, (uint32_t*)out); });
  timeit("loop 16 x (16 x 8 steps), 64 threads", [&] { hipLaunchKernelGGL(k_loop<16>, dim3(1), dim3(64), 0, 0, 7u, 16u, (uint32_t*)out); });
  for (int K : {0, 4, 8}) {
    char nm[64]; snprintf(nm, sizeof nm, "chase K=%d plain loads", K);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chase<false>, dim3(1), dim3(1024), 0, 0, chain, 0, K, out); });
    snprintf(nm, sizeof nm, "chase K=%d agent-scope (sc1) loads", K);
    timeit(nm, [&] { hipLaunchKernelGGL(k_chase<true>, dim3(1), dim3(1024), 0, 0, chain, 0, K, out); });
  }
  for (int G : {1, 2, 4, 8, 16}) {
    char nm[64]; snprintf(nm, sizeof nm, "probe 4096 records (8 KB apart), %d WG", G);
    timeit(nm, [&] { hipLaunchKernelGGL(k_probe, dim3(G), dim3(1024), 0, 0, ring, 4096, 1024, out); });
  }
  for (int G : {1, 2, 4, 8, 16}) {
    char nm[64]; snprintf(nm, sizeof nm, "emit 12288 segs (393 KB), %d WG", G);
    timeit(nm, [&] { hipLaunchKernelGGL(k_emit, dim3(G), dim3(1024), 0, 0, plan, 12288); });
  }
  // a graph of 20 empty kernels: per-node cost
  {
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t prev = nullptr;
    int zero = 0; void* args[1] = {&zero};
    for (int i = 0; i < 20; i++) {
      hipKernelNodeParams np = {}; np.func = (void*)k_empty1024; np.gridDim = dim3(1); np.blockDim = dim3(1024); np.kernelParams = args;
      hipGraphNode_t nd; CK(hipGraphAddKernelNode(&nd, g, prev ? &prev : nullptr, prev ? 1 : 0, &np)); prev = nd;
    }
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ex, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of 20 empty 1x1024 kernels: %.2f us per node\n", ms * 1e3 / 400);
  }
  return 0;
}
