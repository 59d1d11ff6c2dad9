// TEST INFRASTRUCTURE.  Self-test of the wave emulator (tests/cc/wave_emu.h) on small kernels whose results are
// known in closed form: the cross-lane primitives, the DPP prefix sums of csrc/grdma_devfn.h (the SAME functions the
// GPU runs), a block-wide scan across four waves, an LDS hand-off behind a barrier, early-exiting lanes, and the
// detection of a cross-lane operation that the lanes of a wave reach from different places.
#include "wave_emu.h"

#include <cstdio>
#include <vector>

#include "../../grpc-rdma_amd/csrc/grdma_devfn.h"

namespace {
__global__ void k_prims(uint64_t* out) {
  const int lane = threadIdx.x & 63;
  const uint64_t b = __ballot(lane % 3 == 0);
  const uint32_t up = __shfl_up((uint32_t)(lane * 10), 2, 64);
  const uint32_t x = __shfl_xor((uint32_t)lane, 5, 64);
  const uint64_t wide = __shfl((uint64_t)lane << 40, 63 - lane, 64);
  const int first = __builtin_amdgcn_readfirstlane(lane + 7);
  const int rl = __builtin_amdgcn_readlane(lane * 3, 17);
  out[threadIdx.x * 8 + 0] = b;
  out[threadIdx.x * 8 + 1] = up;
  out[threadIdx.x * 8 + 2] = x;
  out[threadIdx.x * 8 + 3] = wide;
  out[threadIdx.x * 8 + 4] = (uint64_t)first;
  out[threadIdx.x * 8 + 5] = (uint64_t)rl;
  out[threadIdx.x * 8 + 6] = wave_incl_scan_u32((uint32_t)(lane * lane + 1));
  out[threadIdx.x * 8 + 7] = wave_incl_scan((uint64_t)lane << 33, lane);
}

__global__ void k_block_scan(const uint64_t* in, uint64_t* out, uint64_t* total) {
  __shared__ uint64_t s_wave[PLAN_THREADS / 64];
  uint64_t t;
  out[threadIdx.x] = block_excl_scan(in[threadIdx.x], s_wave, &t);
  if (threadIdx.x == 0) *total = t;
}

__global__ void k_handoff(uint32_t* out) {
  __shared__ uint32_t s_x[256];
  s_x[threadIdx.x] = threadIdx.x * 2 + 1;
  __syncthreads();
  out[threadIdx.x] = s_x[255 - threadIdx.x];  // written by a lane of ANOTHER wave
  if ((threadIdx.x & 63) >= 32) return;       // half of every wave leaves
  const uint64_t b = __ballot(1);             // the rest still meets
  out[threadIdx.x] += (uint32_t)__builtin_popcountll(b) * 1000;
}

__global__ void k_divergent(uint32_t* out) {
  const int lane = threadIdx.x & 63;
  uint32_t v;
  if (lane < 10) v = __shfl((uint32_t)lane, 0, 64);      // the lanes of one wave at two different operations
  else v = (uint32_t)__builtin_popcountll(__ballot(lane & 1));
  out[lane] = v;
}
}  // namespace

extern "C" int emu_selftest(int which) {
  if (which == 99) {  // must abort with the emulator's diagnostic
    std::vector<uint32_t> o(64);
    uint32_t* po = o.data();
    emu::launch(dim3(1), dim3(64), [=] { k_divergent(po); });
    return 0;
  }
  int bad = 0;
  {
    std::vector<uint64_t> o(64 * 8);
    uint64_t* po = o.data();
    emu::launch(dim3(1), dim3(64), [=] { k_prims(po); });
    uint64_t expect_b = 0, s32 = 0, s64 = 0;
    for (int l = 0; l < 64; l++)
      if (l % 3 == 0) expect_b |= 1ull << l;
    for (int l = 0; l < 64; l++) {
      s32 += (uint32_t)(l * l + 1);
      s64 += (uint64_t)l << 33;
      bad += o[l * 8 + 0] != expect_b;
      bad += o[l * 8 + 1] != (uint64_t)(l >= 2 ? (l - 2) * 10 : l * 10);
      bad += o[l * 8 + 2] != (uint64_t)(l ^ 5);
      bad += o[l * 8 + 3] != ((uint64_t)(63 - l) << 40);
      bad += o[l * 8 + 4] != 7;
      bad += o[l * 8 + 5] != 51;
      bad += o[l * 8 + 6] != s32;
      bad += o[l * 8 + 7] != s64;
    }
  }
  {
    std::vector<uint64_t> in(256), o(256);
    uint64_t total = 0, acc = 0;
    for (int i = 0; i < 256; i++) in[i] = (uint64_t)i * i + 3;
    const uint64_t* pi = in.data();
    uint64_t *po = o.data(), *pt = &total;
    emu::launch(dim3(1), dim3(256), [=] { k_block_scan(pi, po, pt); });
    for (int i = 0; i < 256; i++) {
      bad += o[i] != acc;
      acc += in[i];
    }
    bad += total != acc;
  }
  {
    std::vector<uint32_t> o(256);
    uint32_t* po = o.data();
    emu::launch(dim3(1), dim3(256), [=] { k_handoff(po); });
    for (int i = 0; i < 256; i++) {
      const uint32_t base = (255 - i) * 2 + 1;
      bad += o[i] != ((i & 63) < 32 ? base + 32000 : base);
    }
  }
  return bad;
}
