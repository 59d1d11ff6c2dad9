// k_tx_index / txf_body (grdma_tx_fast.h): the send plan of a streaming job (PairPollable::Send arithmetic, pair.cc:645-734, and
// the rdma_flush cursor, rdma_bp_posix.cc:470-524) over an INDEX of the slice buffer being written.
//
// rdma_write hands one grpc_slice_buffer to a loop of Sends; each Send takes a prefix of what is left.  The
// general planner (k_tx_plan, grdma_tx_body.h) prices every Send from scratch: slice table into LDS, a block-wide
// prefix sum of the encoded sizes, the budget test, a second prefix sum for the tiles -- 30 us for 4095 records,
// all of it on the path between the credit that frees the ring and the gather.  But the encoded offset of slice k
// inside ANY Send of the same write is a difference of two entries of ONE prefix sum over the whole buffer:
//
//   k_tx_index   once per write (the first node of a job's step): enc_pre[k] = sum of 16 + round_up8(len_j), j < k;
//                len_pre[k] = sum of len_j; tile_pre[k] = sum of ceil(len_j / tile).
//   txf_body     per Send (one workgroup, sixteen records per thread), no scan at all: st_i = enc_pre[start + i] -
//                enc_pre[start] (+ a correction for the bytes of the first slice already sent); record i goes out
//                whole exactly when st_{i+1} + 8 <= min(staging, free) -- CalculateWritableSize
//                (ring_buffer.h:185-189) is round_down8(space - 24), so "len_i <= W(room - st_i)" is
//                "st_i + 16 + round_up8(len_i) + 8 <= room", monotone in i -- so the number of whole records is a
//                COUNT (one ballot per wave), the short record is the one behind them, and every thread writes its
//                segment and tile-prefix entry straight from the index.
//
// Same plan, wire requests, cursor, counters and result block as k_tx_plan.  What it does not take (a slice of
// zero bytes anywhere in the buffer -- it ends a Send like the reference's `break` --, a direct wire, the latency
// path) is left to the general planner, which k_tx_plan_job runs in the same launch when the body returns false.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"

namespace {

#define TXF_THREADS 1024
#define TXF_PER 4
#define TXF_WAVES (TXF_THREADS / 64)

// exclusive scan of a u64 per thread over the 1024-thread block
__device__ __forceinline__ uint64_t txf_scan64(uint64_t v, uint64_t* s_w, uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t incl = wave_incl_scan(v, lane);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TXF_WAVES; w++) {
    const uint64_t s = s_w[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// ---------------------------------------------------------------------------------------------------------
// k_tx_index: prefix sums over the slice buffer of a write.  One workgroup per connection; thread t owns the
// contiguous run [t * per, (t + 1) * per).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TXF_THREADS) void k_tx_index(grdma_txf_ctl* ctls) {
  grdma_txf_ctl* ctl = &ctls[blockIdx.x];
  const uint32_t tid = threadIdx.x;
  const grdma_sge* sl = ctl->slices;
  const uint64_t n = ctl->n;
  const uint32_t ts = ctl->tile_shift;
  uint64_t* const enc_pre = ctl->enc_pre;
  uint64_t* const len_pre = ctl->len_pre;
  uint32_t* const tile_pre = ctl->tile_pre;
  __shared__ uint64_t s_w[TXF_WAVES];
  __shared__ uint32_t s_bad;
  if (tid == 0) s_bad = 0;
  const uint64_t per = (n + TXF_THREADS - 1) / TXF_THREADS;
  const uint64_t k0 = (uint64_t)tid * per, k1 = k0 + per < n ? k0 + per : n;
  uint64_t e = 0, l = 0, t = 0;
  bool bad = false;
  for (uint64_t k = k0; k < k1; k++) {
    const uint64_t len = sl[k].len;
    bad |= len == 0 || len >= (1ull << 31);
    e += enc_size(len);
    l += len;
    t += (len + (1ull << ts) - 1) >> ts;
  }
  __syncthreads();
  if (bad) s_bad = 1;
  uint64_t te, tl, tt;
  uint64_t xe = txf_scan64(e, s_w, &te);
  uint64_t xl = txf_scan64(l, s_w, &tl);
  uint64_t xt = txf_scan64(t, s_w, &tt);
  for (uint64_t k = k0; k < k1; k++) {
    const uint64_t len = sl[k].len;  // (second pass over my run: served by the cache)
    enc_pre[k] = xe;
    len_pre[k] = xl;
    tile_pre[k] = (uint32_t)xt;
    xe += enc_size(len);
    xl += len;
    xt += (len + (1ull << ts) - 1) >> ts;
  }
  if (tid == 0) {
    enc_pre[n] = te;
    len_pre[n] = tl;
    tile_pre[n] = (uint32_t)tt;
    ctl->valid = (s_bad == 0 && tt < (1ull << 32)) ? 1u : 0u;
  }
}

}  // namespace

extern "C" {
__attribute__((visibility("hidden"))) const void* grdma_kernel_fn_tx_index(void) { return reinterpret_cast<const void*>(&k_tx_index); }
__attribute__((visibility("hidden"))) uint32_t grdma_tx_index_threads(void) { return TXF_THREADS; }
__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_index(grdma_txf_ctl* d_ctls, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_index, dim3(n), dim3(TXF_THREADS), 0, s, d_ctls);
  return hipGetLastError();
}
}
