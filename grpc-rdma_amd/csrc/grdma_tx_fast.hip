// k_tx_index / txf_body (grdma_tx_fast.h): the send plan of a streaming job (PairPollable::Send arithmetic, pair.cc:645-734, and
// the rdma_flush cursor, rdma_bp_posix.cc:470-524) over an INDEX of the slice buffer being written.
//
// rdma_write hands one grpc_slice_buffer to a loop of Sends; each Send takes a prefix of what is left.  The
// general planner (k_tx_plan, grdma_tx_body.h) prices every Send from scratch: slice table into LDS, a block-wide
// prefix sum of the encoded sizes, the budget test, a second prefix sum for the tiles -- 30 us for 4095 records,
// all of it on the path between the credit that frees the ring and the gather.  But the encoded offset of slice k
// inside ANY Send of the same write is a difference of two entries of ONE prefix sum over the whole buffer:
//
//   k_tx_index   once per write (the first node of a job's step; grid-wide, no inter-workgroup wait): enc_pre[k] = sum of 16 + round_up8(len_j), j < k;
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
// zero bytes anywhere in the buffer -- it ends a Send like the reference's `break` --, the latency
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

// ---------------------------------------------------------------------------------------------------------
// k_tx_index: prefix sums over the slice buffer of a write.  Grid (ceil(n / 1024), connections): workgroup b owns
// entries [1024 b, 1024 (b + 1)), one per thread, and needs no word from any other workgroup -- it sums the entries IN
// FRONT of its own itself (b coalesced 16-byte loads per thread, at most 32 for the bench's 33 280 slices; 8.7 MB
// of L2 reads in total) instead of waiting for its predecessors' totals.  A single workgroup walking 33 slices per
// thread took 116-123 us per step (profiles/r03: 14 % of the step); this form is a handful of microseconds.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TXF_THREADS) void k_tx_index(grdma_txf_ctl* ctls) {
  grdma_txf_ctl* ctl = &ctls[blockIdx.y];
  const uint32_t tid = threadIdx.x;
  const grdma_sge* sl = ctl->slices;
  const uint64_t n = ctl->n;
  const uint32_t ts = ctl->tile_shift;
  const uint64_t first = (uint64_t)blockIdx.x * TXF_THREADS;
  if (first >= n && !(first == 0 && n == 0)) return;  // (grids are sized for the longest list of the job)
  __shared__ uint32_t s_bad;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  auto tiles_of = [&](uint64_t len) -> uint64_t { return (len + (1ull << ts) - 1) >> ts; };
  // what lies in front of this workgroup's entries (and whether any of it is unusable)
  uint64_t pe = 0, pl = 0, pt = 0;
  bool bad = false;
  // (eight loads in flight per pass -- thirty-two were slower, most of them clamped duplicates for the front workgroups: a plain loop is a chain of load -> add, one L2 round trip per entry -- 16 us for
  // the workgroup that has 32 of them in front)
  for (uint64_t k0 = tid; k0 < first; k0 += 8 * TXF_THREADS) {
    uint64_t lens[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t k = k0 + (uint64_t)j * TXF_THREADS;
      lens[j] = sl[k < first ? k : 0].len;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t k = k0 + (uint64_t)j * TXF_THREADS;
      if (k >= first) continue;
      const uint64_t len = lens[j];
      bad |= len == 0 || len >= (1ull << 31);
      pe += enc_size(len);
      pl += len;
      pt += tiles_of(len);
    }
  }
  const uint64_t k = first + tid;
  const uint64_t len = k < n ? sl[k].len : 0;
  if (k < n) bad |= len == 0 || len >= (1ull << 31);
  const uint64_t e = k < n ? enc_size(len) : 0, l = len, t = k < n ? tiles_of(len) : 0;
  // one pass for all six numbers: wave scans of my entry's three values (the other lanes' values in front of mine)
  // and wave sums of what lies in front of the workgroup, combined across the 16 waves through LDS
  __shared__ uint64_t s_x[6][TXF_WAVES];
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t ie = wave_incl_scan(e, lane), il = wave_incl_scan(l, lane), it = wave_incl_scan(t, lane);
  uint64_t re = pe, rl = pl, rt = pt;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    re += __shfl_xor(re, d, 64);
    rl += __shfl_xor(rl, d, 64);
    rt += __shfl_xor(rt, d, 64);
  }
  if (lane == 63) {
    s_x[0][wave] = ie;
    s_x[1][wave] = il;
    s_x[2][wave] = it;
    s_x[3][wave] = re;
    s_x[4][wave] = rl;
    s_x[5][wave] = rt;
  }
  if (bad) s_bad = 1;
  __syncthreads();
  uint64_t be = 0, bl = 0, bt = 0, te = 0, tl = 0, tt = 0, we = 0, wl = 0, wt = 0;
#pragma unroll
  for (int w = 0; w < TXF_WAVES; w++) {
    const uint64_t a0 = s_x[0][w], a1 = s_x[1][w], a2 = s_x[2][w];
    if (w < wave) { we += a0; wl += a1; wt += a2; }
    te += a0; tl += a1; tt += a2;
    be += s_x[3][w]; bl += s_x[4][w]; bt += s_x[5][w];
  }
  const uint64_t xe = be + we + ie - e, xl = bl + wl + il - l, xt = bt + wt + it - t;
  if (k < n) {
    ctl->enc_pre[k] = xe;
    ctl->len_pre[k] = xl;
    ctl->tile_pre[k] = (uint32_t)xt;
  }
  if (first + TXF_THREADS >= n && tid == 0) {  // the workgroup that holds the last entry has seen them all
    ctl->enc_pre[n] = be + te;
    ctl->len_pre[n] = bl + tl;
    ctl->tile_pre[n] = (uint32_t)(bt + tt);
    ctl->valid = (s_bad == 0 && bt + tt < (1ull << 32)) ? 1u : 0u;
  }
}

}  // namespace

extern "C" {
__attribute__((visibility("hidden"))) const void* grdma_kernel_fn_tx_index(void) { return reinterpret_cast<const void*>(&k_tx_index); }
__attribute__((visibility("hidden"))) uint32_t grdma_tx_index_threads(void) { return TXF_THREADS; }
// blocks = workgroups per connection: ceil(longest slice list / 1024)
__attribute__((visibility("hidden"))) hipError_t grdma_launch_tx_index(grdma_txf_ctl* d_ctls, uint32_t n, uint32_t blocks, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_index, dim3(blocks ? blocks : 1, n), dim3(TXF_THREADS), 0, s, d_ctls);
  return hipGetLastError();
}
}
