// CDNA4 (gfx950) kernels of the ring-buffer endpoint data plane.
//
// The reference moves every byte with CPU memcpy/memset inside
// PairPollable::Send (src/core/lib/ibverbs/pair.cc:645-734) and
// RingBufferPollable::Read (src/core/lib/ibverbs/ring_buffer.cc:122-191).  Here
// the same protocol is split the way the hardware wants it:
//
//   k_tx_plan   one workgroup per connection: the head/tail credit arithmetic of
//               Send() as a block-wide prefix scan (all records priced at once,
//               first short record found with an LDS atomic-min), record
//               header/footer tags, the rdma_flush slice cursor, the ≤2 wire
//               work requests.  Emits a list of byte-copy segments.
//   k_copy      the only kernel that touches payload: every wave takes 4 KiB
//               tiles of the segment list; 16-byte destination-aligned stores,
//               source realigned in registers (two aligned 16-byte loads + a
//               funnel shift), so arbitrary grpc_slice alignment costs no
//               extra HBM transactions.  Used for slice gather (TX), the
//               loop-back wire, and ring->slice scatter (RX).
//   k_rx_plan   one wave per connection: walks the record chain from head_
//               (header tag + footer tag = message-ready test of
//               GetReadableSize, ring_buffer.cc:67-97) 64 speculative probes per
//               memory round trip, replays the endpoint_read loop of
//               rdma_bp_posix.cc:180-326 to decide slice boundaries, does the
//               credit accounting of Recv() (pair.cc:264-286), clears the tags.
//   k_rx_apply  K4: scatters the payload to the slices, clears it behind itself
//               (reader zero-fill invariant, ring_buffer.cc:146,160,164,173-180);
//               the last workgroup posts the 16-byte status report.
//   k_poll      K3 batched: one lane per connection, 64 connections per wave,
//               __ballot() of the ready set (HasMessage / GetReadableSize).
//
// No MFMA anywhere: this is HBM-bound byte shuffling.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grdma_dev.h"
#include "grdma_devfn.h"
#include "grdma_ops.h"

namespace {

// ----------------------------------------------------------------------------
// k_tx_plan: PairPollable::Send arithmetic + rdma_flush cursor, one block per op
// ----------------------------------------------------------------------------
// All records of a Send are priced at once: enc_i = 16 + round_up8(len_i) is
// prefix-summed across the block (st_i), every record tests its own budget
// pay_i = min(len_i, W(S - st_i), W(free0 - st_i)) under the assumption that all
// earlier records went out whole, and an LDS atomic-min finds the first record
// that comes up short -- which is exactly where the reference's sequential loop
// stops (pair.cc:671-707).  Global loads/stores are
// striped over the block (record i -> thread i % 256) so they coalesce.
__global__ __launch_bounds__(PLAN_THREADS) void k_tx_plan(const grdma_tx_op* ops) {
  const grdma_tx_op op = ops[blockIdx.x];
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint64_t s_wave[PLAN_THREADS / 64];
  __shared__ uint64_t s_len[GRDMA_TX_MAX_RECORDS];       // len_i, later pay_i
  __shared__ uint64_t s_excl[GRDMA_TX_MAX_RECORDS + 1];  // st_i
  __shared__ unsigned int s_first_short;
  __shared__ unsigned int s_wrap_rec;
  const unsigned tid = threadIdx.x;
  uint64_t tdbg[8];
  tdbg[0] = __builtin_amdgcn_s_memtime();

  const uint64_t cap = c->cap, mask = cap - 1;
  const uint64_t S = c->staging_cap;
  const uint64_t tail0 = c->remote_tail;
  // get_remote_head(), pair.h:229-233
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  uint64_t start = op.use_cursor == 1 ? c->tx_slice_idx : 0;
  const uint64_t byte_idx =
      op.use_cursor == 1 ? c->tx_byte_idx : (op.use_cursor ? 0 : op.byte_idx);
  if (start > op.nslices) start = op.nslices;
  const uint64_t avail = op.nslices - start;
  const grdma_sge* sl = op.slices + start;

  if (tid == 0) {
    s_first_short = 0xFFFFFFFFu;
    s_wrap_rec = 0xFFFFFFFFu;
  }

  // total bytes offered (pair.cc:660-663).  A streaming job keeps the running
  // remainder in the connection instead of re-summing the whole list per round.
  uint64_t offered;
  if (op.use_cursor == 1) {
    offered = c->tx_remaining;
    __syncthreads();
  } else {
    uint64_t part = 0;
    for (uint64_t i = tid; i < avail; i += PLAN_THREADS) part += sl[i].len;
    block_excl_scan(part, s_wave, &offered);
    offered = sat_sub(offered, byte_idx);
  }

  uint64_t m = avail;
  if (m > c->max_sge) m = c->max_sge;
  if (m > GRDMA_TX_MAX_RECORDS - 1) m = GRDMA_TX_MAX_RECORDS - 1;
  if (!connected) m = 0;

  tdbg[1] = __builtin_amdgcn_s_memtime();
  // lengths, striped
  for (uint64_t i = tid; i < m; i += PLAN_THREADS) {
    uint64_t l = sl[i].len;
    if (i == 0) l = sat_sub(l, byte_idx);
    s_len[i] = l;
  }
  __syncthreads();

  // st_i: each thread scans a contiguous run of `per` records out of LDS
  const uint64_t per = (m + PLAN_THREADS - 1) / PLAN_THREADS;
  {
    uint64_t chunk = 0;
    for (uint64_t k = 0; k < per; k++) {
      const uint64_t i = tid * per + k;
      if (i < m) {
        const uint64_t l = s_len[i];
        // clamp so that sums cannot overflow; anything above 2*cap cannot fit anyway
        chunk += enc_size(l < (cap << 1) ? l : (cap << 1));
      }
    }
    uint64_t total_enc;
    uint64_t st = block_excl_scan(chunk, s_wave, &total_enc);
    for (uint64_t k = 0; k < per; k++) {
      const uint64_t i = tid * per + k;
      if (i < m) {
        s_excl[i] = st;
        const uint64_t l = s_len[i];
        st += enc_size(l < (cap << 1) ? l : (cap << 1));
      }
    }
    if (tid == PLAN_THREADS - 1 || (tid * per < m && (tid + 1) * per >= m)) s_excl[m] = st;
    if (m == 0 && tid == 0) s_excl[0] = 0;
  }
  __syncthreads();

  tdbg[2] = __builtin_amdgcn_s_memtime();
  // budget test, striped
  const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
  const uint64_t free0 = cap - occupied0;
  for (uint64_t i = tid; i < m; i += PLAN_THREADS) {
    const uint64_t st = s_excl[i];
    const uint64_t a = writable_of(sat_sub(S, st));
    const uint64_t b = writable_of(sat_sub(free0, st));
    const uint64_t l = s_len[i];
    uint64_t p = l;
    if (a < p) p = a;
    if (b < p) p = b;
    // a zero payload ends the send exactly like the reference's `break`.
    // One LDS atomic per wave: the lowest short lane of a wave holds its lowest i.
    const bool is_short = p < l || l == 0;
    const uint64_t bm = __ballot(is_short);
    if (bm != 0 && (tid & 63) == (unsigned)__builtin_ctzll(bm)) atomicMin(&s_first_short, (unsigned int)i);
  }
  __syncthreads();
  const uint64_t fs = s_first_short;
  const uint64_t nrec = (fs != 0xFFFFFFFFu) ? fs : m;  // records [0, nrec) go out whole
  uint64_t short_pay = 0;
  if (fs != 0xFFFFFFFFu) {
    const uint64_t st = s_excl[fs];
    const uint64_t a = writable_of(sat_sub(S, st));
    const uint64_t b = writable_of(sat_sub(free0, st));
    short_pay = s_len[fs];
    if (a < short_pay) short_pay = a;
    if (b < short_pay) short_pay = b;
  }
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  // Σ enc over the whole records, plus the short one if any
  const uint64_t staged = s_excl[nrec] + (short_pay > 0 ? enc_size(short_pay) : 0);
  __syncthreads();
  if (tid == 0 && short_pay > 0) s_len[nrec] = short_pay;  // s_len[i] is pay_i from here on
  __syncthreads();

  // destination of record i: staging + st_i, or the peer ring itself at
  // (tail0 + st_i) & mask when the wire is direct.
  const bool direct = c->wire_direct != 0;
  uint8_t* const dbase = direct ? c->peer_ring : c->staging;
  if (direct) {
    for (uint64_t i = tid; i < nrec_total; i += PLAN_THREADS) {
      const uint64_t pstart = (tail0 + s_excl[i] + 8) & mask;
      if (pstart + s_len[i] > cap) atomicMin(&s_wrap_rec, (unsigned int)i);
    }
  }
  __syncthreads();
  const uint64_t wrap_rec = s_wrap_rec;

  tdbg[3] = __builtin_amdgcn_s_memtime();
  // tags + segments, striped
  uint64_t sent_part = 0;
  for (uint64_t i = tid; i < nrec_total; i += PLAN_THREADS) {
    const uint64_t p = s_len[i];
    const uint64_t st = s_excl[i];
    sent_part += p;
    const uint64_t hdr_off = direct ? ((tail0 + st) & mask) : st;
    const uint64_t pay_off = direct ? ((hdr_off + 8) & mask) : st + 8;
    const uint64_t foot_off = direct ? ((hdr_off + 8 + round_up8(p)) & mask) : st + 8 + round_up8(p);
    // AppendHeader / AppendFooter, ring_buffer.h:84-99
    *reinterpret_cast<uint64_t*>(dbase + hdr_off) = p;
    *reinterpret_cast<uint64_t*>(dbase + foot_off) = GRDMA_FOOTER;
    // deterministic zero padding (the reference leaves stale staging bytes there)
    for (uint64_t q = p; q < round_up8(p); q++)
      dbase[direct ? ((pay_off + q) & mask) : pay_off + q] = 0;
    const uint8_t* src = sl[i].ptr + (i == 0 ? byte_idx : 0);
    const uint64_t seg = i + (i > wrap_rec ? 1 : 0);
    if (i == wrap_rec) {
      const uint64_t l1 = cap - pay_off;
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, l1, 0};
      plan->segs[seg + 1] = {(uint64_t)dbase, (uint64_t)(src + l1), p - l1, 0};
    } else {
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, p, 0};
    }
  }
  uint64_t sent;
  block_excl_scan(sent_part, s_wave, &sent);

  tdbg[4] = __builtin_amdgcn_s_memtime();
  // tile prefix per segment (contiguous runs again, out of LDS)
  uint64_t ntiles;
  {
    const uint64_t per2 = (nrec_total + PLAN_THREADS - 1) / PLAN_THREADS;
    auto tiles_of = [&](uint64_t i, uint64_t* t1) -> uint64_t {
      const uint64_t p = s_len[i];
      if (i == wrap_rec) {
        const uint64_t pay_off = (tail0 + s_excl[i] + 16) & mask;  // (hdr_off + 8) & mask
        const uint64_t l1 = cap - ((tail0 + s_excl[i] + 8) & mask);
        (void)pay_off;
        *t1 = (l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
        return *t1 + (p - l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
      }
      *t1 = (p + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
      return *t1;
    };
    uint64_t chunk = 0, t1;
    for (uint64_t k = 0; k < per2; k++) {
      const uint64_t i = tid * per2 + k;
      if (i < nrec_total) chunk += tiles_of(i, &t1);
    }
    uint64_t x = block_excl_scan(chunk, s_wave, &ntiles);
    for (uint64_t k = 0; k < per2; k++) {
      const uint64_t i = tid * per2 + k;
      if (i < nrec_total) {
        const uint64_t seg = i + (i > wrap_rec ? 1 : 0);
        const uint64_t t = tiles_of(i, &t1);
        plan->tile_prefix[seg] = (uint32_t)x;
        if (i == wrap_rec) plan->tile_prefix[seg + 1] = (uint32_t)(x + t1);
        x += t;
      }
    }
  }
  const uint64_t nsegs = nrec_total + ((wrap_rec != 0xFFFFFFFFu) ? 1 : 0);

  tdbg[5] = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    plan->nsegs = (uint32_t)nsegs;
    plan->ntiles = (uint32_t)ntiles;
    plan->tile_prefix[nsegs] = (uint32_t)ntiles;
    plan->bytes = sent;
    const uint64_t new_tail = (tail0 + staged) & mask;
    // the ≤2 RDMA WRITEs of GetWriteRequests(sg_list), ring_buffer.cc:261-330
    uint64_t seg1 = staged < cap - tail0 ? staged : cap - tail0;
    grdma_tx_result* r = op.result;
    r->wr_count = 0;
    r->wr_off[0] = r->wr_off[1] = r->wr_len[0] = r->wr_len[1] = 0;
    if (staged > 0) {
      r->wr_off[0] = tail0;
      r->wr_len[0] = seg1;
      r->wr_count = 1;
      if (tail0 + staged >= cap) {  // a record reached (or crossed) the ring end
        r->wr_off[1] = 0;
        r->wr_len[1] = staged - seg1;
        r->wr_count = 2;
      }
    }
    grdma_plan* wp = op.wire_plan;
    if (wp != nullptr) {
      uint32_t ns = 0, nt = 0;
      if (!direct && staged > 0) {
        wp->segs[0] = {(uint64_t)(c->peer_ring + tail0), (uint64_t)c->staging, seg1, 0};
        wp->tile_prefix[0] = 0;
        nt = (uint32_t)((seg1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES);
        ns = 1;
        if (staged > seg1) {
          wp->segs[1] = {(uint64_t)c->peer_ring, (uint64_t)(c->staging + seg1), staged - seg1, 0};
          wp->tile_prefix[1] = nt;
          nt += (uint32_t)((staged - seg1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES);
          ns = 2;
        }
      }
      wp->nsegs = ns;
      wp->ntiles = nt;
      wp->tile_prefix[ns] = nt;
      wp->bytes = direct ? 0 : staged;
    }
    // rdma_flush cursor walk, rdma_bp_posix.cc:480-493
    uint64_t idx = start + nrec;
    uint64_t bidx = 0;
    if (short_pay > 0) bidx = (nrec == 0 ? byte_idx : 0) + short_pay;
    else if (nrec == 0) bidx = byte_idx;
    c->remote_tail = new_tail;
    c->partial_write = sent < offered ? 1 : 0;  // pair.cc:709
    c->total_written += sent;
    c->tx_records += nrec_total;
    if (nrec_total) c->tx_rounds++;
    if (op.use_cursor) {
      c->tx_slice_idx = idx;
      c->tx_byte_idx = bidx;
      c->tx_remaining = offered - sent;
    }
    r->sent = sent;
    r->records = nrec_total;
    r->staged = staged;
    r->partial = sent < offered ? 1 : 0;
    r->new_remote_tail = new_tail;
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = (idx >= op.nslices) ? 1 : 0;
    tdbg[6] = __builtin_amdgcn_s_memtime();
    for (int q = 0; q < 7; q++) r->dbg[q] = tdbg[q];
    r->dbg[7] = m;
    __threadfence_system();
    __hip_atomic_store(&r->seq, r->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ----------------------------------------------------------------------------
// k_copy: segment-list byte mover (slice gather, wire, ring scatter)
// ----------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 funnel16(u32x4 a, u32x4 b, unsigned shift) {
  // bytes [shift, shift+16) of the 32-byte little-endian concatenation a|b
  uint64_t q0 = (uint64_t)a.x | ((uint64_t)a.y << 32);
  uint64_t q1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
  uint64_t q2 = (uint64_t)b.x | ((uint64_t)b.y << 32);
  uint64_t q3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
  if (shift & 8) {
    q0 = q1;
    q1 = q2;
    q2 = q3;
  }
  unsigned s = (shift & 7) * 8;
  uint64_t o0 = q0, o1 = q1;
  if (s) {
    o0 = (q0 >> s) | (q1 << (64 - s));
    o1 = (q1 >> s) | (q2 << (64 - s));
  }
  u32x4 o;
  o.x = (uint32_t)o0;
  o.y = (uint32_t)(o0 >> 32);
  o.z = (uint32_t)o1;
  o.w = (uint32_t)(o1 >> 32);
  return o;
}

// One wave moves n (<= GRDMA_TILE_BYTES) bytes src -> dst, any alignment.
__device__ __forceinline__ void wave_copy_tile(uint8_t* dst, const uint8_t* src, uint64_t n,
                                               int lane) {
  uint64_t head = (16 - ((uint64_t)dst & 15)) & 15;
  if (head > n) head = n;
  if ((uint64_t)lane < head) dst[lane] = src ? src[lane] : 0;
  dst += head;
  if (src) src += head;
  n -= head;
  const uint64_t units = n >> 4;
  const unsigned shift = (unsigned)((uint64_t)src & 15);
  const u32x4* sa = reinterpret_cast<const u32x4*>((uint64_t)src & ~15ull);
  u32x4* da = reinterpret_cast<u32x4*>(dst);
  if (src == nullptr) {
    for (uint64_t u = lane; u < units; u += 64) da[u] = u32x4{0, 0, 0, 0};
  } else if (shift == 0) {
#pragma unroll 4
    for (uint64_t u = lane; u < units; u += 64) da[u] = __builtin_nontemporal_load(sa + u);
  } else {
#pragma unroll 4
    for (uint64_t u = lane; u < units; u += 64) {
      u32x4 a = __builtin_nontemporal_load(sa + u);
      u32x4 b = __builtin_nontemporal_load(sa + u + 1);
      da[u] = funnel16(a, b, shift);
    }
  }
  const uint64_t tail = n & 15;
  if ((uint64_t)lane < tail) {
    uint64_t o = (units << 4) + lane;
    dst[o] = src ? src[o] : 0;
  }
}

// Reader zero-fill (ring_buffer.cc:160,164): clear exactly [p, p+n).
__device__ __forceinline__ void wave_zero_tile(uint8_t* p, uint64_t n, int lane) {
  uint64_t head = (16 - ((uint64_t)p & 15)) & 15;
  if (head > n) head = n;
  if ((uint64_t)lane < head) p[lane] = 0;
  p += head;
  n -= head;
  const uint64_t units = n >> 4;
  u32x4* q = reinterpret_cast<u32x4*>(p);
  for (uint64_t u = lane; u < units; u += 64) q[u] = u32x4{0, 0, 0, 0};
  const uint64_t tail = n & 15;
  if ((uint64_t)lane < tail) p[(units << 4) + lane] = 0;
}

#define PREFIX_LDS 2048

// Every workgroup stages the tile prefix in LDS (one coalesced load), then each
// wave maps its tiles to segments with an LDS binary search: no dependent global
// loads between picking a tile and issuing its first payload load.
__device__ __forceinline__ void run_plan_tiles(const grdma_plan* plan, uint32_t wave,
                                               uint32_t nwaves, int lane) {
  __shared__ uint32_t s_prefix[PREFIX_LDS + 1];
  const uint32_t nsegs = plan->nsegs;
  const uint32_t ntiles = plan->ntiles;
  const bool in_lds = nsegs <= PREFIX_LDS;
  if (in_lds)
    for (uint32_t i = threadIdx.x; i <= nsegs; i += COPY_THREADS) s_prefix[i] = plan->tile_prefix[i];
  __syncthreads();
  for (uint32_t t = wave; t < ntiles; t += nwaves) {
    uint32_t lo = 0, hi = nsegs;  // invariant: prefix[lo] <= t < prefix[hi]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      const uint32_t pm = in_lds ? s_prefix[mid] : plan->tile_prefix[mid];
      if (pm <= t) lo = mid; else hi = mid;
    }
    const grdma_seg sg = plan->segs[lo];
    const uint32_t p0 = in_lds ? s_prefix[lo] : plan->tile_prefix[lo];
    const uint64_t off = (uint64_t)(t - p0) * GRDMA_TILE_BYTES;
    uint64_t n = sg.len - off;
    if (n > GRDMA_TILE_BYTES) n = GRDMA_TILE_BYTES;
    uint8_t* src = sg.src ? reinterpret_cast<uint8_t*>(sg.src + off) : nullptr;
    wave_copy_tile(reinterpret_cast<uint8_t*>(sg.dst + off), src, n, lane);
    if ((sg.flags & GRDMA_SEG_ZERO_SRC) && src) {
      // every load of this tile has returned (its data fed the stores above);
      // make that explicit before the source bytes are overwritten
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wave_zero_tile(src, n, lane);
    }
  }
}

__global__ __launch_bounds__(COPY_THREADS) void k_copy(const grdma_plan* const* plans) {
  const grdma_plan* plan = plans[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  run_plan_tiles(plan, wave, nwaves, lane);
}

// ----------------------------------------------------------------------------
// k_rx_apply: K4 in one launch -- copy the payload out, clear it behind, and let
// the last workgroup post the credit (status report) once every byte is free.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(COPY_THREADS) void k_rx_apply(const grdma_rx_op* ops) {
  const grdma_rx_op op = ops[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * COPY_THREADS + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * COPY_THREADS) >> 6;
  run_plan_tiles(op.plan, wave, nwaves, lane);
  // arrival: my stores have been issued and acknowledged (vmcnt(0)); count in,
  // the last workgroup publishes.  Consumers on this device run in later
  // kernels of the stream (a kernel boundary makes the writes visible); a ring
  // registered for a NIC is uncached memory, where acknowledged stores are
  // already at their destination -- so no L2 write-back fence is paid here.
  __shared__ unsigned int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int prev = __hip_atomic_fetch_add(&op.conn->rx_blocks_done, 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    grdma_rx_result* res = op.result;
    if (res->credit_sent) {
      grdma_status_report* ps = op.conn->peer_status;
      if (ps != nullptr)
        __hip_atomic_store(&ps->remote_head, res->credit_head, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&res->commit_seq, res->commit_seq + 1, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct ring_probe {
  uint64_t n;      // header value at `pos`
  bool ready;      // header valid and footer tag present
};

__device__ __forceinline__ ring_probe probe_record(const uint8_t* ring, uint64_t cap,
                                                   uint64_t pos) {
  // GetReadableSize, ring_buffer.cc:67-97
  ring_probe r;
  r.n = ld_tag(ring + pos);
  r.ready = false;
  if (r.n == 0 || r.n > cap - GRDMA_RESERVED) return r;  // empty / torn header
  uint64_t f = (pos + 8 + round_up8(r.n)) & (cap - 1);
  r.ready = ld_tag(ring + f) == GRDMA_FOOTER;
  return r;
}

// ----------------------------------------------------------------------------
// k_poll: batched message-ready detection (K3), one lane per connection
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_poll(grdma_conn* const* conns, uint32_t nconns,
                                             uint64_t* readable_out, uint64_t* ready_mask,
                                             uint64_t* has_msg_mask) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  uint64_t readable = 0;
  bool has = false;
  if (i < nconns) {
    const grdma_conn* c = conns[i];
    if (c->remain > 0) {  // HasMessage / GetReadableSize fast path
      readable = c->remain;
      has = true;
    } else {
      ring_probe pr = probe_record(c->ring, c->cap, c->head);
      has = pr.n > 0;  // HasMessage, ring_buffer.cc:56-65: header only
      readable = pr.ready ? pr.n : 0;
    }
    readable_out[i] = readable;
  }
  // wavefront ballots: 64 connections -> two 64-bit words
  const uint64_t m_ready = __ballot(readable > 0);
  const uint64_t m_has = __ballot(has);
  if (threadIdx.x == 0) {
    ready_mask[blockIdx.x] = m_ready;
    has_msg_mask[blockIdx.x] = m_has;
  }
}

}  // namespace

// ------------------------------------------------------------------ launchers
extern "C" {

hipError_t grdma_launch_tx_plan(const grdma_tx_op* d_ops, uint32_t nops, hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tx_plan, dim3(nops), dim3(PLAN_THREADS), 0, s, d_ops);
  return hipGetLastError();
}

hipError_t grdma_launch_copy(const grdma_plan* const* d_plans, uint32_t nplans,
                             uint32_t blocks_per_plan, hipStream_t s) {
  if (nplans == 0) return hipSuccess;
  hipLaunchKernelGGL(k_copy, dim3(blocks_per_plan, nplans), dim3(COPY_THREADS), 0, s, d_plans);
  return hipGetLastError();
}

hipError_t grdma_launch_rx_apply(const grdma_rx_op* d_ops, uint32_t nops, uint32_t blocks_per_op,
                                 hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_apply, dim3(blocks_per_op, nops), dim3(COPY_THREADS), 0, s, d_ops);
  return hipGetLastError();
}

hipError_t grdma_launch_poll(grdma_conn* const* d_conns, uint32_t nconns, uint64_t* d_readable,
                             uint64_t* d_ready_mask, uint64_t* d_has_mask, hipStream_t s) {
  if (nconns == 0) return hipSuccess;
  hipLaunchKernelGGL(k_poll, dim3((nconns + 63) / 64), dim3(64), 0, s, d_conns, nconns,
                     d_readable, d_ready_mask, d_has_mask);
  return hipGetLastError();
}

}  // extern "C"
