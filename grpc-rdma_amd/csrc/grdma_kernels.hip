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
#include "grdma_ops.h"

#define PLAN_THREADS 256
#define COPY_THREADS 256

namespace {

__device__ __forceinline__ uint64_t round_up8(uint64_t v) { return (v + 7ull) & ~7ull; }
__device__ __forceinline__ uint64_t round_down8(uint64_t v) { return v & ~7ull; }
__device__ __forceinline__ uint64_t enc_size(uint64_t pay) { return 16ull + round_up8(pay); }
// CalculateWritableSize, ring_buffer.h:185-189
__device__ __forceinline__ uint64_t writable_of(uint64_t space) {
  return space > GRDMA_RESERVED ? round_down8(space - GRDMA_RESERVED) : 0ull;
}
__device__ __forceinline__ uint64_t sat_sub(uint64_t a, uint64_t b) { return a > b ? a - b : 0ull; }

// Tag words are written by another agent (the wire kernel of a peer, a NIC):
// relaxed agent-scope atomic loads bypass the per-CU L1 (never refreshed by other
// writers) and are served by L2 / memory.  A ring registered for NIC writes must
// be allocated uncached (fine-grained), where the same load reaches memory.
__device__ __forceinline__ uint64_t ld_tag(const uint8_t* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// Inclusive wave64 prefix sum of a 32-bit value on the DPP network (row shifts
// 1,2,4,8, then row_bcast:15 / row_bcast:31 across the four 16-lane rows): six
// VALU instructions, no LDS traffic.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}

// Exclusive scan of one value per thread over a 256-thread block; returns the
// exclusive prefix and the block total (via *total).
__device__ __forceinline__ uint64_t block_excl_scan(uint64_t v, uint64_t* wave_sums,
                                                    uint64_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t incl = wave_incl_scan(v, lane);
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < PLAN_THREADS / 64; w++) {
    uint64_t s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

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
  __shared__ uint64_t s_len[GRDMA_MAX_SEGS];       // len_i, later pay_i
  __shared__ uint64_t s_excl[GRDMA_MAX_SEGS + 1];  // st_i, later the tile prefix
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
  if (m > GRDMA_MAX_SEGS - 1) m = GRDMA_MAX_SEGS - 1;
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
// k_rx_plan: message-ready test + record chain walk + endpoint_read replay
// ----------------------------------------------------------------------------
// The record chain is a linked list (each header gives the next offset), so a
// naive walk costs one dependent HBM/L2 round trip per record.  Here one wave
// probes 64 *predicted* positions per round trip: record sizes on a gRPC
// connection repeat with period 2 (9-byte DATA frame header slice, 16 KiB
// payload slice), so lane j loads the tag words at the offset the chain reaches
// after j records if the last two sizes keep alternating.  Every lane then
// checks its own link (header valid, its size equals the prediction) and the
// footer word in front of it; two __ballot()s give the verified prefix.  A round
// always resolves at least one record, and a mispredicted record's size is
// carried into the next round so its footer probe is exact.
#define CHAIN_CAP 128

struct chain_walker {
  const uint8_t* ring;
  uint64_t cap;
  uint64_t pos;   // ring offset of the first unverified record
  uint64_t e0;    // encoded size of the record at pos when its header is already known
  uint64_t h2, h1;  // encoded sizes of the two records before pos (0 = unknown)
  bool dry;       // pos holds no complete record
};

// One probe round.  Stores the payload sizes of the verified records in
// chain[0..v) and returns v.
__device__ __forceinline__ uint32_t chain_round(chain_walker* w, uint64_t* chain, int lane) {
  const uint64_t cap = w->cap, mask = cap - 1;
  // history with the already-known first record folded in
  const uint64_t H2 = w->e0 ? w->h1 : w->h2;
  const uint64_t H1 = w->e0 ? w->e0 : w->h1;
  const uint64_t A = H2 ? H2 : H1, B = H1;  // predicted sizes alternate A, B, A, ...
  // rel(j): predicted offset of record j from pos; k = index among the predicted ones
  auto rel_of = [&](uint64_t j) -> uint64_t {
    uint64_t base = 0, k = j;
    if (w->e0) {
      if (j == 0) return 0;
      base = w->e0;
      k = j - 1;
    }
    return base + (k >> 1) * (A + B) + ((k & 1) ? A : 0);
  };
  const bool have_pattern = (A != 0);
  const uint64_t rel = have_pattern ? rel_of(lane) : 0;
  const uint64_t rel_next = have_pattern ? rel_of(lane + 1) : 0;
  // The sender never lets the ring hold more than cap - 8 bytes (W() keeps 24
  // free before a write), so a record can only exist where it ends by cap - 8,
  // and the footer in front of lane j only matters if record j-1 ends by then.
  const bool probe_hdr = (lane == 0) || (have_pattern && rel_next <= cap - 8);
  const bool probe_prev = lane > 0 && have_pattern && rel <= cap - 8;
  const uint64_t my_pos = (w->pos + rel) & mask;
  uint64_t hdr = 0, prev = 0;
  if (probe_hdr) hdr = ld_tag(w->ring + my_pos);
  if (probe_prev) prev = ld_tag(w->ring + ((my_pos + cap - 8) & mask));  // footer of record j-1
  const bool valid = probe_hdr && hdr != 0 && hdr <= cap - GRDMA_RESERVED;
  const uint64_t enc = 16 + round_up8(hdr);
  const bool link_ok = valid && have_pattern && lane < 63 && enc == rel_next - rel;
  const uint64_t m_link = __ballot(link_ok);
  const uint64_t m_foot = __ballot(probe_prev && prev == GRDMA_FOOTER) >> 1;  // bit j: footer of j
  const uint64_t m_fprobed = __ballot(probe_prev) >> 1;
  const uint64_t m_hprobed = __ballot(probe_hdr);
  const uint64_t good = m_link & m_foot;
  const uint32_t v = (good == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~good);
  if ((uint32_t)lane < v) chain[lane] = hdr;
  // state for the next round: lane v is the first unverified record
  const uint64_t m_valid = __ballot(valid);
  const uint64_t rel_v = __shfl(rel, v < 64 ? v : 63, 64);
  const uint64_t enc_v = __shfl(enc, v < 64 ? v : 63, 64);
  const uint64_t enc_l1 = __shfl(enc, v >= 1 ? v - 1 : 0, 64);
  const uint64_t enc_l2 = __shfl(enc, v >= 2 ? v - 2 : 0, 64);
  if (v >= 2) {
    w->h2 = enc_l2;
    w->h1 = enc_l1;
  } else if (v == 1) {
    w->h2 = w->e0 ? w->h1 : w->h1;
    w->h1 = enc_l1;
  }
  // (v == 1 keeps the older size as h2: the record before lane 0)
  w->pos = (w->pos + rel_v) & mask;
  w->e0 = 0;
  if (v < 64) {
    const bool v_hprobed = (m_hprobed >> v) & 1;
    const bool v_valid = (m_valid >> v) & 1;
    const bool v_link = (m_link >> v) & 1;
    const bool v_fprobed = (m_fprobed >> v) & 1;
    if (!v_hprobed) {
      // not looked at (prediction ran past the ring): probe it as lane 0 next round
    } else if (!v_valid) {
      w->dry = true;                 // no (or torn) header: nothing more is ready
    } else if (!v_link || !v_fprobed) {
      w->e0 = enc_v;                 // header known, exact footer probe next round
    } else {
      w->dry = true;                 // size as predicted but the footer has not landed
    }
  }
  return v;
}

// transition of the endpoint-read state over one record of n bytes:
// s = bytes of space left in an open 256-byte read (0 = between reads)
__device__ __forceinline__ uint64_t read_space_after(uint64_t n, uint64_t s) {
  if (s == 0) return n >= GRDMA_MIN_READ_SLICE ? 0 : GRDMA_MIN_READ_SLICE - n;
  if (n < s) return s - n;
  if (n == s) return 0;
  const uint64_t r = n - s;
  return r >= GRDMA_MIN_READ_SLICE ? 0 : GRDMA_MIN_READ_SLICE - r;
}

__global__ __launch_bounds__(64) void k_rx_plan(const grdma_rx_op* ops) {
  const uint64_t t_begin = __builtin_amdgcn_s_memtime();
  const grdma_rx_op op = ops[blockIdx.x];
  const int lane = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;
  uint8_t* ring = c->ring;
  uint64_t n_rounds = 0, n_fast = 0, n_scalar = 0, t_refill = 0, t_fast = 0;
  const uint64_t cap = c->cap, mask = cap - 1;
  uint64_t head = c->head, mh = c->moving_head, remain = c->remain;
  uint64_t irs = c->internal_read_size, leftover = c->leftover_cap;
  const uint64_t mh0 = mh;
  uint64_t nslices = 0, nsegs = 0, ntiles = 0, bytes = 0, consumed_total = 0, records = 0;
  uint64_t a_off = 0, would_block = 0, credit = 0, credit_head = 0;
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  grdma_slice_out* out_slices = op.slices;
  uint64_t max_slices = GRDMA_MAX_SLICES;
  if (op.append == 2 && lane == 0) {  // first round of a streaming job
    c->rx_arena_off = 0;
    c->rx_slice_idx = 0;
  }
  if (op.append) {  // streaming job: keep filling the caller's buffer / slice table
    const uint64_t s_idx = op.append == 2 ? 0 : c->rx_slice_idx;
    a_off = op.append == 2 ? 0 : c->rx_arena_off;
    out_slices = op.slices + s_idx;
    const uint64_t room = op.slices_cap > s_idx ? op.slices_cap - s_idx : 0;
    if (room < max_slices) max_slices = room;
  }
  if (op.max_reads < max_slices) max_slices = op.max_reads;

  __shared__ uint64_t s_chain[CHAIN_CAP];
  chain_walker w = {ring, cap, head, 0, 0, 0, false};
  uint32_t chain_n = 0, chain_i = 0;

  auto refill = [&]() {
    while (chain_i == chain_n && !w.dry) {
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      chain_n = chain_round(&w, s_chain, lane);
      chain_i = 0;
      __syncthreads();
      n_rounds++;
      t_refill += __builtin_amdgcn_s_memtime() - t0;
    }
  };
  // keep at least 64 verified records queued while the ring has more
  auto top_up = [&]() {
    while (chain_n - chain_i < 64 && !w.dry) {
      const uint64_t t0 = __builtin_amdgcn_s_memtime();
      const uint32_t k = chain_n - chain_i;
      uint64_t keep = 0;
      if ((uint32_t)lane < k) keep = s_chain[chain_i + lane];
      __syncthreads();
      if ((uint32_t)lane < k) s_chain[lane] = keep;
      chain_i = 0;
      chain_n = k;
      chain_n += chain_round(&w, s_chain + k, lane);
      __syncthreads();
      n_rounds++;
      t_refill += __builtin_amdgcn_s_memtime() - t0;
    }
  };
  // size of the next unopened record if it is completely there, else 0
  auto next_ready = [&]() -> uint64_t {
    refill();
    return chain_i < chain_n ? s_chain[chain_i] : 0;
  };

  // PairPollable::Recv -> RingBufferPollable::Read(dst, capacity)
  // (pair.cc:264-286, ring_buffer.cc:122-191); returns the bytes copied.
  auto recv_step = [&](uint64_t dst, uint64_t capacity) -> uint64_t {
    uint64_t avail = remain;
    if (avail == 0) avail = next_ready();
    const uint64_t cpy = avail < capacity ? avail : capacity;
    if (cpy == 0) return 0;
    const uint64_t prev_mh = mh;
    if (remain == 0) {  // open the record, ring_buffer.cc:133-146
      if (lane == 0) *reinterpret_cast<uint64_t*>(ring + head) = 0;  // clear header
      mh = (head + 8) & mask;
      head = (head + 16 + round_up8(avail)) & mask;
      records++;
      chain_i++;
    }
    // payload bytes [mh, mh+cpy) -> dst, at most two pieces at the wrap; the
    // copying wave clears them behind itself (ring_buffer.cc:160,164)
    const uint64_t l1 = cpy < cap - mh ? cpy : cap - mh;
    if (lane == 0) {
      plan->segs[nsegs] = {dst, (uint64_t)(ring + mh), l1, GRDMA_SEG_ZERO_SRC};
      plan->tile_prefix[nsegs] = (uint32_t)ntiles;
    }
    ntiles += (l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
    nsegs++;
    if (cpy > l1) {
      if (lane == 0) {
        plan->segs[nsegs] = {dst + l1, (uint64_t)ring, cpy - l1, GRDMA_SEG_ZERO_SRC};
        plan->tile_prefix[nsegs] = (uint32_t)ntiles;
      }
      ntiles += (cpy - l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
      nsegs++;
    }
    mh = (mh + cpy) & mask;
    remain = avail - cpy;
    if (remain == 0) {  // finish the record, ring_buffer.cc:169-182
      const uint64_t pad_end = round_up8(mh);
      if (lane == 0) {
        for (uint64_t q = mh; q < pad_end; q++) ring[q & mask] = 0;  // clear padded space
        *reinterpret_cast<uint64_t*>(ring + (pad_end & mask)) = 0;    // clear footer
      }
      mh = pad_end & mask;
      mh = (mh + 8) & mask;
    }
    const uint64_t consumed = (mh + cap - prev_mh) & mask;
    consumed_total += consumed;
    // credit return every cap/2 consumed bytes, pair.cc:276-284
    irs += consumed;
    if (irs >= cap / 2) {
      credit_head = mh;
      credit++;
      irs = 0;
    }
    return cpy;
  };

  // ---- data-parallel replay of whole records, one lane per record -------------
  // Between reads (remain == 0, no retained slice) the endpoint-read loop is a
  // tiny state machine over the record sizes: s = space left in an open 256-byte
  // read.  Any record of >= 511 bytes forces s back to 0 whatever came before, so
  // every lane finds its own incoming state by looking back to the nearest such
  // record and replaying the few small records in between.  Offsets, slice and
  // segment indices then follow from wave prefix sums.  Returns the number of
  // records consumed (0: fall back to the scalar path).
  auto fast_chunk = [&]() -> uint32_t {
    if (cap > (1ull << 31)) return 0;  // 32-bit DPP scans below
    top_up();
    uint32_t k = chain_n - chain_i;
    if (k == 0) return 0;
    if (k > 64) k = 64;
    // conservative room checks for up to 64 records
    if (nslices + 128 > max_slices || nsegs + 256 + 520 > GRDMA_MAX_SEGS) return 0;
    const bool act0 = (uint32_t)lane < k;
    const uint64_t n = act0 ? s_chain[chain_i + lane] : 0;
    // incoming read state
    const uint64_t resets = __ballot(act0 && n >= 2 * GRDMA_MIN_READ_SLICE - 1);
    const uint64_t below = resets & ((1ull << lane) - 1ull);
    const uint32_t from = below ? (64 - __builtin_clzll(below)) : 0;  // first record after the reset
    uint64_t s_in = 0;
    for (uint32_t i = from; i < (uint32_t)lane && act0; i++)
      s_in = read_space_after(s_chain[chain_i + i], s_in);
    const uint64_t s_out = read_space_after(n, s_in);
    // stop after the last record that leaves the state clean
    const uint64_t clean = __ballot(act0 && s_out == 0);
    if (clean == 0) return 0;
    const uint32_t cnt = 64 - __builtin_clzll(clean);
    const bool act = (uint32_t)lane < cnt;
    const uint32_t enc = act ? (uint32_t)(16 + round_up8(n)) : 0;

    // what this record does to the read sequence
    uint64_t c1 = 0, c2 = 0;          // bytes of the two Recv steps
    uint64_t sl_len[2] = {0, 0};      // slices completed here, in order
    uint32_t sl_cnt = 0;
    if (act) {
      if (s_in == 0) {
        c1 = n;
        if (n >= GRDMA_MIN_READ_SLICE) sl_len[sl_cnt++] = n;
      } else if (n <= s_in) {
        c1 = n;
        if (n == s_in) sl_len[sl_cnt++] = GRDMA_MIN_READ_SLICE;
      } else {
        c1 = s_in;
        c2 = n - s_in;
        sl_len[sl_cnt++] = GRDMA_MIN_READ_SLICE;
        if (c2 >= GRDMA_MIN_READ_SLICE) sl_len[sl_cnt++] = c2;
      }
    }
    const uint32_t done_bytes =
        (uint32_t)(((sl_len[0] + 15) & ~15ull) + ((sl_len[1] + 15) & ~15ull));
    const uint32_t i_enc = wave_incl_scan_u32(enc);
    const uint32_t i_bytes = wave_incl_scan_u32(done_bytes);
    const uint32_t i_n = wave_incl_scan_u32(act ? (uint32_t)n : 0);
    const uint64_t tot_n = __shfl(i_n, 63, 64);
    // arena room: every slice start is 16-byte aligned
    if (a_off + tot_n + 32ull * cnt + 512 > op.arena_cap) return 0;
    const uint64_t x_enc = i_enc - enc, x_bytes = i_bytes - done_bytes;
    const uint64_t pos = (head + x_enc) & mask;           // header of my record
    const uint64_t pay = (pos + 8) & mask;
    const uint64_t A = a_off + x_bytes;                   // start of the open / next slice
    const uint64_t filled = s_in ? GRDMA_MIN_READ_SLICE - s_in : 0;
    const uint64_t dst1 = (uint64_t)op.arena + A + filled;
    const uint64_t dst2 = (uint64_t)op.arena + A + GRDMA_MIN_READ_SLICE;
    // segments: each step is one piece, two when it crosses the ring end
    uint64_t sg_dst[4], sg_src[4], sg_len[4];
    uint32_t sg_cnt = 0;
    auto add_step = [&](uint64_t dst, uint64_t off, uint64_t len) {
      if (len == 0) return;
      const uint64_t p0 = (pay + off) & mask;
      const uint64_t l1 = len < cap - p0 ? len : cap - p0;
      sg_dst[sg_cnt] = dst; sg_src[sg_cnt] = (uint64_t)(ring + p0); sg_len[sg_cnt] = l1; sg_cnt++;
      if (len > l1) {
        sg_dst[sg_cnt] = dst + l1; sg_src[sg_cnt] = (uint64_t)ring; sg_len[sg_cnt] = len - l1; sg_cnt++;
      }
    };
    if (act) {
      add_step(dst1, 0, c1);
      add_step(dst2, c1, c2);
    }
    uint32_t my_tiles = 0;
    for (uint32_t q = 0; q < sg_cnt; q++)
      my_tiles += (uint32_t)((sg_len[q] + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES);
    const uint32_t packed = sl_cnt | (sg_cnt << 16);
    const uint32_t i_packed = wave_incl_scan_u32(packed);
    const uint32_t i_tiles = wave_incl_scan_u32(my_tiles);
    const uint64_t x_slices = (i_packed & 0xFFFFu) - sl_cnt;
    const uint64_t x_segs = (i_packed >> 16) - sg_cnt;
    uint64_t x_tiles = i_tiles - my_tiles;
    if (act) {
      for (uint32_t q = 0; q < sg_cnt; q++) {
        plan->segs[nsegs + x_segs + q] = {sg_dst[q], sg_src[q], sg_len[q], GRDMA_SEG_ZERO_SRC};
        plan->tile_prefix[nsegs + x_segs + q] = (uint32_t)(ntiles + x_tiles);
        x_tiles += (sg_len[q] + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
      }
      uint64_t so = A;
      for (uint32_t q = 0; q < sl_cnt; q++) {
        out_slices[nslices + x_slices + q].off = so;
        out_slices[nslices + x_slices + q].len = sl_len[q];
        so += (sl_len[q] + 15) & ~15ull;
      }
      // clear header, padding and footer (ring_buffer.cc:146,173-180)
      *reinterpret_cast<uint64_t*>(ring + pos) = 0;
      for (uint64_t q = n; q < round_up8(n); q++) ring[(pay + q) & mask] = 0;
      *reinterpret_cast<uint64_t*>(ring + ((pay + round_up8(n)) & mask)) = 0;
    }
    // credit accounting over the Recv steps, in order (pair.cc:276-284): a
    // record's steps consume enc bytes in total, so the running sum after its
    // last step is the inclusive enc scan
    const uint64_t pad_foot = round_up8(n) - n + 8;
    const uint64_t cons2 = act && c2 ? c2 + pad_foot : 0;
    const uint64_t mh1 = c2 == 0 ? (pos + enc) & mask : (pay + c1) & mask;
    const uint64_t mh2 = (pos + enc) & mask;
    const uint64_t C2 = i_enc;
    const uint64_t C1 = C2 - cons2;
    const uint64_t Ctot = __shfl(i_enc, 63, 64);
    uint64_t base = 0, thr = cap / 2 - irs;
    bool crossed = false;
    for (;;) {
      const uint64_t hit = __ballot(act && C2 >= thr);
      if (hit == 0) break;
      const int f = __builtin_ctzll(hit);
      const uint64_t fC1 = __shfl(C1, f, 64), fC2 = __shfl(C2, f, 64);
      const uint64_t fmh1 = __shfl(mh1, f, 64), fmh2 = __shfl(mh2, f, 64);
      const bool first = fC1 >= thr;
      credit_head = first ? fmh1 : fmh2;
      base = first ? fC1 : fC2;
      credit++;
      crossed = true;
      thr = base + cap / 2;
    }
    irs = crossed ? Ctot - base : irs + Ctot;

    const uint32_t t_packed = __shfl(i_packed, 63, 64);
    // lanes >= cnt contributed nothing, so lane 63 holds the totals
    head = (head + Ctot) & mask;
    mh = head;
    consumed_total += Ctot;
    bytes += tot_n;
    records += cnt;
    nslices += t_packed & 0xFFFFu;
    nsegs += t_packed >> 16;
    ntiles += __shfl(i_tiles, 63, 64);
    a_off += __shfl(i_bytes, 63, 64);
    chain_i += cnt;
    return cnt;
  };

  if (connected && op.raw_cap > 0) {
    // grdma_pair_recv(): exactly one Recv(buf, capacity)
    uint64_t n = recv_step((uint64_t)op.arena, op.raw_cap);
    if (lane == 0) {
      out_slices[0].off = 0;
      out_slices[0].len = n;
    }
    nslices = n ? 1 : 0;
    bytes = n;
    a_off = n;
  } else {
    while (connected && nslices < max_slices && nsegs + 520 <= GRDMA_MAX_SEGS) {
      {
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        const bool took = remain == 0 && leftover == 0 && fast_chunk() > 0;
        t_fast += __builtin_amdgcn_s_memtime() - t0;
        if (took) { n_fast++; continue; }
      }
      n_scalar++;
      // rdma_continue_read, rdma_bp_posix.cc:306-317
      uint64_t readable = remain;
      if (readable == 0) readable = next_ready();
      const uint64_t alloc =
          leftover ? leftover
                   : (readable > GRDMA_MIN_READ_SLICE ? readable : GRDMA_MIN_READ_SLICE);
      if (a_off + alloc > op.arena_cap) break;  // receive arena exhausted
      uint64_t total = 0;
      // rdma_do_read loop, rdma_bp_posix.cc:195-277
      while (total < alloc) {
        uint64_t n = recv_step((uint64_t)(op.arena + a_off + total), alloc - total);
        if (n == 0) break;
        total += n;
      }
      if (total == 0) {  // nothing ready: notify_on_read, the slice stays allocated
        leftover = alloc;
        would_block = 1;
        break;
      }
      leftover = alloc - total;  // grpc_slice_buffer_trim_end -> last_read_buffer
      if (lane == 0) {
        out_slices[nslices].off = a_off;
        out_slices[nslices].len = total;
      }
      nslices++;
      bytes += total;
      a_off = (a_off + total + 15) & ~15ull;
    }
  }

  if (lane != 0) return;
  plan->nsegs = (uint32_t)nsegs;
  plan->ntiles = (uint32_t)ntiles;
  plan->tile_prefix[nsegs] = (uint32_t)ntiles;
  plan->bytes = bytes;

  c->head = head;
  c->moving_head = mh;
  c->remain = remain;
  c->internal_read_size = irs;
  c->leftover_cap = leftover;
  c->total_read += bytes;
  c->credit_msgs += credit;
  c->rx_records += records;
  if (nslices) c->rx_rounds++;
  if (op.append) {
    c->rx_arena_off = a_off;
    c->rx_slice_idx = (op.append == 2 ? 0 : c->rx_slice_idx) + nslices;
  }
  c->rx_blocks_done = 0;
  // updateStatus() (pair.cc:624-641) must not overtake the copy-out and the
  // zero-fill of the bytes it grants: the 16-byte report is posted by the last
  // workgroup of k_rx_apply.
  if (credit) c->status_send.remote_head = credit_head;
  res->credit_head = credit_head;
  res->nslices = nslices;
  res->bytes = bytes;
  res->consumed = consumed_total;
  res->records = records;
  res->would_block = would_block;
  res->credit_sent = credit;
  res->head = head;
  res->moving_head = mh;
  res->remain = remain;
  res->arena_used = a_off;
  res->dbg[0] = t_begin;
  res->dbg[1] = __builtin_amdgcn_s_memtime();
  res->dbg[2] = n_rounds;
  res->dbg[3] = n_fast;
  res->dbg[4] = n_scalar;
  res->dbg[5] = t_refill;
  res->dbg[6] = t_fast;
  // consumed ring bytes are always the contiguous range [mh0, mh)
  res->zero_off[0] = res->zero_off[1] = res->zero_len[0] = res->zero_len[1] = 0;
  if (consumed_total > 0) {
    if (mh > mh0) {
      res->zero_off[0] = mh0;
      res->zero_len[0] = mh - mh0;
    } else {
      res->zero_off[0] = mh0;
      res->zero_len[0] = cap - mh0;
      res->zero_off[1] = 0;
      res->zero_len[1] = mh;
    }
  }
  __threadfence_system();
  __hip_atomic_store(&res->seq, res->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

hipError_t grdma_launch_rx_plan(const grdma_rx_op* d_ops, uint32_t nops, hipStream_t s) {
  if (nops == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rx_plan, dim3(nops), dim3(64), 0, s, d_ops);
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
