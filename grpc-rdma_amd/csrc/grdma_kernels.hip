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
#define PLAN_ITEMS (GRDMA_MAX_SEGS / PLAN_THREADS)  // 16 slices per thread
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

// Tag words are polled across agents (a NIC or a peer GPU writes them): use
// system-scope relaxed atomics so they are never served from a stale L1 line.
__device__ __forceinline__ uint64_t ld_tag(const uint8_t* p) {
  return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
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
__global__ __launch_bounds__(PLAN_THREADS) void k_tx_plan(const grdma_tx_op* ops) {
  const grdma_tx_op op = ops[blockIdx.x];
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  __shared__ uint64_t s_wave[PLAN_THREADS / 64];
  __shared__ uint64_t s_excl[GRDMA_MAX_SEGS + 1];  // staging offset of record i
  __shared__ unsigned int s_first_short;
  __shared__ unsigned int s_wrap_rec;
  const int tid = threadIdx.x;

  const uint64_t cap = c->cap, mask = cap - 1;
  const uint64_t S = c->staging_cap;
  const uint64_t tail0 = c->remote_tail;
  // get_remote_head(), pair.h:229-233
  const uint64_t rhead = __hip_atomic_load(&c->status_recv.remote_head, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_SYSTEM);
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  uint64_t start = op.use_cursor == 1 ? c->tx_slice_idx : 0;
  uint64_t byte_idx = op.use_cursor == 1 ? c->tx_byte_idx : (op.use_cursor ? 0 : op.byte_idx);
  if (start > op.nslices) start = op.nslices;
  const uint64_t avail = op.nslices - start;
  const grdma_sge* sl = op.slices + start;

  if (tid == 0) {
    s_first_short = 0xFFFFFFFFu;
    s_wrap_rec = 0xFFFFFFFFu;
  }

  // total bytes offered (pair.cc:660-663)
  uint64_t part = 0;
  for (uint64_t i = tid; i < avail; i += PLAN_THREADS) part += sl[i].len;
  uint64_t offered;
  block_excl_scan(part, s_wave, &offered);
  offered = sat_sub(offered, byte_idx);

  uint64_t m = avail;
  if (m > c->max_sge) m = c->max_sge;
  if (m > GRDMA_MAX_SEGS - 1) m = GRDMA_MAX_SEGS - 1;
  if (!connected) m = 0;

  // per-thread contiguous chunk of PLAN_ITEMS slices
  uint64_t len[PLAN_ITEMS], enc[PLAN_ITEMS];
  uint64_t chunk = 0;
  const uint64_t i0 = (uint64_t)tid * PLAN_ITEMS;
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++) {
    uint64_t i = i0 + k;
    uint64_t l = 0;
    if (i < m) {
      l = sl[i].len;
      if (i == 0) l = sat_sub(l, byte_idx);
    }
    len[k] = l;
    // clamp so that sums cannot overflow; anything above 2*cap is "too big" anyway
    uint64_t e = (i < m) ? enc_size(l < (cap << 1) ? l : (cap << 1)) : 0;
    enc[k] = e;
    chunk += e;
  }
  uint64_t total_enc;
  uint64_t excl = block_excl_scan(chunk, s_wave, &total_enc);

  // Budget test with "every earlier record went out whole" (pair.cc:676-685).
  // The first record that does not fit whole ends the send (Appendix A.4).
  const uint64_t occupied0 = (tail0 + cap - rhead) & mask;
  const uint64_t free0 = cap - occupied0;
  uint64_t pay[PLAN_ITEMS];
  {
    uint64_t st = excl;
#pragma unroll
    for (int k = 0; k < PLAN_ITEMS; k++) {
      uint64_t i = i0 + k;
      if (i <= m) s_excl[i] = st;
      uint64_t a = writable_of(sat_sub(S, st));
      uint64_t b = writable_of(sat_sub(free0, st));
      uint64_t p = len[k];
      if (a < p) p = a;
      if (b < p) p = b;
      pay[k] = p;
      if (i < m && p < len[k]) atomicMin(&s_first_short, (unsigned int)i);
      // zero-length slices cannot occur (grpc never queues them); a zero
      // payload ends the send exactly like the reference's `break`.
      if (i < m && len[k] == 0) atomicMin(&s_first_short, (unsigned int)i);
      st += enc[k];
    }
  }
  __syncthreads();
  const uint64_t fs = s_first_short;
  // number of records and the (possibly short) last payload
  uint64_t nrec = m;
  if (fs != 0xFFFFFFFFu) nrec = fs;  // records [0, fs) whole; fs itself maybe short
  __shared__ uint64_t s_last_pay;
  if (tid == 0) s_last_pay = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++)
    if (i0 + k == fs && fs < m) s_last_pay = pay[k];
  __syncthreads();
  const uint64_t short_pay = (fs != 0xFFFFFFFFu && fs < m) ? s_last_pay : 0;
  const uint64_t nrec_total = nrec + (short_pay > 0 ? 1 : 0);
  // Σ enc over the whole records, plus the short one if any
  const uint64_t staged = s_excl[nrec] + (short_pay > 0 ? enc_size(short_pay) : 0);

  // destination of record i: staging + st_i, or the peer ring itself at
  // (tail0 + st_i) & mask when the wire is direct.
  const bool direct = c->wire_direct != 0;
  uint8_t* const dbase = direct ? c->peer_ring : c->staging;

  // find the single record whose payload crosses the ring end (direct mode)
  uint64_t my_pay[PLAN_ITEMS];
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++) {
    uint64_t i = i0 + k;
    uint64_t p = 0;
    if (i < nrec) p = len[k];
    else if (i == nrec && short_pay > 0) p = short_pay;
    my_pay[k] = p;
    if (direct && p > 0) {
      uint64_t pstart = (tail0 + s_excl[i] + 8) & mask;
      if (pstart + p > cap) atomicMin(&s_wrap_rec, (unsigned int)i);
    }
  }
  __syncthreads();
  const uint64_t wrap_rec = s_wrap_rec;

  // tags + segments
  uint64_t tiles_chunk = 0;
  uint64_t seg_tiles[PLAN_ITEMS][2];
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++) {
    uint64_t i = i0 + k;
    seg_tiles[k][0] = seg_tiles[k][1] = 0;
    uint64_t p = my_pay[k];
    if (p == 0) continue;
    uint64_t st = s_excl[i];
    uint64_t hdr_off = direct ? ((tail0 + st) & mask) : st;
    uint64_t pay_off = direct ? ((hdr_off + 8) & mask) : st + 8;
    uint64_t foot_off = direct ? ((hdr_off + 8 + round_up8(p)) & mask) : st + 8 + round_up8(p);
    // AppendHeader / AppendFooter, ring_buffer.h:84-99
    *reinterpret_cast<uint64_t*>(dbase + hdr_off) = p;
    *reinterpret_cast<uint64_t*>(dbase + foot_off) = GRDMA_FOOTER;
    // deterministic zero padding (the reference leaves stale staging bytes there)
    for (uint64_t q = p; q < round_up8(p); q++) dbase[direct ? ((pay_off + q) & mask) : pay_off + q] = 0;
    const uint8_t* src = sl[i].ptr + (i == 0 ? byte_idx : 0);
    uint64_t seg = i + (i > wrap_rec ? 1 : 0);
    if (i == wrap_rec) {
      uint64_t l1 = cap - pay_off;
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, l1, 0};
      plan->segs[seg + 1] = {(uint64_t)dbase, (uint64_t)(src + l1), p - l1, 0};
      seg_tiles[k][0] = (l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
      seg_tiles[k][1] = (p - l1 + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
    } else {
      plan->segs[seg] = {(uint64_t)(dbase + pay_off), (uint64_t)src, p, 0};
      seg_tiles[k][0] = (p + GRDMA_TILE_BYTES - 1) / GRDMA_TILE_BYTES;
    }
    tiles_chunk += seg_tiles[k][0] + seg_tiles[k][1];
  }
  uint64_t ntiles;
  uint64_t texcl = block_excl_scan(tiles_chunk, s_wave, &ntiles);
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++) {
    uint64_t i = i0 + k;
    if (my_pay[k] == 0) continue;
    uint64_t seg = i + (i > wrap_rec ? 1 : 0);
    plan->tile_prefix[seg] = (uint32_t)texcl;
    texcl += seg_tiles[k][0];
    if (i == wrap_rec) {
      plan->tile_prefix[seg + 1] = (uint32_t)texcl;
      texcl += seg_tiles[k][1];
    }
  }
  const uint64_t nsegs = nrec_total + ((wrap_rec != 0xFFFFFFFFu) ? 1 : 0);

  // payload total
  uint64_t sent_part = 0;
#pragma unroll
  for (int k = 0; k < PLAN_ITEMS; k++) sent_part += my_pay[k];
  uint64_t sent;
  block_excl_scan(sent_part, s_wave, &sent);

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
    }
    r->sent = sent;
    r->records = nrec_total;
    r->staged = staged;
    r->partial = sent < offered ? 1 : 0;
    r->new_remote_tail = new_tail;
    r->slice_idx = idx;
    r->byte_idx = bidx;
    r->done = (idx >= op.nslices) ? 1 : 0;
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

__device__ __forceinline__ void run_plan_tiles(const grdma_plan* plan, uint32_t wave,
                                               uint32_t nwaves, int lane) {
  const uint32_t nsegs = plan->nsegs;
  const uint32_t ntiles = plan->ntiles;
  // tile -> segment: binary search of the tile prefix (wave-uniform)
  for (uint32_t t = wave; t < ntiles; t += nwaves) {
    uint32_t lo = 0, hi = nsegs;  // invariant: prefix[lo] <= t < prefix[hi]
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (plan->tile_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    const grdma_seg sg = plan->segs[lo];
    const uint64_t off = (uint64_t)(t - plan->tile_prefix[lo]) * GRDMA_TILE_BYTES;
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
// payload slice), so lane j loads the tag words at the position the chain
// reaches after j records if the last two sizes keep alternating.  The chain
// is then verified lane by lane in registers (readlane), stopping at the first
// misprediction; a round always resolves at least one record.
#define CHAIN_CAP 128

struct chain_state {
  uint64_t e_prev1;  // encoded size of the most recent verified record (0 = unknown)
  uint64_t e_prev2;  // the one before it
};

// Verifies up to CHAIN_CAP ready records starting at ring offset `pos` and
// stores their payload sizes in chain[0..count).  Returns count; *exhausted is
// set when the walk stopped on a position that holds no complete record.
__device__ __forceinline__ uint32_t chain_refill(const uint8_t* ring, uint64_t cap, uint64_t pos,
                                                 chain_state* cs, uint64_t* chain,
                                                 bool* exhausted, int lane) {
  const uint64_t mask = cap - 1;
  uint32_t count = 0;
  *exhausted = false;
  while (count + 64 <= CHAIN_CAP) {
    // predicted offsets: pos, pos+ea, pos+ea+eb, ... (ea = two back, eb = one back)
    uint64_t ea = cs->e_prev2 ? cs->e_prev2 : cs->e_prev1;
    uint64_t eb = cs->e_prev1;
    uint64_t rel = (uint64_t)(lane >> 1) * (ea + eb) + ((lane & 1) ? ea : 0);
    if (ea == 0) rel = 0;  // nothing known yet: only lane 0 is meaningful
    const uint64_t my_pos = (pos + rel) & mask;
    const uint64_t hdr = ld_tag(ring + my_pos);
    const uint64_t prev = ld_tag(ring + ((my_pos + cap - 8) & mask));  // footer of the previous record
    bool stop = false;
    uint64_t cur = pos;
    int s = 0;
    for (; s < 64; s++) {
      const uint64_t n = __shfl(hdr, s, 64);
      if (n == 0 || n > cap - GRDMA_RESERVED) {  // empty or torn header: not ready
        stop = true;
        *exhausted = true;
        break;
      }
      const uint64_t enc = 16 + round_up8(n);
      const uint64_t nxt = (cur + enc) & mask;
      uint64_t foot;
      bool predicted = false;
      if (s < 63) {
        const uint64_t npos = __shfl(my_pos, s + 1, 64);
        if (npos == nxt && (ea != 0)) {
          foot = __shfl(prev, s + 1, 64);
          predicted = true;
        }
      }
      if (!predicted) foot = ld_tag(ring + ((nxt + cap - 8) & mask));
      if (foot != GRDMA_FOOTER) {  // header landed, footer not yet: not ready
        stop = true;
        *exhausted = true;
        break;
      }
      if (lane == 0) chain[count] = n;
      count++;
      cs->e_prev2 = cs->e_prev1;
      cs->e_prev1 = enc;
      cur = nxt;
      if (!predicted) {  // continue from here with the corrected pattern
        s++;
        break;
      }
    }
    pos = cur;
    if (stop) break;
  }
  return count;
}

__global__ __launch_bounds__(64) void k_rx_plan(const grdma_rx_op* ops) {
  const grdma_rx_op op = ops[blockIdx.x];
  const int lane = threadIdx.x;
  grdma_conn* c = op.conn;
  grdma_plan* plan = op.plan;
  grdma_rx_result* res = op.result;
  uint8_t* ring = c->ring;
  const uint64_t cap = c->cap, mask = cap - 1;
  uint64_t head = c->head, mh = c->moving_head, remain = c->remain;
  uint64_t irs = c->internal_read_size, leftover = c->leftover_cap;
  const uint64_t mh0 = mh;
  uint64_t nslices = 0, nsegs = 0, ntiles = 0, bytes = 0, consumed_total = 0, records = 0;
  uint64_t a_off = 0, would_block = 0, credit = 0, credit_head = 0;
  const bool connected = c->status == GRDMA_PAIR_CONNECTED;
  grdma_slice_out* out_slices = op.slices;
  uint64_t max_slices = GRDMA_MAX_SLICES;
  if (op.append == 2) {  // first round of a streaming job
    if (lane == 0) {
      c->rx_arena_off = 0;
      c->rx_slice_idx = 0;
    }
  }
  if (op.append) {  // streaming job: keep filling the caller's buffer / slice table
    const uint64_t s_idx = op.append == 2 ? 0 : c->rx_slice_idx;
    a_off = op.append == 2 ? 0 : c->rx_arena_off;
    out_slices = op.slices + s_idx;
    const uint64_t room = op.slices_cap > s_idx ? op.slices_cap - s_idx : 0;
    if (room < max_slices) max_slices = room;
  }

  __shared__ uint64_t s_chain[CHAIN_CAP];
  chain_state cs = {0, 0};
  uint32_t chain_n = 0, chain_i = 0;
  bool chain_dry = false;  // the walk hit a position without a complete record
  uint64_t chain_pos = head;  // ring offset of the next unverified record

  // size of the next unopened record if it is completely there, else 0
  auto next_ready = [&]() -> uint64_t {
    if (chain_i == chain_n) {
      if (chain_dry) return 0;
      chain_n = chain_refill(ring, cap, chain_pos, &cs, s_chain, &chain_dry, lane);
      chain_i = 0;
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      // advance chain_pos past the verified records
      for (uint32_t k = 0; k < chain_n; k++) chain_pos = (chain_pos + 16 + round_up8(s_chain[k])) & mask;
      if (chain_n == 0) return 0;
    }
    return s_chain[chain_i];
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
    while (connected && nslices < op.max_reads && nslices < max_slices &&
           nsegs + 520 <= GRDMA_MAX_SEGS) {
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
  // arrival: release my stores, count in, last one publishes
  __shared__ unsigned int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int prev = __hip_atomic_fetch_add(&op.conn->rx_blocks_done, 1u, __ATOMIC_ACQ_REL,
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
