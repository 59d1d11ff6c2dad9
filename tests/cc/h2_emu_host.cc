// TEST INFRASTRUCTURE.  k_h2_deframe -- the real kernel source, csrc/grdma_h2_kernels.h -- run on the CPU
// under the wave emulator (tests/cc/wave_emu.h): 512 emulated threads, the parsing wave with its ballots and
// prefix sums, the seven staging waves and the LDS look-ahead ring between them.  tests/test_h2_emu.py
// compares the events with the oracle's, with and without the boundary step and GRDMA_H2_BULK_PAIRS.
#include "wave_emu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../grpc-rdma_amd/csrc/grdma_h2_kernels.h"

// flags: grdma_h2_parser_flags (include/grdma_amd.h).  slices: {offset, length} into arena.  Returns the number
// of events (or -1 when the event array overflowed); *h2_error = connection error; stats = {bulk steps, frames in
// bulk steps, boundary steps, slices parsed}.
extern "C" int64_t h2_emu_deframe(int flags, uint32_t max_frame, const uint32_t* open_ids, uint32_t n_open,
                                  const uint8_t* arena, uint64_t arena_len, const uint64_t* table, uint64_t nslices,
                                  uint32_t* ev_out /* 6 words per event */, uint64_t cap, int* h2_error, uint64_t stats[4]) {
  // parser block as grdma_h2_parser_create_ex builds it
  const uint32_t table_slots = 4096;
  std::vector<grdma_h2_stream_dev> tab(table_slots);
  memset(tab.data(), 0, sizeof(grdma_h2_stream_dev) * table_slots);
  grdma_h2_parser_dev P;
  memset(&P, 0, sizeof(P));
  P.is_server = (flags & GRDMA_H2_SERVER) ? 1 : 0;
  P.is_first_frame = (flags & GRDMA_H2_FIRST_FRAME) ? 1 : 0;
  P.state = P.is_server ? 0 : 24;
  P.max_frame_size = max_frame;
  P.max_concurrent = 0xffffffffu;
  P.tab_mask = table_slots - 1;
  P.boundary_step = (flags & GRDMA_H2_BOUNDARY_STEP) ? 1 : 0;
  P.bulk_pairs = (flags & GRDMA_H2_BULK_PAIRS) ? 1 : 0;
  P.ticks = (flags & GRDMA_H2_TICKS) ? 1 : 0;
  P.tab = tab.data();
  if (n_open) {
    std::vector<grdma_h2_table_op> ops(n_open);
    for (uint32_t i = 0; i < n_open; i++) ops[i] = {1, open_ids[i], 0, 0};
    grdma_h2_parser_dev* gp = &P;
    grdma_h2_table_op* po = ops.data();
    emu::launch(dim3(1), dim3(1), [&] { k_h2_table_ops(gp, po, n_open); });
    for (uint32_t i = 0; i < n_open; i++)
      if (ops[i].rc != 0) return -2;
  }
  // the arena with slack behind it (the staging waves read whole 16-byte blocks)
  std::vector<uint8_t> mem(arena_len + 128 + 16);
  uint8_t* base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  memcpy(base, arena, arena_len);
  std::vector<grdma_slice_out> sl(nslices ? nslices : 1);
  for (uint64_t i = 0; i < nslices; i++) {
    sl[i].off = table[2 * i];
    sl[i].len = table[2 * i + 1];
  }
  std::vector<grdma_h2_event> ev(cap ? cap : 1);
  grdma_h2_deframe_result res;
  memset(&res, 0, sizeof(res));
  {
    grdma_h2_parser_dev* gp = &P;
    const uint8_t* a = base;
    const grdma_slice_out* s = sl.data();
    grdma_h2_event* e = ev.data();
    grdma_h2_deframe_result* r = &res;
    emu::launch(dim3(1), dim3(H2_DEFRAME_THREADS), [&] { k_h2_deframe(gp, a, s, nslices, e, cap, r); });
  }
  const uint64_t m = res.nevents < cap ? res.nevents : cap;
  for (uint64_t i = 0; i < m; i++) {
    uint32_t* o = ev_out + 6 * i;
    o[0] = ev[i].kind; o[1] = ev[i].a; o[2] = ev[i].b; o[3] = ev[i].c; o[4] = ev[i].d; o[5] = ev[i].slice;
  }
  if (h2_error) *h2_error = (int)res.error;
  if (stats) {
    stats[0] = res.bulk_steps;
    stats[1] = res.bulk_frames;
    stats[2] = res.boundary_steps;
    stats[3] = res.slices_done;
  }
  return res.overflow ? -1 : (int64_t)m;
}
