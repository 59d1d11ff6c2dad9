// TEST INFRASTRUCTURE.  Host check of the deframer's message-boundary step
// (grpc-rdma_amd/csrc/grdma_h2_fast.h): the same two functions k_h2_deframe calls are run here
// inside the CPU oracle's parser -- wherever the kernel's preconditions hold and the slice
// matches, the step's events are taken and the oracle's state is advanced the way the kernel
// advances its registers; everything else is fed to the oracle's automaton.  The caller
// (tests/test_h2_fast_host.py) compares the result with the oracle alone.
#include <stdint.h>
#include <string.h>

extern "C" {
#include "../../oracle/grdma_oracle.h"
}
#include "../../grpc-rdma_amd/csrc/grdma_h2_fast.h"

namespace {
orc_h2_stream* find_stream(orc_h2_parser* p, uint32_t id) {
  for (uint64_t i = 0; i < p->nstreams; i++)
    if (p->streams[i].stream_id == id) return &p->streams[i];
  return nullptr;
}
uint64_t le64(const uint8_t* b, uint64_t have) {
  uint64_t v = 0;
  for (uint64_t i = 0; i < 8 && i < have; i++) v |= (uint64_t)b[i] << (8 * i);
  return v;
}
}  // namespace

extern "C" int h2fast_hybrid_parse(int flags, uint32_t max_frame, uint32_t max_concurrent, const uint32_t* open_ids,
                                   uint64_t n_open, const uint8_t* data, const uint64_t* lens, uint64_t nslices,
                                   int use_step, uint32_t* ev_out /* 6 words per event */, uint64_t cap,
                                   uint64_t* nev_out, uint64_t* steps_out) {
  orc_h2_parser p;
  orc_h2_parser_init_ex(&p, flags, max_frame, max_concurrent);
  for (uint64_t i = 0; i < n_open; i++) orc_h2_parser_open_stream(&p, open_ids[i]);
  orc_h2_event* tmp = new orc_h2_event[cap ? cap : 1];
  uint64_t nev = 0, steps = 0, off = 0;
  int rc = 0;
  for (uint64_t s = 0; s < nslices && rc == 0; s++) {
    const uint8_t* sl = data + off;
    const uint64_t len = lens[s];
    // the kernel's preconditions: at a frame header, no header block open, not the first
    // frame, a current stream that is open for reads
    orc_h2_stream* D = find_stream(&p, p.incoming_stream_id);
    if (use_step && p.state == 24 && p.expect_continuation_stream_id == 0 && !p.is_first_frame && D &&
        !D->read_closed && (D->state == 0 || (D->state == 5 && D->frame_size - 1u < 9u))) {
      uint64_t c[4];
      for (int q = 0; q < 4; q++) c[q] = len > 8ull * q ? le64(sl + 8 * q, len - 8ull * q) : 0;
      const uint64_t next_len = s + 1 < nslices ? lens[s + 1] : ~0ull;
      const h2_bstep B = h2_boundary_match(c[0], c[1], c[2], c[3], len, next_len, D->state, D->frame_size,
                                           D->stream_id, p.max_frame_size);
      if (B.ok && nev + B.nev <= cap) {
        for (uint32_t k = 0; k < B.nev; k++) h2_boundary_event(B, D->stream_id, (uint32_t)s, k, ev_out + 6 * (nev + k));
        nev += B.nev;
        // what the kernel's registers hold after the step
        D->state = B.rem ? 5 : 0;
        D->frame_size = B.rem;
        D->compressed = (int)B.comp;
        p.incoming_frame_size = 0;
        p.incoming_frame_type = 0;
        p.incoming_frame_flags = 0;
        p.incoming_stream_id = D->stream_id;
        p.cur_parser = 1;
        p.received_last_frame = 0;
        steps++;
        off += len;
        if (B.nslices == 2) {
          off += lens[s + 1];
          s++;
        }
        continue;
      }
    }
    uint64_t n = 0;
    rc = orc_h2_parser_feed(&p, sl, len, tmp, cap - nev, &n);
    for (uint64_t k = 0; k < n; k++) {
      uint32_t* e = ev_out + 6 * (nev + k);
      e[0] = tmp[k].kind; e[1] = tmp[k].a; e[2] = tmp[k].b; e[3] = tmp[k].c; e[4] = tmp[k].d; e[5] = (uint32_t)s;
    }
    nev += n;
    off += len;
  }
  delete[] tmp;
  orc_h2_parser_free(&p);
  *nev_out = nev;
  if (steps_out) *steps_out = steps;
  return rc;
}
