// A few seconds of GPU time: device-side checks of code paths that are switched on by a flag, each
// against the kernel's own verified path on the same input (no Python, no oracle: the process starts in
// well under a second, which is what is left of a round's GPU budget when this is needed).
//
//   h2 boundary step   k_h2_deframe with GRDMA_H2_BOUNDARY_STEP vs GRDMA_H2_NO_BOUNDARY_STEP: identical
//                      event lists on the bench shape (1 MiB messages as the receiving side sees them),
//                      on mixed sizes in sender and receiver shape, with slices at odd arena offsets;
//                      kernel time of both.
//
// usage: gpu_quickcheck [out_file]      exit code 0 = every check passed
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "grdma_amd.h"

typedef std::vector<uint8_t> bytes;
static FILE* g_out = nullptr;
static int g_fail = 0;
#define SAY(...)                       \
  do {                                 \
    printf(__VA_ARGS__);               \
    fflush(stdout);                    \
    if (g_out) {                       \
      fprintf(g_out, __VA_ARGS__);     \
      fflush(g_out);                   \
    }                                  \
  } while (0)

static void on_alarm(int) {
  const char m[] = "gpu_quickcheck: TIMEOUT\n";
  if (write(1, m, sizeof(m) - 1) < 0) {}
  _exit(3);
}

static bytes frame_header(uint32_t len, uint8_t type, uint8_t flags, uint32_t sid) {
  bytes h = {(uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, type, flags,
             (uint8_t)(sid >> 24), (uint8_t)(sid >> 16), (uint8_t)(sid >> 8), (uint8_t)sid};
  return h;
}

// slices as chttp2 hands them to the endpoint (the 5-byte message header merged into the first frame's
// inlined header slice, payload slices by reference)
static std::vector<bytes> sender_slices(const std::vector<uint32_t>& sizes, uint32_t sid, bool end_stream) {
  std::vector<bytes> out;
  for (size_t i = 0; i < sizes.size(); i++) {
    const uint32_t n = sizes[i];
    bytes body(5 + (size_t)n);
    body[0] = 0;
    body[1] = (uint8_t)(n >> 24); body[2] = (uint8_t)(n >> 16); body[3] = (uint8_t)(n >> 8); body[4] = (uint8_t)n;
    for (uint32_t j = 0; j < n; j++) body[5 + j] = (uint8_t)((j * 13 + i) % 251);
    size_t off = 0;
    while (off < body.size()) {
      const size_t k = std::min<size_t>(16384, body.size() - off);
      const bool last = end_stream && i + 1 == sizes.size() && off + k == body.size();
      bytes fh = frame_header((uint32_t)k, 0, last ? 1 : 0, sid);
      if (off == 0) {
        fh.insert(fh.end(), body.begin(), body.begin() + 5);
        out.push_back(fh);
        if (k > 5) out.emplace_back(body.begin() + 5, body.begin() + k);
      } else {
        out.push_back(fh);
        out.emplace_back(body.begin() + off, body.begin() + off + k);
      }
      off += k;
    }
  }
  return out;
}

// what endpoint reads deliver for those records (reads of max(256, first record), rdma_bp_posix.cc:308)
static std::vector<bytes> receiver_slices(const std::vector<bytes>& tx) {
  const size_t first = 256;
  std::vector<bytes> out;
  bytes cur;
  size_t room = first;
  for (const bytes& r : tx) {
    size_t pos = 0;
    while (pos < r.size()) {
      if (room == 0) { out.push_back(cur); cur.clear(); room = first; }
      if (cur.empty() && pos == 0 && r.size() > first) { out.push_back(r); pos = r.size(); continue; }
      const size_t take = std::min(room, r.size() - pos);
      cur.insert(cur.end(), r.begin() + pos, r.begin() + pos + take);
      room -= take;
      pos += take;
      if (room == 0 && pos < r.size()) {
        out.push_back(cur);
        out.emplace_back(r.begin() + pos, r.end());
        cur.clear(); room = first; pos = r.size();
      }
    }
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

struct run_result {
  std::vector<grdma_h2_event> ev;
  int err = 0;
  int64_t n = 0;
  double us = 0;
  uint64_t steps = 0;
  uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

static run_result deframe(const std::vector<bytes>& chunks, bool odd_offsets, int flags) {
  run_result R;
  bytes arena;
  std::vector<grdma_read_slice> table;
  uint32_t g = 1;
  for (const bytes& c : chunks) {
    if (odd_offsets) { arena.insert(arena.end(), 1 + (g = g * 7 % 15), 0xEE); }
    else arena.resize((arena.size() + 15) & ~(size_t)15);
    grdma_read_slice s;
    memset(&s, 0, sizeof(s));
    s.off = arena.size();
    s.len = c.size();
    table.push_back(s);
    arena.insert(arena.end(), c.begin(), c.end());
  }
  arena.resize(arena.size() + 64);
  void* d = grdma_device_alloc(arena.size());
  if (!d || grdma_copy_to_device(d, arena.data(), arena.size()) != 0) { R.n = -1000; return R; }
  grdma_h2_parser* p = grdma_h2_parser_create_ex(flags, 16384, 0xffffffffu, 0);
  const uint32_t one = 1;
  if (!p || grdma_h2_parser_open_streams(p, &one, 1) != 0) { R.n = -1001; return R; }
  const uint64_t cap = 8 * chunks.size() + 4096;
  R.ev.resize(cap);
  R.n = grdma_h2_deframe(p, d, table.data(), table.size(), R.ev.data(), cap, &R.err);
  R.us = grdma_h2_last_kernel_us();
  R.steps = grdma_h2_last_boundary_steps();
  grdma_h2_last_deframe_stats(R.st);
  if (R.n >= 0) R.ev.resize((size_t)R.n);
  grdma_h2_parser_destroy(p);
  grdma_device_free(d);
  return R;
}

static void check_h2(const char* name, const std::vector<bytes>& body, bool odd, uint64_t min_steps) {
  std::vector<bytes> chunks;
  bytes h = frame_header(1, 1, 4, 1);  // HEADERS(END_HEADERS) on stream 1, one HPACK byte
  h.push_back(0x82);
  chunks.push_back(h);
  chunks.insert(chunks.end(), body.begin(), body.end());
  const run_result a = deframe(chunks, odd, GRDMA_H2_NO_BOUNDARY_STEP);
  const run_result b = deframe(chunks, odd, GRDMA_H2_BOUNDARY_STEP);
  bool ok = a.n > 0 && b.n == a.n && a.err == 0 && b.err == 0 && a.steps == 0 && b.steps >= min_steps;
  size_t first_diff = 0;
  if (ok) {
    for (size_t i = 0; i < a.ev.size(); i++)
      if (memcmp(&a.ev[i], &b.ev[i], sizeof(grdma_h2_event)) != 0) { ok = false; first_diff = i; break; }
  }
  SAY("h2_boundary %-28s %s  slices %zu events %lld/%lld err %d/%d steps %llu  kernel_us off %.1f on %.1f\n", name,
      ok ? "PASS" : "FAIL", chunks.size(), (long long)a.n, (long long)b.n, a.err, b.err,
      (unsigned long long)b.steps, a.us, b.us);
  SAY("  ticks on:  bulk steps %llu (%llu frames) boundary %llu | wait %llu bulk %llu boundary %llu bytewise %llu total %llu\n",
      (unsigned long long)b.st[0], (unsigned long long)b.st[1], (unsigned long long)b.st[2], (unsigned long long)b.st[3],
      (unsigned long long)b.st[4], (unsigned long long)b.st[5], (unsigned long long)b.st[6], (unsigned long long)b.st[7]);
  SAY("  ticks off: bulk steps %llu (%llu frames) | wait %llu bulk %llu bytewise %llu total %llu\n",
      (unsigned long long)a.st[0], (unsigned long long)a.st[1], (unsigned long long)a.st[3], (unsigned long long)a.st[4],
      (unsigned long long)a.st[6], (unsigned long long)a.st[7]);
  if (!ok) {
    g_fail++;
    if (a.n > 0 && b.n > 0 && first_diff < a.ev.size() && first_diff < b.ev.size()) {
      const grdma_h2_event &x = a.ev[first_diff], &y = b.ev[first_diff];
      SAY("  first difference at event %zu: off {%u %u %u %u %u sl %u} on {%u %u %u %u %u sl %u}\n", first_diff, x.kind,
          x.a, x.b, x.c, x.d, x.slice, y.kind, y.a, y.b, y.c, y.d, y.slice);
    }
  }
}

int main(int argc, char** argv) {
  signal(SIGALRM, on_alarm);
  alarm(argc > 2 ? atoi(argv[2]) : 12);
  if (argc > 1) g_out = fopen(argv[1], "w");
  if (grdma_init(0) != 0) {
    SAY("gpu_quickcheck: grdma_init failed: %s\n", grdma_last_error());
    return 2;
  }
  SAY("gpu_quickcheck: device ready\n");
  {
    std::vector<uint32_t> sizes(24, 1u << 20);
    const std::vector<bytes> tx = sender_slices(sizes, 1, false);
    check_h2("bench shape, receiver", receiver_slices(tx), false, 24);
    check_h2("bench shape, sender", tx, false, 24);
  }
  {
    const std::vector<uint32_t> sizes = {1u << 20, 16384 * 3 - 5, 40000, 16384 - 5, 7, 16384 * 70 + 123, 1, 300000,
                                         5, 2, 16379, 16380, 9, 100, 20000};
    const std::vector<bytes> tx = sender_slices(sizes, 1, true);
    check_h2("mixed sizes, sender", tx, false, 6);
    check_h2("mixed sizes, receiver, odd", receiver_slices(tx), true, 4);
    // some slices cut in two
    std::vector<bytes> cut;
    size_t j = 0;
    for (const bytes& s : tx) {
      if (s.size() > 100 && (j++ % 5) == 2) {
        cut.emplace_back(s.begin(), s.begin() + 77);
        cut.emplace_back(s.begin() + 77, s.end());
      } else {
        cut.push_back(s);
      }
    }
    check_h2("mixed sizes, cut slices, odd", cut, true, 1);
  }
  SAY("gpu_quickcheck: %s\n", g_fail ? "FAILED" : "ALL PASS");
  if (g_out) fclose(g_out);
  return g_fail ? 1 : 0;
}
