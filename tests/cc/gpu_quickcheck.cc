// TEST INFRASTRUCTURE.  A few seconds of GPU time: device-side checks of new code paths, each against
// the kernel's own verified path on the same input or against the CPU oracle (oracle/, linked here as
// the checker).  No Python: the process starts in well under a second, which is what is left of a
// round's GPU budget when this is needed.
//
//   h2 boundary step   k_h2_deframe with GRDMA_H2_BOUNDARY_STEP vs GRDMA_H2_NO_BOUNDARY_STEP: identical
//                      event lists on the bench shape (1 MiB messages as the receiving side sees them),
//                      on mixed sizes in sender and receiver shape, with slices at odd arena offsets;
//                      kernel time of both.
//   zero-copy send     grdma_pair_allocate_send_buffer / grdma_pair_send_zerocopy (k_tx_plan_zc + k_copy)
//                      against orc_pair_allocate_send_buffer / orc_pair_send_zerocopy: seeded sequences of
//                      zero-copy sends (buffer slices, ordinary slices, both), plain Sends and Recvs;
//                      accepted bytes, work requests, entry counts, ring image, state, delivered bytes.
//
// usage: gpu_quickcheck [out_file]      exit code 0 = every check passed
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "grdma_amd.h"
extern "C" {
#include "oracle/grdma_oracle.h"
}

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
  const run_result a = deframe(chunks, odd, GRDMA_H2_NO_BOUNDARY_STEP | GRDMA_H2_TICKS);
  const run_result b = deframe(chunks, odd, GRDMA_H2_BOUNDARY_STEP | GRDMA_H2_TICKS);
  const run_result c = deframe(chunks, odd, GRDMA_H2_BOUNDARY_STEP | GRDMA_H2_BULK_PAIRS);
  const run_result d = deframe(chunks, odd, GRDMA_H2_BOUNDARY_STEP);  // (no clock samples: the product's default)
  SAY("h2 without clock samples %-16s kernel_us %.1f (boundary step), %.1f (+ bulk pairs)\n", name, d.us, c.us);
  {
    bool okc = c.n == a.n && c.err == 0;
    for (size_t i = 0; okc && i < a.ev.size(); i++)
      if (memcmp(&a.ev[i], &c.ev[i], sizeof(grdma_h2_event)) != 0) okc = false;
    SAY("h2_bulk_pairs %-26s %s  events %lld/%lld bulk steps %llu (%llu frames)  kernel_us %.1f\n", name, okc ? "PASS" : "FAIL",
        (long long)a.n, (long long)c.n, (unsigned long long)c.st[0], (unsigned long long)c.st[1], c.us);
    if (!okc) g_fail++;
  }
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

// ---- zero-copy send against the oracle ------------------------------------------------------------
static uint32_t g_rng = 12345;
static uint32_t rnd() { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }

// every differing byte must be a padding byte (the device writes zeros there, the oracle's come from its
// staging history): it is zero on the device and the next word is the footer in both images
static bool ring_eq(const bytes& x, const bytes& y) {
  const size_t R = x.size();
  for (size_t i = 0; i < R; i++) {
    if (x[i] == y[i]) continue;
    if (x[i] != 0) return false;
    const size_t nxt = ((i & ~(size_t)7) + 8) % R;
    for (size_t k = 0; k < 8; k++)
      if (x[nxt + k] != 0xFF || y[nxt + k] != 0xFF) return false;
  }
  return true;
}

static bool zc_sequence(uint64_t R, int sge, uint64_t Z, uint32_t seed, int steps, std::string* why) {
  g_rng = seed;
  grdma_pair* a = grdma_pair_create(R, sge, 0);
  grdma_pair* b = grdma_pair_create(R, sge, 0);
  orc_pair oa, ob;
  if (!a || !b || grdma_pair_connect(a, b) != 0 || grdma_pair_enable_zerocopy(a, Z) != 0 ||
      orc_pair_init(&oa, R, sge) || orc_pair_init(&ob, R, sge) || orc_pair_enable_zerocopy(&oa, Z)) {
    *why = "setup failed";
    return false;
  }
  orc_pair_connect(&oa, &ob);
  uint8_t* zc_dev = nullptr;  // base of the device zero-copy buffer (the first allocation returns it)
  bool ok = true;
  const uint64_t sizes[] = {1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257, R / 3, R, Z / 2};
  for (int step = 0; step < steps && ok; step++) {
    const uint32_t op = rnd() % 100;
    char where[96];
    snprintf(where, sizeof(where), "R %llu sge %d Z %llu seed %u step %d", (unsigned long long)R, sge,
             (unsigned long long)Z, seed, step);
    if (op < 55) {
      const int n = 1 + rnd() % 5;
      std::vector<grdma_slice> dsl;
      std::vector<orc_slice> osl;
      std::vector<void*> dev_allocs;
      std::vector<bytes> host_keep;
      host_keep.reserve(n);
      for (int i = 0; i < n; i++) {
        uint64_t len = sizes[rnd() % (sizeof(sizes) / sizeof(sizes[0]))];
        if (len == 0) len = 1;
        const bool want_zc = rnd() % 2 == 0;
        bytes data(len);
        for (auto& v : data) v = (uint8_t)rnd();
        if (want_zc && len <= Z) {
          void* dptr = grdma_pair_allocate_send_buffer(a, len);
          uint8_t* optr = orc_pair_allocate_send_buffer(&oa, len);
          if ((dptr == nullptr) != (optr == nullptr)) { ok = false; *why = std::string("allocate disagrees, ") + where; break; }
          uint64_t off;
          if (dptr) {
            if (!zc_dev) zc_dev = static_cast<uint8_t*>(dptr);
            off = (uint64_t)(static_cast<uint8_t*>(dptr) - zc_dev);
            if (off != (uint64_t)(optr - oa.zc_buf)) { ok = false; *why = std::string("allocate offset, ") + where; break; }
          } else if (zc_dev && rnd() % 2) {
            off = rnd() % (Z - len + 1);  // any range of the buffer counts as inside
          } else {
            off = ~0ull;
          }
          if (off != ~0ull) {
            grdma_copy_to_device(zc_dev + off, data.data(), len);
            memcpy(oa.zc_buf + off, data.data(), len);
            dsl.push_back({zc_dev + off, len});
            osl.push_back({oa.zc_buf + off, len});
            continue;
          }
        }
        const uint64_t shift = rnd() % 16;
        uint8_t* d = static_cast<uint8_t*>(grdma_device_alloc(len + 16));
        dev_allocs.push_back(d);
        grdma_copy_to_device(d + shift, data.data(), len);
        host_keep.push_back(data);
        dsl.push_back({d + shift, len});
        osl.push_back({host_keep.back().data(), len});
      }
      if (!ok) break;
      const uint64_t bi = (rnd() % 10 < 3) ? rnd() % dsl[0].len : 0;
      const int64_t sd = grdma_pair_send_zerocopy(a, dsl.data(), dsl.size(), bi, 0);
      const uint64_t so = orc_pair_send_zerocopy(&oa, osl.data(), osl.size(), bi);
      uint64_t wr[2][2] = {{0, 0}, {0, 0}}, zs[4];
      const int nwr = grdma_pair_last_wrs(a, wr);
      grdma_pair_zerocopy_state(a, zs);
      if (sd != (int64_t)so) { ok = false; *why = std::string("accepted bytes differ, ") + where; }
      else if (nwr != oa.wr_count) { ok = false; *why = std::string("work request count, ") + where; }
      else if (zs[0] != oa.zc_tail || zs[1] != oa.zc_bytes || zs[2] != oa.copy_bytes || zs[3] != oa.sge_count) {
        ok = false;
        *why = std::string("zero-copy state (tail / bytes / copied / entries), ") + where;
      }
      for (int k = 0; ok && k < nwr; k++)
        if (wr[k][0] != oa.wr[k][0] || wr[k][1] != oa.wr[k][1]) { ok = false; *why = std::string("work requests, ") + where; }
      for (void* d : dev_allocs) grdma_device_free(d);
    } else if (op < 65) {
      const uint64_t len = sizes[rnd() % 14];
      bytes data(len);
      for (auto& v : data) v = (uint8_t)rnd();
      uint8_t* d = static_cast<uint8_t*>(grdma_device_alloc(len + 16));
      grdma_copy_to_device(d, data.data(), len);
      grdma_slice ds = {d, len};
      orc_slice os = {data.data(), len};
      const int64_t sd = grdma_pair_send(a, &ds, 1, 0, 0);
      const uint64_t so = orc_pair_send(&oa, &os, 1, 0);
      if (sd != (int64_t)so) { ok = false; *why = std::string("plain Send differs, ") + where; }
      grdma_device_free(d);
    } else {
      const uint64_t caps[] = {1, 8, 64, 256, R};
      const uint64_t cap = caps[rnd() % 5];
      bytes gd(cap), go(cap);
      const int64_t nd = grdma_pair_recv(b, gd.data(), cap, GRDMA_MEM_HOST);
      const uint64_t no = orc_pair_recv(&ob, go.data(), cap);
      if (nd != (int64_t)no || memcmp(gd.data(), go.data(), no) != 0) { ok = false; *why = std::string("Recv differs, ") + where; }
    }
    if (!ok) break;
    bytes ring(R), oring(ob.ring.buf, ob.ring.buf + R);
    grdma_pair_state sa, sb;
    if (grdma_pair_peek_ring(b, 0, ring.data(), R) != 0 || grdma_pair_state_get(a, &sa) != 0 ||
        grdma_pair_state_get(b, &sb) != 0) { ok = false; *why = std::string("peek failed, ") + where; break; }
    if (!ring_eq(ring, oring)) { ok = false; *why = std::string("ring image differs, ") + where; break; }
    if (sa.remote_tail != oa.remote_tail || sa.partial_write != (uint64_t)oa.partial_write ||
        sa.remote_head != oa.status_recv.remote_head || sb.head != ob.ring.head || sb.moving_head != ob.ring.moving_head ||
        sb.remain != ob.ring.remain || sb.internal_read_size != ob.internal_read_size || sb.credit_msgs != ob.credit_msgs) {
      ok = false;
      *why = std::string("state differs, ") + where;
    }
  }
  grdma_pair_destroy(a);
  grdma_pair_destroy(b);
  orc_pair_destroy(&oa);
  orc_pair_destroy(&ob);
  return ok;
}

static void check_zerocopy() {
  const struct { uint64_t R; int sge; uint64_t Z; } cfg[] = {
      {4096, 30, 8192}, {256, 5, 64}, {65536, 8, 4096}, {1024, 4, 2048}, {64, 100, 128}, {1u << 20, 30, 1u << 21}};
  int pass = 0, total = 0;
  for (uint32_t seed = 1; seed <= 3; seed++)
    for (const auto& c : cfg) {
      std::string why;
      total++;
      if (zc_sequence(c.R, c.sge, c.Z, seed * 7919u, 40, &why)) pass++;
      else {
        g_fail++;
        SAY("zerocopy FAIL: %s\n", why.c_str());
      }
    }
  SAY("zerocopy sequences vs oracle: %d / %d %s\n", pass, total, pass == total ? "PASS" : "FAIL");
}

// ---- armed read through the latency engine against the oracle ---------------------------------------
// unary ping-pong, [14 B][66 B] each way; armed = a standing read order carried out by a watcher workgroup (k_watch)
// State and rings must equal the oracle's after the same traffic, and
// equal the un-armed run's; prints both p50s.
static bool pingpong_run(bool armed, uint64_t iters, uint64_t* p50_ns, grdma_pair_state st[2], bytes rings[2],
                         int64_t* hits) {
  const uint64_t R = 4u << 20;
  grdma_pair* a = grdma_pair_create(R, 30, 0);
  grdma_pair* b = grdma_pair_create(R, 30, 0);
  if (!a || !b || grdma_pair_connect(a, b) != 0 || grdma_pair_set_latency_mode(a, 1) != 0 ||
      grdma_pair_set_latency_mode(b, 1) != 0)
    return false;
  if (armed && (grdma_pair_arm_read(a, 64) != 0 || grdma_pair_arm_read(b, 64) != 0)) return false;
  uint8_t m0[14], m1[66];
  for (int i = 0; i < 14; i++) m0[i] = (uint8_t)(i * 7 + 1);
  for (int i = 0; i < 66; i++) m1[i] = (uint8_t)(i * 5 + 3);
  const grdma_slice sl[2] = {{m0, 14}, {m1, 66}};
  std::vector<uint64_t> rtt(iters);
  uint64_t ph[4];
  bool ok = grdma_engine_start() == 0 &&
            grdma_pingpong(a, b, sl, 2, sl, 2, GRDMA_MEM_HOST, iters, 20, rtt.data(), ph) == 0;
  if (ok && armed) {
    *hits = grdma_pair_watch_hits(a) + grdma_pair_watch_hits(b);
    ok = grdma_pair_arm_read(a, 0) == 0 && grdma_pair_arm_read(b, 0) == 0;
  }
  grdma_engine_stop();
  if (ok) {
    std::sort(rtt.begin(), rtt.end());
    *p50_ns = rtt[iters / 2];
    grdma_pair* pp[2] = {a, b};
    for (int k = 0; k < 2 && ok; k++) {
      rings[k].resize(R);
      ok = grdma_pair_peek_ring(pp[k], 0, rings[k].data(), R) == 0 && grdma_pair_state_get(pp[k], &st[k]) == 0;
    }
  }
  grdma_pair_destroy(a);
  grdma_pair_destroy(b);
  return ok;
}

static void check_armed_read() {
  const uint64_t iters = 300, R = 4u << 20;
  uint64_t p50[2] = {0, 0};
  grdma_pair_state st[2][2];
  bytes rings[2][2];
  int64_t hits = 0;
  for (int armed = 0; armed < 2; armed++)
    if (!pingpong_run(armed != 0, iters, &p50[armed], st[armed], rings[armed], &hits)) {
      g_fail++;
      SAY("armed read FAIL: %s run: %s\n", armed ? "armed" : "plain", grdma_last_error());
      return;
    }
  // the oracle through the same traffic
  orc_pair o[2];
  if (orc_pair_init(&o[0], R, 30) || orc_pair_init(&o[1], R, 30)) { g_fail++; SAY("armed read FAIL: oracle setup\n"); return; }
  orc_pair_connect(&o[0], &o[1]);
  uint8_t m0[14], m1[66];
  for (int i = 0; i < 14; i++) m0[i] = (uint8_t)(i * 7 + 1);
  for (int i = 0; i < 66; i++) m1[i] = (uint8_t)(i * 5 + 3);
  const orc_slice os[2] = {{m0, 14}, {m1, 66}};
  bytes buf(4096);
  bool ok = true;
  for (uint64_t it = 0; it < iters + 20 && ok; it++)
    for (int dir = 0; dir < 2 && ok; dir++) {
      ok = orc_pair_send(&o[dir], os, 2, 0) == 80;
      uint64_t alloc = 0;
      while (orc_endpoint_read(&o[1 - dir], buf.data(), &alloc) != 0) {}
    }
  for (int armed = 0; armed < 2 && ok; armed++)
    for (int k = 0; k < 2 && ok; k++) {
      const grdma_pair_state& s = st[armed][k];
      const bytes oring(o[k].ring.buf, o[k].ring.buf + R);
      if (!ring_eq(rings[armed][k], oring)) { ok = false; SAY("armed read FAIL: ring image, armed %d side %d\n", armed, k); }
      if (s.remote_tail != o[k].remote_tail || s.head != o[k].ring.head || s.moving_head != o[k].ring.moving_head ||
          s.remain != o[k].ring.remain || s.internal_read_size != o[k].internal_read_size ||
          s.credit_msgs != o[k].credit_msgs || s.remote_head != o[k].status_recv.remote_head) {
        ok = false;
        SAY("armed read FAIL: state, armed %d side %d\n", armed, k);
      }
    }
  if (ok && hits != (int64_t)(2 * (iters + 20))) { ok = false; SAY("armed read FAIL: %lld watcher completions\n", (long long)hits); }
  orc_pair_destroy(&o[0]);
  orc_pair_destroy(&o[1]);
  if (!ok) g_fail++;
  SAY("armed read vs oracle (unary 64 B, %llu round trips): %s; RTT p50 plain %.2f us, armed %.2f us\n",
      (unsigned long long)iters, ok ? "PASS" : "FAIL", p50[0] / 1e3, p50[1] / 1e3);
}

int main(int argc, char** argv) {
  signal(SIGALRM, on_alarm);
  alarm(argc > 2 ? atoi(argv[2]) : 20);
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
  check_zerocopy();
  check_armed_read();  // last: a resident kernel; the alarm ends the run if it ever wedged
  SAY("gpu_quickcheck: %s\n", g_fail ? "FAILED" : "ALL PASS");
  if (g_out) fclose(g_out);
  return g_fail ? 1 : 0;
}
