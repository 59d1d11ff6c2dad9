"""GPU parity: the HIP pair (ring codec + credit accounting + endpoint read
replay) against the CPU oracle on the same seeded inputs, bit-exact.

Mirrors what a pair-level test of the reference would check
(src/core/lib/ibverbs/pair.cc Send/Recv, ring_buffer.cc Read/Write): accepted
byte counts, the ring image after every step, head/tail/credit state, delivered
slices, the zero-after-read invariant.
"""
import random

import pytest

from oracle import pyorc

pytestmark = pytest.mark.gpu

STATE_KEYS = ["head", "moving_head", "remain", "remote_tail", "remote_head",
              "internal_read_size", "credit_msgs", "partial_write"]


def mk_link(g, R, sge, flags=0):
    a, b = g.Pair(R, sge, flags), g.Pair(R, sge, flags)
    g.connect_pairs(a, b)
    return a, b


def dev_slices(g, slices, rng):
    return [g.DeviceBuffer(data=s, offset=rng.randrange(16)) for s in slices]


def check_state(a, b, o):
    sa, sb = a.state(), b.state()
    oa, ob = o.state(0), o.state(1)
    for k in STATE_KEYS:
        assert sa[k] == oa[k], ("side0", k, sa, oa)
        assert sb[k] == ob[k], ("side1", k, sb, ob)


@pytest.mark.parametrize("flags", [0, 2, 4, 6], ids=["staged", "direct", "staged-finegrained", "direct-finegrained"])
@pytest.mark.parametrize("seed", range(6))
def test_random_ops_match_oracle(gpu, seed, flags):
    g = gpu
    rng = random.Random(1000 + seed)
    R = rng.choice([64, 256, 4096, 65536])
    sge = rng.choice([1, 3, 30, 200])
    a, b = mk_link(g, R, sge, flags)
    o = pyorc.OracleLink(R, sge)
    sizes = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257, R // 3, R]
    for step in range(40):
        op = rng.random()
        if op < 0.5:
            n = rng.randint(1, 8)
            sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes))) for _ in range(n)]
            bi = rng.randrange(len(sl[0])) if rng.random() < 0.3 else 0
            bufs = dev_slices(g, sl, rng)
            s_g = a.Send(bufs, bi)
            s_o = o.send(0, sl, bi)
            assert s_g == s_o, (seed, step, s_g, s_o)
            assert a.last_wrs() == o.last_wrs(0)
            if not flags:
                used = sum(16 + ((x + 7) & ~7) for x in [0])  # placeholder, real check below
                st_o = o.staging_mem(0)
                st_g = a.staging_mem(len(st_o))
                # padding bytes: the reference leaves stale staging bytes, the HIP
                # encoder writes zeros; the oracle's staging starts zeroed and sees
                # the same history, so compare with the pad masked out.
                assert _mask_pads(st_g) == _mask_pads(st_o)
        elif op < 0.75:
            cap = rng.choice([1, 3, 8, 64, 256, R])
            assert b.Recv(cap) == o.recv(1, cap)
        else:
            got, _wb = b.endpoint_read(1)
            exp, _alloc = o.endpoint_read(1)
            assert (got[0] if got else b"") == exp
        assert _ring_eq(b.ring_mem(), o.ring_mem(1)), (seed, step)
        check_state(a, b, o)
        assert b.GetReadableSize() == o.readable(1)
        assert b.HasMessage() == o.has_message(1)
        assert a.GetWritableSize() == o.writable(0)
    a.close(); b.close(); o.close()


def _records(buf):
    """Yield (offset, payload_len) of back-to-back records in a staging image."""
    off = 0
    while off + 16 <= len(buf):
        n = int.from_bytes(buf[off:off + 8], "little")
        if n == 0:
            break
        yield off, n
        off += 16 + ((n + 7) & ~7)


def _mask_pads(buf):
    out = bytearray(buf)
    for off, n in _records(buf):
        for q in range(off + 8 + n, off + 8 + ((n + 7) & ~7)):
            out[q] = 0
    return bytes(out)


def _ring_eq(x, y):
    # Oracle staging starts zeroed and both sides see the same send history, but
    # the oracle's pad bytes can hold stale bytes of an earlier, longer record.
    # Compare everything except pad bytes, which are located from the record tags
    # still in the ring (both images must agree on every tag word).
    if x == y:
        return True
    R = len(x)
    diff = [i for i in range(R) if x[i] != y[i]]
    # every differing byte must be a pad byte: the HIP side holds 0 there
    for i in diff:
        if x[i] != 0:
            return False
        w = i & ~7
        # pad bytes live in the last payload word of a record: the next word is the footer
        nxt = (w + 8) % R
        if x[nxt:nxt + 8] != b"\xff" * 8 or y[nxt:nxt + 8] != b"\xff" * 8:
            return False
    return True


def test_streaming_one_mib_messages(gpu):
    """Config 3 shape at parity-test size: three 1 MiB messages framed at 16 KiB,
    4 MiB ring, max_sge 30; accepted bytes, ring image and delivered slices
    identical to the oracle at every step."""
    g = gpu
    R = 4 << 20
    a, b = mk_link(g, R, 30)
    o = pyorc.OracleLink(R, 30)
    rng = random.Random(7)
    msg = bytes(i % 251 for i in range(1 << 20))
    wire, lens = pyorc.h2_frame_message(msg, stream_id=1)
    assert len(lens) == 130
    slices, off = [], 0
    for n in lens:
        slices.append(wire[off:off + n])
        off += n
    bufs = dev_slices(g, slices, rng)
    for _ in range(3):
        idx, bidx = 0, 0
        while idx < len(slices):
            s_g = a.Send(bufs[idx:], bidx)
            s_o = o.send(0, slices[idx:], bidx)
            assert s_g == s_o
            assert _ring_eq(b.ring_mem(), o.ring_mem(1))
            sent = s_g
            while sent > 0:
                left = len(slices[idx]) - bidx
                if sent >= left:
                    sent -= left; idx += 1; bidx = 0
                else:
                    bidx += sent; sent = 0
            got, _ = b.endpoint_read(4096)
            exp = []
            while True:
                s, _al = o.endpoint_read(1)
                if not s:
                    break
                exp.append(s)
            # the oracle's final would-block attempt is also replayed by the drain
            assert got == exp
            assert b.ring_mem() == o.ring_mem(1)
            check_state(a, b, o)
    assert b.ring_mem() == bytes(R)
    a.close(); b.close(); o.close()


def test_poll_batch_ballot(gpu):
    """K3 batched: 100 connections, a known subset has a complete record, another
    subset only a header (HasMessage true, readable 0)."""
    g = gpu
    R = 4096
    links = [mk_link(g, R, 30) for _ in range(100)]
    rng = random.Random(3)
    expect_r, expect_h = [], []
    for i, (a, b) in enumerate(links):
        if i % 3 == 0:
            n = rng.randint(1, 500)
            a.Send([g.DeviceBuffer(data=bytes(n))])
            expect_r.append(n); expect_h.append(True)
        else:
            expect_r.append(0); expect_h.append(False)
    rd, hm = g.poll_pairs([b for _, b in links])
    assert rd == expect_r and hm == expect_h


@pytest.mark.parametrize("flags", [0, 2], ids=["staged", "direct"])
@pytest.mark.parametrize("R,sizes,seed", [
    (1 << 16, [1, 9, 14, 100, 255, 256, 257, 510, 511, 512, 600, 2000], 1),
    (1 << 16, [9, 16384], 2),
    (1 << 20, [9, 14, 16379, 16384, 255, 256], 3),
    (1 << 12, [1, 2, 3, 8, 9], 4),
    (1 << 18, [300, 511, 700, 5000, 16384], 5),
])
def test_drain_many_records(gpu, R, sizes, seed, flags):
    """Fill the ring with many records, then drain with one device pass
    (max_reads large): exercises the 64-probe chain walker, the one-lane-per-
    record replay, ring wrap and the cap/2 credit rule over several cycles."""
    g = gpu
    rng = random.Random(seed)
    a, b = mk_link(g, R, 4095, flags)
    o = pyorc.OracleLink(R, 4095)
    pending = []
    for cycle in range(6):
        # queue slices until the send comes up short
        while True:
            sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes)))
                  for _ in range(rng.randint(1, 60))]
            bufs = dev_slices(g, sl, rng)
            s_g = a.Send(bufs); s_o = o.send(0, sl)
            assert s_g == s_o
            if s_g < sum(len(x) for x in sl) or rng.random() < 0.15:
                break
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))
        # sometimes leave the reader mid-record first
        if rng.random() < 0.5:
            cap = rng.choice([1, 5, 100, 256, 300])
            assert b.Recv(cap) == o.recv(1, cap)
        limit = rng.choice([4096, 4096, 7, 1])
        got, wb = b.endpoint_read(limit)
        exp = []
        while len(exp) < limit:
            s, _al = o.endpoint_read(1)
            if not s:
                break
            exp.append(s)
        assert [len(x) for x in got] == [len(x) for x in exp]
        assert got == exp
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))  # unread records may differ in pad bytes only
        check_state(a, b, o)
    a.close(); b.close(); o.close()


@pytest.mark.parametrize("pattern,R", [
    ([9, 16384], 1 << 20),                 # period 2
    ([14, 1000, 9, 1000, 9, 200, 9, 9], 1 << 18),  # period 8 with small-record runs
    ([700], 1 << 16),                      # period 1
    ([9, 300, 9, 300, 9, 255], 1 << 17),
])
def test_periodic_stream_bulk_tier(gpu, pattern, R):
    """A strictly periodic record stream (what a gRPC stream of equal messages
    looks like) sent in bursts that never split a record: after the history has
    seen a few periods the 256-thread bulk tier takes over.  Several ring laps so
    the wrap and the cap/2 credit rule are crossed inside bulk passes."""
    g = gpu
    rng = random.Random(11)
    a, b = mk_link(g, R, 4095)
    o = pyorc.OracleLink(R, 4095)
    period_bytes = sum(16 + ((n + 7) & ~7) for n in pattern)
    bufs_one = [bytes(rng.getrandbits(8) for _ in range(n)) for n in pattern]
    dev_one = dev_slices(g, bufs_one, rng)
    for lap in range(10):
        reps = max(1, min(4000 // len(pattern), (R // 2 - 64) // period_bytes, o.writable(0) // period_bytes))
        sl = bufs_one * reps
        s_g = a.Send(dev_one * reps); s_o = o.send(0, sl)
        assert s_g == s_o == sum(len(x) for x in sl)
        got, wb = b.endpoint_read(8192)
        exp = []
        while True:
            s, _al = o.endpoint_read(1)
            if not s:
                break
            exp.append(s)
        assert [len(x) for x in got] == [len(x) for x in exp]
        assert got == exp
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))
        check_state(a, b, o)
    a.close(); b.close(); o.close()


class _GpuLinkAdapter:
    """OracleLink-shaped view of two HIP pairs so that the golden replayer of
    tests/test_golden_ring.py drives the device path."""

    def __init__(self, g, R, sge):
        self.g = g
        self.a, self.b = mk_link(g, R, sge)
        self.rng = random.Random(99)

    def send(self, side, slices, byte_idx=0):
        return self.a.Send(dev_slices(self.g, slices, self.rng), byte_idx)

    def recv(self, side, cap):
        return self.b.Recv(cap)

    def endpoint_read(self, side):
        before = self.b.state()["leftover_cap"]
        readable = self.b.GetReadableSize()
        got, _ = self.b.endpoint_read(1)
        return (got[0] if got else b""), (before or max(256, readable))

    def last_wrs(self, side):
        return self.a.last_wrs()

    def ring_mem(self, side):
        return self.b.ring_mem()

    def staging_mem(self, side):
        return b""

    def state(self, side):
        return (self.a if side == 0 else self.b).state()

    def readable(self, side):
        return self.b.GetReadableSize()

    def has_message(self, side):
        return self.b.HasMessage()

    def writable(self, side):
        return self.a.GetWritableSize()


def test_golden_reference_traces_on_gpu(gpu):
    """The traces generated from the reference-built codec (tests/golden/ring_*.json)
    replayed on the HIP pair: accepted bytes, work requests, delivered bytes (sha256),
    reader/sender state, readable/writable after every step; ring image where the
    vector carries it (pad bytes excepted: the reference leaves stale staging bytes
    there, the HIP encoder writes zeros)."""
    import glob, json, os
    from tests.test_golden_ring import replay
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ring_*.json"))):
        doc = json.load(open(path))
        link = _GpuLinkAdapter(gpu, doc["ring_size"], doc["max_sge"])
        replay(doc, link, ring_exact=False, mask=_ring_eq)
        link.a.close(); link.b.close()


@pytest.mark.parametrize("flags", [0, 2], ids=["staged", "direct"])
@pytest.mark.parametrize("seed", range(4))
def test_latency_mode_matches_oracle(gpu, seed, flags):
    """Latency mode (one fused launch per call, pinned arena, host bytes through the
    bounce buffer) must be byte-identical to the batched path and the oracle."""
    g = gpu
    rng = random.Random(500 + seed)
    R = rng.choice([4096, 65536, 4 << 20])
    a, b = mk_link(g, R, 30, flags)
    a.set_latency_mode(True); b.set_latency_mode(True)
    o = pyorc.OracleLink(R, 30)
    sizes = [1, 9, 14, 64, 66, 255, 256, 300, 1000, 5000]
    for step in range(40):
        if rng.random() < 0.55:
            sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(sizes))) for _ in range(rng.randint(1, 6))]
            assert a.Send(sl) == o.send(0, sl)           # host bytes
        else:
            got, _ = b.endpoint_read(rng.choice([1, 3, 64]))
            exp = []
            while len(exp) < len(got) or (not got and not exp):
                s_, _al = o.endpoint_read(1)
                if not s_:
                    break
                exp.append(s_)
            assert got == exp
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))
        check_state(a, b, o)
    a.close(); b.close(); o.close()


def test_pingpong_unary_64b(gpu):
    """Config 2 shape: 64-byte unary ping-pong, request and response each framed as
    chttp2 would ([14-byte inlined header slice][66-byte proto])."""
    g = gpu
    from grpc_rdma_amd import h2
    msg = bytes([0x0A, 64]) + bytes(range(64))
    items = h2.frame_message(len(msg), 1)
    slices = [i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items]
    assert [len(s) for s in slices] == [14, 66]
    a, b = mk_link(g, 4 << 20, 30)
    a.set_latency_mode(True); b.set_latency_mode(True)
    rtt, ph = g.pingpong(a, b, slices, slices, iters=200, warmup=20)
    assert len(rtt) == 200 and min(rtt) > 0
    # state after 220 round trips equals the oracle's
    o = pyorc.OracleLink(4 << 20, 30)
    for _ in range(220):
        for src, dst in ((0, 1), (1, 0)):
            assert o.send(src, slices) == 80
            while True:
                s_, _al = o.endpoint_read(dst)
                if not s_:
                    break
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)


@pytest.mark.parametrize("sizes", [[14, 66], [5, 9], [80], [14, 66, 30, 1], [200, 40]],
                         ids=["unary64", "tiny", "one", "four", "near256"])
def test_express_drain_matches_oracle(gpu, sizes):
    """Latency mode, small unary messages: the single-wave express drain must leave exactly
    what the reference's rdma_continue_read / rdma_do_read loop leaves -- the delivered
    slices, the unfilled tail kept in last_read_buffer (it shrinks from drain to drain until
    a record no longer fits and is cut, which falls back to the general tiers), ring image,
    credit and head state -- message after message."""
    g = gpu
    lib = g.load()
    import ctypes as C
    lib.grdma_express_drains.restype = C.c_uint64
    before = lib.grdma_express_drains()
    rng = random.Random(77 + len(sizes))
    R = 8192  # small ring: the cap/2 credit rule is crossed many times
    a, b = mk_link(g, R, 30)
    a.set_latency_mode(True); b.set_latency_mode(True)
    o = pyorc.OracleLink(R, 30)
    for it in range(120):
        sl = [bytes(rng.getrandbits(8) for _ in range(n)) for n in sizes]
        assert a.Send(sl) == o.send(0, sl)
        got, wb = b.endpoint_read(64)
        exp = []
        while True:
            s_, _al = o.endpoint_read(1)
            if not s_:
                break
            exp.append(s_)
        assert got == exp, it
        assert wb
        assert _ring_eq(b.ring_mem(), o.ring_mem(1))
        check_state(a, b, o)
        assert b.state()["leftover_cap"] == o.p[1].leftover_cap, it
        if it % 10 == 9:  # a drain that finds nothing
            got, wb = b.endpoint_read(64)
            s_, _al = o.endpoint_read(1)
            assert got == [] and not s_ and wb
            check_state(a, b, o)
            assert b.state()["leftover_cap"] == o.p[1].leftover_cap, it
    # (the probe round predicts sizes that alternate, and a message only takes the express path
    # while it still fits the shrinking open read: small two-record messages do most of the time)
    if len(sizes) <= 2 and sum(sizes) <= 128:
        assert lib.grdma_express_drains() - before >= 60, "the express path was not exercised"
    a.close(); b.close(); o.close()


def test_ring_exports_as_dmabuf(gpu):
    """The HBM ring exported as a dma-buf fd (what ibv_reg_dmabuf_mr takes, where the reference
    registers host memory with ibv_reg_mr, pair.cc:107-119): a valid, closable descriptor; two
    exports give two descriptors."""
    import os
    g = gpu
    lib = g.load()
    a = g.Pair(1 << 20, 30, 4)
    fd = lib.grdma_pair_export_ring_dmabuf(a.h)
    if fd < 0:
        msg = lib.grdma_last_error().decode()
        a.close()
        pytest.skip("dma-buf export not available on this kernel / driver: " + msg)
    fd2 = lib.grdma_pair_export_ring_dmabuf(a.h)
    assert fd >= 3 and fd2 >= 3 and fd2 != fd
    st = os.fstat(fd)
    assert st.st_size in (0, 1 << 20) or st.st_size >= (1 << 20)
    os.close(fd)
    os.close(fd2)
    a.close()
