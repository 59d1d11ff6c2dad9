"""GPU: arrival-triggered reads (k_watch).  A standing read order (grdma_pair_arm_read) is carried out by a resident
WATCHER workgroup of the latency engine the moment the sender's arrival report -- on an ordered wire: a complete
record -- shows up in the connection's own ring, whoever wrote it; grdma_endpoint_read is then a look at pinned host
memory.  What the reference's busy-polling thread is to an outstanding grpc_endpoint_read
(ring_buffer.cc:56-97, ev_epollex_rdma_bpev_linux.cc:1105-1149, poller.cc:84; rdma_bp_posix.cc:343-376).

Every record goes THROUGH the ring (the only way there is: round 4's drain chained into the peer's send command, with
its cut-through of unary-sized records, was retired when the watchers came);
bytes, connection state, record-size histories and both ring images equal the oracle's after the same sequence of
sends and reads.  Runs under the emulator too (tests/test_emu_gpu_suite.py), where engine and watcher are threads."""
import ctypes as C
import random
import time

import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import STATE_KEYS, mk_link

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


def read_when_ready(p, max_reads=64, timeout=20.0):
    """The completion of the standing order, once the watcher has produced it."""
    t0 = time.time()
    while not p.armed_ready():
        assert time.time() - t0 < timeout, "the watcher never delivered"
    return p.endpoint_read(max_reads)


def oracle_pingpong(o, rounds):
    open_read = {0: None, 1: None}
    for sa, sb, iters in rounds:
        for _ in range(iters):
            for src, dst, sl in ((0, 1, sa), (1, 0, sb)):
                assert o.send(src, sl) == sum(len(x) for x in sl)
                while True:
                    got, alloc = o.endpoint_read(dst)
                    if not got:
                        open_read[dst] = alloc
                        break
    return open_read


@pytest.mark.parametrize("sizes", [[14, 66], [14, 600], [9, 2000, 5]], ids=["unary64", "two_records", "beyond_the_fast_lane"])
def test_watched_reads_go_through_the_ring_and_match_the_oracle(gpu, sizes, monkeypatch):
    g = gpu
    lib = g.load()
    slices = [bytes((i * 11 + k) % 253 for i in range(n)) for k, n in enumerate(sizes)]
    total = sum(sizes)
    a, b = mk_link(g, 4 << 20, 30)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    a.arm_read(64)          # (armed before the engine is up: the order reaches its slot with the first command)
    b.arm_read(64)
    g._lib.check(lib.grdma_engine_start())
    o = pyorc.OracleLink(4 << 20, 30)
    oracle_pingpong(o, [(slices, slices, 25)])

    def expect(m):      # what the oracle's reader gets for message m: the reads of one drain
        assert o.send(0, m) == sum(len(x) for x in m)
        out = []
        while True:
            got, _alloc = o.endpoint_read(1)
            if not got:
                return out
            out.append(got)
    try:
        rtt, _ph = g.pingpong(a, b, slices, slices, iters=20, warmup=5)
        assert len(rtt) == 20 and min(rtt) > 0
        assert a.watch_hits() == 25 and b.watch_hits() == 25
        # a completion is handed out once, in order, with its bytes
        msg = [b"hello, ", b"watched read"]
        a.endpoint_write(msg)
        assert read_when_ready(b) == (expect(msg), True)
        assert b.endpoint_read(64) == ([], True)     # nothing there: the read stays outstanding, no device work
        # a second message lands while the first completion waits: its drain does not run before that one is taken
        a.endpoint_write([b"first"])
        t0 = time.time()
        while not b.armed_ready():
            assert time.time() - t0 < 20
        a.endpoint_write([b"second, longer"])
        time.sleep(0.05)
        assert b.endpoint_read(64) == (expect([b"first"]), True)
        assert read_when_ready(b) == (expect([b"second, longer"]), True)
        a.arm_read(0)
        b.arm_read(0)
        hits = b.watch_hits()
        a.endpoint_write(msg)
        assert b.endpoint_read(64)[0] == expect(msg)   # an ordinary drain command again
        assert b.watch_hits() == hits
    finally:
        lib.grdma_engine_stop()
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)
    a.close(); b.close(); o.close()


@pytest.mark.parametrize("fast", ["1", "0"], ids=["single_wave_drain", "plan_body_only"])
@pytest.mark.parametrize("flags,ring", [(0, 1 << 20), (2, 1 << 20), (4, 1 << 16), (0, 1 << 12)],
                         ids=["staged_r1m", "direct_r1m", "finegrained_r64k", "staged_r4k_wraps"])
def test_watched_reads_on_a_random_sequence(gpu, flags, ring, fast, monkeypatch):
    """One to six slices of 1 .. 5000 bytes each way (express drains, the general tiers, records that do not fit the
    open read, rings that wrap and return credit many times): after the sequence every field of both connections'
    state, both record-size histories and both rings equal the oracle's."""
    g = gpu
    lib = g.load()
    # (the watcher's own single-wave drain of unary-sized messages, rxw_fast, or every drain through the plan body:
    #  both against the oracle -- read when the engine is launched)
    monkeypatch.setenv("GRDMA_WATCH_FAST", fast)
    lib.grdma_watch_fast_drains.restype = C.c_uint64
    fd0 = int(lib.grdma_watch_fast_drains())
    rng = random.Random(20260923 + flags + ring)
    big = [1, 5, 9, 14, 66, 100, 200, 256, 257, 600, 1500, 5000]
    if ring <= 4096:
        big = [1, 5, 9, 14, 66, 100, 200, 256]      # (a message stays below ring / 2: one Send takes it whole)
    rounds = []
    for _ in range(16):
        mk = lambda: [bytes(rng.getrandbits(8) for _ in range(rng.choice(big))) for _ in range(rng.randint(1, 3 if ring <= 4096 else 6))]
        rounds.append((mk(), mk(), rng.randint(1, 5)))
    a, b = mk_link(g, ring, 30, flags)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    g._lib.check(lib.grdma_engine_start())
    try:
        a.arm_read(64)      # (armed while the engine is resident)
        b.arm_read(64)
        for sa_, sb_, iters in rounds:
            g.pingpong(a, b, sa_, sb_, iters=iters, warmup=0)
        n = sum(r[2] for r in rounds)
        assert a.watch_hits() >= n and b.watch_hits() >= n
    finally:
        lib.grdma_engine_stop()
    fd = int(lib.grdma_watch_fast_drains()) - fd0
    assert (fd > 0) if fast == "1" else (fd == 0), fd
    hist = []
    for p in (a, b):
        h = (C.c_uint32 * 4096)()
        cnt, per = C.c_uint64(), C.c_uint32()
        lib.grdma_pair_debug_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        assert lib.grdma_pair_debug_hist(p.h, h, C.byref(cnt), C.byref(per)) == 0
        hist.append(int(cnt.value))
    o = pyorc.OracleLink(ring, 30)
    open_read = oracle_pingpong(o, rounds)
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert sa["leftover_cap"] == open_read[0] and sb["leftover_cap"] == open_read[1]
    assert a.ring_mem() == o.ring_mem(0) == bytes(ring) and b.ring_mem() == o.ring_mem(1) == bytes(ring)
    nrec = sum((len(r[0]) + len(r[1])) * r[2] for r in rounds)
    assert hist[0] + hist[1] == nrec, (hist, nrec)
    a.close(); b.close(); o.close()


def test_watcher_on_an_ordered_wire_finds_the_record_by_its_tags(gpu, monkeypatch):
    """GRDMA_WIRE_ORDERED (what a NIC is: bytes in address order, footer last, no arrival report): the watcher polls the
    header at the head and the footer it points at (GetReadableSize, ring_buffer.cc:67-97).  Single records of at most
    256 bytes -- the sender's unary branch stores a record's footer behind everything else of it."""
    g = gpu
    lib = g.load()
    a, b = mk_link(g, 1 << 16, 30, 8)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    a.arm_read(64)
    b.arm_read(64)
    g._lib.check(lib.grdma_engine_start())
    rng = random.Random(7)
    rounds = [([bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 8, 64, 200, 256])))],
               [bytes(rng.getrandbits(8) for _ in range(rng.choice([3, 64, 129])))], rng.randint(1, 4)) for _ in range(10)]
    try:
        for sa_, sb_, iters in rounds:
            g.pingpong(a, b, sa_, sb_, iters=iters, warmup=0)
        assert a.watch_hits() == sum(r[2] for r in rounds)
    finally:
        lib.grdma_engine_stop()
    o = pyorc.OracleLink(1 << 16, 30)
    oracle_pingpong(o, rounds)
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)
    a.close(); b.close(); o.close()


def test_the_engine_comes_back_with_its_standing_orders(gpu, monkeypatch):
    """Stop and start between messages: the orders are posted afresh; a completion nobody took before the stop is
    still handed out after it."""
    g = gpu
    lib = g.load()
    a, b = mk_link(g, 1 << 18, 30)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    a.arm_read(64)
    b.arm_read(64)
    o = pyorc.OracleLink(1 << 18, 30)
    msgs = [[b"one"], [b"two", b"2"], [b"x" * 700], [b"four"]]

    def expect(m):
        assert o.send(0, m) == sum(len(x) for x in m)
        out = []
        while True:
            got, _alloc = o.endpoint_read(1)
            if not got:
                return out
            out.append(got)
    try:
        g._lib.check(lib.grdma_engine_start())
        a.endpoint_write(msgs[0])
        assert read_when_ready(b) == (expect(msgs[0]), True)
        a.endpoint_write(msgs[1])
        t0 = time.time()
        while not b.armed_ready():
            assert time.time() - t0 < 20
        lib.grdma_engine_stop()                       # the completion of "two2" waits in the result block
        assert b.endpoint_read(64) == (expect(msgs[1]), True)
        a.endpoint_write(msgs[2])                     # engine stopped: launches of their own
        assert b.endpoint_read(64)[0] == expect(msgs[2])
        g._lib.check(lib.grdma_engine_start())
        a.endpoint_write(msgs[3])
        assert read_when_ready(b) == (expect(msgs[3]), True)
    finally:
        lib.grdma_engine_stop()
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)
    a.close(); b.close(); o.close()
