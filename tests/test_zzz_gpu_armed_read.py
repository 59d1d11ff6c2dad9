"""GPU: the armed read (grdma_pair_arm_read) -- a small send carries the local peer's drain in ONE command of the
resident latency engine (GRDMA_ENGINE_SEND_INLINE_DRAIN).  Checked so far against the emulated library only
(tests/test_emu_gpu_suite.py), so the file sorts last of all and carries a hard timeout: if the resident kernel
ever wedged on hardware, the run ends here instead of hanging in grdma_engine_stop."""
import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import STATE_KEYS, mk_link

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180, method="thread")]


@pytest.mark.parametrize("sizes", [[14, 66], [14, 600], [9, 2000, 5]], ids=["unary64", "two_records", "beyond_the_fast_lane"])
def test_armed_reads_ride_in_the_send_command(gpu, sizes, monkeypatch):
    """grdma_pair_arm_read under GRDMA_ENGINE_CHAIN=1 (round 4's way; by default a watcher workgroup carries the order
    out, tests/test_zzz_gpu_watch_read.py): the drain of the local peer runs behind the send in ONE engine command
    (GRDMA_ENGINE_SEND_INLINE_DRAIN); delivered bytes, state and rings are those of the separate commands."""
    g = gpu
    lib = g.load()
    monkeypatch.setenv("GRDMA_ENGINE_CHAIN", "1")
    slices = [bytes((i * 11 + k) % 253 for i in range(n)) for k, n in enumerate(sizes)]
    total = sum(sizes)
    a, b = mk_link(g, 4 << 20, 30)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    a.arm_read(64)
    b.arm_read(64)
    g._lib.check(lib.grdma_engine_start())
    import ctypes as C
    lib.grdma_cut_through_drains.restype = C.c_uint64
    ct0 = int(lib.grdma_cut_through_drains())
    try:
        g.pingpong(a, b, slices, slices, iters=20, warmup=5)
        inline = total <= 1024
        assert a.armed_hits() == (25 if inline else 0) and b.armed_hits() == (25 if inline else 0)
        # (unary-sized commands are cut through: the records of an armed send + drain never touch the ring -- three
        #  commands in four here: the read a drain leaves open takes 80 bytes three times, then 16 bytes are left of its
        #  256 and the general tiers split the next message; a record of 600 bytes takes the ring, its sizes still
        #  reach the drain through LDS)
        ct = int(lib.grdma_cut_through_drains()) - ct0
        assert (ct >= 30) if sizes == [14, 66] else (ct == 0), ct
        # an armed completion is handed out once, in order, with its bytes
        msg = [b"hello, ", b"armed read"]
        a.endpoint_write(msg)
        assert b.endpoint_read(64) == ([b"".join(msg)], True)   # one Recv takes both records
        assert b.endpoint_read(64) == ([], True)     # nothing left: an ordinary drain that would block
        a.arm_read(0)
        b.arm_read(0)
        hits = b.armed_hits()
        a.endpoint_write(msg)
        assert b.armed_hits() == hits and b.endpoint_read(64)[0] == [b"".join(msg)]
    finally:
        lib.grdma_engine_stop()
    o = pyorc.OracleLink(4 << 20, 30)
    for _ in range(25):
        for src, dst in ((0, 1), (1, 0)):
            assert o.send(src, slices) == total
            while o.endpoint_read(dst)[0]:
                pass
    for _ in range(2):
        assert o.send(0, msg) == 17
        while o.endpoint_read(1)[0]:
            pass
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)
    a.close(); b.close(); o.close()


def test_cut_through_commands_leave_the_state_the_two_bodies_leave(gpu, monkeypatch):
    """The armed send + drain command with its round-4 hand-over (sizes through LDS, unary-sized records cut through, one
    release) against the same command running its two bodies the way separate commands do (GRDMA_ENGINE_CUT_THROUGH=0):
    after the same sequence of ping-pongs -- one to four slices of 1 .. 256 bytes, now and then a record that does not
    fit the open read, one beyond 256 bytes -- every field of both connections' state, both record-size histories and
    both rings are equal, and equal to the oracle's."""
    import ctypes as C
    import random
    g = gpu
    lib = g.load()
    lib.grdma_cut_through_drains.restype = C.c_uint64
    rng = random.Random(20260922)
    rounds = []
    for _ in range(14):
        mk = lambda: [bytes(rng.getrandbits(8) for _ in range(rng.choice([1, 5, 9, 14, 66, 100, 200, 256])))
                      for _ in range(rng.randint(1, 4))]
        rounds.append((mk(), mk(), rng.randint(1, 6)))
    rounds.insert(5, ([b"x" * 14, b"y" * 300], [b"z" * 9], 2))      # a record the unary branch does not take
    outcomes = []
    monkeypatch.setenv("GRDMA_ENGINE_CHAIN", "1")
    for mode in ("1", "0"):
        monkeypatch.setenv("GRDMA_ENGINE_CUT_THROUGH", mode)
        a, b = mk_link(g, 1 << 20, 30)
        a.set_latency_mode(True)
        b.set_latency_mode(True)
        a.arm_read(64)
        b.arm_read(64)
        g._lib.check(lib.grdma_engine_start())
        ct0 = int(lib.grdma_cut_through_drains())
        try:
            for sa, sb, iters in rounds:
                g.pingpong(a, b, sa, sb, iters=iters, warmup=0)
        finally:
            lib.grdma_engine_stop()
        ct = int(lib.grdma_cut_through_drains()) - ct0
        hist = []
        for p in (a, b):
            h = (C.c_uint32 * 4096)()
            cnt, per = C.c_uint64(), C.c_uint32()
            lib.grdma_pair_debug_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
            assert lib.grdma_pair_debug_hist(p.h, h, C.byref(cnt), C.byref(per)) == 0
            n = min(int(cnt.value), 64)
            hist.append((int(cnt.value), int(per.value), [int(h[(int(cnt.value) - n + i) % 1024]) for i in range(n)]))
        outcomes.append((ct, a.state(), b.state(), hist, a.ring_mem() == bytes(1 << 20), b.ring_mem() == bytes(1 << 20),
                         a.armed_hits(), b.armed_hits()))
        a.close(); b.close()
    on, off = outcomes
    assert on[0] > 0 and off[0] == 0, (on[0], off[0])
    assert on[1] == off[1] and on[2] == off[2], "connection state differs between the two ways of running the command"
    assert on[3] == off[3], "record-size history differs"
    assert on[4:] == off[4:] and on[4] and on[5]
    o = pyorc.OracleLink(1 << 20, 30)
    open_read = {0: None, 1: None}   # the read the endpoint keeps open behind a would-block: its size (leftover_cap)
    for sa, sb, iters in rounds:
        for _ in range(iters):
            for src, dst, sl in ((0, 1, sa), (1, 0, sb)):
                assert o.send(src, sl) == sum(len(x) for x in sl)
                while True:
                    got, alloc = o.endpoint_read(dst)
                    if not got:
                        open_read[dst] = alloc
                        break
    for k in STATE_KEYS:
        assert on[1][k] == o.state(0)[k] and on[2][k] == o.state(1)[k], k
    assert on[1]["leftover_cap"] == open_read[0] and on[2]["leftover_cap"] == open_read[1]
    o.close()
