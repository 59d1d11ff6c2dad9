"""GPU: the armed read (grdma_pair_arm_read) -- a small send carries the local peer's drain in ONE command of the
resident latency engine (GRDMA_ENGINE_SEND_INLINE_DRAIN).  Checked so far against the emulated library only
(tests/test_emu_gpu_suite.py), so the file sorts last of all and carries a hard timeout: if the resident kernel
ever wedged on hardware, the run ends here instead of hanging in grdma_engine_stop."""
import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import STATE_KEYS, mk_link

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180, method="thread")]


@pytest.mark.parametrize("sizes", [[14, 66], [14, 600], [9, 2000, 5]], ids=["unary64", "two_records", "beyond_the_fast_lane"])
def test_armed_reads_ride_in_the_send_command(gpu, sizes):
    """grdma_pair_arm_read: the drain of the local peer runs behind the send in ONE engine command
    (GRDMA_ENGINE_SEND_INLINE_DRAIN); delivered bytes, state and rings are those of the separate commands."""
    g = gpu
    lib = g.load()
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
