"""GPU: unary ping-pong THROUGH the resident latency engine (k_engine: commands in the pinned mailbox, small ones in
its fast lane, the plan bodies called out of line) -- the path bench.py's RTT leg times.  After the round trips
the state and both rings equal the oracle's.  (So far checked against the emulated library, where the engine
runs in a thread of its own: tests/test_emu_gpu_suite.py; the file sorts last.)"""
import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import STATE_KEYS, mk_link

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sizes", [[14, 66], [14, 600], [9, 2000, 5]], ids=["unary64", "two_records", "beyond_the_fast_lane"])
def test_pingpong_through_the_latency_engine(gpu, sizes):
    g = gpu
    lib = g.load()
    slices = [bytes((i * 7 + k) % 251 for i in range(n)) for k, n in enumerate(sizes)]
    total = sum(sizes)
    a, b = mk_link(g, 4 << 20, 30)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    g._lib.check(lib.grdma_engine_start())
    try:
        rtt, ph = g.pingpong(a, b, slices, slices, iters=50, warmup=10)
    finally:
        lib.grdma_engine_stop()
    assert len(rtt) == 50 and min(rtt) > 0
    o = pyorc.OracleLink(4 << 20, 30)
    for _ in range(60):
        for src, dst in ((0, 1), (1, 0)):
            assert o.send(src, slices) == total
            while True:
                s_, _al = o.endpoint_read(dst)
                if not s_:
                    break
    sa, sb = a.state(), b.state()
    for k in STATE_KEYS:
        assert sa[k] == o.state(0)[k] and sb[k] == o.state(1)[k], k
    assert a.ring_mem() == o.ring_mem(0) and b.ring_mem() == o.ring_mem(1)
    a.close(); b.close(); o.close()
