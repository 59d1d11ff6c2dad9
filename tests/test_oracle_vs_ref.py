"""CPU: the C restatement (oracle/grdma_oracle.c) against the REFERENCE's own ring
codec (oracle/_ref/libref_ring.so, built from /root/reference/src/core/lib/ibverbs/
ring_buffer.cc).  This is what pins the oracle; the reference tree itself ships no
ring/pair unit tests (test/core/ibverbs/ is absent)."""
import random

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import pyorc

pytestmark = pytest.mark.skipif(not pyorc.ref_available(),
                                reason="oracle/_ref/libref_ring.so not built (no /root/reference)")

SIZES = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 100, 255, 256, 257]


def same(a, b, step):
    assert a.ring_mem(1) == b.ring_mem(1), step
    assert a.state(0) == b.state(0), (step, a.state(0), b.state(0))
    assert a.state(1) == b.state(1), (step, a.state(1), b.state(1))
    assert a.readable(1) == b.readable(1) and a.has_message(1) == b.has_message(1)
    assert a.writable(0) == b.writable(0)


def test_statics_match_reference():
    r, o = pyorc.ref(), pyorc.lib()
    for v in list(range(0, 200)) + [4095, 4096, 1 << 20, (1 << 22) - 1]:
        assert o.orc_calc_writable(v) == r.ref_calc_writable(v)
        if v:
            assert o.orc_encoded_size(v) == r.ref_encoded_size(v)
    assert r.ref_reserved_space() == 24
    assert r.ref_sizeof_grpc_slice() == 32          # include/grpc/impl/codegen/slice.h:60-75
    assert r.ref_sizeof_grpc_slice_buffer() == 296  # :82-94
    assert r.ref_slice_inlined_size() == 23


@pytest.mark.parametrize("seed", range(40))
def test_random_sequences(seed):
    rng = random.Random(seed)
    R = rng.choice([64, 128, 256, 1024, 4096, 65536])
    sge = rng.choice([1, 2, 3, 30, 100])
    a, b = pyorc.OracleLink(R, sge), pyorc.RefLink(R, sge)
    for step in range(80):
        op = rng.random()
        if op < 0.5:
            sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(SIZES + [R // 3, R])))
                  for _ in range(rng.randint(1, 8))]
            bi = rng.randrange(len(sl[0])) if rng.random() < 0.3 else 0
            assert a.send(0, sl, bi) == b.send(0, sl, bi)
            assert a.staging_mem(0) == b.staging_mem(0)
            assert a.last_wrs(0) == b.last_wrs(0)
        elif op < 0.8:
            cap = rng.choice([1, 3, 8, 64, 256, R])
            assert a.recv(1, cap) == b.recv(1, cap)
        else:
            assert a.endpoint_read(1) == b.endpoint_read(1)
        same(a, b, (seed, step))
    a.close(); b.close()


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(st.data())
def test_hypothesis_sends_and_reads(data):
    R = data.draw(st.sampled_from([64, 256, 4096]))
    sge = data.draw(st.sampled_from([1, 4, 30]))
    a, b = pyorc.OracleLink(R, sge), pyorc.RefLink(R, sge)
    for step in range(data.draw(st.integers(1, 25))):
        kind = data.draw(st.sampled_from(["send", "recv", "epread"]))
        if kind == "send":
            sl = data.draw(st.lists(st.binary(min_size=1, max_size=R), min_size=1, max_size=6))
            bi = data.draw(st.integers(0, len(sl[0]) - 1))
            assert a.send(0, sl, bi) == b.send(0, sl, bi)
            assert a.last_wrs(0) == b.last_wrs(0)
        elif kind == "recv":
            cap = data.draw(st.integers(1, R))
            assert a.recv(1, cap) == b.recv(1, cap)
        else:
            assert a.endpoint_read(1) == b.endpoint_read(1)
        same(a, b, step)
    a.close(); b.close()


def test_inlined_slices_are_read_through_the_accessor_macros():
    """Slices <= 23 bytes are inlined in the grpc_slice (slice.h:47-48,67-70); the
    reference reads them with GRPC_SLICE_START_PTR -- same record bytes."""
    b1, b2 = pyorc.RefLink(4096, 30), pyorc.RefLink(4096, 30)
    sl = [b"123456789", b"x" * 14, b"y" * 23, b"z" * 24, b"w" * 300]
    assert b1.send(0, sl, 0, inline_small=0) == b2.send(0, sl, 0, inline_small=1)
    assert b1.ring_mem(1) == b2.ring_mem(1)
    b1.close(); b2.close()


@pytest.mark.parametrize("ring", [1 << 14, 1 << 18, 4 << 20])
@pytest.mark.parametrize("msg_len", [1, 300, 70000, 1 << 20])
def test_stream_baseline_port_equals_reference_codec(ring, msg_len):
    """The streaming loop bench.py times as the CPU baseline: the plain-C port and the same
    loop over the reference-built ring codec deliver the same bytes (count and the
    first/last-byte checksum of every endpoint read)."""
    if not pyorc.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    msg = bytes((i * 37 + 11) & 0xFF for i in range(msg_len))
    wire, lens = pyorc.h2_frame_message(msg, 1)
    n_port, _s, chk_port = pyorc.stream_baseline(ring, 30, wire, lens, 5, with_checksum=True)
    n_ref, _s2, chk_ref = pyorc.ref_stream_baseline(ring, 30, wire, lens, 5)
    assert n_port == n_ref == 5 * len(wire)
    assert chk_port == chk_ref
