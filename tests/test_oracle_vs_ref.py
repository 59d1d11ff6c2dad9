"""CPU: the C restatement (oracle/grdma_oracle.c) against REFERENCE CODE built here from /root/reference, unmodified
(oracle/Makefile).  This is what pins the oracle; the reference tree itself ships no ring / pair unit tests
(test/core/ibverbs/ is absent).  Five builds, in the order of the tests below:
  * oracle/_ref/libref_ring.so          ring_buffer.cc (+ a driver that transcribes pair.cc's call order around it)
  * oracle/_ref/ref_pair_trace          pair.cc itself -- PairPollable::Send / Recv / credit / SendZerocopy -- with
                                        ring_buffer.cc, device.cc, memory_region.cc, buffer.cc, address.cc, config.cc over
                                        the software verbs of oracle/fakeverbs
  * oracle/_ref/ref_h2_trace            frame_data.cc's grpc_chttp2_encode_data over slice.cc / slice_buffer.cc
  * oracle/_ref/ref_h2_deframe_trace    frame_data.cc's grpc_deframe_unprocessed_incoming_frames over the same slice layer
  * oracle/_ref/ref_h2_perform_read_trace  parsing.cc's grpc_chttp2_perform_read (frame headers, init_frame_parser, the stream-map
                                        rules) over stream_map.cc, frame_data.cc, frame_rst_stream.cc and the slice layer
  * oracle/_ref/ref_endpoint_trace      rdma_bp_posix.cc itself -- grpc_rdma_bp_create, the endpoint's read and write
                                        paths -- over pair.cc and the slice layer
Every test feeds the same seeded operations to the oracle and to the reference build and compares step by step."""
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


@pytest.mark.parametrize("seed", range(40))
def test_zerocopy_send_matches_reference_driver(seed):
    """PairPollable::AllocateSendBuffer / SendZerocopy (pair.cc:305-323, 793-941): slices inside the
    zero-copy buffer travel as header / payload / padding / footer scatter-gather entries, other slices
    are staged, Send and SendZerocopy interleave.  The oracle's restatement against the driver's
    transcription over the reference-built ring codec: accepted bytes, ring image (padding bytes
    included: both read them from identically evolving staging buffers), staged bytes, work requests,
    scatter-gather entry count, buffer tail, counters, state."""
    rng = random.Random(1000 + seed)
    R = rng.choice([64, 128, 256, 1024, 4096, 65536])
    sge = rng.choice([1, 3, 4, 5, 8, 30, 100])
    Z = rng.choice([64, 4096, 2 * R])
    a, b = pyorc.OracleLink(R, sge), pyorc.RefLink(R, sge)
    a.enable_zerocopy(0, Z)
    b.enable_zerocopy(0, Z)
    for step in range(60):
        op = rng.random()
        if op < 0.45:
            sl = []
            for _ in range(rng.randint(1, 6)):
                n = rng.choice(SIZES + [R // 3, R, Z // 2])
                if rng.random() < 0.5:
                    n = min(n, Z)
                    off_a, off_b = a.allocate_send_buffer(0, n), b.allocate_send_buffer(0, n)
                    assert off_a == off_b
                    if off_a is None:  # the buffer is not empty: AllocateSendBuffer refuses
                        # an arbitrary range of the buffer still counts as "inside" for SendZerocopy
                        if rng.random() < 0.5:
                            off_a = rng.randrange(0, Z - n + 1)
                        else:
                            sl.append(bytes(rng.getrandbits(8) for _ in range(n)))
                            continue
                    data = bytes(rng.getrandbits(8) for _ in range(n))
                    a.zerocopy_write(0, off_a, data)
                    b.zerocopy_write(0, off_a, data)
                    sl.append(("zc", off_a, n))
                else:
                    sl.append(bytes(rng.getrandbits(8) for _ in range(n)))
            first = sl[0][2] if isinstance(sl[0], tuple) else len(sl[0])
            bi = rng.randrange(first) if rng.random() < 0.3 else 0
            assert a.send_zerocopy(0, sl, bi) == b.send_zerocopy(0, sl, bi)
            assert a.staging_mem(0) == b.staging_mem(0)
            assert a.last_wrs(0) == b.last_wrs(0)
            assert a.zerocopy_state(0) == b.zerocopy_state(0)
        elif op < 0.55:
            sl = [bytes(rng.getrandbits(8) for _ in range(rng.choice(SIZES + [R // 3]))) for _ in range(rng.randint(1, 4))]
            assert a.send(0, sl) == b.send(0, sl)
        elif op < 0.8:
            cap = rng.choice([1, 8, 64, 256, R])
            assert a.recv(1, cap) == b.recv(1, cap)
        else:
            assert a.endpoint_read(1) == b.endpoint_read(1)
        same(a, b, (seed, step))
    a.close(); b.close()


def test_zerocopy_rules():
    """The rules spelled out: the allocator only serves an empty buffer; a zero-copy record is limited
    by the receiver's credit and not by the staging buffer; it needs four free entries."""
    R = 4096
    for L in (pyorc.OracleLink(R, 30), pyorc.RefLink(R, 30)):
        L.enable_zerocopy(0, 8192)
        assert L.allocate_send_buffer(0, 0) is None
        assert L.allocate_send_buffer(0, 8193) is None
        off = L.allocate_send_buffer(0, 3000)
        assert off == 0 and L.allocate_send_buffer(0, 16) is None      # tail != 0
        L.zerocopy_write(0, 0, bytes(range(256)) * 12)
        # 3000 bytes > W(staging = 2048) = 2024: Send would cut the record, SendZerocopy does not
        assert L.send_zerocopy(0, [("zc", 0, 3000)]) == 3000
        st = L.zerocopy_state(0)
        assert st["tail"] == 0 and st["zerocopy_bytes"] == 3000 and st["sges"] == 3   # 3000 % 8 == 0: no padding entry
        assert L.allocate_send_buffer(0, 16) == 0                      # empty again
        got = L.recv(1, 4096)
        assert got == (bytes(range(256)) * 12)[:3000]
        L.close()
    for L in (pyorc.OracleLink(R, 3), pyorc.RefLink(R, 3)):           # three entries: never enough
        L.enable_zerocopy(0, 64)
        assert L.allocate_send_buffer(0, 10) == 0
        assert L.send_zerocopy(0, [("zc", 0, 10)]) == 0
        assert L.zerocopy_state(0)["tail"] == 10
        L.close()


# ---- the oracle against the reference's pair.cc ITSELF -------------------------------------------------------------
# oracle/_ref/ref_pair_trace = the reference's unmodified pair.cc + ring_buffer.cc + device.cc + memory_region.cc +
# buffer.cc + address.cc + config.cc over the software verbs of oracle/fakeverbs (oracle/Makefile): two PairPollable
# objects connected to each other replay an operation list.  The same list goes through the plain-C oracle; every
# return value, the credit-dependent writable size after every step, partial_write_, the bytes received, and at the end
# remote_tail_, internal_read_size_, the head the sender has been told and the whole ring image must be the same.
def _pat(seed, i, n):
    import numpy as np
    j = np.arange(n, dtype=np.uint64)
    return ((seed * 131 + i * 17 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8).tobytes()


def _fnv(b):
    import zlib
    return zlib.crc32(b) & 0xFFFFFFFF


def _random_ops(rng, ring, max_sge, n_ops):
    ops = []
    for _ in range(n_ops):
        side = rng.randrange(2)
        r = rng.random()
        if r < 0.5:
            n = rng.choice([1, 1, 2, 3, max_sge - 1, max_sge, max_sge + 3, rng.randrange(1, 2 * max_sge)])
            kind = rng.random()
            if kind < 0.3:      # frame-like: small header slices between payload slices
                lens = [rng.choice([9, 5, 14, rng.randrange(1, 300)]) if k % 2 == 0 else rng.randrange(1, ring // 6)
                        for k in range(n)]
            elif kind < 0.6:    # bulk
                lens = [rng.randrange(1, ring // 3) for _ in range(n)]
            else:
                lens = [rng.randrange(1, 2000) for _ in range(n)]
            byte_idx = rng.randrange(lens[0]) if rng.random() < 0.3 else 0
            ops.append(("S", side, byte_idx, rng.randrange(1 << 16), lens))
        else:
            cap = rng.choice([1, 7, 255, 256, 257, 4096, rng.randrange(1, ring), 2 * ring])
            ops.append(("R", side, cap))
    # drain both sides completely at the end (the ring image then shows what the reader zeroed)
    for side in (0, 1):
        for _ in range(8):
            ops.append(("R", side, 2 * ring))
    return ops


def _run_ref_pair_trace(ops, ring_kb, max_sge):
    import os
    import subprocess
    text = []
    for op in ops:
        if op[0] == "S":
            _, side, byte_idx, seed, lens = op
            text.append("S %d %d %d %d %s" % (side, byte_idx, seed, len(lens), " ".join(map(str, lens))))
        else:
            text.append("R %d %d" % (op[1], op[2]))
    text.append("Q")
    env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge))
    p = subprocess.run([pyorc.REF_PAIR_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True,
                       timeout=120, env=env)
    assert p.returncode == 0, p.stderr[-400:]
    return [ln.split() for ln in p.stdout.strip().splitlines()]


@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (256, 30), (64, 4), (1024, 64), (4096, 30)])
@pytest.mark.parametrize("seed", range(10))
def test_oracle_equals_the_reference_pairpollable_itself(seed, ring_kb, max_sge):
    import os
    if not os.path.exists(pyorc.REF_PAIR_TRACE):
        pytest.skip("oracle/_ref/ref_pair_trace not built (no reference tree here)")
    ring = ring_kb * 1024
    rng = random.Random(1000 * seed + ring_kb + max_sge)
    ops = _random_ops(rng, ring, max_sge, 160 if ring_kb <= 256 else 60)
    ref = _run_ref_pair_trace(ops, ring_kb, max_sge)
    o = pyorc.OracleLink(ring, max_sge)
    try:
        for k, (op, line) in enumerate(zip(ops, ref)):
            if op[0] == "S":
                _, side, byte_idx, sd, lens = op
                sent = o.send(side, [_pat(sd, i, n) for i, n in enumerate(lens)], byte_idx)
                got = ("S", sent, o.writable(side), int(o.p[side].partial_write))
                want = ("S", int(line[1]), int(line[2]), int(line[3]))
            else:
                _, side, cap = op
                b = o.recv(side, cap)
                got = ("R", len(b), _fnv(b), o.readable(side), int(o.has_message(side)), o.writable(1 - side))
                want = ("R", int(line[1]), int(line[2]), int(line[3]), int(line[4]), int(line[5]))
            assert got == want, "op %d %r: oracle %r, pair.cc %r" % (k, op[:3], got, want)
        q = ref[len(ops):]
        assert len(q) == 2
        for side in (0, 1):
            st = o.state(side)
            want = [int(x) for x in q[side][2:]]
            got = [st["remote_tail"], st["internal_read_size"], st["remote_head"], st["moving_head"], _fnv(o.ring_mem(side))]
            assert got == want, "side %d {remote_tail, internal_read_size, remote_head, get_head() = moving_head_, ring image}: %r vs %r" % (
                side, got, want)
    finally:
        o.close()


@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (64, 5), (256, 8), (1024, 30)])
@pytest.mark.parametrize("seed", range(10))
def test_oracle_zerocopy_equals_the_reference_pairpollable_itself(seed, ring_kb, max_sge):
    """PairPollable::AllocateSendBuffer / SendZerocopy of the reference's own pair.cc (pair.cc:305-323, 793-941; its
    zero-copy buffer is as large as the ring: config.cc reads the ring-size variable for it) against the oracle:
    allocator answers, accepted bytes, writable size, partial_write_, the buffer tail after every Send, received bytes,
    final state and ring image.  Plain Sends and Recvs interleave."""
    import os
    if not os.path.exists(pyorc.REF_PAIR_TRACE):
        pytest.skip("oracle/_ref/ref_pair_trace not built (no reference tree here)")
    ring = ring_kb * 1024
    Z = ring
    rng = random.Random(7000 + 100 * seed + ring_kb + max_sge)
    o = pyorc.OracleLink(ring, max_sge)
    o.enable_zerocopy(0, Z)
    o.enable_zerocopy(1, Z)
    text, checks = [], []  # the op list for the reference; what the oracle said, in the same order
    try:
        for _ in range(120):
            side = rng.randrange(2)
            r = rng.random()
            if r < 0.45:
                sl, spec = [], []
                sd = rng.randrange(1 << 16)
                for i in range(rng.randint(1, 6)):
                    n = rng.choice([1, 9, 100, 255, 256, 257, 4000, ring // 7, ring // 3])
                    if rng.random() < 0.55:
                        off = o.allocate_send_buffer(side, n)
                        text.append("A %d %d" % (side, n))
                        checks.append(("A", -1 if off is None else off))
                        if off is None:
                            if rng.random() < 0.5:   # any range of the buffer counts as "inside" for SendZerocopy
                                off = rng.randrange(0, Z - n + 1)
                            else:
                                sl.append(_pat(sd, i, n))
                                spec.append("%d -1" % n)
                                continue
                        wseed = rng.randrange(1 << 16)
                        o.zerocopy_write(side, off, _pat(wseed, 0, n))
                        text.append("W %d %d %d %d" % (side, off, wseed, n))
                        sl.append(("zc", off, n))
                        spec.append("%d %d" % (n, off))
                    else:
                        sl.append(_pat(sd, i, n))
                        spec.append("%d -1" % n)
                first = sl[0][2] if isinstance(sl[0], tuple) else len(sl[0])
                bi = rng.randrange(first) if rng.random() < 0.3 else 0
                sent = o.send_zerocopy(side, sl, bi)
                text.append("Z %d %d %d %d %s" % (side, bi, sd, len(sl), " ".join(spec)))
                checks.append(("Z", sent, o.writable(side), int(o.p[side].partial_write), o.zerocopy_state(side)["tail"]))
            elif r < 0.6:
                sd = rng.randrange(1 << 16)
                lens = [rng.choice([5, 9, 300, 5000, ring // 5]) for _ in range(rng.randint(1, 4))]
                sent = o.send(side, [_pat(sd, i, n) for i, n in enumerate(lens)], 0)
                text.append("S %d 0 %d %d %s" % (side, sd, len(lens), " ".join(map(str, lens))))
                checks.append(("S", sent, o.writable(side), int(o.p[side].partial_write)))
            else:
                cap = rng.choice([1, 8, 256, 4096, 2 * ring])
                b = o.recv(side, cap)
                text.append("R %d %d" % (side, cap))
                checks.append(("R", len(b), _fnv(b), o.readable(side), int(o.has_message(side)), o.writable(1 - side)))
        text.append("Q")
        import subprocess
        env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge))
        p = subprocess.run([pyorc.REF_PAIR_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True,
                           timeout=120, env=env)
        assert p.returncode == 0, p.stderr[-400:]
        lines = [ln.split() for ln in p.stdout.strip().splitlines()]
        assert len(lines) == len(checks) + 2
        for k, (want, line) in enumerate(zip(checks, lines)):
            got = tuple([line[0]] + [int(x) for x in line[1:len(want)]])
            assert got == want, "step %d: pair.cc %r, oracle %r" % (k, got, want)
        for side in (0, 1):
            st = o.state(side)
            ref = [int(x) for x in lines[len(checks) + side][2:]]
            mine = [st["remote_tail"], st["internal_read_size"], st["remote_head"], st["moving_head"], _fnv(o.ring_mem(side))]
            assert mine == ref, "side %d: oracle %r, pair.cc %r" % (side, mine, ref)
    finally:
        o.close()


# ---- the oracle's DATA framing against the reference's grpc_chttp2_encode_data ITSELF -----------------------------------
def _pat1(seed, n):
    import numpy as np
    j = np.arange(n, dtype=np.uint64)
    return ((seed * 131 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8).tobytes()


@pytest.mark.parametrize("seed", range(12))
def test_oracle_framing_equals_the_reference_encode_data_itself(seed):
    """oracle/_ref/ref_h2_trace = the reference's unmodified frame_data.cc (grpc_chttp2_encode_data), slice.cc and
    slice_buffer.cc (tiny_add, the inlined-slice merge rule of grpc_slice_buffer_add, the splits of
    grpc_slice_buffer_move_first_no_ref) driven through the two call sites of chttp2_transport.cc:1502-1510 and
    writing.cc:344-355.  Batches of messages queued on one outbuf -- empty, tiny, exactly a frame, many frames,
    compressed flag, END_STREAM, several max_frame_size values -- must give the oracle's slice list (every length, in
    order) and the oracle's bytes."""
    import os
    import subprocess
    if not os.path.exists(pyorc.REF_H2_TRACE):
        pytest.skip("oracle/_ref/ref_h2_trace not built (no reference tree here)")
    rng = random.Random(4242 + seed)
    text, want = [], []
    for _ in range(25):
        max_frame = rng.choice([16384, 16384, 16384, 32768, 20000, 1 << 20])
        msgs, sids, flags = [], [], []
        for _m in range(rng.randint(1, 6)):
            n = rng.choice([0, 1, 7, 18, 19, 23, 100, max_frame - 5, max_frame - 4, max_frame, 3 * max_frame + 11,
                            rng.randrange(0, 200000)])
            sd = rng.randrange(1 << 16)
            sid = 2 * rng.randrange(1, 1 << 20) + 1
            compressed, end_stream = int(rng.random() < 0.2), int(rng.random() < 0.3)
            msgs.append(_pat1(sd, n))
            sids.append(sid)
            flags.append(compressed | (end_stream << 1))
            text.append("M %d %d %d %d %d %d" % (sid, compressed, end_stream, max_frame, n, sd))
        text.append("F")
        wire, lens = pyorc.h2_frame_batch(msgs, sids, flags, max_frame=max_frame)
        want.append((len(lens), _fnv(wire), lens))
    p = subprocess.run([pyorc.REF_H2_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-400:]
    lines = [ln.split() for ln in p.stdout.strip().splitlines()]
    assert len(lines) == len(want)
    for k, (line, (n, crc, lens)) in enumerate(zip(lines, want)):
        got_lens = [int(x) for x in line[5:]]
        assert int(line[1]) == n and got_lens == lens, "batch %d: encode_data %r, oracle %r" % (k, got_lens, lens)
        assert int(line[2]) == crc, "batch %d: the bytes differ" % k
        # (grpc_transport_one_way_stats: 9 framing bytes per DATA frame, the rest is data)
        assert int(line[3]) + int(line[4]) == sum(lens)


# ---- the oracle's message deframer against the reference's grpc_deframe_unprocessed_incoming_frames ITSELF ----------
@pytest.mark.parametrize("seed", range(40))
def test_oracle_deframer_equals_the_reference_deframer_itself(seed):
    """oracle/_ref/ref_h2_deframe_trace = the reference's unmodified frame_data.cc deframer (the FH_0 .. FH_4 / FRAME state
    machine over a slice buffer) on its own slice layer.  A stream of framed gRPC messages -- empty, tiny, large,
    compressed flag -- is cut into DATA frames and the wire into arbitrary chunks; the oracle's parser is fed the chunks,
    the reference's deframer the payload pieces (what grpc_chttp2_data_parser_parse stores: the part of every chunk that
    is DATA payload).  Message begin (flags, length), every payload piece handed on, message end: the same sequence."""
    import os
    import struct
    import subprocess
    if not os.path.exists(pyorc.REF_H2_DEFRAME_TRACE):
        pytest.skip("oracle/_ref/ref_h2_deframe_trace not built (no reference tree here)")
    rng = random.Random(9100 + seed)
    msgs, flags = [], []
    for _ in range(rng.randint(3, 30)):
        n = rng.choice([0, 1, 4, 5, 6, 100, 1000, 16379, 16384, 40000, rng.randrange(0, 70000)])
        msgs.append(_pat1(rng.randrange(1 << 16), n))
        flags.append(int(rng.random() < 0.25))          # bit 0 = compressed; no END_STREAM: one stream carries them all
    max_frame = rng.choice([16384, 16384, 1000, 70000])
    wire, _lens = pyorc.h2_frame_batch(msgs, [1] * len(msgs), flags, max_frame=max_frame)
    # arbitrary chunks: some tiny (a header byte at a time), some large
    chunks, off = [], 0
    while off < len(wire):
        n = rng.choice([1, 1, 2, 3, 5, 9, 14, 100, 4096, 8192, 20000, rng.randrange(1, 50000)])
        chunks.append(wire[off:off + n])
        off += n
    parser = pyorc.H2Parser(expect_client_prefix=False, max_frame_size=max_frame)
    parser.open_stream(1)
    want, pieces = [], []
    for ch in chunks:
        rc, evs = parser.feed(ch)
        assert rc == 0
        for kind, a, b, c, d in evs:
            if kind == pyorc.EV_PAYLOAD:
                pieces.append(ch[a:a + b])
            elif kind == pyorc.EV_MSG_BEGIN:
                want.append("B %d %d" % (0x80000000 if a else 0, b))   # GRPC_WRITE_INTERNAL_COMPRESS
            elif kind == pyorc.EV_MSG_BYTES:
                want.append("Y %d" % b)
            elif kind == pyorc.EV_MSG_END:
                want.append("E")
    assert b"".join(pieces) == b"".join(bytes([f & 1]) + struct.pack(">I", len(m)) + m for m, f in zip(msgs, flags))
    data = struct.pack("<I", len(pieces)) + b"".join(struct.pack("<I", len(p)) + p for p in pieces)
    r = subprocess.run([pyorc.REF_H2_DEFRAME_TRACE], input=data, capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr[-300:]
    got = r.stdout.decode().strip().splitlines()
    assert got[-1].startswith("S ")
    framing, data_bytes = [int(x) for x in got[-1].split()[1:]]
    assert got[:-1] == want
    assert framing == 5 * len(msgs) and data_bytes == sum(len(m) for m in msgs)
    assert want.count("E") == len(msgs)


# ---- the oracle's endpoint-read loop against the reference's rdma_bp_posix.cc ITSELF -------------------------------------
@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (256, 30), (1024, 64), (64, 4)])
@pytest.mark.parametrize("seed", range(8))
def test_oracle_endpoint_read_equals_the_reference_endpoint_itself(seed, ring_kb, max_sge):
    """oracle/_ref/ref_endpoint_trace = the reference's unmodified rdma_bp_posix.cc -- two endpoints made by
    grpc_rdma_bp_create over a socketpair, reads through rdma_read / rdma_handle_read / rdma_continue_read / rdma_do_read --
    on its own pair.cc and slice layer.  Sends of framed-looking and bulk slice lists, endpoint reads in between (also when
    nothing has arrived: the read then keeps its 256-byte slice for the next edge).  Every read: would-block or the bytes
    delivered, the readable size left, the peer's writable size (the credit the reads have returned)."""
    import os
    import subprocess
    if not os.path.exists(pyorc.REF_ENDPOINT_TRACE):
        pytest.skip("oracle/_ref/ref_endpoint_trace not built (no reference tree here)")
    ring = ring_kb * 1024
    rng = random.Random(31000 + 100 * seed + ring_kb + max_sge)
    o = pyorc.OracleLink(ring, max_sge)
    text, want = [], []
    try:
        for _ in range(150):
            side = rng.randrange(2)
            if rng.random() < 0.4:
                n = rng.choice([1, 2, 3, max_sge, max_sge + 2, rng.randrange(1, 2 * max_sge)])
                if rng.random() < 0.5:
                    lens = [rng.choice([9, 5, 14, 100, 255, 256, 257]) if k % 2 == 0 else rng.randrange(1, ring // 5)
                            for k in range(n)]
                else:
                    lens = [rng.choice([1, 9, 200, 256, 300, 511, 512, 5000]) for _ in range(n)]
                sd = rng.randrange(1 << 16)
                bi = rng.randrange(lens[0]) if rng.random() < 0.25 else 0
                sent = o.send(side, [_pat(sd, i, m) for i, m in enumerate(lens)], bi)
                text.append("S %d %d %d %d %s" % (side, bi, sd, len(lens), " ".join(map(str, lens))))
                want.append(("S", sent))
            else:
                b, _alloc = o.endpoint_read(side)
                text.append("E %d" % side)
                want.append(("E", len(b) if b else -1, _fnv(b) if b else 0, o.readable(side), o.writable(1 - side)))
        env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge))
        p = subprocess.run([pyorc.REF_ENDPOINT_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True,
                           timeout=120, env=env)
        assert p.returncode == 0, p.stderr[-400:]
        lines = [ln.split() for ln in p.stdout.strip().splitlines()]
        assert len(lines) == len(want)
        for k, (line, w) in enumerate(zip(lines, want)):
            if w[0] == "S":
                got = ("S", int(line[1]))
            else:
                got = ("E", int(line[1]), int(line[2]), int(line[4]), int(line[5]))
            assert got == w, "step %d (%s): rdma_bp_posix.cc %r, oracle %r" % (k, text[k][:40], got, w)
    finally:
        o.close()


@pytest.mark.parametrize("ring_kb,max_sge", [(64, 30), (256, 30), (64, 4), (1024, 64)])
@pytest.mark.parametrize("seed", range(8))
def test_oracle_write_loop_equals_the_reference_endpoint_itself(seed, ring_kb, max_sge):
    """The reference's own write path -- rdma_write / rdma_flush / rdma_handle_write (rdma_bp_posix.cc:470-586) on its own
    pair.cc: one Send from the cursor per flush, the slices that went out whole dropped from the buffer, the rest waits for
    the writable edge -- against the oracle's Send driven by the same cursor walk.  Writes larger than the ring, writes
    cut by max_sge, edges that find no credit, reads in between that return it: after every write / edge, whether the write
    completed, what the peer can read, what the sender may still write, HasPendingWrites; every read as in the test above."""
    import os
    import subprocess
    if not os.path.exists(pyorc.REF_ENDPOINT_TRACE):
        pytest.skip("oracle/_ref/ref_endpoint_trace not built (no reference tree here)")
    ring = ring_kb * 1024
    rng = random.Random(52000 + 100 * seed + ring_kb + max_sge)
    o = pyorc.OracleLink(ring, max_sge)
    pending = [None, None]   # per side: [slices, idx, byte] of the write that waits
    text, want = [], []

    def flush(side):  # rdma_flush, rdma_bp_posix.cc:470-524
        sl, idx, byte = pending[side]
        left = o.send(side, sl[idx:], byte)
        while left > 0:
            room = len(sl[idx]) - byte
            if left >= room:
                left -= room
                idx += 1
                byte = 0
            else:
                byte += left
                break
        done = idx == len(sl) and byte == 0
        pending[side] = None if done else [sl, idx, byte]
        return ("w", 1 if done else 0, o.readable(1 - side), o.writable(side), int(o.p[side].partial_write))

    try:
        for _ in range(160):
            side = rng.randrange(2)
            r = rng.random()
            if r < 0.3 and pending[side] is None:
                n = rng.choice([1, 2, 5, max_sge, max_sge + 3, 3 * max_sge])
                lens = [rng.choice([9, 14, 100, 256, 257, 4000, ring // 6, ring // 2, ring]) for _ in range(n)]
                sd = rng.randrange(1 << 16)
                pending[side] = [[_pat(sd, i, m) for i, m in enumerate(lens)], 0, 0]
                text.append("W %d %d %d %s" % (side, sd, len(lens), " ".join(map(str, lens))))
                want.append(flush(side))
            elif r < 0.5:
                text.append("F %d" % side)
                want.append(flush(side) if pending[side] is not None else ("w", None))
            else:
                b, _alloc = o.endpoint_read(side)
                text.append("E %d" % side)
                want.append(("E", len(b) if b else -1, _fnv(b) if b else 0, o.readable(side), o.writable(1 - side)))
        env = dict(os.environ, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb), FAKEVERBS_MAX_SGE=str(max_sge))
        p = subprocess.run([pyorc.REF_ENDPOINT_TRACE], input="\n".join(text) + "\n", capture_output=True, text=True,
                           timeout=120, env=env)
        assert p.returncode == 0, p.stderr[-400:]
        lines = [ln.split() for ln in p.stdout.strip().splitlines()]
        assert len(lines) == len(want)
        for k, (line, w) in enumerate(zip(lines, want)):
            if w[0] == "w":
                got = ("w", None) if line[1] == "-" else ("w", int(line[1]), int(line[2]), int(line[3]), int(line[4]))
            else:
                got = ("E", int(line[1]), int(line[2]), int(line[4]), int(line[5]))
            assert got == w, "step %d (%s): rdma_bp_posix.cc %r, oracle %r" % (k, text[k][:40], got, w)
    finally:
        o.close()


# ---- the oracle's frame parser + stream map (K8) against the reference's grpc_chttp2_perform_read ITSELF ---------------
_H2_ERRORS = [("Connect string mismatch", 1), ("Frame size", 2), ("Expected CONTINUATION frame, got", 5),
              ("Expected CONTINUATION frame for", 6), ("Unexpected CONTINUATION", 7),
              ("Expected SETTINGS frame as the first frame", 8), ("Max stream count exceeded", 9),
              ("invalid rst_stream", 10), ("Settings frame received for grpc_chttp2_stream", 11),
              ("non-empty settings ack frame received", 12), ("invalid flags on settings frame", 13),
              ("settings frames must be a multiple of six bytes", 14), ("invalid ping", 15), ("invalid window update", 16),
              ("goaway frame too short", 17), ("Too many trailer frames", 18)]
_PREFACE = b"PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n"


def _h2f(ftype, flags, sid, payload=b""):
    import struct
    return struct.pack(">I", len(payload))[1:] + bytes([ftype, flags]) + struct.pack(">I", sid) + payload


def _perform_read_case(seed):
    """A seeded connection: (is_server, first_frame, max_streams, ops) with ops = ('o', id) | ('w', id) | ('f', bytes)."""
    import struct
    rng = random.Random(52000 + seed)
    server = rng.random() < 0.6
    max_streams = rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 3, 6])
    # (the acknowledged MAX_FRAME_SIZE, checked against every frame header -- parsing.cc:195-205; 0: flow control
    #  disabled, no check)
    max_frame = rng.choice([0, 0, 16384, 16384, 2000, 1000])
    malformed = 0.012 if seed % 3 else 0.0  # (a malformed control frame ends the connection: not in every case)
    wire = bytearray()
    pre_ops = []
    if server:
        wire += _PREFACE
        if rng.random() < 0.92:
            wire += _h2f(4, 0, 0, bytes(6 * rng.randint(0, 3)))     # SETTINGS first
    known, hdr_blocks, pending, next_new = [], {}, {}, 1
    if not server:
        for _ in range(rng.randint(1, 6)):
            pre_ops.append(("o", next_new))
            known.append(next_new)
            next_new += 2

    def some_stream():
        r = rng.random()
        if known and r < 0.8:
            return rng.choice(known)
        if r < 0.9:
            return rng.choice([2, 4, 1000001, next_new + 40])    # even / far away / unknown
        return 0

    def header_block(sid, allow_bad):
        nonlocal wire
        n_cont = rng.choice([0, 0, 0, 1, 2])
        flags = (1 if rng.random() < 0.2 else 0) | (0x20 if rng.random() < 0.15 else 0)
        if hdr_blocks.get(sid, 0) >= 2 and rng.random() < 0.7:
            n_cont = 0       # (a third block without END_HEADERS on one stream fails the connection: "Too many trailer frames")
        hdr_blocks[sid] = hdr_blocks.get(sid, 0) + 1
        wire += _h2f(1, flags | (4 if n_cont == 0 else 0), sid, bytes(rng.randrange(256) for _ in range(rng.randint(0, 30))))
        for i in range(n_cont):
            if allow_bad and rng.random() < 0.04:
                wire += _h2f(rng.choice([0, 1, 6]), 0, sid, b"12345678")       # not a CONTINUATION
                return
            csid = sid if not (allow_bad and rng.random() < 0.04) else sid + 2
            wire += _h2f(9, 4 if i == n_cont - 1 else 0, csid, bytes(rng.randint(0, 12)))

    for _ in range(rng.randint(20, 120)):
        r = rng.random()
        if r < 0.18:                                            # HEADERS: a new stream, or one more block on a known one
            if server and rng.random() < 0.6:
                sid = next_new if rng.random() < 0.9 else max(1, next_new - 4)
                if sid == next_new:
                    next_new += rng.choice([2, 2, 4])
                    known.append(sid)
            else:
                sid = some_stream() or 1
            header_block(sid, True)
        elif r < 0.68:                                          # DATA: the next bytes of that stream's message queue
            sid = some_stream() or 3
            q = pending.setdefault(sid, bytearray())
            while len(q) < 3000:
                n = rng.choice([0, 1, 5, 40, 700, 2500])
                q += bytes([rng.randrange(2)]) + struct.pack(">I", n) + bytes((j * 7 + sid) & 255 for j in range(n))
            k = rng.choice([0, 1, 2, 5, 9, 64, 1000, 2900])
            flags = 1 if rng.random() < 0.08 else (8 if rng.random() < 0.03 else 0)
            wire += _h2f(0, flags, sid, bytes(q[:k]))
            del q[:k]
        elif r < 0.74:
            wire += _h2f(3, 0, some_stream() or 5, struct.pack(">I", rng.choice([0, 8, 2])) if rng.random() < 0.96 else b"123")
        elif r < 0.80:
            bad = rng.random() < 4 * malformed                                                # WINDOW_UPDATE
            wire += _h2f(8, rng.choice([0, 1, 4]) if bad else 0, rng.choice([0, some_stream()]),
                         struct.pack(">I", 1000)[:rng.choice([4, 4, 3, 0])] + (b"x" * rng.choice([0, 0, 1]) if bad else b"") if bad
                         else struct.pack(">I", 1000))
        elif r < 0.85:
            bad = rng.random() < 4 * malformed                                                # PING (also on a stream: not checked)
            wire += _h2f(6, rng.choice([0, 1, 2, 0x80]) if bad else rng.choice([0, 1]), rng.choice([0, 0, some_stream()]) if bad else 0,
                         b"pingpong"[:rng.choice([8, 8, 7, 0])] + (b"!" * rng.choice([0, 1]) if bad else b"") if bad else b"pingpong")
        elif r < 0.89:
            if rng.random() < 5 * malformed:                                                  # SETTINGS: on a stream, bad ack, flags, length
                wire += rng.choice([_h2f(4, 0, some_stream() or 1, bytes(6)), _h2f(4, 1, 0, bytes(6)), _h2f(4, 1, 0, b"x"),
                                    _h2f(4, rng.choice([2, 3, 0x81]), 0, bytes(6)), _h2f(4, 0, 0, bytes(rng.choice([1, 5, 7, 13]))),
                                    _h2f(4, 1, some_stream() or 3, b"")])
            else:
                wire += _h2f(4, 1, 0, b"") if rng.random() < 0.5 else _h2f(4, 0, 0, bytes(6))
        elif r < 0.92:
            short = rng.random() < 5 * malformed                                              # GOAWAY (flags and stream id are not checked)
            wire += _h2f(7, rng.choice([0, 0, 1]), rng.choice([0, 0, 5]), (bytes(8) + b"bye")[:rng.choice([7, 4, 0]) if short else 11])
        elif r < 0.97:
            wire += _h2f(rng.choice([0x0a, 0x0b, 0x42]), rng.randrange(256), some_stream(), bytes(rng.randint(0, 20)))
        elif r < 0.985:
            wire += _h2f(9, 4, some_stream() or 1, b"")          # CONTINUATION out of the blue
        else:
            wire += _h2f(0, 0, some_stream() or 1, b"")
    ops, off = list(pre_ops), 0
    while off < len(wire):
        n = rng.choice([1, 1, 2, 3, 5, 9, 14, 33, 100, 400, 4096, rng.randrange(1, 3000)])
        ops.append(("f", bytes(wire[off:off + n])))
        off += n
        if known and rng.random() < 0.05:
            ops.append(("w", rng.choice(known)))
    return server, max_streams, max_frame, ops


@pytest.mark.parametrize("seed", range(300))
def test_oracle_frame_parser_equals_the_reference_perform_read_itself(seed):
    """oracle/_ref/ref_h2_perform_read_trace = the reference's unmodified parsing.cc (grpc_chttp2_perform_read, init_frame_parser,
    init_{data,header,rst_stream,settings,window_update,ping,goaway,skip}_frame_parser, parse_frame_slice) over its own
    stream_map.cc, frame_data.cc, frame_rst_stream.cc and slice layer; the rest of the transport is stand-ins that keep what
    the parser's control flow depends on (oracle/ref_h2_perform_read_trace.cc says which).  A seeded connection -- preface
    and SETTINGS on a server, HEADERS / CONTINUATION blocks that open, continue and end streams, DATA carrying gRPC messages
    on known, unknown, closed streams, with END_STREAM and with bad flags, RST_STREAM (also of a bad length), PING, SETTINGS,
    WINDOW_UPDATE, GOAWAY, unknown types, stray CONTINUATIONs, a small MAX_CONCURRENT_STREAMS, the write side closing in
    between -- is cut at arbitrary points; the same sequence of accepted streams, closed streams (and whether they left
    the map), message begin / bytes / end per stream, the parser state behind every slice, the connection error and the
    number of live streams at the end.

    Round 4: the malformed-peer corners are in the generator too -- a SETTINGS frame on a stream, a non-empty SETTINGS
    ack, SETTINGS with other flags or a length that is no multiple of six, PING / WINDOW_UPDATE of a wrong length or with
    flags, a GOAWAY shorter than eight bytes (the begin_frame functions of the reference's frame_settings.cc,
    frame_ping.cc, frame_window_update.cc, frame_goaway.cc, compiled unmodified into the driver), a third header block
    on one stream without END_HEADERS ("Too many trailer frames": parsing.cc's skipping header parser + the driver's
    restatement of hpack_parser.cc:1746-1790), and frames larger than the acknowledged MAX_FRAME_SIZE (parsing.cc:195-205,
    with a flow-control object in the driver that only answers "enabled").  What stays outside: the VALUES inside
    SETTINGS / WINDOW_UPDATE payloads (their parse functions need the transport's flow control and are skipped)."""
    import os
    import struct
    import subprocess
    if not os.path.exists(pyorc.REF_H2_PERFORM_READ_TRACE):
        pytest.skip("oracle/_ref/ref_h2_perform_read_trace not built (no reference tree here)")
    server, max_streams, max_frame, ops = _perform_read_case(seed)
    # (no check = a limit no 24-bit frame length exceeds)
    parser = pyorc.H2Parser(flags=(pyorc.H2_SERVER | pyorc.H2_FIRST_FRAME) if server else 0, max_frame_size=max_frame or (1 << 24),
                            max_concurrent_streams=max_streams)
    data = struct.pack("<IIIII", 0 if server else 1, 1 if server else 0, max_streams, 1001, max_frame)
    want, dead, rst = [], False, 0
    for op in ops:
        if op[0] == "f":
            data += b"f" + struct.pack("<I", len(op[1])) + op[1]
        else:
            data += op[0].encode() + struct.pack("<I", op[1])
        if dead:
            continue
        if op[0] == "o":
            assert parser.open_stream(op[1]) == 0
        elif op[0] == "w":
            before = parser.live_streams()
            if parser.close_writes(op[1]) == 0:
                want.append("C %d %d" % (op[1], before - parser.live_streams()))
        else:
            rc, evs = parser.feed(op[1])
            for kind, a, b, c, d in evs:
                if kind == pyorc.EV_STREAM_OPEN:
                    want.append("O %d" % c)
                elif kind == pyorc.EV_STREAM_CLOSED:
                    want.append("C %d %d" % (c, a))
                elif kind == pyorc.EV_MSG_BEGIN:
                    want.append("B %d %d %d" % (c, 0x80000000 if a else 0, b))
                elif kind == pyorc.EV_MSG_BYTES:
                    want.append("Y %d %d" % (c, b))
                elif kind == pyorc.EV_MSG_END:
                    want.append("E %d" % c)
                elif kind == pyorc.EV_FRAME and a == 0xff:
                    want.append("G %d" % c)     # a gRPC message flag byte > 1 (bytes of a skipped DATA frame are missing)
                elif kind == pyorc.EV_FRAME and a == 0 and (b >> 8):
                    rst += 1            # DATA with bad flags: RST_STREAM queued (parsing.cc:388-394)
            if rc != 0:
                want.append("X %d" % rc)
                dead = True
            else:
                st = parser.p.state
                want.append("P %d %d" % (st, parser.p.incoming_frame_size if st == 33 else 0))
    want.append("S %d %d" % (rst, parser.live_streams()))
    r = subprocess.run([pyorc.REF_H2_PERFORM_READ_TRACE], input=data, capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr[-300:]
    got = []
    for line in r.stdout.decode().strip().splitlines():
        if line.startswith("X "):
            codes = [c for pre, c in _H2_ERRORS if line[2:].startswith(pre)]
            assert codes, line
            line = "X %d" % codes[0]
        got.append(line)


    def split(lines):
        """-> (stream / connection events in order, {stream: its B / Y / E sequence with adjacent payload pieces summed}):
        when the surface pulls message bytes is not the parser's business (the driver pulls behind every slice)."""
        control, per_stream = [], {}
        for ln in lines:
            f = ln.split()
            if f[0] in "BYEG":
                seq = per_stream.setdefault(int(f[1]), [])
                if f[0] == "Y" and seq and seq[-1][0] == "Y":
                    seq[-1] = ("Y", seq[-1][1] + int(f[2]))
                elif f[0] == "Y":
                    if int(f[2]):
                        seq.append(("Y", int(f[2])))
                else:
                    seq.append(tuple(f[:1] + [int(x) for x in f[2:]]))
            else:
                control.append(ln)
        return control, per_stream

    got_c, got_m = split(got)
    want_c, want_m = split(want)
    assert got_c == want_c, next((i, g, w) for i, (g, w) in enumerate(zip(got_c + ["-"], want_c + ["-"])) if g != w)
    assert got_m == want_m
    assert any(got_m.values()) or any(ln[0] in "OCX" for ln in got_c)
