"""The deframer over chunks (csrc/grdma_h2_kernels.h: k_h2_deframe_chunks / k_h2_merge_or_deframe): a list of >= 2048
delivered slices is cut at slices in which a message starts, the chunks
are parsed side by side from the state the boundary step recorded, and merged only when the chain of end states holds.
Whatever the list holds, the events, the parser state (checked through the NEXT call's events) and the stream map are
the sequential parser's, i.e. the oracle's (parsing.cc:111-250 + frame_data.cc:92-276 restated in oracle/)."""
import random

import numpy as np
import pytest

from oracle import pyorc
from tests.h2_helpers import PREFACE, frame, grpc_msg
from tests.test_h2_fast_host import receiver_slices

pytestmark = pytest.mark.gpu


def fast_sender_slices(sizes, sid=1, seed=0, max_frame=16384):
    """tests/test_h2_fast_host.py::sender_slices with numpy payloads (thousands of slices)."""
    out = []
    for i, n in enumerate(sizes):
        pay = ((np.arange(n, dtype=np.uint32) * 13 + i + seed) % 251).astype(np.uint8).tobytes()
        body = b"\x00" + n.to_bytes(4, "big") + pay
        off = 0
        while off < len(body):
            k = min(max_frame, len(body) - off)
            fh = k.to_bytes(3, "big") + bytes([0, 0]) + sid.to_bytes(4, "big")
            if off == 0:
                out.append(fh + body[:5])
                if k > 5:
                    out.append(body[5:k])
            else:
                out += [fh, body[off:off + k]]
            off += k
    return out


class Both:
    """The same calls on a device parser and on the oracle's."""

    def __init__(self, g, prefix, streams=(), chunks=None, gap_seed=None):
        from grpc_rdma_amd import h2dev
        self.g = g
        self.dev = h2dev.Parser(prefix, boundary_step=True, chunks=chunks)
        self.orc = pyorc.H2Parser(expect_client_prefix=prefix)
        if streams:
            assert self.dev.open_streams(streams) == 0
            for sid in streams:
                assert self.orc.open_stream(sid) == 0
        self.rng = random.Random(gap_seed) if gap_seed is not None else None

    def call(self, slices):
        arena, table = bytearray(), []
        for s in slices:
            if self.rng is not None:
                arena += b"\xee" * self.rng.randrange(1, 16)
            else:
                arena += bytes((-len(arena)) % 16)
            table.append((len(arena), len(s)))
            arena += s
        buf = self.g.DeviceBuffer(data=bytes(arena) + bytes(64))
        err, ev = self.dev.deframe(buf.ptr, table, cap=8 * len(slices) + 4096)
        exp = []
        for i, s in enumerate(slices):
            rc, e = self.orc.feed(s, cap=1024)
            assert rc == 0
            exp += [(k, a, b, c, d, i) for k, a, b, c, d in e]
        assert err == 0
        assert len(ev) == len(exp)
        if ev != exp:
            bad = next(i for i in range(len(ev)) if ev[i] != exp[i])
            raise AssertionError("event %d: device %r, oracle %r" % (bad, ev[bad], exp[bad]))
        return len(ev)

    def close(self):
        self.dev.close()


@pytest.mark.parametrize("shape", ["sender", "receiver"])
@pytest.mark.parametrize("gaps", [False, True], ids=["aligned", "unaligned"])
def test_chunked_deframer_streaming_shapes(gpu, shape, gaps):
    """A streaming call of equal messages, as the sending side's slice buffer and as the receiving endpoint's reads:
    the first call leaves the hint, every later long list is planned, verified and merged."""
    b = Both(gpu, False, streams=(1,), gap_seed=5 if gaps else None)
    mk = (lambda tx: tx) if shape == "sender" else receiver_slices
    warm = mk(fast_sender_slices([100000] * 3))
    b.call([frame(1, 4, 1, b"\x82")] + warm)
    assert b.dev.chunk_stats() == (0, 0)  # (too short, and no hint before it)
    for rep in range(2):
        body = mk(fast_sender_slices([100000] * 330, seed=rep))
        assert len(body) >= 4096
        b.call(body)
        assert b.dev.chunk_stats() == (rep + 1, rep + 1)
    # a short list goes the sequential way and finds the state the merge installed
    b.call(mk(fast_sender_slices([5, 100000, 16379], seed=9)))
    assert b.dev.chunk_stats() == (2, 2)
    b.close()


def test_chunked_deframer_mixed_sizes_and_control_frames(gpu):
    """Message sizes that end on and off frame boundaries, PING frames between messages: still one stream in one state
    at every message start, so the chain holds and the call is merged."""
    rng = random.Random(3)
    b = Both(gpu, True)
    b.call([PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82")] + fast_sender_slices([7, 40000, 16379]))
    sizes = [rng.choice([1, 9, 300, 16379, 16380, 40000, 65536, 200000]) for _ in range(500)]
    body = []
    for i, n in enumerate(sizes):
        body += fast_sender_slices([n], seed=i)
        if i % 50 == 17:
            body.append(frame(6, 0, 0, bytes(8)))
    assert len(body) >= 2048
    b.call(body)
    assert b.dev.chunk_stats() == (1, 1)
    b.call(receiver_slices(fast_sender_slices([70000] * 400)))
    assert b.dev.chunk_stats() == (2, 2)
    b.close()


@pytest.mark.parametrize("shape", ["sender", "receiver"])
def test_chunked_deframer_messages_longer_than_a_chunk(gpu, shape):
    """Messages of ~1200 slices (small frames): several quantiles lie in front of the same message start, so some chunks
    are EMPTY -- they end where and as they start -- and the chain still holds.  One message far longer than the
    search span has no cut at all: the call is the sequential deframer's.  (Message lengths are multiples of the frame
    size: the receiving side then sees the closing 5-byte frame and the next message start in one slice, the shape the
    boundary step -- and with it the cut search -- knows.)"""
    mk = (lambda tx: tx) if shape == "sender" else receiver_slices
    b = Both(gpu, False, streams=(1,))
    b.call([frame(1, 4, 1, b"\x82")] + mk(fast_sender_slices([3072] * 3, max_frame=1024)))
    body = mk(fast_sender_slices([600064] * 8, seed=2, max_frame=1024))
    assert len(body) >= 8 * 1000
    b.call(body)
    assert b.dev.chunk_stats() == (1, 1)
    b.call(mk(fast_sender_slices([300032, 5120, 900096, 70656] * 3, seed=3, max_frame=1024)))
    assert b.dev.chunk_stats() == (2, 2)
    b.call(mk(fast_sender_slices([6000640], seed=4, max_frame=1024)))  # ~11 700 slices, one message start
    assert b.dev.chunk_stats()[1] == 2
    b.call(mk(fast_sender_slices([100352] * 40, seed=5, max_frame=1024)))
    assert b.dev.chunk_stats()[1] == 3
    b.close()


def test_chunked_deframer_declines_what_it_cannot_verify(gpu):
    """Lists on which the chain of end states does NOT hold are the sequential parser's: a second stream that is
    mid-message at a cut, a stream that opens in the middle of the list, a stream that closes in it, a connection
    error.  The events are the oracle's every time and nothing of the chunks' work shows."""
    b = Both(gpu, True)
    b.call([PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82"), frame(1, 4, 3, b"\x82")] + fast_sender_slices([50000] * 3))
    # (a) two streams interleaved, a frame of stream 3 behind every message of stream 1: stream 3 is in the middle of
    # a message at every cut
    one = fast_sender_slices([60000] * 300, sid=1)
    three = fast_sender_slices([6000000], sid=3, seed=4)
    mix, j = [], 0
    for i in range(0, len(one), 8):
        mix += one[i:i + 8]
        if j < len(three):
            mix += three[j:j + 2]
            j += 2
    mix += three[j:]
    assert len(mix) >= 2048
    b.call(mix)
    planned, merged = b.dev.chunk_stats()
    assert planned == 1 and merged == 0
    # (b) a stream opens in the middle of the list (HEADERS of stream 5): live_streams / last_new_stream_id move
    body = fast_sender_slices([60000] * 150, seed=1) + [frame(1, 4, 5, b"\x82")] + fast_sender_slices([60000] * 150, seed=2)
    b.call(body)
    assert b.dev.chunk_stats()[1] == 0
    # (c) the hinted stream ends in the middle of the list (END_STREAM), stream 3 carries on behind it: no message of
    # stream 1 starts behind its last one, so the chunk that begins there runs to the end of the list -- and what
    # happens inside the LAST chunk is not constrained (a stream may close or open there): merged
    body = fast_sender_slices([60000] * 150, seed=3) + [frame(0, 1, 1, grpc_msg(bytes(100)))]
    body += fast_sender_slices([60000] * 150, sid=3, seed=6)
    assert len(body) >= 2048
    b.call(body)
    assert b.dev.chunk_stats()[1] == 1
    # (d) the hint now names stream 3; a stream opens inside the last chunk; the merged call leaves the state the next
    # (short, sequential) call continues from
    b.call(fast_sender_slices([60000] * 300, sid=3, seed=7) + [frame(1, 4, 7, b"\x82")] +
           fast_sender_slices([60000], sid=7, seed=8) + fast_sender_slices([60000] * 5, sid=3, seed=9))
    assert b.dev.chunk_stats()[1] == 2
    b.call(fast_sender_slices([100, 60000], sid=7, seed=10) + fast_sender_slices([60000] * 2, sid=3, seed=11))
    assert b.dev.live_streams() == b.orc.live_streams()
    b.close()


def test_chunked_and_sequential_deframer_agree_call_by_call(gpu):
    """The same calls on a parser with chunks and on one without: identical events (both checked against the oracle),
    through alternating long and short lists, so that each path starts from the state the other left."""
    rng = random.Random(11)
    on = Both(gpu, False, streams=(1,), chunks=True)
    off = Both(gpu, False, streams=(1,), chunks=False)
    first = [frame(1, 4, 1, b"\x82")]
    for rep in range(5):
        n = rng.choice([3, 40, 260])
        size = rng.choice([16379, 100000, 1 << 20]) if n < 100 else 150000
        tx = fast_sender_slices([size] * n, seed=rep)
        body = first + (receiver_slices(tx) if rep % 2 else tx)
        first = []
        assert on.call(body) == off.call(body)
    assert off.dev.chunk_stats() == (0, 0)
    assert on.dev.chunk_stats()[1] >= 1
    on.close()
    off.close()


def test_h2_pipe_at_the_bench_configuration_matches_the_oracle(gpu):
    """What bench.py's value_with_h2 leg runs, at its size: 256 messages of 1 MiB per step framed on the device
    (k_h2_frame_index / k_h2_frame_emit), carried through a 128 MiB ring by the streaming job, deframed over chunks --
    framing and deframing as nodes of the job's graph, two jobs over one connection taking turns.  Every step's events
    equal the oracle's over the slices that step delivered (33 k slices, 83 k events per step), the delivered bytes are
    the framed messages, and every step after the first (which leaves the hint) was merged from chunks."""
    g = gpu
    from grpc_rdma_amd import h2 as h2host, h2dev, stream as gs
    n_msgs, msg_len = 256, 1 << 20
    payload = ((np.arange(n_msgs * msg_len, dtype=np.uint32) * 7 + 3) % 251).astype(np.uint8)
    pbuf = g.DeviceBuffer(data=payload.tobytes())
    msgs = [(pbuf.ptr + i * msg_len, msg_len, 1, 0) for i in range(n_msgs)]
    lens = [len(it[1]) if it[0] == "inl" else it[1][1] for it in h2host.frame_message(msg_len, 1, 16384)] * n_msgs
    scratch = g.DeviceBuffer(nbytes=max(lens) + 64)
    sge = [(scratch.ptr, n) for n in lens]  # placeholders: the framing kernels overwrite the table
    R = 128 << 20
    tx, rx = g.Pair(R, 4095), g.Pair(R, 4095)
    g.connect_pairs(tx, rx)
    N = sum(lens)
    scap = 2 * len(lens) + 64 + N // 256
    dst_cap = N + 16 * scap + 4096
    parser = h2dev.Parser(False)
    assert parser.open_streams([1]) == 0
    jobs, pipes, dsts = [], [], []
    for _ in range(2):
        dst = g.DeviceBuffer(nbytes=dst_cap)
        job = gs.StreamJob(tx, rx, sge, dst.ptr, dst_cap, scap, 16)
        job.set_pipeline(True)
        r = job.run(gs.RUN_EAGER)
        job.set_rounds(int(max(r.tx_rounds, r.rx_rounds)))
        r = job.run(gs.RUN_GRAPH)
        assert r.done and r.bytes_delivered == N
        pipes.append(h2dev.Pipe(job, msgs, parser, len(job.delivered_slices(0)), 4 * len(lens) + 1024))
        jobs.append(job)
        dsts.append(dst)
    po = pyorc.H2Parser(expect_client_prefix=False)
    assert po.open_stream(1) == 0
    # the framed stream the sender builds: per message 64 frames of 16384 bytes and one of 5
    hdr = lambda n: n.to_bytes(3, "big") + b"\x00\x00" + (1).to_bytes(4, "big")
    steps = 3
    for step in range(steps):
        p_, job, dst = pipes[step % 2], jobs[step % 2], dsts[step % 2]
        p_.enqueue()
        res = p_.sync(want_events=True)
        assert res["h2_error"] == 0 and res["framed"] == len(lens) and not res["frame_overflow"] and not res["deframe_overflow"]
        ds = job.delivered_slices(0)
        assert res["parsed"] == len(ds)
        got = dst.read(dst_cap)
        ev_o = []
        for i, (o, n) in enumerate(ds):
            rc, ev = po.feed(got[o:o + n], cap=1024)
            assert rc == 0
            ev_o += [(k, a, b, c, d, i) for k, a, b, c, d in ev]
        assert len(res["event_list"]) == len(ev_o)
        assert res["event_list"] == ev_o, "step %d" % step
        if step == 0:
            stream = b"".join(got[o:o + n] for o, n in ds)
            exp = bytearray()
            for i in range(n_msgs):
                body = b"\x00" + msg_len.to_bytes(4, "big") + payload[i * msg_len:(i + 1) * msg_len].tobytes()
                for off in range(0, len(body), 16384):
                    piece = body[off:off + 16384]
                    exp += hdr(len(piece)) + piece
            assert stream == bytes(exp)
    planned, merged = parser.chunk_stats()
    assert merged == steps - 1 and planned >= merged
    for p_ in pipes:
        p_.close()
    for j_ in jobs:
        j_.close()
    parser.close()
    tx.close()
    rx.close()


def test_chunked_deframer_random_streams(gpu):
    """Seeded random calls: message sizes from one byte to a few hundred KiB, frames smaller than the maximum, control
    frames between messages, now and then a message of a second stream, the receiving side's reads, slices cut in
    two at random.  Whatever the chunked deframer decides -- merge or decline -- every call's events are the oracle's
    (a merge that should not have happened shows here as a wrong event list), and over the seeds it does merge."""
    merges = []
    for seed in range(8):
        merges.append(_random_stream_calls(gpu, seed))
    assert sum(merges) >= 4, merges


def _random_stream_calls(gpu, seed):
    rng = random.Random(1000 + seed)
    b = Both(gpu, True, gap_seed=seed if seed % 2 else None)
    b.call([PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82"), frame(1, 4, 3, b"\x82")] +
           fast_sender_slices([rng.randrange(1, 50000) for _ in range(3)], seed=seed))
    for call in range(3):
        max_frame = rng.choice([16384, 16384, 4096, 1000])
        sizes_pool = rng.choice([[1, 9, 300, 16379, 16380, 40000, 65536, 200000], [70000], [5, 16384 * 3 - 5, 16384 - 5],
                                 [max_frame * 8, max_frame * 40]])
        second_stream = rng.random() < 0.3
        body = []
        while len(body) < 2600:
            body += fast_sender_slices([rng.choice(sizes_pool)], seed=len(body), max_frame=max_frame)
            r = rng.random()
            if r < 0.05:
                body.append(frame(6, 0, 0, bytes(8)))            # PING
            elif r < 0.08:
                body.append(frame(8, 0, 0, (1000).to_bytes(4, "big")))  # WINDOW_UPDATE on the connection
            elif second_stream and r < 0.15:
                body += fast_sender_slices([rng.randrange(1, 3000)], sid=3, seed=len(body), max_frame=max_frame)
        if rng.random() < 0.5:
            body = receiver_slices(body)
        if rng.random() < 0.3:
            cut = []
            for s_ in body:
                if len(s_) > 2 and rng.random() < 0.02:
                    k = rng.randrange(1, len(s_))
                    cut += [s_[:k], s_[k:]]
                else:
                    cut.append(s_)
            body = cut
        b.call(body)
    planned, merged = b.dev.chunk_stats()
    assert merged <= planned
    b.close()
    return merged

