"""CPU: the HTTP/2 DATA framing / deframing restatement against the reference's
own byte vectors (tests/golden/h2_bad_client.json, extracted from
test/core/bad_client/tests/*.cc) and the constructive framing of
test/cpp/microbenchmarks/bm_chttp2_transport.cc:504-560."""
import json
import os
import random

import pytest

import grpc_rdma_amd  # noqa: F401  (import shim)
from grpc_rdma_amd import h2
from oracle import pyorc
from oracle.pyorc import EV_FRAME, EV_MSG_BEGIN, EV_MSG_BYTES, EV_MSG_END, EV_PAYLOAD

VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "h2_bad_client.json")))["vectors"]


def feed_chunks(data, cuts, prefix=True, streams=()):
    """Feed `data` split at `cuts`; -> (rc, events with absolute stream offsets).  `streams`:
    ids the caller opened (a client's calls; a server learns its streams from HEADERS)."""
    p = pyorc.H2Parser(expect_client_prefix=prefix)
    for sid in streams:
        assert p.open_stream(sid) == 0
    out, base = [], 0
    bounds = [0] + sorted(cuts) + [len(data)]
    for a, b in zip(bounds, bounds[1:]):
        rc, ev = p.feed(data[a:b])
        if rc:
            return rc, out
        for k, x, y, z, w in ev:
            if k in (EV_PAYLOAD, EV_MSG_BYTES):
                out.append((k, x + a, y, z, w))
            else:
                out.append((k, x, y, z, w))
    return 0, out


def coalesce(events):
    """Chunking-independent view of a parse: the frame / message events in order,
    and the byte intervals handed out as frame payload and as message bytes."""
    def merged(iv):
        out = []
        for a, n in sorted(iv):
            if n == 0:
                continue
            if out and out[-1][0] + out[-1][1] == a:
                out[-1] = (out[-1][0], out[-1][1] + n)
            else:
                out.append((a, n))
        return out
    control = [e for e in events if e[0] not in (EV_PAYLOAD, EV_MSG_BYTES)]
    payload = merged([(e[1], e[2]) for e in events if e[0] == EV_PAYLOAD])
    frame_ends = sorted(e[1] + e[2] for e in events if e[0] == EV_PAYLOAD and e[3] == 1)
    msg = {}
    for e in events:
        if e[0] == EV_MSG_BYTES:
            msg.setdefault(e[3], []).append((e[1], e[2]))
    return control, payload, frame_ends, {k: merged(v) for k, v in msg.items()}


@pytest.mark.parametrize("vec", VEC, ids=[v["name"] for v in VEC])
def test_reference_vectors_parse_as_documented(vec):
    data = bytes.fromhex(vec["hex"])
    rc, ev = feed_chunks(data, [])
    assert rc == 0
    frames = [[a, b & 0xFF, c, d] for k, a, b, c, d in ev if k == EV_FRAME and a != 0xFF]
    assert frames == vec["frames"]
    begins = [(c, a, b) for k, a, b, c, d in ev if k == EV_MSG_BEGIN]
    assert begins == [(m["stream"], m["compressed"], m["length"]) for m in vec["messages"]]
    ends = [c for k, a, b, c, d in ev if k == EV_MSG_END]
    assert ends == [m["stream"] for m in vec["messages"] if m["complete"]]
    for m in vec["messages"]:
        if m["complete"]:
            body = b"".join(data[a:a + b] for k, a, b, c, d in ev if k == EV_MSG_BYTES and c == m["stream"])
            assert body == bytes.fromhex(m["payload_byte"]) * m["length"]
    err = vec["stream_error"]
    if err and err["code"] == "bad_grpc_frame_type":
        assert any(k == EV_FRAME and a == 0xFF and d == 4 and c == err["stream"] for k, a, b, c, d in ev)
    if err and err["code"] == "data_flags":
        assert any(k == EV_FRAME and a == 0 and (b >> 8) == 3 for k, a, b, c, d in ev)


@pytest.mark.parametrize("vec", VEC, ids=[v["name"] for v in VEC])
def test_every_split_point_gives_the_same_parse(vec):
    """h2_sockpair_1byte.cc feeds chttp2 one byte at a time; the state machines
    must be resumable at any byte (parsing.cc:56-253, frame_data.cc:92-276)."""
    data = bytes.fromhex(vec["hex"])
    rc0, whole = feed_chunks(data, [])
    rc1, single = feed_chunks(data, list(range(1, len(data))))
    assert rc0 == rc1 == 0
    assert coalesce(single) == coalesce(whole)
    rng = random.Random(5)
    for _ in range(20):
        cuts = sorted(rng.sample(range(1, len(data)), rng.randint(1, 12)))
        rc, ev = feed_chunks(data, cuts)
        assert rc == 0 and coalesce(ev) == coalesce(whole)


def create_incoming_data_slice(length, frame_size):
    """bm_chttp2_transport.cc:504-560: 5-byte message header + 'a' * length cut into
    DATA frames of frame_size on stream 1 (the last frame holds the rest, > 0 bytes)."""
    unframed = bytes([0]) + length.to_bytes(4, "big") + b"a" * length
    out = bytearray()
    while len(unframed) > frame_size:
        out += frame_size.to_bytes(3, "big") + bytes([0, 0, 0, 0, 0, 1]) + unframed[:frame_size]
        unframed = unframed[frame_size:]
    out += len(unframed).to_bytes(3, "big") + bytes([0, 0, 0, 0, 0, 1]) + unframed
    return bytes(out)


@pytest.mark.parametrize("length", [0, 1, 4, 5, 16378, 16379, 16380, 16384, 100000, 1 << 20])
def test_tx_framing_equals_the_reference_benchmark_framing(length):
    msg = b"a" * length
    wire, lens = pyorc.h2_frame_message(msg, stream_id=1, max_frame=16384)
    assert wire == create_incoming_data_slice(length, 16384)
    assert sum(lens) == len(wire)
    # and it parses back to exactly that message
    rc, ev = feed_chunks(wire, [], prefix=False, streams=(1,))
    assert rc == 0
    assert [(a, b) for k, a, b, c, d in ev if k == EV_MSG_BEGIN] == [(0, length)]
    assert b"".join(wire[a:a + b] for k, a, b, c, d in ev if k == EV_MSG_BYTES) == msg


def test_one_mib_message_slice_list_is_the_documented_one():
    """130 slices, N = 1 049 170, E = 1 051 712."""
    msg = bytes(1048580)
    wire, lens = pyorc.h2_frame_message(msg)
    assert len(lens) == 130 and lens[0] == 14 and lens[1] == 16379 and lens[-2:] == [9, 9]
    assert sum(lens) == 1049170
    assert h2.ring_bytes_for(lens) == 1051712


@pytest.mark.parametrize("M", [0, 1, 4, 5, 6, 17, 18, 19, 100, 16379, 16380, 16384, 16385, 70000])
@pytest.mark.parametrize("F", [1, 3, 5, 9, 100, 16384])
def test_host_layout_matches_oracle(M, F):
    if M // F > 4000:
        pytest.skip("too many frames for a unit test")
    rng = random.Random(M * 31 + F)
    msg = bytes(rng.getrandbits(8) for _ in range(M))
    wire, lens = pyorc.h2_frame_message(msg, 7, F)
    items = h2.frame_message(M, 7, F)
    assert [len(i[1]) if i[0] == "inl" else i[1][1] for i in items] == lens
    assert b"".join(i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items) == wire


def test_oversized_frame_and_bad_prefix_are_connection_errors():
    p = pyorc.H2Parser(expect_client_prefix=True)
    rc, _ = p.feed(b"PRI * HTTP/2.0\r\n\r\nSM\r\n\r\nX"[:24][:-1] + b"X")
    assert rc == 1  # connect string mismatch, parsing.cc:91-104
    p = pyorc.H2Parser(expect_client_prefix=False, max_frame_size=16384)
    rc, _ = p.feed((16385).to_bytes(3, "big") + bytes([0, 0, 0, 0, 0, 1]))
    assert rc == 2  # parsing.cc:195-205


# ---------------------------------------------------------------------------------------------
# The stream map the DATA path reads (parsing.cc init_data_frame_parser / init_header_frame_parser,
# chttp2_transport.cc grpc_chttp2_mark_stream_closed): lookup only, streams accepted from HEADERS on a
# server, read-closed by END_STREAM, removed once both sides are closed.
from h2_helpers import PREFACE, frame, grpc_msg, messages_of, unary_call  # noqa: E402


def test_two_hundred_sequential_unary_streams_and_forty_interleaved():
    """Every RPC is a new stream: a connection must keep deframing after any number of them
    (the reference looks the stream up in the transport's map, parsing.cc:352-370)."""
    rng = random.Random(11)
    data = bytearray(PREFACE + frame(4, 0, 0))
    expect = []
    sid = 1
    for i in range(200):
        body = bytes(rng.getrandbits(8) for _ in range(rng.randrange(0, 300)))
        data += unary_call(sid, body)
        expect.append((sid, body))
        sid += 2
    # 40 concurrent streams, each message cut into three DATA frames, frames interleaved
    ids = list(range(sid, sid + 80, 2))
    bodies = {s_: bytes(rng.getrandbits(8) for _ in range(rng.randrange(10, 2000))) for s_ in ids}
    for s_ in ids:
        data += frame(1, 4, s_, b"\x82")
    pieces = {s_: grpc_msg(bodies[s_]) for s_ in ids}
    for part in range(3):
        order = ids[:]
        rng.shuffle(order)
        for s_ in order:
            m = pieces[s_]
            cut = [0, len(m) // 3, 2 * len(m) // 3, len(m)]
            data += frame(0, 1 if part == 2 else 0, s_, m[cut[part]:cut[part + 1]])
    data = bytes(data)
    p = pyorc.H2Parser(expect_client_prefix=True)
    rc, ev = p.feed(data, cap=200000)
    assert rc == 0
    got = messages_of(ev, data)
    assert got[:200] == expect
    assert sorted(got[200:]) == sorted(bodies.items())
    opened = [c for k, a, b, c, d in ev if k == pyorc.EV_STREAM_OPEN]
    closed = [c for k, a, b, c, d in ev if k == pyorc.EV_STREAM_CLOSED]
    assert opened == list(range(1, sid + 80, 2)) and sorted(closed) == opened
    # read-closed streams stay in the map until their write side closes too
    assert p.live_streams() == 240
    for s_ in opened:
        assert p.close_writes(s_) == 0
    assert p.live_streams() == 0
    # and the same bytes, with the write side of every finished call closed as it completes
    p = pyorc.H2Parser(expect_client_prefix=True, max_concurrent_streams=64)
    pos, peak = 0, 0
    for n in [len(PREFACE) + 9] + [7] * ((len(data) - len(PREFACE) - 9) // 7 + 1):
        chunk = data[pos:pos + n]
        pos += n
        rc, e2 = p.feed(chunk)
        assert rc == 0
        for k, a, b, c, d in e2:
            if k == pyorc.EV_STREAM_CLOSED and a == 0:
                p.close_writes(c)
        peak = max(peak, p.live_streams())
    assert p.live_streams() == 0 and peak <= 41


def test_data_for_unknown_closed_or_refused_streams_is_skipped():
    body = b"x" * 40
    p = pyorc.H2Parser(expect_client_prefix=True)
    data = PREFACE + frame(4, 0, 0)
    data += frame(0, 0, 5, grpc_msg(body))              # no HEADERS seen: unknown stream -> skip
    data += unary_call(7, body)                         # accepted, read-closed by END_STREAM
    data += frame(0, 0, 7, grpc_msg(body))              # DATA after END_STREAM -> skip (parsing.cc:372-374)
    data += frame(1, 4, 3, b"\x82") + frame(0, 1, 3, grpc_msg(body))  # id below last_new_stream_id -> ignored
    data += frame(1, 4, 8, b"\x82") + frame(0, 1, 8, grpc_msg(body))  # even id -> ignored
    data += frame(1, 4, 9, b"\x82") + frame(3, 0, 9, b"\0\0\0\x08")  # RST_STREAM removes stream 9
    data += frame(0, 1, 9, grpc_msg(body))              # -> unknown again
    rc, ev = p.feed(data)
    assert rc == 0
    assert messages_of(ev, data) == [(7, body)]
    assert [c for k, a, b, c, d in ev if k == pyorc.EV_STREAM_OPEN] == [7, 9]
    assert [(c, a) for k, a, b, c, d in ev if k == pyorc.EV_STREAM_CLOSED] == [(7, 0), (9, 1)]
    assert p.live_streams() == 1


def test_client_side_streams_are_opened_by_the_caller():
    body = b"y" * 100
    data = frame(0, 0, 1, grpc_msg(body)) + frame(1, 5, 1, b"\x88")  # DATA, then trailers with END_STREAM
    p = pyorc.H2Parser(expect_client_prefix=False)
    rc, ev = p.feed(data)
    assert rc == 0 and messages_of(ev, data) == []      # the call was never started here
    p = pyorc.H2Parser(expect_client_prefix=False)
    assert p.open_stream(1) == 0 and p.open_stream(1) == -1
    rc, ev = p.feed(data)
    assert rc == 0 and messages_of(ev, data) == [(1, body)]
    assert [(c, a) for k, a, b, c, d in ev if k == pyorc.EV_STREAM_CLOSED] == [(1, 0)]
    assert p.close_writes(1) == 0 and p.live_streams() == 0


def test_continuation_and_first_frame_rules():
    H = lambda rc_expected, data, **kw: (pyorc.H2Parser(**kw).feed(data)[0] == rc_expected)
    srv = dict(expect_client_prefix=True)
    assert H(8, PREFACE + frame(0, 0, 1, b"abc"), **srv)                  # first frame must be SETTINGS
    ok = PREFACE + frame(4, 0, 0)
    assert H(0, ok + frame(1, 0, 1, b"\x82") + frame(9, 4, 1, b"\x86") + frame(0, 1, 1, grpc_msg(b"z")), **srv)
    assert H(5, ok + frame(1, 0, 1, b"\x82") + frame(0, 0, 1, b""), **srv)  # expected CONTINUATION
    assert H(6, ok + frame(1, 0, 1, b"\x82") + frame(9, 4, 3, b""), **srv)  # CONTINUATION for another stream
    assert H(7, ok + frame(9, 4, 1, b""), **srv)                          # unexpected CONTINUATION
    assert H(10, ok + frame(3, 0, 1, b"\0\0\0"), **srv)                  # RST_STREAM length != 4
    p = pyorc.H2Parser(expect_client_prefix=True, max_concurrent_streams=2)
    rc, _ = p.feed(ok + frame(1, 4, 1, b"") + frame(1, 4, 3, b"") + frame(1, 4, 5, b""))
    assert rc == 9                                                        # Max stream count exceeded
    # END_STREAM on HEADERS closes reads once the header block ends (END_HEADERS on the CONTINUATION)
    p = pyorc.H2Parser(expect_client_prefix=True)
    rc, ev = p.feed(ok + frame(1, 1, 1, b"\x82") + frame(9, 4, 1, b"\x86"))
    assert rc == 0 and [(c, a) for k, a, b, c, d in ev if k == pyorc.EV_STREAM_CLOSED] == [(1, 0)]


def _grpcio_capture():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h2_grpcio_capture.json")
    d = json.load(open(path))
    data = bytes.fromhex(d["client_bytes_hex"])
    from oracle import gen_h2_grpcio_capture
    unary, stream = gen_h2_grpcio_capture.payloads()   # the payloads that went into the capture
    assert [len(p) for p in unary + stream] == d["payload_lengths"]
    return data, unary + stream


def test_oracle_deframes_what_a_real_grpc_client_sends():
    """tests/golden/h2_grpcio_capture.json holds every byte a stock gRPC C-core client (grpcio) put
    on the wire for four unary calls and one client-streaming call with known payloads
    (oracle/gen_h2_grpcio_capture.py): preface, SETTINGS, HPACK HEADERS, WINDOW_UPDATE, PING,
    RST_STREAM and 29 DATA frames.  The oracle's server-side parser, fed the bytes whole and cut at
    arbitrary points, must hand back exactly those payloads in order -- K8/K9 pinned against bytes
    produced by real chttp2 code, not by hand."""
    import random
    data, exp = _grpcio_capture()
    assert data.startswith(PREFACE)
    for seed in range(6):
        rng = random.Random(seed)
        cuts = [] if seed == 0 else sorted(rng.sample(range(1, len(data)), rng.choice([1, 7, 60, 400])))
        bounds = [0] + cuts + [len(data)]
        p = pyorc.H2Parser(expect_client_prefix=True)
        events, base = [], 0
        for a, b in zip(bounds, bounds[1:]):
            rc, ev = p.feed(data[a:b])
            assert rc == 0
            events += [(k, x + base if k == pyorc.EV_MSG_BYTES else x, y, z, w) for k, x, y, z, w in ev]
            base = b
        msgs = messages_of(events, data)
        assert [m for _, m in msgs] == exp
        # four unary calls on streams 1,3,5,7 and the streaming call on 9
        assert [sid for sid, _ in msgs] == [1, 3, 5, 7, 9, 9, 9, 9, 9]
