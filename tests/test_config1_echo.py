"""BASELINE config 1 / the reference's end-to-end echo check: seeded RPCs whose request sizes are
drawn like examples/cpp/test/common.h:4-31 (uniform in [1, 4 MiB - 1 KiB]), echoed by the server,
compared byte for byte by the client.

* CPU (`-m "not gpu"`): 1000 RPCs through the oracle's ring-buffer connection -- every request is
  framed as chttp2 frames it (HEADERS + DATA frames of <= 16384 bytes, the slices one
  grpc_endpoint_write carries), sent with the pair's Send loop, read with the endpoint-read loop,
  deframed by the oracle's HTTP/2 parser, echoed back the same way on the other direction of the
  connection.  Reference-default knobs (4 MiB ring, max_sge 30).
* CPU: 100 of the same RPCs through a real gRPC stack over loop-back TCP (the grpcio wheel of the
  image: what GRPC_PLATFORM_TYPE=TCP runs) -- the "reference TCP endpoint on the same inputs".
* GPU (`-m gpu`): the same 1000 RPCs through the HIP pair (device rings, kernels for every byte),
  deframed by the device parser.
"""
import random

import pytest

from oracle import pyorc
from h2_helpers import PREFACE, frame, messages_of

MAX_SIZE = (4 << 20) - 1024


def rpc_sizes(n, seed=0):
    rng = random.Random(seed)
    return [rng.randint(1, MAX_SIZE) for _ in range(n)]


def payload_of(i, n):
    # cheap, position dependent, different per RPC
    blk = bytes((j * 31 + i * 7 + 3) % 251 for j in range(4096))
    return (blk * (n // 4096 + 1))[:n]


def framed_call(sid, body, with_headers=True):
    """-> list of slices of one unary call on stream sid: HEADERS(END_HEADERS), then the DATA frames
    of the message as grpc_chttp2_encode_data cuts them, END_STREAM on the last one"""
    wire, lens = pyorc.h2_frame_message(body, stream_id=sid, end_stream=1)
    out, off = ([frame(1, 4, sid, b"\x82\x86")] if with_headers else []), 0
    for n in lens:
        out.append(wire[off:off + n])
        off += n
    return out


class OracleConn:
    """One direction pair of an oracle link with an HTTP/2 parser behind each reader."""

    def __init__(self, ring, sge):
        self.o = pyorc.OracleLink(ring, sge)
        # side 1 = server (accepts streams from HEADERS), side 0 = client (opens them itself)
        self.parser = {1: pyorc.H2Parser(expect_client_prefix=True), 0: pyorc.H2Parser(expect_client_prefix=False)}

    def transfer(self, src, slices):
        """write all slices from side `src`, draining at the other side whenever the ring is full;
        -> the messages the receiving parser completed"""
        dst = 1 - src
        events, data = [], bytearray()
        pending, byte_idx = list(slices), 0
        while pending:
            sent = self.o.send(src, pending[:4000], byte_idx)
            # rdma_flush cursor walk (rdma_bp_posix.cc:480-493)
            while sent > 0:
                left = len(pending[0]) - byte_idx
                if sent >= left:
                    sent -= left
                    pending.pop(0)
                    byte_idx = 0
                else:
                    byte_idx += sent
                    sent = 0
            while True:  # rdma_do_read until it would block
                s, _ = self.o.endpoint_read(dst)
                if not s:
                    break
                base = len(data)
                rc, ev = self.parser[dst].feed(s)
                assert rc == 0
                events += [(k, a + base if k == pyorc.EV_MSG_BYTES else a, b, c, d) for k, a, b, c, d in ev]
                data += s
        return messages_of(events, bytes(data))


def test_1000_seeded_echo_rpcs_through_the_oracle_connection():
    sizes = rpc_sizes(1000)
    conn = OracleConn(4 << 20, 30)
    total = 0
    assert conn.transfer(0, [PREFACE + frame(4, 0, 0)]) == []   # connection preface + SETTINGS
    for i, n in enumerate(sizes):
        sid = 2 * i + 1
        req = payload_of(i, n)
        assert conn.parser[0].open_stream(sid) == 0   # the client starts the call
        got = conn.transfer(0, framed_call(sid, req))
        assert got == [(sid, req)], "request %d (%d bytes) differs at the server" % (i, n)
        # the server echoes on the same stream, END_STREAM closes it on both sides
        back = conn.transfer(1, framed_call(sid, got[0][1]))
        assert back == [(sid, req)], "response %d differs at the client" % i
        conn.parser[1].close_writes(sid)
        conn.parser[0].close_writes(sid)
        total += n
    assert conn.parser[0].live_streams() == 0 and conn.parser[1].live_streams() == 0
    st = conn.o.state(0), conn.o.state(1)
    assert st[0]["remain"] == 0 and st[1]["remain"] == 0
    assert conn.o.ring_mem(0) == bytes(4 << 20) and conn.o.ring_mem(1) == bytes(4 << 20)
    assert total == sum(sizes)


def test_seeded_echo_rpcs_over_loopback_tcp_with_grpcio():
    """The same requests through a stock gRPC stack over loop-back TCP: byte-identical echo."""
    grpc = pytest.importorskip("grpc")
    from concurrent import futures
    ident = lambda b: b  # noqa: E731
    handler = grpc.method_handlers_generic_handler("helloworld.Greeter", {
        "SayHello": grpc.unary_unary_rpc_method_handler(lambda req, ctx: req, ident, ident)})
    opts = [("grpc.max_receive_message_length", -1), ("grpc.max_send_message_length", -1)]
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=2), options=opts)
    server.add_generic_rpc_handlers((handler,))
    port = server.add_insecure_port("127.0.0.1:0")
    server.start()
    try:
        ch = grpc.insecure_channel("127.0.0.1:%d" % port, options=opts)
        try:
            grpc.channel_ready_future(ch).result(timeout=20)
        except Exception:
            pytest.skip("no loop-back networking in this sandbox")
        call = ch.unary_unary("/helloworld.Greeter/SayHello", request_serializer=ident, response_deserializer=ident)
        for i, n in enumerate(rpc_sizes(100)):
            req = payload_of(i, n)
            assert call(req) == req
        ch.close()
    finally:
        server.stop(0)


@pytest.mark.gpu
def test_1000_seeded_echo_rpcs_through_the_hip_pair(gpu):
    """The same RPCs through device rings: endpoint writes from host slices, endpoint reads,
    device-side deframing (k_h2_deframe with its stream map), echo, byte-identical at the client;
    rings zero and stream maps empty at the end."""
    g = gpu
    from grpc_rdma_amd import h2dev
    sizes = rpc_sizes(1000)
    R = 4 << 20
    a, b = g.Pair(R, 30), g.Pair(R, 30)      # a = client end, b = server end
    g.connect_pairs(a, b)
    parser = {"srv": h2dev.Parser(True, max_concurrent_streams=100, table_slots=256),
              "cli": h2dev.Parser(False, table_slots=256)}

    def transfer(tx, rx, who, slices):
        evs, data = [], bytearray()
        assert len(slices) <= 4000
        _steps, done = tx.endpoint_write(slices)
        while True:
            got, _wb = rx.endpoint_read(max_reads=512)
            if got:
                # one deframe call over the delivered slices of this pass
                table, blob = [], bytearray()
                for s in got:
                    table.append((len(blob), len(s)))
                    blob += s + bytes((-len(s)) % 16)
                buf = g.DeviceBuffer(data=bytes(blob) + bytes(64))
                base = len(data)
                err, ev = parser[who].deframe(buf.ptr, table)
                assert err == 0
                offs = []
                o_ = 0
                for s in got:
                    offs.append(base + o_)
                    o_ += len(s)
                evs += [(k, a_ + offs[sl] if k == pyorc.EV_MSG_BYTES else a_, b_, c_, d_) for k, a_, b_, c_, d_, sl in ev]
                for s in got:
                    data += s
                buf.free()
            if done and not got:
                break
            if not done:
                _steps, done = tx.endpoint_write_continue()
        return messages_of(evs, bytes(data))

    assert transfer(a, b, "srv", [PREFACE + frame(4, 0, 0)]) == []
    for i, n in enumerate(sizes):
        sid = 2 * i + 1
        req = payload_of(i, n)
        assert parser["cli"].open_streams([sid]) == 0
        got = transfer(a, b, "srv", framed_call(sid, req))
        assert got == [(sid, req)], "request %d (%d bytes) differs at the server" % (i, n)
        back = transfer(b, a, "cli", framed_call(sid, got[0][1]))
        assert back == [(sid, req)], "response %d differs at the client" % i
        parser["srv"].close_writes([sid])
        parser["cli"].close_writes([sid])
    assert parser["srv"].live_streams() == 0 and parser["cli"].live_streams() == 0
    assert a.ring_mem() == bytes(R) and b.ring_mem() == bytes(R)
    a.close()
    b.close()
