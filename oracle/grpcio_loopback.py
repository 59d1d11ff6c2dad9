#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product).

gRPC over loop-back TCP on this host, measured with the grpcio wheel of the image (gRPC C-core
with its stock TCP endpoint: the transport the reference's GRPC_PLATFORM_TYPE=TCP mode uses,
tcp_posix.cc), in the two shapes of the reference's micro-benchmark
(examples/cpp/micro_benchmark/mb_client.cc: client-streaming of fixed-size messages; unary
ping-pong):

    python oracle/grpcio_loopback.py stream <seconds> <payload_bytes>
    python oracle/grpcio_loopback.py unary  <seconds> <payload_bytes>

Messages are SimpleRequest{bytes message}-shaped byte strings passed through identity
serializers (no protobuf work on either side).  The Python binding costs CPU that the
reference's C++ client does not pay: read the figure as "a real gRPC TCP stack on these cores",
next to oracle/tcp_floor.c (raw sendmsg/recvmsg, the upper bound for any TCP transport here).
Prints one JSON line.
"""
import json
import os
import sys
import time
from concurrent import futures


def main():
    import grpc
    mode, seconds, payload = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    ident = lambda b: b  # noqa: E731

    def client_stream(request_iterator, context):
        n = 0
        for m in request_iterator:
            n += len(m)
        return n.to_bytes(8, "little")

    def unary(request, context):
        return request

    handler = grpc.method_handlers_generic_handler("mb.BenchmarkService", {
        "ClientStream": grpc.stream_unary_rpc_method_handler(client_stream, ident, ident),
        "Unary": grpc.unary_unary_rpc_method_handler(unary, ident, ident)})
    opts = [("grpc.max_receive_message_length", -1), ("grpc.max_send_message_length", -1)]
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=2), options=opts)
    server.add_generic_rpc_handlers((handler,))
    port = server.add_insecure_port("127.0.0.1:0")
    server.start()
    ch = grpc.insecure_channel("127.0.0.1:%d" % port, options=opts)
    grpc.channel_ready_future(ch).result(timeout=20)
    msg = bytes([0x0A]) + bytes((i * 7 + 3) % 251 for i in range(min(payload, 4096))) * (payload // 4096 + 1)
    msg = msg[:payload]
    out = {"mode": mode, "payload": payload, "grpcio": grpc.__version__, "nproc": os.cpu_count()}
    if mode == "stream":
        call = ch.stream_unary("/mb.BenchmarkService/ClientStream", request_serializer=ident, response_deserializer=ident)
        sent = [0]
        t_end = [0.0]

        def gen():
            stop = time.perf_counter() + seconds
            while time.perf_counter() < stop:
                sent[0] += 1
                yield msg
        t0 = time.perf_counter()
        got = int.from_bytes(call(gen()), "little")
        sec = time.perf_counter() - t0
        assert got == sent[0] * payload
        out.update({"msgs": sent[0], "seconds": round(sec, 3), "GiBps": round(got / sec / (1 << 30), 4)})
    else:
        call = ch.unary_unary("/mb.BenchmarkService/Unary", request_serializer=ident, response_deserializer=ident)
        for _ in range(200):
            call(msg)
        rtt = []
        stop = time.perf_counter() + seconds
        while time.perf_counter() < stop:
            t0 = time.perf_counter()
            r = call(msg)
            rtt.append(time.perf_counter() - t0)
        assert r == msg
        rtt.sort()
        n = len(rtt)
        out.update({"iters": n, "p50_us": round(1e6 * rtt[n // 2], 2), "p95_us": round(1e6 * rtt[int(n * .95)], 2),
                    "p99_us": round(1e6 * rtt[int(n * .99)], 2)})
    ch.close()
    server.stop(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
