#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Captures what a REAL gRPC client puts on the wire.

A stock gRPC C-core (the grpcio wheel of the image; chttp2 is the transport the reference forks)
connects to a raw TCP socket opened by this script.  The script speaks just enough HTTP/2 for the
channel to become READY (its own SETTINGS, an ACK of the client's), records every byte the client
sends -- connection preface, SETTINGS, WINDOW_UPDATE / PING, HPACK HEADERS, and the DATA frames of a
few unary requests and one client-streaming call with known payloads -- and never answers the
calls (they end by deadline).  The capture and the payloads that went in are written to
tests/golden/h2_grpcio_capture.json; tests/test_h2_oracle.py and tests/test_gpu_h2.py feed the
bytes, whole and cut at arbitrary points, to the CPU oracle's deframer and to k_h2_deframe and
require exactly those payloads back.  This pins the deframer against bytes produced
by real chttp2 code instead of hand-written expectations.

Run in the build container:  python oracle/gen_h2_grpcio_capture.py
"""
import json
import os
import socket
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "h2_grpcio_capture.json")


def payloads():
    out = [bytes([0x0A, 64]) + bytes(range(64)),                       # SimpleRequest{bytes message = 64 B}
           b"",                                                         # an empty message
           bytes((i * 7 + 1) % 251 for i in range(40000)),             # spans three DATA frames
           bytes((i * 13 + 5) % 251 for i in range(16379))]            # exactly fills one frame with its 5-byte header
    stream = [bytes((i * 3 + k) % 251 for i in range(n)) for k, n in enumerate((1, 70000, 5, 16384, 30000))]
    return out, stream


def main():
    import grpc
    unary, stream = payloads()
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    port = srv.getsockname()[1]
    captured = bytearray()
    stop = threading.Event()

    def serve():
        conn, _ = srv.accept()
        conn.settimeout(0.2)
        # our SETTINGS: a large initial window so that the client never waits for WINDOW_UPDATEs
        # (SETTINGS_INITIAL_WINDOW_SIZE = 0x4, 2^30), then an ACK of the client's SETTINGS, and a
        # connection-level WINDOW_UPDATE
        conn.sendall(bytes([0, 0, 6, 4, 0, 0, 0, 0, 0]) + bytes([0, 4]) + (1 << 30).to_bytes(4, "big"))
        conn.sendall(bytes([0, 0, 0, 4, 1, 0, 0, 0, 0]))
        conn.sendall(bytes([0, 0, 4, 8, 0, 0, 0, 0, 0]) + ((1 << 30) - 70000).to_bytes(4, "big"))
        while not stop.is_set():
            try:
                d = conn.recv(1 << 16)
            except socket.timeout:
                continue
            except OSError:
                break
            if not d:
                break
            captured.extend(d)
            # answer PINGs (type 6, no ACK flag) so that keepalive / BDP probes do not stall the client
            # (a scan over the fresh bytes is enough for this capture: PING frames arrive whole)
            i = 0
            while i + 9 <= len(d):
                if d[i:i + 3] == b"\x00\x00\x08" and d[i + 3] == 6 and d[i + 4] == 0 and d[i + 5:i + 9] == b"\x00\x00\x00\x00" and i + 17 <= len(d):
                    try:
                        conn.sendall(bytes([0, 0, 8, 6, 1, 0, 0, 0, 0]) + d[i + 9:i + 17])
                    except OSError:
                        pass
                    i += 17
                else:
                    i += 1
        conn.close()

    th = threading.Thread(target=serve, daemon=True)
    th.start()
    ident = lambda b: b  # noqa: E731
    ch = grpc.insecure_channel("127.0.0.1:%d" % port, options=[("grpc.max_send_message_length", -1)])
    grpc.channel_ready_future(ch).result(timeout=20)
    call = ch.unary_unary("/mb.BenchmarkService/Unary", request_serializer=ident, response_deserializer=ident)
    for p in unary:
        try:
            call(p, timeout=0.5)
        except grpc.RpcError:
            pass  # nobody answers: DEADLINE_EXCEEDED after the request went out
    scall = ch.stream_unary("/mb.BenchmarkService/ClientStream", request_serializer=ident, response_deserializer=ident)
    try:
        scall(iter(stream), timeout=1.5)
    except grpc.RpcError:
        pass
    time.sleep(0.5)
    ch.close()
    stop.set()
    th.join(timeout=5)
    srv.close()
    doc = {"source": "grpcio %s client -> raw socket, oracle/gen_h2_grpcio_capture.py" % grpc.__version__,
           "payloads": "payloads() of the generating script: four unary requests, then the five messages of one client-streaming call",
           "payload_lengths": [len(p) for p in unary + stream],
           "client_bytes_hex": bytes(captured).hex()}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(doc, f)
    print("captured %d bytes -> %s" % (len(captured), OUT))


if __name__ == "__main__":
    main()
