"""Byte-level builders shared by the HTTP/2 deframing tests (CPU oracle and GPU parity)."""
from oracle.pyorc import EV_MSG_BEGIN, EV_MSG_BYTES, EV_MSG_END

PREFACE = b"PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n"


def frame(ftype, flags, sid, payload=b""):
    return len(payload).to_bytes(3, "big") + bytes([ftype, flags]) + sid.to_bytes(4, "big") + payload


def grpc_msg(body, compressed=0):
    return bytes([compressed]) + len(body).to_bytes(4, "big") + body


def unary_call(sid, body, hdr=b"\x82\x86"):
    """HEADERS(END_HEADERS) + DATA(END_STREAM) of one unary request on stream sid."""
    return frame(1, 4, sid, hdr) + frame(0, 1, sid, grpc_msg(body))


def messages_of(events, data):
    out, cur = [], {}
    for k, a, b, c, d in events:
        if k == EV_MSG_BEGIN:
            cur[c] = bytearray()
        elif k == EV_MSG_BYTES:
            cur[c] += data[a:a + b]
        elif k == EV_MSG_END:
            out.append((c, bytes(cur.pop(c))))
    return out
