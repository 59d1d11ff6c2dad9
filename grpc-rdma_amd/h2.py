"""Host-side mirror of the chttp2 DATA framing that produces the slice list an
endpoint_write receives (src/core/ext/transport/chttp2/transport/frame_data.cc:64-90
grpc_chttp2_encode_data, chttp2_transport.cc:1502-1510 message header,
slice_buffer.cc:136-171 inlined-slice merging, :270-313 move_first_no_ref).

Pure layout arithmetic (no payload bytes are touched here): used to lay a
workload out in HBM; the byte work happens in the HIP kernels."""

GRPC_HEADER_SIZE = 5          # internal.h:778
FRAME_HEADER_SIZE = 9
SLICE_INLINED_SIZE = 23       # include/grpc/impl/codegen/slice.h:47-48 (LP64)
DEFAULT_MAX_FRAME_SIZE = 16384  # http2_settings.cc:56


def grpc_message_header(length, compressed=False):
    return bytes([1 if compressed else 0]) + length.to_bytes(4, "big")


def data_frame_header(length, stream_id, end_stream=False):
    assert length < (1 << 24)
    return length.to_bytes(3, "big") + bytes([0, 1 if end_stream else 0]) + \
        (stream_id & 0x7FFFFFFF).to_bytes(4, "big")


class _SliceBuffer:
    """(kind, payload) entries; kind 'inl' carries bytes, 'ref' carries (offset, length)
    into the message."""

    def __init__(self):
        self.items = []

    def add_indexed(self, item):
        self.items.append(item)

    def add(self, item):
        # grpc_slice_buffer_add: two consecutive inlined slices are concatenated
        if item[0] == "inl" and self.items and self.items[-1][0] == "inl" and \
                len(self.items[-1][1]) < SLICE_INLINED_SIZE:
            back = self.items[-1][1]
            if len(back) + len(item[1]) <= SLICE_INLINED_SIZE:
                self.items[-1] = ("inl", back + item[1])
            else:
                cp1 = SLICE_INLINED_SIZE - len(back)
                self.items[-1] = ("inl", back + item[1][:cp1])
                self.items.append(("inl", item[1][cp1:]))
            return
        self.items.append(item)


def _ilen(item):
    return len(item[1]) if item[0] == "inl" else item[1][1]


def frame_message(msg_len, stream_id=1, max_frame=DEFAULT_MAX_FRAME_SIZE, compressed=False,
                  end_stream=False):
    """-> list of slices for ONE message sent alone on `stream_id`:
    ('inl', bytes) for inlined header slices, ('ref', (offset, length)) for
    sub-slices of the message payload."""
    fcb = [("inl", grpc_message_header(msg_len, compressed))]
    if msg_len:
        fcb.append(("ref", (0, msg_len)))
    fcb_len = GRPC_HEADER_SIZE + msg_len
    out = _SliceBuffer()
    while fcb_len > 0:
        send = min(fcb_len, max_frame)
        is_last = end_stream and send == fcb_len
        out.add(("inl", data_frame_header(send, stream_id, is_last)))
        n = send
        if fcb_len == n:            # grpc_slice_buffer_move_into
            for it in fcb:
                out.add(it)
            fcb = []
        else:
            while fcb:
                it = fcb[0]
                sl = _ilen(it)
                if n > sl:
                    out.add(it); n -= sl; fcb.pop(0)
                elif n == sl:
                    out.add(it); fcb.pop(0)
                    break
                else:               # split: the head is appended un-merged
                    if it[0] == "inl":
                        out.add_indexed(("inl", it[1][:n]))
                        fcb[0] = ("inl", it[1][n:])
                    else:
                        off, ln = it[1]
                        out.add_indexed(("ref", (off, n)))
                        fcb[0] = ("ref", (off + n, ln - n))
                    break
        fcb_len -= send
    return out.items


def ring_bytes_for(slice_lens):
    """Σ (16 + round_up8(n)): encoded size E of a slice list on the ring."""
    return sum(16 + ((n + 7) & ~7) for n in slice_lens)
