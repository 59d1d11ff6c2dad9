"""A remote writer lands records in the ring WHILE the receiver is looking at it.

On the reference's wire a record arrives as one RDMA WRITE whose bytes become visible in address
order: the 8-byte length first, then the payload, the padding, and the 0xFF.. footer last
(ring_buffer.h:84-99).  The receiver may look at any moment in between; GetReadableSize /
HasMessage (ring_buffer.cc:48-97) and Read must treat a record whose footer has not landed as
"nothing there".  Here a second host thread plays the NIC: it writes every record into the
receiver's ring in that order, in small pieces with pauses, through plain device copies, while
the main thread keeps calling the message-ready tests (single pair and the batched k_poll) and
endpoint reads.  Whatever the interleaving: only whole records are ever delivered, in order, the
ring is zero behind them, and the head at the end is where the CPU oracle's arithmetic puts it.
"""
import ctypes as C
import random
import struct
import threading
import time

import pytest

from oracle import pyorc

pytestmark = pytest.mark.gpu

FOOTER = struct.pack("<Q", 0xFFFFFFFFFFFFFFFF)


def encode(payload):
    pad = (-len(payload)) % 8
    return struct.pack("<Q", len(payload)) + payload + bytes(pad) + FOOTER


class RingWriter(threading.Thread):
    """Writes encoded records at the ring's tail, header first / footer last, in pieces."""

    def __init__(self, g, pair, ring_size, payloads, consumed, seed):
        super().__init__(daemon=True)
        self.lib = g.load()
        self.base = self.lib.grdma_pair_ring_device_ptr(pair.h)
        self.R, self.payloads, self.consumed = ring_size, payloads, consumed
        self.rng = random.Random(seed)
        self.tail = 0           # ring offset of the next record
        self.written = 0        # encoded bytes written so far
        self.error = None

    def put(self, off, data):
        """bytes -> ring[off ...], split at the ring end"""
        off %= self.R
        first = min(len(data), self.R - off)
        for o, chunk in ((off, data[:first]), (0, data[first:])):
            if chunk:
                buf = C.create_string_buffer(chunk, len(chunk))
                rc = self.lib.grdma_copy_to_device(C.c_void_p(self.base + o), buf, len(chunk))
                if rc < 0:
                    raise RuntimeError("copy_to_device failed")

    def run(self):
        try:
            for p in self.payloads:
                enc = encode(p)
                body = len(enc) - 8                 # header + payload + padding (the footer word comes last)
                # never let unread bytes exceed half the ring (the reader clears what it consumed
                # before `consumed` moves, so the space in front of the tail is free AND zero)
                while self.written + len(enc) - self.consumed[0] > self.R // 2:
                    time.sleep(0.0002)
                pieces = []
                cuts = sorted({8, body} | {self.rng.randrange(8, body + 1) for _ in range(self.rng.randrange(0, 4))})
                prev = 0
                for c in cuts:
                    if c > prev:
                        pieces.append((prev, c))
                        prev = c
                for a, b in pieces:                 # the header goes first: (0, 8)
                    self.put(self.tail + a, enc[a:b])
                    if self.rng.random() < 0.5:
                        time.sleep(self.rng.random() * 0.0008)
                self.put(self.tail + body, FOOTER)  # the footer lands last
                self.tail = (self.tail + len(enc)) % self.R
                self.written += len(enc)
        except Exception as e:  # surfaced by the main thread
            self.error = e


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_concurrent_writer_never_yields_partial_records(gpu, seed):
    g = gpu
    R = 1 << 16
    rng = random.Random(seed)
    payloads = []
    for i in range(90):
        n = rng.choice([1, 7, 8, 9, 200, 255, 256, 257, 1000, 4096, 9000])
        payloads.append(bytes((i * 31 + j) % 251 for j in range(n)))
    # (the second thread plays a NIC: bytes in address order, the footer last, no arrival report -- GRDMA_WIRE_ORDERED)
    a, b = g.Pair(R, 30, flags=8), g.Pair(R, 30, flags=8)
    g.connect_pairs(a, b)
    consumed = [0]
    w = RingWriter(g, b, R, payloads, consumed, seed)
    expected = b"".join(payloads)
    got = bytearray()
    enc_prefix, e = {0: 0}, 0
    acc = 0
    for p in payloads:
        acc += len(p)
        e += 16 + len(p) + ((-len(p)) % 8)
        enc_prefix[acc] = e
    w.start()
    deadline = time.time() + 60
    polls = 0
    while len(got) < len(expected):
        assert time.time() < deadline, "reader starved"
        assert w.error is None, w.error
        polls += 1
        # the three message-ready tests the event engines use, at arbitrary moments
        has = b.HasMessage()
        readable = b.GetReadableSize()
        rd, hm = g.poll_pairs([b])
        if readable:
            # a non-zero readable size is the payload length of a COMPLETE first record
            nxt = expected[len(got):]
            assert readable <= len(nxt)
        slices, _wb = b.endpoint_read(max_reads=4)
        for s in slices:
            got += s
        assert bytes(got) == expected[:len(got)], "delivered bytes differ from what the writer sent"
        # everything delivered so far ends on a record boundary or inside a record whose footer
        # had landed; report the consumed encoded bytes of WHOLE records to the writer
        whole = max(k for k in enc_prefix if k <= len(got))
        consumed[0] = enc_prefix[whole]
        del has, rd, hm  # (their values race with the writer by design; they must only never fault)
    w.join(timeout=20)
    assert w.error is None, w.error
    assert bytes(got) == expected
    assert b.ring_mem() == bytes(R), "ring not zero behind the delivered records"
    # head bookkeeping after the same records (ring_buffer.cc:99-191): every encoded byte consumed
    sb = b.state()
    assert sb["head"] == e % R and sb["remain"] == 0, sb
    # and the CPU oracle fed the same ring image record by record delivers the same bytes
    o = pyorc.OracleLink(R, 30)
    stream = bytearray()
    for p in payloads:
        sent = 0
        while sent < len(p):
            n = o.send(0, [p[sent:]])
            sent += n
            while True:
                s_, _ = o.endpoint_read(1)
                if not s_:
                    break
                stream += s_
    assert bytes(stream) == expected
    assert polls > 0
    a.close()
    b.close()
