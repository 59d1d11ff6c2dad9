"""The deframer's message-boundary step (grpc-rdma_amd/csrc/grdma_h2_fast.h) on the CPU: the two
functions k_h2_deframe calls, compiled for the host and run inside the oracle's parser
(tests/cc/h2_fast_host.cc), must give the oracle's events on every stream shape -- the ones the step
is made for (sender-side and receiver-side message starts) and the ones it must leave alone."""
import ctypes as C
import os
import random
import subprocess

import pytest

from tests.h2_helpers import PREFACE, frame, grpc_msg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libh2fast_host.so")
SRCS = [os.path.join(ROOT, "tests", "cc", "h2_fast_host.cc"),
        os.path.join(ROOT, "grpc-rdma_amd", "csrc", "grdma_h2_fast.h"),
        os.path.join(ROOT, "oracle", "grdma_oracle.c"), os.path.join(ROOT, "oracle", "grdma_oracle.h")]


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS):
        obj = os.path.join(BUILD, "h2fast_oracle.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-c", SRCS[2], "-o", obj])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fPIC", "-shared", SRCS[0], obj,
                               "-o", SO])
    L = C.CDLL(SO)
    L.h2fast_hybrid_parse.restype = C.c_int
    L.h2fast_hybrid_parse.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64,
                                      C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.c_int,
                                      C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]
    return L


def parse(L, chunks, use_step, server=True, streams=(), max_frame=16384):
    data = b"".join(chunks)
    lens = (C.c_uint64 * max(1, len(chunks)))(*[len(c) for c in chunks])
    cap = 8 * len(chunks) + len(data) // 4 + 64
    ev = (C.c_uint32 * (6 * cap))()
    nev, steps = C.c_uint64(0), C.c_uint64(0)
    ids = (C.c_uint32 * max(1, len(streams)))(*streams)
    rc = L.h2fast_hybrid_parse(3 if server else 0, max_frame, 0xffffffff, ids, len(streams), data, lens, len(chunks),
                               1 if use_step else 0, ev, cap, C.byref(nev), C.byref(steps))
    out = [tuple(ev[6 * i:6 * i + 6]) for i in range(nev.value)]
    return rc, out, steps.value


def sender_slices(sizes, sid=1, end_stream=True, seed=0):
    """Slices as chttp2 hands them to the endpoint: the message header rides in the inlined slice of the
    first frame header (grpc_slice_buffer_add merge), payload slices by reference."""
    out = []
    for i, n in enumerate(sizes):
        body = grpc_msg(bytes((j * 13 + i + seed) % 251 for j in range(n)))
        off = 0
        while off < len(body):
            k = min(16384, len(body) - off)
            last = end_stream and i == len(sizes) - 1 and off + k == len(body)
            fh = k.to_bytes(3, "big") + bytes([0, 1 if last else 0]) + sid.to_bytes(4, "big")
            if off == 0:
                out.append(fh + body[:5])
                if k > 5:
                    out.append(body[5:k])
            else:
                out += [fh, body[off:off + k]]
            off += k
    return out


def receiver_slices(tx, first=256):
    """What endpoint reads deliver for those records (rdma_bp_posix.cc:180-326): a read sized to
    max(256, first record) takes whole records and a piece of the next, the following read its rest."""
    out, cur, room = [], b"", first
    for rec in tx:
        while rec:
            if room == 0:
                out.append(cur)
                cur, room = b"", first
            if not cur and len(rec) > first:
                out.append(rec)  # a read sized to the record itself
                rec = b""
                continue
            take = rec[:room]
            cur += take
            room -= len(take)
            rec = rec[len(take):]
            if room == 0 and rec:
                out.append(cur)
                out.append(rec)  # the next read is sized to what is left of the record
                cur, room, rec = b"", first, b""
    if cur:
        out.append(cur)
    return out


PRE = [PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82")]


@pytest.mark.parametrize("shape", ["sender", "receiver"])
def test_boundary_step_matches_the_oracle_on_streaming_shapes(lib, shape):
    sizes = [1 << 20, 16384 * 3 - 5, 40000, 16384 - 5, 7, 16384 * 70 + 123, 1, 300000, 5, 2, 16379, 16380]
    tx = sender_slices(sizes)
    body = tx if shape == "sender" else receiver_slices(tx)
    assert b"".join(body) == b"".join(tx)
    chunks = PRE + body
    rc0, ev0, _ = parse(lib, chunks, False)
    rc1, ev1, steps = parse(lib, chunks, True)
    assert rc0 == 0 and rc1 == 0
    assert ev1 == ev0
    assert steps >= (len(sizes) // 2 if shape == "sender" else 4), "the step hardly ever matched: %d" % steps


def test_boundary_step_bench_shape_takes_every_message_start(lib):
    """256 x 1 MiB messages as the receiving side sees them: every message start but the first goes
    through the step (closing 5-byte frame + first frame in one 256-byte slice)."""
    n = 24
    tx = sender_slices([1 << 20] * n, end_stream=False)
    rx = receiver_slices(tx)
    chunks = [frame(1, 4, 1, b"\x82")] + rx
    rc0, ev0, _ = parse(lib, chunks, False, server=False, streams=(1,))
    rc1, ev1, steps = parse(lib, chunks, True, server=False, streams=(1,))
    assert rc0 == 0 and rc1 == 0 and ev1 == ev0
    assert steps == n
    # the merged slice really has the shape the step is made for
    assert any(len(s) == 256 and s[:3] == (5).to_bytes(3, "big") and s[14:17] == (16384).to_bytes(3, "big") for s in rx)


def test_boundary_step_random_cuts_and_streams(lib):
    """Random message sizes on three interleaved streams, slices cut at random places, unknown
    streams, END_STREAM and padding-free control frames in between: events equal the oracle's."""
    rng = random.Random(11)
    for trial in range(60):
        parts = [PREFACE + frame(4, 0, 0)]
        sids = [1, 3, 5]
        for sid in sids:
            parts.append(frame(1, 4, sid, b"\x82\x86"))
        per = {sid: sender_slices([rng.choice([1, 4, 5, 9, 100, 16379, 16384, 20000, 70000]) for _ in range(rng.randrange(1, 5))],
                                  sid=sid, end_stream=rng.random() < 0.5, seed=trial) for sid in sids}
        # interleave whole frames of the streams (a frame = a header slice + its payload slice(s))
        groups = {sid: [] for sid in sids}
        for sid, sl in per.items():
            i = 0
            while i < len(sl):
                fs = int.from_bytes(sl[i][:3], "big")
                have = len(sl[i]) - 9
                g = [sl[i]]
                i += 1
                while have < fs:
                    g.append(sl[i])
                    have += len(sl[i])
                    i += 1
                groups[sid].append(g)
        order = []
        while any(groups.values()):
            sid = rng.choice([s for s in sids if groups[s]])
            run = rng.randrange(1, 6) if trial % 2 else len(groups[sid])
            for _ in range(min(run, len(groups[sid]))):
                order += groups[sid].pop(0)
            if rng.random() < 0.2:
                order.append(frame(6, 0, 0, bytes(8)))       # PING
            if rng.random() < 0.1:
                order.append(frame(0, 0, 9, b"\0\0\0\0\1x"))  # DATA for a stream that is not in the map
        body = order if trial % 3 else receiver_slices(order)
        if trial % 4 == 3:  # cut some slices in two
            cut = []
            for s_ in body:
                if len(s_) > 2 and rng.random() < 0.3:
                    k = rng.randrange(1, len(s_))
                    cut += [s_[:k], s_[k:]]
                else:
                    cut.append(s_)
            body = cut
        chunks = parts + body
        rc0, ev0, _ = parse(lib, chunks, False)
        rc1, ev1, _ = parse(lib, chunks, True)
        assert rc0 == rc1 == 0, trial
        assert ev1 == ev0, trial


def test_boundary_step_leaves_the_odd_cases_alone(lib):
    """Flags, a frame that spans two messages, an empty message, a bad compression byte, a frame larger
    than the message, a slice that goes on behind the frame: no match, the automaton's events."""
    hdr = frame(1, 4, 1, b"\x82")
    cases = [
        [frame(0, 1, 1, grpc_msg(b"abc"))],                                      # END_STREAM
        [frame(0, 0, 1, grpc_msg(b"abc") + grpc_msg(b"defg"))],                  # two messages in one frame
        [frame(0, 0, 1, grpc_msg(b"")), frame(0, 0, 1, grpc_msg(b"xy"))],        # empty message
        [frame(0, 0, 1, b"\x02\0\0\0\1z")],                                      # "Bad GRPC frame type"
        [frame(0, 0, 1, grpc_msg(b"abc")) + frame(0, 0, 1, grpc_msg(b"q"))],     # slice goes on behind the frame
        [frame(0, 8, 1, grpc_msg(b"abc"))],                                      # PADDED flag: stream error
        [frame(0, 0, 1, grpc_msg(b"abcdefgh")[:9]), grpc_msg(b"abcdefgh")[9:]],  # frame header lies about the size
        [(20).to_bytes(3, "big") + b"\0\0" + (1).to_bytes(4, "big") + grpc_msg(b"0123456789")[:10], b"x" * 3],
    ]
    for body in cases:
        chunks = [PREFACE + frame(4, 0, 0), hdr] + body
        rc0, ev0, _ = parse(lib, chunks, False)
        rc1, ev1, _ = parse(lib, chunks, True)
        assert rc1 == rc0 and ev1 == ev0, body
