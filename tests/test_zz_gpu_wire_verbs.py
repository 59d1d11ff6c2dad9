"""The NIC wire back end as code (csrc/grdma_wire_verbs.cc): two pairs whose rings are written by an "HCA" -- the verbs
stand-in of oracle/fakeverbs, 21 verbs over process memory with the checks of a real one (regions, keys, queue-pair
states, completions) -- instead of by the loop-back copy kernel.  Each pair registers its ring (through the dma-buf
call) and status block as remote-writable, staging buffer and status_send as local regions, and brings its queue pair
to RTS against the peer's address; a Send's <= 2 chained RDMA WRITEs are the {ring offset, length} pairs the device's
send planner left in the result block (K2: GetWriteRequests, ring_buffer.cc:261-330; pair.cc:709-734), the credit
report is the 16-byte status write of updateStatus (pair.cc:624-641), every completion is reaped.

Replayed through it: the endpoint traces the REFERENCE produced (tests/golden/ref_endpoint_*.json: rdma_bp_posix.cc +
pair.cc, unmodified) -- accepted bytes, delivered bytes (CRC), readable size and returned credit step by step --, and
after every Send the receiver's ring image equals the oracle's (pinned to the reference-built pair.cc).

Runs where the library carries the NIC wire: the emulated library of the CPU suite (tests/test_emu_gpu_suite.py), and
ON THE MI355X the product sources built for gfx950 with the same back end over the stand-in's fabric in its HIP form
(tests/cc/build_fakeverbs_hip.sh -> oracle/_build/libgrdma_amd_fakeverbs.so: an RDMA WRITE is a copy-engine copy into the
registered HBM ring in address order, the last eight bytes last) -- the product library itself is built without verbs
(no header, no HCA in these images), so each test re-runs itself in a child pytest against that variant: real k_tx_plan /
k_copy / k_rx_plan / k_rx_apply on both sides of the queue pair, the ring registered through its real dma-buf fd."""
import ctypes as C
import glob
import json
import os
import random
import zlib

import numpy as np
import pytest

from oracle import pyorc
from tests.test_gpu_pair_parity import _ring_eq   # (equal up to the pad bytes: the reference leaves stale staging bytes there)

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_endpoint_*.json")))


class VerbsAddress(C.Structure):
    _fields_ = [("qpn", C.c_uint32), ("psn", C.c_uint32), ("lid", C.c_uint16), ("pad0", C.c_uint16), ("ring_rkey", C.c_uint32),
                ("gid", C.c_uint8 * 16), ("ring_addr", C.c_uint64), ("ring_size", C.c_uint64), ("status_addr", C.c_uint64),
                ("status_rkey", C.c_uint32), ("status_size", C.c_uint32)]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "oracle", "_build", "libgrdma_amd_fakeverbs.so")


def wire_here_or_in_the_variant(lib, request):
    """True: this process's library has the NIC wire, go on.  False: the test has just passed in a child pytest that
    loaded the gfx950 build with the wire over the HIP fabric.  Skips only where neither exists."""
    lib.grdma_verbs_supported.restype = C.c_int
    if lib.grdma_verbs_supported():
        return True
    if os.environ.get("GRDMA_LIB_PATH") or not os.path.exists(VARIANT):
        pytest.skip("this library was built without the NIC wire and oracle/_build/libgrdma_amd_fakeverbs.so is absent")
    import subprocess
    import sys
    env = dict(os.environ, GRDMA_LIB_PATH=VARIANT, GRDMA_TEST_ALLOW_EMU="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", request.node.nodeid],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    import re
    assert p.returncode == 0 and re.search(r"(^|\s)1 passed", p.stdout), (p.stdout + p.stderr)[-3000:]
    return False


def pattern(seed, i, n):
    j = np.arange(n, dtype=np.uint64)
    return ((seed * 131 + i * 17 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8).tobytes()


def verbs_link(g, lib, R, sge):
    a, b = g.Pair(R, sge, 8), g.Pair(R, sge, 8)          # GRDMA_WIRE_ORDERED: a NIC writes these rings
    lib.grdma_pair_verbs_open.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.grdma_pair_verbs_address.argtypes = [C.c_void_p, C.POINTER(VerbsAddress)]
    lib.grdma_pair_verbs_connect.argtypes = [C.c_void_p, C.POINTER(VerbsAddress)]
    lib.grdma_pair_verbs_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    for p in (a, b):
        g._lib.check(lib.grdma_pair_verbs_open(p.h, None, 1, 0))
    aa, ab = VerbsAddress(), VerbsAddress()
    g._lib.check(lib.grdma_pair_verbs_address(a.h, C.byref(aa)))
    g._lib.check(lib.grdma_pair_verbs_address(b.h, C.byref(ab)))
    assert aa.qpn != ab.qpn and aa.ring_rkey != ab.ring_rkey and aa.ring_size == R
    g._lib.check(lib.grdma_pair_verbs_connect(a.h, C.byref(ab)))
    g._lib.check(lib.grdma_pair_verbs_connect(b.h, C.byref(aa)))
    assert a.get_status() == 2 and b.get_status() == 2
    return a, b


def writable(R, st):   # GetWritableSize(), pair.cc:294-301, from the connection block (credit lands there by DMA)
    free = R - ((st["remote_tail"] + R - st["remote_head"]) & (R - 1))
    return free - 24 if free > 24 else 0


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[len("ref_endpoint_"):-5] for f in FILES])
def test_reference_made_endpoint_trace_over_the_verbs_wire(gpu, path, request):
    g = gpu
    lib = g.load()
    if not wire_here_or_in_the_variant(lib, request):
        return
    doc = json.load(open(path))
    R = doc["ring_kib"] * 1024
    a, b = verbs_link(g, lib, R, doc["max_sge"])
    o = pyorc.OracleLink(R, doc["max_sge"])
    rng = random.Random(3)
    sends = wraps = credits = 0
    try:
        for k, (op, want) in enumerate(zip(doc["ops"], doc["results"])):
            if op[0] == "S":
                _, bi, seed, lens = op
                sl = [pattern(seed, i, n) for i, n in enumerate(lens)]
                bufs = [g.DeviceBuffer(data=s, offset=rng.randrange(16)) for s in sl]
                got = [a.Send(bufs, bi)]
                assert o.send(0, sl, bi) == got[0]
                wrs = a.last_wrs()
                assert wrs == o.last_wrs(0), "step %d: the write requests differ from GetWriteRequests'" % k
                sends += 1 if got[0] else 0
                wraps += 1 if len(wrs) == 2 else 0
                assert _ring_eq(b.ring_mem(), o.ring_mem(1)), "step %d: ring image after the RDMA WRITEs" % k
            else:
                slices, _wb = b.endpoint_read(1)
                data = slices[0] if slices else b""
                od, _alloc = o.endpoint_read(1)
                assert (od or b"") == data
                got = [len(data) if data else -1, (zlib.crc32(data) & 0xFFFFFFFF) if data else 0,
                       b.GetReadableSize(), writable(R, a.state())]
                assert _ring_eq(b.ring_mem(), o.ring_mem(1)), "step %d: ring image after the read" % k
            assert got == want, "step %d %r" % (k, op[:3])
        cnt = (C.c_uint64 * 3)()
        lib.grdma_pair_verbs_counts(a.h, cnt)
        assert cnt[0] == sends + wraps and cnt[2] == cnt[0] + cnt[1], list(cnt)       # every posted write was reaped
        lib.grdma_pair_verbs_counts(b.h, cnt)
        credits = b.state()["credit_msgs"]
        assert cnt[1] == credits and cnt[0] == 0, (list(cnt), credits)                 # one status write per credit report
        sa, sb = a.state(), b.state()
        for key in ("remote_tail", "remote_head", "partial_write"):
            assert sa[key] == o.state(0)[key], key
        for key in ("head", "moving_head", "remain", "internal_read_size", "credit_msgs"):
            assert sb[key] == o.state(1)[key], key
    finally:
        a.close()
        b.close()
        o.close()


def test_verbs_wire_checks(gpu, request):
    """What Init() / Connect() refuse: a ring that is not marked NIC-written, a peer with another ring size; and
    Disconnect() tells the peer through the status write (pair.cc:332-336 => kHalfClosed, :349-356)."""
    g = gpu
    lib = g.load()
    if not wire_here_or_in_the_variant(lib, request):
        return
    lib.grdma_pair_verbs_open.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    plain = g.Pair(1 << 16, 30, 0)
    assert lib.grdma_pair_verbs_open(plain.h, None, 1, 0) < 0 and b"GRDMA_WIRE_ORDERED" in lib.grdma_last_error()
    plain.close()
    a, b = verbs_link(g, lib, 1 << 16, 30)
    c = g.Pair(1 << 17, 30, 8)
    g._lib.check(lib.grdma_pair_verbs_open(c.h, None, 1, 0))
    ac = VerbsAddress()
    lib.grdma_pair_verbs_address(a.h, C.byref(ac))
    assert lib.grdma_pair_verbs_connect(c.h, C.byref(ac)) < 0 and b"ring sizes differ" in lib.grdma_last_error()
    msg = [b"over the wire", b"!" * 300]
    bufs = [g.DeviceBuffer(data=m) for m in msg]
    assert a.Send(bufs) == 313
    got, _ = b.endpoint_read(4)
    assert b"".join(got) == b"".join(msg)
    # what has no way to reach the queue pair is refused, not silently accounted (the engine's commands, asynchronous
    # launch chains and device-resident jobs write through a peer-ring pointer this wire does not have)
    lib.grdma_pair_set_latency_mode.argtypes = [C.c_void_p, C.c_int]
    lib.grdma_endpoint_set_async.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
    lib.grdma_pair_arm_read.argtypes = [C.c_void_p, C.c_uint64]
    assert lib.grdma_pair_set_latency_mode(a.h, 1) < 0 and b"NIC wire" in lib.grdma_last_error()
    assert lib.grdma_endpoint_set_async(a.h, 0, 0) < 0 and b"NIC wire" in lib.grdma_last_error()
    assert lib.grdma_pair_arm_read(b.h, 4) < 0 and b"NIC wire" in lib.grdma_last_error()
    from importlib import import_module
    stream = import_module(g.Pair.__module__.rsplit(".", 1)[0] + ".stream")
    dst = g.DeviceBuffer(1 << 16)
    with pytest.raises(Exception, match="NIC wire"):
        stream.StreamJob(a, b, [(x.ptr, x.nbytes) for x in bufs], dst.ptr, 1 << 16, 64, 4)
    a.Disconnect()
    import time
    t0 = time.time()
    while b.get_status() != 3:
        assert time.time() - t0 < 10, "the peer never saw the disconnect"
        time.sleep(0.01)
    a.close(); b.close(); c.close()


def test_a_failed_write_gives_the_wire_up_at_once(gpu, request):
    """A work completion in error (here: the fabric refuses the write, as a peer that has gone does): the Send reports it
    without waiting ten seconds for completions that will never come, nothing stays counted as pending, and the next
    Send is refused immediately -- the queue pair is in the error state (pair.cc:500-558 turns this into kError)."""
    g = gpu
    lib = g.load()
    if not wire_here_or_in_the_variant(lib, request):
        return
    import time
    a, b = verbs_link(g, lib, 1 << 16, 30)
    bufs = [g.DeviceBuffer(data=b"x" * 100)]
    assert a.Send(bufs) == 100
    lib.fakeverbs_fail_next_writes.argtypes = [C.c_int]
    lib.fakeverbs_fail_next_writes(1)
    t0 = time.time()
    with pytest.raises(Exception, match="work completion"):
        a.Send(bufs)
    with pytest.raises(Exception, match="error state"):
        a.Send(bufs)
    assert time.time() - t0 < 5
    cnt = (C.c_uint64 * 3)()
    lib.grdma_pair_verbs_counts(a.h, cnt)
    assert cnt[0] == 2 and cnt[2] == 1, list(cnt)     # two posted, the good one reaped; the failed one is not waited for
    a.close(); b.close()
