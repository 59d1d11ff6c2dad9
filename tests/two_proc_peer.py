#!/usr/bin/env python3
"""One end of a two-PROCESS connection (helper of tests/test_gpu_two_process.py).

usage: two_proc_peer.py <client|server> <fd> <hip_device> <ring_kib> <num_bytes> <write_size> <slice_size>

Bootstraps a pair over the inherited socket fd exactly like grpc_rdma_bp_create does
(exchange_data + Connect, rdma_bp_posix.cc:763-784), then runs the reference's endpoint
conformance shape (test/core/iomgr/endpoint_tests.cc:341-355: a stream of bytes i % 256 written
in `write_size` pieces of `slice_size` slices) client -> server, echoed back server -> client and
byte-checked on return.  Prints one line 'ok ...' and exits 0, or exits non-zero."""
import ctypes as C
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import grpc_rdma_amd as g  # noqa: E402
from grpc_rdma_amd._lib import ReadSlice, Slice, check  # noqa: E402


class Writer:
    """rdma_write + rdma_flush retries, non-blocking: one Send per step()."""

    def __init__(self, pair):
        self.pair, self.lib, self.active, self.keep = pair, pair.lib, False, None

    def begin(self, slices):
        arr, keep, _ = g.Pair._slices(slices)
        self.keep = (arr, keep)          # the pair reads these buffers until the write is done
        check(self.lib.grdma_endpoint_write_begin(self.pair.h, arr, len(slices), 1))
        self.active = True

    def step(self):
        done = C.c_int(0)
        n = check(self.lib.grdma_endpoint_write_step(self.pair.h, C.byref(done)))
        if done.value:
            self.active, self.keep = False, None
        return n


def on_the_clock(role, pair, w, seconds, ring_kib, write_size, slice_size):
    """The same echo for `seconds` of wall time instead of a byte count (small rings: a credit report crosses the
    process boundary every ring/2 bytes, each way).  The client checks every echoed byte as it arrives."""
    t_end = time.time() + seconds
    deadline = t_end + 120
    if role == "client":
        pos = got = nslices = 0
        while True:
            assert time.time() < deadline, "client timed out at %d/%d" % (got, pos)
            if not w.active and time.time() < t_end:
                data = bytes((pos + i) & 0xFF for i in range(write_size))
                w.begin([data[o:o + slice_size] for o in range(0, write_size, slice_size)])
                pos += write_size
            if w.active:
                w.step()
            sl, _ = pair.endpoint_read(256)
            nslices += len(sl)
            for s_ in sl:
                assert s_ == bytes((got + i) & 0xFF for i in range(len(s_))), "echo differs near byte %d" % got
                got += len(s_)
            if not w.active and time.time() >= t_end and got == pos:
                break
        assert pair.ring_mem() == bytes(ring_kib * 1024), "client ring not zero after the echo"
        st = pair.state()
        pair.Disconnect()
        print("ok client %d bytes echoed in %d slices, %d credit reports sent" % (got, nslices, st["credit_msgs"]))
    else:
        pending, echoed, seen = deque(), 0, 0
        while True:
            assert time.time() < deadline, "server timed out at %d" % echoed
            sl, _ = pair.endpoint_read(256)
            for s_ in sl:
                assert all(b == ((seen + i) & 0xFF) for i, b in enumerate(s_))
                seen += len(s_)
                pending.append(s_)
            if not w.active and pending:
                batch = [pending.popleft() for _ in range(min(len(pending), 2000))]
                w.batch_bytes = sum(len(b) for b in batch)
                w.begin(batch)
            if w.active:
                w.step()
                if not w.active:
                    echoed += w.batch_bytes
            elif not pending and not sl and pair.get_status() == 3:
                break
        assert pair.ring_mem() == bytes(ring_kib * 1024), "server ring not zero after the echo"
        print("ok server %d bytes echoed, %d credit reports sent" % (echoed, pair.state()["credit_msgs"]))
    pair.close()


def unary_pingpong(role, pair, iters, ring_kib):
    """Unary 64-byte ping-pong ACROSS the process boundary on the arrival-triggered path: each process runs its own
    latency engine; a read is a standing order (grdma_pair_arm_read) that a watcher workgroup of the READER's engine
    carries out when the bytes the OTHER process wrote land in its ring (k_watch) -- no command of the reading process,
    nothing both ends would have to share but the rings.  [14 B][66 B] each way, sizes and byte sums checked per end,
    the rings all zero afterwards; the client prints its round-trip percentiles."""
    lib = pair.lib
    pair.set_latency_mode(True)
    pair.arm_read(64)
    check(lib.grdma_engine_start())
    is_client = role == "pp_client"
    msg = [bytes((i * 7 + (3 if is_client else 5)) % 251 for i in range(14)), bytes((i * 11 + (1 if is_client else 2)) % 253 for i in range(66))]
    peer = [bytes((i * 7 + (5 if is_client else 3)) % 251 for i in range(14)), bytes((i * 11 + (2 if is_client else 1)) % 253 for i in range(66))]
    arr, keep, _ = g.Pair._slices(msg)
    warmup = max(10, iters // 10)
    rtt = (C.c_uint64 * iters)()
    bsum = C.c_uint64(0)
    try:
        check(lib.grdma_pingpong_end(pair.h, 1 if is_client else 0, arr, 2, 80, 1, iters, warmup, rtt, C.byref(bsum)))
        hits = pair.watch_hits()
        if is_client:
            pair.Disconnect()
        else:
            for _ in range(4000):
                if pair.get_status() == 3:
                    break
                time.sleep(0.005)
    finally:
        lib.grdma_engine_stop()
    expect = sum(sum(x) for x in peer) * (iters + warmup)
    assert bsum.value == expect, "byte sum of what arrived: %d, expected %d" % (bsum.value, expect)
    assert hits == iters + warmup, hits
    assert pair.ring_mem() == bytes(ring_kib * 1024), "ring not zero after the ping-pong"
    r = sorted(rtt)
    print("ok %s %d round trips, watch hits %d, rtt p50 %.2f us p95 %.2f us p99 %.2f us" % (
        "client" if is_client else "server", iters, hits, r[iters // 2] / 1e3, r[int(iters * .95)] / 1e3, r[int(iters * .99)] / 1e3))
    pair.close()


def main():
    role, fd, dev, ring_kib, num_bytes, write_size, slice_size = sys.argv[1], *map(int, sys.argv[2:8])
    g.init(dev)
    pair = g.Pair(ring_kib * 1024, 30, int(os.environ.get("GRDMA_TEST_PAIR_FLAGS", "0")))
    pair.bootstrap_fd(fd)
    assert pair.get_status() == 2
    if role in ("victim", "watcher"):
        # peer-death detection: the victim is killed with SIGKILL, the watcher must see kHalfClosed
        print("connected", flush=True)
        if role == "victim":
            time.sleep(600)
            sys.exit(1)
        t0 = time.time()
        while pair.get_status() == 2:
            assert time.time() - t0 < 30, "the watcher never saw its peer die"
            time.sleep(0.002)
        assert pair.get_status() == 3, pair.get_status()
        print("ok watcher: half closed %.3f s after the connection came up" % (time.time() - t0), flush=True)
        pair.close()
        return
    if role in ("pp_client", "pp_server"):
        return unary_pingpong(role, pair, num_bytes, ring_kib)
    w = Writer(pair)
    seconds = float(os.environ.get("GRDMA_TEST_SECONDS", "0"))
    if seconds > 0:
        return on_the_clock(role, pair, w, seconds, ring_kib, write_size, slice_size)
    deadline = time.time() + 240
    if role == "client":
        writes, pos = deque(), 0
        while pos < num_bytes:
            n = min(write_size, num_bytes - pos)
            data = bytes((pos + i) & 0xFF for i in range(n))
            sl = [data[o:o + slice_size] for o in range(0, n, slice_size)]
            for o in range(0, len(sl), 4000):   # the ABI takes at most 4095 slices per write context
                writes.append(sl[o:o + 4000])   # (grdma_endpoint.cc windows a slice buffer the same way)
            pos += n
        got, nslices = bytearray(), 0
        while len(got) < num_bytes:
            assert time.time() < deadline, "client timed out at %d/%d" % (len(got), num_bytes)
            if not w.active and writes:
                w.begin(writes.popleft())
            if w.active:
                w.step()
            sl, _ = pair.endpoint_read(256)
            nslices += len(sl)
            for s in sl:
                got += s
        assert len(got) == num_bytes
        bad = next((i for i in range(num_bytes) if got[i] != (i & 0xFF)), None)
        assert bad is None, "echo differs at byte %d" % bad
        assert pair.ring_mem() == bytes(ring_kib * 1024), "client ring not zero after the echo"
        pair.Disconnect()
        print("ok client %d bytes echoed in %d slices" % (num_bytes, nslices))
    else:
        pending, echoed, seen = deque(), 0, 0
        while echoed < num_bytes:
            assert time.time() < deadline, "server timed out at %d/%d" % (echoed, num_bytes)
            sl, _ = pair.endpoint_read(256)
            for s in sl:
                # (records are the client's slices; small ones share a 256-byte read, rdma_bp_posix.cc:306-317)
                assert all(b == ((seen + i) & 0xFF) for i, b in enumerate(s))
                seen += len(s)
                pending.append(s)
            if not w.active and pending:
                batch = [pending.popleft() for _ in range(min(len(pending), 2000))]
                w.batch_bytes = sum(len(b) for b in batch)
                w.begin(batch)
            if w.active:
                w.step()
                if not w.active:
                    echoed += w.batch_bytes
        # the client disconnects when it has everything: this end turns half closed (pair.cc:349-356)
        for _ in range(2000):
            if pair.get_status() == 3:
                break
            time.sleep(0.005)
        assert pair.get_status() == 3, "server never saw the peer exit"
        print("ok server %d bytes echoed" % num_bytes)
    pair.close()


if __name__ == "__main__":
    main()
