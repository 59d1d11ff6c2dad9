#!/usr/bin/env python3
"""Where a unary round trip's microseconds go: the host-side phases of grdma_pingpong next to the resident engine's own
cycle counters (command-block load / body per command type) and the phase ticks of the one-wave small send."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import grpc_rdma_amd as g
    from grpc_rdma_amd import h2
    lib = g.load()
    g.init(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    # "watch": standing reads carried out by the engine's watcher workgroups (grdma_pair_arm_read); without it every
    # read is a command of its own (round 4's configuration)
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    if "prof" in sys.argv[3:]:   # the phase stamps of the latency paths: off unless asked for
        os.environ["GRDMA_PROFILE_TICKS"] = "1"
    armed = mode == "watch"
    msg = bytes([0x0A, 64]) + bytes(range(64))
    items = h2.frame_message(len(msg), 1)
    slices = [i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items]
    a, b = g.Pair(4 << 20, 30), g.Pair(4 << 20, 30)
    g.connect_pairs(a, b)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    g._lib.check(lib.grdma_engine_start())
    if armed:
        lib.grdma_pair_arm_read.argtypes = [C.c_void_p, C.c_uint64]
        lib.grdma_pair_arm_read(a.h, 64)
        lib.grdma_pair_arm_read(b.h, 64)
    g.pingpong(a, b, slices, slices, iters=2000, warmup=200)
    e0 = (C.c_uint64 * 5)()
    t0 = (C.c_uint64 * 8)()
    lib.grdma_engine_debug(e0)
    lib.grdma_tx_small_ticks(t0)
    x0 = (C.c_uint64 * 9)()
    lib.grdma_rx_express_ticks(x0)
    w0 = (C.c_uint64 * 12)()
    lib.grdma_watch_ticks(w0)
    rtt, ph = g.pingpong(a, b, slices, slices, iters=iters, warmup=0)
    w1 = (C.c_uint64 * 12)()
    lib.grdma_watch_ticks(w1)
    x1 = (C.c_uint64 * 9)()
    lib.grdma_rx_express_ticks(x1)
    e1 = (C.c_uint64 * 5)()
    t1 = (C.c_uint64 * 8)()
    lib.grdma_engine_debug(e1)
    lib.grdma_tx_small_ticks(t1)
    rtt.sort()
    print("rtt p50 %.2f us  p95 %.2f  (%d round trips%s)" % (rtt[len(rtt) // 2] / 1e3, rtt[int(len(rtt) * .95)] / 1e3, iters,
                                                            ", reads: " + mode if armed else ""))
    if armed:
        print("watch hits %d / %d, watcher workgroups %d" % (a.watch_hits(), b.watch_hits(), lib.grdma_engine_watchers()))
    print("host phases us: client write %.2f, server read %.2f, server write %.2f, client read %.2f" % tuple(p / 1e3 / iters for p in ph))
    de = [int(e1[i]) - int(e0[i]) for i in range(5)]
    print("engine ticks per round trip: odd-type commands load %d body %d; even-type commands load %d body %d" % tuple(x // iters for x in de[:4]))
    n = int(t1[6]) - int(t0[6])
    if n:
        names = ["slice loads", "pricing", "copies issued", "copies acked", "bookkeeping", "release"]
        print("small send wave, ticks per send (%d sends): " % n + ", ".join("%s %d" % (names[i], (int(t1[i]) - int(t0[i])) // n) for i in range(6)))
    nx = int(x1[8]) - int(x0[8])
    if nx:
        names = ["state loaded", "records known", "payload loaded", "stores issued", "stores acknowledged", "commit: counters loaded",
                 "commit: stores issued", "released"]
        print("express drain, ticks per drain (%d drains): " % nx + ", ".join("%s %d" % (names[i], (int(x1[i]) - int(x0[i])) // nx) for i in range(8)))
    nw = int(w1[3]) - int(w0[3])
    if nw:
        print("watchers, us per drain (%d drains): arrival report published -> found %.2f, found -> plan body entered %.2f, found -> drain done %.2f" % (
            nw, (int(w1[1]) - int(w0[1])) / nw / 100.0, (int(w1[4]) - int(w0[4])) / nw / 100.0, (int(w1[2]) - int(w0[2])) / nw / 100.0))
        d = lambda i: (int(w1[i]) - int(w0[i])) / nw / 100.0
        print("  single-wave drains, us since found: bytes loaded %.2f, chain walked %.2f, payload in LDS %.2f, stores issued %.2f, sequence word stored %.2f; command off the mailbox -> found %.2f" % (
            d(5), d(10), d(11), d(6), d(7), d(9)))


if __name__ == "__main__":
    main()
