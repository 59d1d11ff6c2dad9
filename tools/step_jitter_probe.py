#!/usr/bin/env python3
"""The headline job's graph step N times in one process: min / median / max, the steps that took more than 1.2 x the
median, and which planners took the drains (grdma_rx_fast_drains: predicted / declined by reason)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch  # noqa: F401
    import grpc_rdma_amd as g
    from grpc_rdma_amd import stream as gs
    g.init(0)
    lib = g.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    ring = 262144 * 1024
    w = bench.Workload(g, 256)
    tx, rx = g.Pair(ring, 4095), g.Pair(ring, 4095)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 16 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], 16)
    job.set_pipeline(True)
    job.set_sends(2)
    r = job.run(gs.RUN_EAGER)
    job.set_rounds(int(max(-(-int(r.tx_rounds) // 2), r.rx_rounds)))
    fd0 = (C.c_uint64 * 6)()
    lib.grdma_rx_fast_drains(fd0)
    for _ in range(5):
        job.run(gs.RUN_GRAPH)
    ts = [1e3 * job.run(gs.RUN_GRAPH).ms_total for _ in range(n)]
    fd1 = (C.c_uint64 * 6)()
    lib.grdma_rx_fast_drains(fd1)
    s = sorted(ts)
    med = s[len(s) // 2]
    slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 1.2 * med]
    print("steps %d: min %.1f median %.1f max %.1f us; slower than 1.2 x median: %s" % (n, s[0], med, s[-1], slow[:20]))
    print("drains: predicted %d, declined %s" % (fd1[0] - fd0[0], [int(fd1[i] - fd0[i]) for i in range(1, 6)]))


if __name__ == "__main__":
    main()
