#!/usr/bin/env python3
"""The bench workload's job with N Sends per round (grdma_stream_job_set_sends): rounds of the passes, which planners
took the drains / Sends (decline reasons), per-launch times of the timed schedule."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counts(g):
    lib = g.load()
    out = (C.c_uint64 * 6)()
    lib.grdma_rx_fast_drains.argtypes = [C.POINTER(C.c_uint64)]
    lib.grdma_rx_fast_drains(out)
    tx = (C.c_uint64 * 2)()
    lib.grdma_tx_fast_sends.argtypes = [C.POINTER(C.c_uint64)]
    lib.grdma_tx_fast_sends(tx)
    return [int(x) for x in out], [int(x) for x in tx]


def main():
    import torch  # noqa: F401
    import grpc_rdma_amd as g
    from grpc_rdma_amd import stream as gs
    g.init(0)
    sends = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    w = bench.Workload(g, 256)
    ring = (int(sys.argv[3]) if len(sys.argv) > 3 else 262144) * 1024
    max_sge = int(sys.argv[4]) if len(sys.argv) > 4 else 4095
    tx, rx = g.Pair(ring, max_sge, flags), g.Pair(ring, max_sge, flags)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 16 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], int(sys.argv[5]) if len(sys.argv) > 5 else 24)
    pipeline = (int(sys.argv[7]) if len(sys.argv) > 7 else 1) != 0
    job.set_pipeline(pipeline)
    if sends > 1:
        job.set_sends(sends)
    if len(sys.argv) > 8 and int(sys.argv[8]):
        job.set_promised_credit(True)
    c0 = counts(g)
    r = job.run(gs.RUN_EAGER)
    c1 = counts(g)
    print("eager: done %d tx_rounds %d rx_rounds %d; drains took/declined-by-reason %s, sends priced/declined %s" % (
        r.done, r.tx_rounds, r.rx_rounds, [a - b for a, b in zip(c1[0], c0[0])], [a - b for a, b in zip(c1[1], c0[1])]))
    job.set_rounds(int(sys.argv[6]) if len(sys.argv) > 6 else max(-(-int(r.tx_rounds) // sends), 1) + 1)
    for i in range(4):
        c0 = counts(g)
        r = job.run(gs.RUN_GRAPH)
        c1 = counts(g)
        print("graph %d: done %d tx_rounds %d rx_rounds %d, %.1f us; drains took/declined-by-reason %s, sends priced/declined %s" % (
            i, r.done, r.tx_rounds, r.rx_rounds, 1e3 * r.ms_total, [a - b for a, b in zip(c1[0], c0[0])], [a - b for a, b in zip(c1[1], c0[1])]))
    inst = job.run(gs.RUN_INSTRUMENTED_SCHEDULE if pipeline else gs.RUN_INSTRUMENTED)
    names = gs.CLASS_NAMES
    print("us per launch:", {names[i]: (int(inst.launches_class[i]), round(1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])), 1)) for i in range(len(names)) if int(inst.launches_class[i])})


if __name__ == "__main__":
    main()
