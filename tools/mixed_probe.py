#!/usr/bin/env python3
"""The mixed-message-size leg of bench.py under the microscope: which planner bodies take its drains and Sends, run by
run (grdma_rx_fast_drains / grdma_tx_fast_sends), and the launches of its timed schedule."""
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
    w = bench.MixedWorkload(g, 64)
    ring, max_sge = 131072 * 1024, 4095
    tx, rx = g.Pair(ring, max_sge), g.Pair(ring, max_sge)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 16 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], max(8, 4 * (w.E // (ring // 2) + 2), 2 * (len(w.lens) // max_sge + 2)))
    job.set_pipeline(True)

    def counts():
        out = (C.c_uint64 * 6)()
        lib.grdma_rx_fast_drains.argtypes = [C.POINTER(C.c_uint64)]
        lib.grdma_rx_fast_drains(out)
        t = (C.c_uint64 * 2)()
        lib.grdma_tx_fast_sends.argtypes = [C.POINTER(C.c_uint64)]
        lib.grdma_tx_fast_sends(t)
        return [int(x) for x in out] + [int(x) for x in t]

    c0 = counts()
    r = job.run(gs.RUN_EAGER)
    c1 = counts()
    print("eager: rounds tx %d rx %d; drains taken %d declined %s; sends fast %d general %d" % (
        r.tx_rounds, r.rx_rounds, c1[0] - c0[0], [a - b for a, b in zip(c1[1:6], c0[1:6])], c1[6] - c0[6], c1[7] - c0[7]))
    job.set_rounds(int(max(r.tx_rounds, r.rx_rounds)))
    for k in range(5):
        c0 = counts()
        r = job.run(gs.RUN_GRAPH)
        c1 = counts()
        print("graph run %d: %.1f us (%.1f GiB/s); drains taken %d declined %s; sends fast %d general %d; period %s" % (
            k, 1e3 * r.ms_total, w.user_bytes / (r.ms_total * 1e-3) / (1 << 30), c1[0] - c0[0],
            [a - b for a, b in zip(c1[1:6], c0[1:6])], c1[6] - c0[6], c1[7] - c0[7], rx.state().get("rx_period")))
    try:
        inst = job.run(gs.RUN_INSTRUMENTED_SCHEDULE)
        names = gs.CLASS_NAMES
        print("us per launch:", {names[i]: round(1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])), 1) for i in range(len(names)) if inst.launches_class[i]})
    except Exception as e:
        print("instrumented schedule:", e)


if __name__ == "__main__":
    main()
