#!/usr/bin/env python3
"""Phase stamps of the two planner kernels (k_tx_plan, k_rx_plan) in the first round of a step at
the bench configuration (s_memtime ticks -> us at the shader clock measured from the kernel's own
total against its HIP-event duration)."""
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
    ring_kb = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    max_sge = int(sys.argv[2]) if len(sys.argv) > 2 else 4095
    w = bench.Workload(g, 256)
    ring = ring_kb * 1024
    tx, rx = g.Pair(ring, max_sge), g.Pair(ring, max_sge)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 16 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], max(8, 4 * (w.E // (ring // 2) + 2), 2 * (len(w.lens) // max_sge + 2)))
    r = job.run(gs.RUN_EAGER)
    job.set_rounds(int(max(r.tx_rounds, r.rx_rounds)))
    job.run(gs.RUN_GRAPH)
    for _ in range(3):
        inst = job.run(gs.RUN_INSTRUMENTED)
    lib = g.load()
    td, rd = (C.c_uint64 * 16)(), (C.c_uint64 * 16)()
    lib.grdma_stream_job_debug.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.grdma_stream_job_debug(job.h, td, rd)
    names = gs.CLASS_NAMES
    us = {names[i]: 1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])) for i in range(len(names))}
    print("kernel us per launch:", {k: round(v, 1) for k, v in us.items()})
    t = [int(x) for x in td]
    r_ = [int(x) for x in rd]
    print("tx dbg:", [t[i] - t[0] for i in range(1, 9)])
    if r_[9] == 0xFA57:
        tot = r_[1] - r_[0]
        print("rxf_body (last round of a parity): total %d ticks; pattern %d, probe %d, read state %d, scans %d, emit %d; V %d P %d"
              % (tot, r_[2], r_[3], r_[4], r_[5], r_[6], r_[7], r_[8]))
    else:
        tot = r_[1] - r_[0]
        print("k_rx_plan (general): total %d ticks; prologue %d, period detection %d, predicted sizes %d, probe %d, pass0 %d, pass1 %d, pass2 %d, loop end at %d"
              % (tot, r_[14], r_[8], r_[15], r_[9], r_[10], r_[11], r_[12], r_[13]))


if __name__ == "__main__":
    main()
