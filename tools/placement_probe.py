#!/usr/bin/env python3
"""Does the step time of the headline job depend on WHERE its buffers landed?  Builds the bench workload's job N times
in one process -- every instance with fresh rings, staging buffers, slices and arenas, the earlier ones kept alive so
that the allocator cannot hand the same memory out again -- and prints each instance's graph step and copy-launch time,
then runs them all once more in the same order (is an instance's time a property of the instance?)."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ring = 262144 * 1024
    keep = []
    names = gs.CLASS_NAMES
    for k in range(n):
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
        keep.append((w, tx, rx, dst, job))
    for rnd in range(2):
        for k, (w, tx, rx, dst, job) in enumerate(keep):
            for _ in range(3):
                job.run(gs.RUN_GRAPH)
            t = sorted(job.run(gs.RUN_GRAPH).ms_total for _ in range(7))[3]
            inst = job.run(gs.RUN_INSTRUMENTED_SCHEDULE)
            per = {names[i]: round(1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])), 1) for i in range(len(names)) if inst.launches_class[i]}
            print("pass %d instance %d: ring %x staging/dst %x  graph step %.1f us  %s" % (rnd, k, rx.ring_ptr() if hasattr(rx, "ring_ptr") else 0, dst.ptr, 1e3 * t, per))


if __name__ == "__main__":
    main()
