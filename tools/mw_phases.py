#!/usr/bin/env python3
"""Phase stamps of the drain plan inside the PAIRED schedule (k_plan_pair_mw): runs the bench
workload's job as its graph, then reads the result blocks of both parities (s_memtime ticks of the committing
workgroup: pattern, probe, read state, totals, emission, end) next to the launch's duration between HIP events."""
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
    ring_kb = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    sends = int(os.environ.get("MW_SENDS", "2"))
    max_sge = int(sys.argv[2]) if len(sys.argv) > 2 else 4095
    w = bench.Workload(g, 256)
    ring = ring_kb * 1024
    tx, rx = g.Pair(ring, max_sge), g.Pair(ring, max_sge)
    g.connect_pairs(tx, rx)
    scap = len(w.lens) * 2 + 64 + w.N // 256
    dst_cap = w.N + 16 * scap + 4096
    dst = g.DeviceBuffer(nbytes=dst_cap)
    job = gs.MultiStreamJob([(tx, rx, w.sge, dst.ptr, dst_cap, scap)], max(8, 4 * (w.E // (ring // 2) + 2), 2 * (len(w.lens) // max_sge + 2)))
    if os.environ.get("MW_WIRE") == "direct":
        pass
    pipeline = os.environ.get("MW_PIPELINE", "1") != "0"
    job.set_pipeline(pipeline)
    if sends > 1:
        job.set_sends(sends)
    if os.environ.get("MW_PROMISE") == "1":
        job.set_promised_credit(True)
    r = job.run(gs.RUN_EAGER)
    job.set_rounds(int(max(-(-int(r.tx_rounds) // sends), r.rx_rounds)))
    for _ in range(4):
        r = job.run(gs.RUN_GRAPH)
    print("graph step %.1f us" % (1e3 * r.ms_total))
    inst = job.run(gs.RUN_INSTRUMENTED_SCHEDULE if pipeline else gs.RUN_INSTRUMENTED)
    names = gs.CLASS_NAMES
    print("us per launch:", {names[i]: round(1e3 * inst.ms_class[i] / max(1, int(inst.launches_class[i])), 1) for i in range(len(names))})
    lib = g.load()
    lib.grdma_stream_job_debug.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    for odd in (0, 1):
        if odd:
            os.environ["GRDMA_DBG_ODD"] = "1"
        else:
            os.environ.pop("GRDMA_DBG_ODD", None)
        td, rd = (C.c_uint64 * 16)(), (C.c_uint64 * 16)()
        lib.grdma_stream_job_debug(job.h, td, rd)
        r_ = [int(x) for x in rd]
        t_ = [int(x) for x in td]
        tot = r_[1] - r_[0]
        print("parity %d drain plan: total %d ticks; pattern %d, probe %d, state %d, totals %d, emit %d, arrived %d, commit loads %d, credit %d; V %d P %d workgroups %d (stamp %x)"
              % (odd, tot, r_[2], r_[3], r_[4], r_[5], r_[6], r_[11], r_[12], r_[13], r_[7], r_[8], r_[10], r_[9]))
        if r_[14]:
            print("parity %d fused round: plan handed over %d ticks after the planners' start, scatter done %.1f us after that" % (odd, r_[14] - r_[0], r_[15] / 100.0))
        print("parity %d send plan: priced %d, end %d ticks; m %d (stamp %x)" % (odd, t_[1] - t_[0], t_[6] - t_[0], t_[7], t_[9]))


if __name__ == "__main__":
    main()
