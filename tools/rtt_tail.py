#!/usr/bin/env python3
"""Where the slow round trips of the unary ping-pong sit: the samples above 1.3 x the median by index (is there a
period?), their share, and a coarse histogram.  usage: rtt_tail.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import grpc_rdma_amd as g
    from grpc_rdma_amd import h2
    lib = g.load()
    g.init(0)
    if not os.environ.get("GRDMA_NO_NUMA_PIN"):
        lib.grdma_host_pin_to_device_node()
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    pay = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    msg = bytes([0x0A, pay]) + bytes(range(pay))
    items = h2.frame_message(len(msg), 1)
    slices = [i[1] if i[0] == "inl" else msg[i[1][0]:i[1][0] + i[1][1]] for i in items]
    a, b = g.Pair(4 << 20, 30), g.Pair(4 << 20, 30)
    g.connect_pairs(a, b)
    a.set_latency_mode(True)
    b.set_latency_mode(True)
    a.arm_read(64)
    b.arm_read(64)
    g._lib.check(lib.grdma_engine_start())
    rtt, ph = g.pingpong(a, b, slices, slices, iters=iters, warmup=2000)
    lib.grdma_engine_stop()
    s = sorted(rtt)
    med = s[len(s) // 2]
    slow = [i for i, v in enumerate(rtt) if v > 1.3 * med]
    print("payload", pay, end=": ")
    print("median %.2f us, %d of %d above 1.3 x median (%.1f %%)" % (med / 1e3, len(slow), len(rtt), 100.0 * len(slow) / len(rtt)))
    gaps = [slow[i + 1] - slow[i] for i in range(len(slow) - 1)]
    from collections import Counter
    print("gaps between slow samples (most common):", Counter(gaps).most_common(12))
    print("first slow indices:", slow[:24])


if __name__ == "__main__":
    main()
