#!/usr/bin/env python3
"""What this box's PCIe link gives a device copy to / from pinned host memory: H2D alone, D2H alone, both at once
(a stream through the endpoint vtable moves every byte over the link once each way).  One JSON line; bench.py reports
value_endpoint_vtable as a fraction of `both_each_GBps` (pcie_ceiling)."""
import json
import sys
import time

import torch

MIB = 1 << 20


def main():
    n = int(sys.argv[1]) * MIB if len(sys.argv) > 1 else 256 * MIB
    reps = 10
    dev = torch.device("cuda:0")
    h_src = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_dst = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(n, dtype=torch.uint8, device=dev)
    d_b = torch.zeros(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def h2d():
        with torch.cuda.stream(s1):
            d_a.copy_(h_src, non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            h_dst.copy_(d_b, non_blocking=True)

    def both():
        h2d()
        d2h()
    t_h2d, t_d2h, t_both = timed(h2d), timed(d2h), timed(both)
    print(json.dumps({"bytes": n, "h2d_GBps": round(n / t_h2d / 1e9, 2), "d2h_GBps": round(n / t_d2h / 1e9, 2),
                      "both_each_GBps": round(n / t_both / 1e9, 2), "both_each_GiBps": round(n / t_both / (1 << 30), 2),
                      "what": "hipMemcpyAsync of %d MiB between pinned host memory and HBM, %d repetitions; both = an H2D and a D2H "
                              "copy in flight together on two streams, rate of EACH direction" % (n // MIB, reps)}))


if __name__ == "__main__":
    main()
