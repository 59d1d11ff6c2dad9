"""Wake-up latency of the background poller: time from the return of Send() on one side
until the receiving pair's wakeup fd is readable (one k_poll pass + eventfd write), with
`n_pairs` registered connections.  Prints p50 / p99 in microseconds."""
import os
import select
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grpc_rdma_amd as g
from grpc_rdma_amd.poller import Poller

g.init(0)
for n_pairs in (1, 64, 512):
    links = []
    for _ in range(n_pairs):
        a, b = g.Pair(1 << 16, 30), g.Pair(1 << 16, 30)
        g.connect_pairs(a, b)
        links.append((a, b))
    pl = Poller(1, 50)
    fds = [pl.add(b) for _, b in links]
    payload = g.DeviceBuffer(data=bytes(range(64)))
    lat = []
    a, b = links[n_pairs // 2]
    fd = fds[n_pairs // 2]
    for it in range(300):
        a.Send([payload])
        t0 = time.perf_counter()
        r, _, _ = select.select([fd], [], [], 2.0)
        t1 = time.perf_counter()
        assert r, "no wakeup"
        b.endpoint_read(4)
        b.lib.grdma_pair_consume_wakeup(b.h)
        # a wakeup written between the drain and the consume: clear it too
        time.sleep(0.0005)
        b.lib.grdma_pair_consume_wakeup(b.h)
        lat.append((t1 - t0) * 1e6)
    lat.sort()
    st = pl.stats()
    print("poller: %4d registered pairs: wake-up p50 %.1f us p99 %.1f us (%d passes, %d wakeups)" % (
        n_pairs, lat[len(lat) // 2], lat[int(len(lat) * 0.99)], st["passes"], st["wakeups"]))
    for _, bb in links:
        pl.remove(bb)
    pl.close()
    for aa, bb in links:
        aa.close(); bb.close()
