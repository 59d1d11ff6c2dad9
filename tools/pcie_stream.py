"""PCIe-inclusive streaming rate of the blocking C ABI (what a gRPC host thread would see):
framed 1 MiB messages in PAGEABLE host memory -> grdma_endpoint_write (pinned bounce, gather
over PCIe) -> ring -> grdma_endpoint_read -> one device-to-host copy of the delivered range.
Prints GiB/s of user payload; DESIGN.md section 5 quotes it (it is never bench.py's `value`)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grpc_rdma_amd as g
from grpc_rdma_amd import h2
from grpc_rdma_amd._lib import Slice, ReadSlice

MIB = 1 << 20
ring_kb = int(os.environ.get("RING_KB", "65536"))
n_msgs = int(os.environ.get("MSGS", "128"))
g.init(0)
lib = g.load()
msg = bytes((i * 131 + 7) & 0xFF for i in range(MIB))
items = h2.frame_message(MIB, 1)
wire = b"".join(it[1] if it[0] == "inl" else msg[it[1][0]:it[1][0] + it[1][1]] for it in items)
lens = [len(it[1]) if it[0] == "inl" else it[1][1] for it in items]
host = C.create_string_buffer(wire * n_msgs, len(wire) * n_msgs)   # pageable host memory
base = C.addressof(host)
sl = (Slice * (len(lens) * n_msgs))()
off = 0
for i in range(n_msgs):
    for j, n in enumerate(lens):
        k = i * len(lens) + j
        sl[k].ptr, sl[k].len = base + off, n
        off += n
a, b = g.Pair(ring_kb << 10, 4095), g.Pair(ring_kb << 10, 4095)
g.connect_pairs(a, b)
out = C.create_string_buffer(len(wire) * n_msgs + 32 * len(lens) * n_msgs + (ring_kb << 11) + 4096)
rs = (ReadSlice * 8192)()
wb, done = C.c_int(0), C.c_int(0)


def run():
    got, idx, total = 0, 0, len(sl)
    while idx < total:
        cnt = min(4000, total - idx)
        win = (Slice * cnt).from_address(C.addressof(sl) + idx * C.sizeof(Slice))
        assert lib.grdma_endpoint_write_begin(a.h, win, cnt, 1) >= 0, lib.grdma_last_error()
        done.value = 0
        while not done.value:
            n = lib.grdma_endpoint_write_step(a.h, C.byref(done))
            assert n >= 0, lib.grdma_last_error()
            # drain what arrived: one device pass, one D2H copy of the delivered range
            while True:
                k = lib.grdma_endpoint_read(b.h, 8192, rs, 8192, C.byref(wb))
                assert k >= 0, lib.grdma_last_error()
                if k == 0:
                    break
                lo = rs[0].off
                hi = max(rs[i].off + rs[i].len for i in range(k))
                assert lib.grdma_pair_arena_copy_out(b.h, lo, C.byref(out, got), hi - lo) >= 0
                got += sum(rs[i].len for i in range(k))
        idx += cnt
    return got


run()  # warm-up (page faults, bounce buffer allocation)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    got = run()
dt = (time.perf_counter() - t0) / reps
assert got == len(wire) * n_msgs, (got, len(wire) * n_msgs)
print("pcie_inclusive: %d x 1 MiB messages, ring %d KiB: %.2f GiB/s user payload (%.1f ms per pass)" % (
    n_msgs, ring_kb, n_msgs * MIB / dt / (1 << 30), dt * 1e3))
