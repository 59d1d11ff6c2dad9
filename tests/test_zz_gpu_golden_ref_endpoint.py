"""GPU parity against vectors the REFERENCE ITSELF produced (tests/golden/ref_endpoint_*.json: the reference's unmodified
rdma_bp_posix.cc + pair.cc replaying seeded Sends and endpoint reads, oracle/gen_ref_endpoint_golden.py): the HIP pair --
Send on one side, the endpoint-read replay of k_rx_plan on the other -- returns what the reference returned, step by step:
accepted bytes, would-block or the bytes delivered, the readable size left, the credit the sender has been returned."""
import glob
import json
import os
import random
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_endpoint_*.json")))


def pattern(seed, i, n):
    j = np.arange(n, dtype=np.uint64)
    return ((seed * 131 + i * 17 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8).tobytes()


@pytest.mark.parametrize("flags", [0, 2], ids=["staged", "direct"])
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[len("ref_endpoint_"):-5] for f in FILES])
def test_hip_pair_reproduces_the_reference_made_endpoint_trace(gpu, path, flags):
    g = gpu
    doc = json.load(open(path))
    R = doc["ring_kib"] * 1024
    a, b = g.Pair(R, doc["max_sge"], flags), g.Pair(R, doc["max_sge"], flags)
    g.connect_pairs(a, b)
    rng = random.Random(3)
    try:
        for k, (op, want) in enumerate(zip(doc["ops"], doc["results"])):
            if op[0] == "S":
                _, bi, seed, lens = op
                bufs = [g.DeviceBuffer(data=pattern(seed, i, n), offset=rng.randrange(16)) for i, n in enumerate(lens)]
                got = [a.Send(bufs, bi)]
            else:
                slices, _wb = b.endpoint_read(1)
                data = slices[0] if slices else b""
                got = [len(data) if data else -1, (zlib.crc32(data) & 0xFFFFFFFF) if data else 0,
                       b.GetReadableSize(), a.GetWritableSize()]
            assert got == want, "step %d %r" % (k, op[:3])
    finally:
        a.close()
        b.close()
