"""Golden traces produced by the reference itself (tests/golden/ref_endpoint_*.json: the reference's unmodified
rdma_bp_posix.cc + pair.cc replaying seeded Sends and endpoint reads, oracle/gen_ref_endpoint_golden.py): the CPU oracle
reproduces every result.  Needs only the JSON -- the reference tree and its build are not required."""
import glob
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pyorc

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_endpoint_*.json")))


def pattern(seed, i, n):
    j = np.arange(n, dtype=np.uint64)
    return ((seed * 131 + i * 17 + j * 7 + (j >> 8)) & 0xFF).astype(np.uint8).tobytes()


def test_there_are_reference_made_traces():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[len("ref_endpoint_"):-5] for f in FILES])
def test_oracle_reproduces_the_reference_made_endpoint_trace(path):
    doc = json.load(open(path))
    assert zlib.crc32(json.dumps(doc["results"]).encode()) & 0xFFFFFFFF == doc["crc_of_results"]
    o = pyorc.OracleLink(doc["ring_kib"] * 1024, doc["max_sge"])
    try:
        for k, (op, want) in enumerate(zip(doc["ops"], doc["results"])):
            if op[0] == "S":
                _, bi, seed, lens = op
                got = [o.send(0, [pattern(seed, i, n) for i, n in enumerate(lens)], bi)]
            else:
                b, _alloc = o.endpoint_read(1)
                got = [len(b) if b else -1, (zlib.crc32(b) & 0xFFFFFFFF) if b else 0, o.readable(1), o.writable(0)]
            assert got == want, "step %d %r" % (k, op[:3])
    finally:
        o.close()
