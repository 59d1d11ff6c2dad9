"""CPU: the oracle replays the traces generated from the reference-built codec
(tests/golden/ring_*.json, made by oracle/gen_golden.py) and must reproduce every
recorded value.  Unlike test_oracle_vs_ref.py this needs neither /root/reference
nor oracle/_ref: the vectors are committed."""
import glob
import hashlib
import json
import os
import random

import pytest

from oracle import pyorc

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ring_*.json")))


def payload(seed, n):
    rng = random.Random(seed)
    return bytes(rng.getrandbits(8) for _ in range(n))


def sha(b):
    return hashlib.sha256(b).hexdigest()


def replay(doc, link, ring_exact=True, mask=None):
    for i, st in enumerate(doc["steps"]):
        if st["op"] == "send":
            sl = [payload(s, n) for s, n in st["slices"]]
            assert link.send(0, sl, st.get("byte_idx", 0)) == st["sent"], (doc["name"], i)
            assert [list(w) for w in link.last_wrs(0)] == st["wrs"]
            if ring_exact:
                assert sha(link.staging_mem(0)) == st["staging_sha256"]
        elif st["op"] == "recv":
            got = link.recv(1, st["cap"])
            assert len(got) == st["got_len"] and sha(got) == st["got_sha256"]
        else:
            got, alloc = link.endpoint_read(1)
            assert len(got) == st["got_len"] and sha(got) == st["got_sha256"]
            assert alloc == st["alloc"]
        if ring_exact:
            assert sha(link.ring_mem(1)) == st["ring_sha256"], (doc["name"], i)
        elif "ring_hex" in st:
            assert mask(link.ring_mem(1), bytes.fromhex(st["ring_hex"])), (doc["name"], i)
        rx, tx = link.state(1), link.state(0)
        for k, v in st["rx_state"].items():
            assert rx[k] == v, (doc["name"], i, "rx", k)
        for k, v in st["tx_state"].items():
            assert tx[k] == v, (doc["name"], i, "tx", k)
        assert link.readable(1) == st["readable"]
        assert link.has_message(1) == st["has_message"]
        assert link.writable(0) == st["writable"]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_reference_traces(path):
    doc = json.load(open(path))
    link = pyorc.OracleLink(doc["ring_size"], doc["max_sge"])
    replay(doc, link)
    link.close()


def test_hello_ring_probe():
    """SURVEY.md section 8c probe: an 11-byte record -> tail 32, readable 11."""
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ring_hello.json")))
    s0 = doc["steps"][0]
    assert s0["sent"] == 11 and s0["tx_state"]["remote_tail"] == 32 and s0["readable"] == 11
    img = bytes.fromhex(s0["ring_hex"])
    assert img[:8] == (11).to_bytes(8, "little") and img[24:32] == b"\xff" * 8
    assert doc["steps"][1]["rx_state"]["head"] == 32


ZC_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "zc_*.json")))


def zc_slices(link, zc_cap, spec, expect_allocs=None):
    """The slice rule of oracle/gen_golden.py::zc_slices."""
    out, allocs = [], []
    for item in spec:
        if item[0] == "zc":
            _, seed, n = item
            off = link.allocate_send_buffer(0, n)
            allocs.append(off)
            if off is None:
                off = (seed * 131) % (zc_cap - n + 1)
            link.zerocopy_write(0, off, payload(seed, n))
            out.append(("zc", off, n))
        else:
            out.append(payload(item[0], item[1]))
    if expect_allocs is not None:
        assert allocs == expect_allocs
    return out


@pytest.mark.parametrize("path", ZC_GOLDEN, ids=[os.path.basename(p) for p in ZC_GOLDEN])
def test_oracle_reproduces_zero_copy_traces(path):
    """AllocateSendBuffer / SendZerocopy traces generated over the reference-built ring codec
    (tests/golden/zc_*.json): allocator answers, accepted bytes, work requests, scatter-gather entry counts,
    buffer tail, staging and ring images, state -- without /root/reference."""
    doc = json.load(open(path))
    assert len(doc["steps"]) >= 50
    link = pyorc.OracleLink(doc["ring_size"], doc["max_sge"])
    link.enable_zerocopy(0, doc["zerocopy_buffer"])
    kinds = set()
    for i, st in enumerate(doc["steps"]):
        if st["op"] == "zc_send":
            sl = zc_slices(link, doc["zerocopy_buffer"], st["slices"], st["allocs"])
            assert link.send_zerocopy(0, sl, st.get("byte_idx", 0)) == st["sent"], (doc["name"], i)
            assert [list(w) for w in link.last_wrs(0)] == st["wrs"]
            assert link.zerocopy_state(0) == st["zc_state"]
            assert sha(link.staging_mem(0)) == st["staging_sha256"]
            kinds.add("wrap" if len(st["wrs"]) == 2 else "flat")
            kinds.add("partial" if st["tx_state"]["partial_write"] else "whole")
        elif st["op"] == "send":
            assert link.send(0, [payload(s, n) for s, n in st["slices"]]) == st["sent"]
        else:
            got = link.recv(1, st["cap"])
            assert len(got) == st["got_len"] and sha(got) == st["got_sha256"]
        assert sha(link.ring_mem(1)) == st["ring_sha256"], (doc["name"], i)
        rx, tx = link.state(1), link.state(0)
        assert all(rx[k] == v for k, v in st["rx_state"].items()), (doc["name"], i)
        assert all(tx[k] == v for k, v in st["tx_state"].items()), (doc["name"], i)
    assert kinds == {"wrap", "flat", "partial", "whole"}
