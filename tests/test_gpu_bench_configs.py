"""GPU parity of the graph schedules at the EXACT configurations bench.py times, and of the burst rounds.

The job must reproduce the sequential execution of the reference's loops -- one Send from the rdma_flush cursor
(rdma_bp_posix.cc:470-524), then endpoint reads until one would block (:180-291) -- slice for slice: same delivered
slices, same number of Sends, same final protocol state, ring all zero.  The oracle side is
oracle/grdma_oracle.c:orc_stream_rounds (checked against the Python-driven OracleLink loop below).
(Until round 4 this file also held the tests of the persistent link engine, k_link -- one resident launch per step, a
third of the graph schedule's rate -- which round 5 retired.)"""
import random

import pytest

from oracle import pyorc

pytestmark = pytest.mark.gpu

PASSES = 3


def framed(n_msgs, msg_len, seed, max_frame=16384):
    """-> (wire bytes, slice lengths) of n_msgs framed messages as chttp2 hands them to the endpoint;
    message i is a rotation of one random block, so every message differs."""
    rng = random.Random(seed)
    block = bytes(rng.getrandbits(8) for _ in range(4099))
    wire, lens = bytearray(), []
    for i in range(n_msgs):
        rot = (i * 131) % len(block)
        body = (block[rot:] + block[:rot]) * (msg_len // len(block) + 1)
        w, l = pyorc.h2_frame_message(body[:msg_len], stream_id=2 * i + 1, max_frame=max_frame)
        wire += w
        lens += l
    return bytes(wire), lens


def mixed(n_msgs, seed, lo=1, hi=(4 << 20) - 1024):
    """Message sizes drawn like the reference's own echo test (examples/cpp/test/common.h:4-31:
    uniform in [1, 4 MiB - 1 KiB]), fixed seed."""
    rng = random.Random(seed)
    block = bytes(rng.getrandbits(8) for _ in range(8191))
    wire, lens = bytearray(), []
    for i in range(n_msgs):
        n = rng.randint(lo, hi)
        rot = rng.randrange(len(block))
        body = (block[rot:] + block[:rot]) * (n // len(block) + 1)
        w, l = pyorc.h2_frame_message(body[:n], stream_id=2 * i + 1)
        wire += w
        lens += l
    return bytes(wire), lens


class Link:
    """One loop-back link with its slices resident in ONE device buffer (slice i at an offset whose
    low four bits vary, like grpc_slice payloads do)."""

    def __init__(self, g, R, max_sge, wire, lens, flags=0, seed=5):
        rng = random.Random(seed)
        self.g, self.R, self.wire, self.lens = g, R, wire, lens
        packed, offs, src = bytearray(), [], 0
        for n in lens:
            packed += bytes(rng.randrange(1, 16))
            offs.append(len(packed))
            packed += wire[src:src + n]
            src += n
        self.buf = g.DeviceBuffer(data=bytes(packed) + bytes(64))
        self.tx, self.rx = g.Pair(R, max_sge, flags), g.Pair(R, max_sge, flags)
        g.connect_pairs(self.tx, self.rx)
        self.N = len(wire)
        self.slices_cap = 2 * len(lens) + 64 + self.N // 256
        self.dst_cap = self.N + 32 * self.slices_cap + 4096
        self.dst = g.DeviceBuffer(nbytes=self.dst_cap)
        self.sge = [(self.buf.ptr + o, n) for o, n in zip(offs, lens)]

    def spec(self):
        return (self.tx, self.rx, self.sge, self.dst.ptr, self.dst_cap, self.slices_cap)

    def check(self, job, li, exp, exact):
        """Delivered slices of link li against the oracle result `exp`."""
        ds = job.delivered_slices(li)
        mem = self.dst.read(self.dst_cap)
        got = [mem[o:o + n] for o, n in ds]
        assert b"".join(got) == self.wire, "delivered byte stream differs from what was written"
        assert self.rx.ring_mem() == bytes(self.R), "ring not zero after the drain"
        tx, rx = self.tx.state(), self.rx.state()
        assert rx["head"] == tx["remote_tail"] and rx["remain"] == 0
        if exact:
            assert [len(x) for x in got] == exp["lens"]
            for k in ("remote_tail", "remote_head", "partial_write"):
                assert tx[k] == exp["state"][k], k
            for k in ("head", "moving_head", "remain", "internal_read_size", "credit_msgs", "leftover_cap"):
                assert rx[k] == exp["state"][k], k

    def close(self):
        self.tx.close()
        self.rx.close()
        self.buf.free()
        self.dst.free()


CASES = [
    # (ring, max_sge, n_msgs, msg_len)
    (1 << 22, 4095, 6, 1 << 20),      # reference default ring, 1 MiB messages: every Send is cut by the staging budget
    (1 << 24, 4095, 40, 70000),       # 16 MiB ring, many medium messages
    (1 << 18, 30, 24, 3000),          # small ring, reference default max_sge = 30
    (1 << 26, 512, 24, 1 << 18),      # big ring, Sends of 512 records
    (1 << 16, 30, 40, 20000),         # 64 KiB ring: every Send is cut by the peer's credit
    (1 << 22, 30, 8, 1 << 20),        # the reference's own operating point: 4 MiB ring, max_sge 30
]
IDS = ["r4m_1mib", "r16m_70k", "r256k_sge30", "r64m_sge512", "r64k_credit", "r4m_sge30"]


def test_c_rounds_oracle_equals_the_python_driven_loop():
    """orc_stream_rounds against OracleLink driven from Python (the loop of test_gpu_stream_job.py)."""
    wire, lens = framed(24, 3000, 3)
    slices, o = [], 0
    for n in lens:
        slices.append(wire[o:o + n])
        o += n
    r = pyorc.stream_rounds(1 << 18, 30, wire, lens, passes=2)
    link = pyorc.OracleLink(1 << 18, 30)
    for _ in range(2):
        idx = byte = 0
        deliv = []
        while idx < len(slices):
            left = link.send(0, slices[idx:], byte)
            while left > 0:
                room = len(slices[idx]) - byte
                if left >= room:
                    left -= room
                    idx += 1
                    byte = 0
                else:
                    byte += left
                    left = 0
            while True:
                s, _ = link.endpoint_read(1)
                if not s:
                    break
                deliv.append(s)
    assert [len(x) for x in deliv] == r["lens"] and r["stream_ok"] and r["ring_zero"]
    assert link.state(1)["head"] == r["state"]["head"] and link.state(0)["remote_tail"] == r["state"]["remote_tail"]
    link.close()


# ---- the exact configurations of bench.py ------------------------------------------------------
def test_bench_config_256x1mib_ring128m(gpu):
    """256 x 1 MiB messages, 128 MiB ring, max_sge 4095 (bench.py's value_ring128m_one_send_per_round): the pipelined and
    the sequential graph schedule against the sequential-rounds oracle."""
    from grpc_rdma_amd import stream as gs
    R, max_sge = 128 << 20, 4095
    wire, lens = framed(256, 1048580, seed=11)
    exp = pyorc.stream_rounds(R, max_sge, wire, lens, passes=1)
    assert exp["stream_ok"] and exp["ring_zero"]
    # at this ring no Send is limited by the credit, so the pipelined schedule makes the same records (hence slices
    # and state) as the sequential one
    for pipeline in (False, True):
        link = Link(gpu, R, max_sge, wire, lens)
        job = gs.MultiStreamJob([link.spec()], 64)
        job.set_pipeline(pipeline)
        r = job.run(gs.RUN_EAGER)
        assert r.done and r.bytes_delivered == len(wire)
        assert int(r.tx_rounds) == exp["rounds"]
        link.check(job, 0, exp, exact=True)
        job.close()
        link.close()


def test_bench_config_32_links_64kib(gpu):
    """32 connections x 64 x 64 KiB messages, 4 MiB rings (BASELINE configs[3] shape): every link against its own
    oracle, the lock-step graph job."""
    from grpc_rdma_amd import stream as gs
    R, max_sge, n = 4 << 20, 4095, 32
    data = [framed(64, 65536 + 3, seed=100 + i) for i in range(n)]
    exps = [pyorc.stream_rounds(R, max_sge, w, l, passes=1) for w, l in data]
    links = [Link(gpu, R, max_sge, w, l, seed=i) for i, (w, l) in enumerate(data)]
    job = gs.MultiStreamJob([l.spec() for l in links], 64)
    r = job.run(gs.RUN_EAGER)
    assert r.done
    for i, (l, e) in enumerate(zip(links, exps)):
        l.check(job, i, e, exact=True)
    job.close()
    for l in links:
        l.close()


BURST_CASES = [
    # (ring, max_sge, n_msgs, msg_len, burst)
    (1 << 22, 30, 8, 1 << 20, 16),     # the reference's default knobs, 16 Sends per round (what bench.py times)
    (1 << 22, 30, 8, 1 << 20, 4),
    (1 << 18, 30, 24, 3000, 8),        # small ring: later Sends of a round find no credit and accept nothing
    (1 << 16, 30, 40, 20000, 3),       # 64 KiB ring: every round is cut by the peer's credit
    (1 << 24, 512, 40, 70000, 2),
]


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("case", BURST_CASES, ids=["r4m_sge30_b16", "r4m_sge30_b4", "r256k_b8", "r64k_b3", "r16m_b2"])
def test_burst_rounds_equal_the_oracle(gpu, case, graph):
    """grdma_stream_job_set_burst: `burst` Sends back to back (one k_tx_plan_seq launch, one gather
    launch over burst plans, one wire launch), then ONE drain.  Delivered slices, number of Sends,
    final protocol state and the zero ring equal oracle/grdma_oracle.c:orc_stream_rounds_burst
    driving the same schedule, pass after pass."""
    from grpc_rdma_amd import stream as gs
    R, sge, n_msgs, msg_len, burst = case
    wire, lens = framed(n_msgs, msg_len, seed=R ^ burst)
    exp = pyorc.stream_rounds(R, sge, wire, lens, passes=PASSES, burst=burst)
    assert exp["stream_ok"] and exp["ring_zero"]
    link = Link(gpu, R, sge, wire, lens)
    job = gs.MultiStreamJob([link.spec()], 4 * (exp["rounds"] + 8))
    job.set_burst(burst)
    first = None
    for p in range(PASSES):
        r = job.run(gs.RUN_EAGER)
        assert r.done and r.bytes_delivered == link.N == r.bytes_sent
        if first is None:
            first = r
        if graph and p == 0:
            job.set_rounds(int(r.rx_rounds) + 1)
            r = job.run(gs.RUN_GRAPH)
            assert r.done and r.bytes_delivered == link.N
            break
    assert int(first.tx_rounds) == exp["rounds"], "number of Sends that accepted bytes"
    if graph:
        exp = pyorc.stream_rounds(R, sge, wire, lens, passes=2, burst=burst)
    link.check(job, 0, exp, exact=True)
    job.close()
    link.close()


@pytest.mark.parametrize("case", [(1 << 22, 30, 8, 1 << 20, 16), (1 << 16, 30, 40, 20000, 3), (1 << 18, 64, 30, 9000, 5)],
                         ids=["r4m_sge30_b16", "r64k_b3", "r256k_sge64_b5"])
def test_burst_rounds_direct_wire(gpu, case):
    """Burst rounds with GRDMA_WIRE_DIRECT: the Sends of a round build their records in the peer
    ring itself (no staging, no wire launch); a record that crosses the ring end is two segments."""
    from grpc_rdma_amd import stream as gs
    R, sge, n_msgs, msg_len, burst = case
    wire, lens = framed(n_msgs, msg_len, seed=R + burst)
    exp = pyorc.stream_rounds(R, sge, wire, lens, passes=PASSES, burst=burst)
    link = Link(gpu, R, sge, wire, lens, flags=2)
    job = gs.MultiStreamJob([link.spec()], 4 * (exp["rounds"] + 8))
    job.set_burst(burst)
    for _ in range(PASSES):
        r = job.run(gs.RUN_EAGER)
        assert r.done and r.bytes_delivered == link.N == r.bytes_sent
    link.check(job, 0, exp, exact=True)
    job.close()
    link.close()


def test_burst_rounds_three_links(gpu):
    """Three connections of different shapes advance in lock step, 6 Sends per round each."""
    from grpc_rdma_amd import stream as gs
    shapes = [(1 << 20, 30, 12, 50000), (1 << 18, 30, 20, 7000), (1 << 22, 30, 3, 1 << 20)]
    links, exps = [], []
    for i, (R, sge, n_msgs, msg_len) in enumerate(shapes):
        wire, lens = framed(n_msgs, msg_len, seed=70 + i)
        exps.append(pyorc.stream_rounds(R, sge, wire, lens, passes=2, burst=6))
        links.append(Link(gpu, R, sge, wire, lens, seed=i))
    job = gs.MultiStreamJob([l.spec() for l in links], 4 * (max(e["rounds"] for e in exps) + 8))
    job.set_burst(6)
    for _ in range(2):
        r = job.run(gs.RUN_EAGER)
        assert r.done and r.bytes_delivered == sum(l.N for l in links)
    for i, l in enumerate(links):
        l.check(job, i, exps[i], exact=True)
    job.close()
    for l in links:
        l.close()
