"""GPU parity of the graph schedules at the EXACT configurations bench.py times.

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
