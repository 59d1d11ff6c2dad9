"""k_h2_deframe with the message-boundary step (GRDMA_H2_BOUNDARY_STEP, csrc/grdma_h2_fast.h): the
events equal the oracle's, and equal the kernel's own events without the step, on the shapes the step
is made for and on the ones it must leave alone.  The step's two functions are checked on the CPU by
tests/test_h2_fast_host.py; this file checks the device glue around them."""
import random

import pytest

from oracle import pyorc
from tests.h2_helpers import PREFACE, frame, grpc_msg
from tests.test_h2_fast_host import receiver_slices, sender_slices

pytestmark = pytest.mark.gpu


def oracle_events(chunks, prefix, streams=()):
    p = pyorc.H2Parser(expect_client_prefix=prefix)
    for sid in streams:
        assert p.open_stream(sid) == 0
    out = []
    for i, s in enumerate(chunks):
        rc, ev = p.feed(s)
        out += [(k, a, b, c, d, i) for k, a, b, c, d in ev]
        assert rc == 0
    return out


def gpu_events(g, chunks, prefix, step, gap_rng=None, streams=()):
    from grpc_rdma_amd import h2dev
    arena, table = bytearray(), []
    for s in chunks:
        if gap_rng is not None:
            arena += b"\xee" * gap_rng.randrange(1, 16)
        else:
            arena += bytes((-len(arena)) % 16)
        table.append((len(arena), len(s)))
        arena += s
    buf = g.DeviceBuffer(data=bytes(arena) + bytes(64))
    p = h2dev.Parser(prefix, boundary_step=step)
    if streams:
        assert p.open_streams(streams) == 0
    err, ev = p.deframe(buf.ptr, table, cap=8 * len(chunks) + 4096)
    steps = p.last_boundary_steps
    p.close()
    assert err == 0
    return ev, steps


PRE = [PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82")]


@pytest.mark.parametrize("shape", ["sender", "receiver"])
@pytest.mark.parametrize("gaps", [False, True], ids=["aligned", "unaligned"])
def test_boundary_step_streaming_shapes(gpu, shape, gaps):
    sizes = [1 << 20, 16384 * 3 - 5, 40000, 16384 - 5, 7, 16384 * 70 + 123, 1, 300000, 5, 2, 16379, 16380]
    tx = sender_slices(sizes)
    chunks = PRE + (tx if shape == "sender" else receiver_slices(tx))
    exp = oracle_events(chunks, True)
    rng = random.Random(9)
    off, n_off = gpu_events(gpu, chunks, True, False, gap_rng=rng if gaps else None)
    on, n_on = gpu_events(gpu, chunks, True, True, gap_rng=rng if gaps else None)
    assert off == exp and n_off == 0
    assert on == exp
    assert n_on >= (6 if shape == "sender" else 4)


def test_boundary_step_bench_shape(gpu):
    """1 MiB messages as the receiving side of the bench sees them: every message start after the first
    goes through the step, and what is left for the byte-wise path is the first message start."""
    n = 24
    rx = receiver_slices(sender_slices([1 << 20] * n, end_stream=False))
    chunks = [frame(1, 4, 1, b"\x82")] + rx
    exp = oracle_events(chunks, False, streams=(1,))
    on, steps = gpu_events(gpu, chunks, False, True, streams=(1,))
    assert on == exp
    assert steps == n


def test_boundary_step_random_streams_and_cuts(gpu):
    rng = random.Random(4)
    for trial in range(12):
        parts = [PREFACE + frame(4, 0, 0)] + [frame(1, 4, sid, b"\x82\x86") for sid in (1, 3)]
        body = []
        for sid in (1, 3):
            body += sender_slices([rng.choice([1, 5, 9, 100, 16379, 16384, 20000, 70000]) for _ in range(rng.randrange(1, 5))],
                                  sid=sid, end_stream=rng.random() < 0.5, seed=trial)
            if rng.random() < 0.5:
                body.append(frame(6, 0, 0, bytes(8)))
        if trial % 2:
            body = receiver_slices(body)
        if trial % 3 == 2:
            cut = []
            for s_ in body:
                if len(s_) > 2 and rng.random() < 0.3:
                    k = rng.randrange(1, len(s_))
                    cut += [s_[:k], s_[k:]]
                else:
                    cut.append(s_)
            body = cut
        chunks = parts + body
        exp = oracle_events(chunks, True)
        on, _ = gpu_events(gpu, chunks, True, True, gap_rng=rng if trial % 2 else None)
        assert on == exp, trial


@pytest.mark.parametrize("pairs", [True, False], ids=["bulk64", "bulk32"])
@pytest.mark.parametrize("shape", ["sender", "receiver"])
def test_bulk_pairs_matches_the_oracle(gpu, shape, pairs):
    """GRDMA_H2_BULK_PAIRS (the default since round 3): every lane of the bulk step owns a frame (64 frames per step);
    GRDMA_H2_NO_BULK_PAIRS: 32 frames per step.  Same events as the oracle on the streaming shapes, with slices at
    odd offsets and with some slices cut in two."""
    from grpc_rdma_amd import h2dev
    sizes = [1 << 20, 16384 * 3 - 5, 40000, 16384 - 5, 7, 16384 * 70 + 123, 1, 300000, 16384 * 130]
    tx = sender_slices(sizes)
    body = tx if shape == "sender" else receiver_slices(tx)
    cut = []
    for j, s_ in enumerate(body):
        if len(s_) > 100 and j % 37 == 5:
            cut += [s_[:77], s_[77:]]
        else:
            cut.append(s_)
    rng = random.Random(2)
    for chunks in (PRE + body, PRE + cut):
        exp = oracle_events(chunks, True)
        arena, table = bytearray(), []
        for s_ in chunks:
            arena += b"\xee" * rng.randrange(1, 16)
            table.append((len(arena), len(s_)))
            arena += s_
        buf = gpu.DeviceBuffer(data=bytes(arena) + bytes(64))
        p = h2dev.Parser(True, boundary_step=True, bulk_pairs=pairs)
        err, ev = p.deframe(buf.ptr, table, cap=8 * len(chunks) + 4096)
        p.close()
        assert err == 0 and ev == exp
