"""k_h2_deframe -- the kernel source itself, csrc/grdma_h2_kernels.h -- run on the CPU under the wave emulator
(tests/cc/wave_emu.h, tests/cc/h2_emu_host.cc): 512 emulated threads, ballots, prefix sums, the LDS look-ahead
ring between the staging waves and the parsing wave.  Events must equal the oracle's with the boundary step off,
on, and with GRDMA_H2_BULK_PAIRS (64 frames per bulk step) -- the variant that has not run on a GPU yet.

This checks the kernel's logic, not its timing or the hardware's memory model; the GPU parity tests stay the
reference (tests/test_gpu_h2.py, tests/test_zz_gpu_h2_boundary.py)."""
import ctypes as C
import json
import os
import random
import subprocess

import pytest

from oracle import pyorc
from tests.h2_helpers import PREFACE, frame
from tests.test_h2_fast_host import receiver_slices, sender_slices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SO = os.path.join(ROOT, "oracle", "_build", "libh2_emu_host.so")
CSRC = os.path.join(ROOT, "grpc-rdma_amd", "csrc")
SRCS = [os.path.join(ROOT, "tests", "cc", "h2_emu_host.cc"), os.path.join(ROOT, "tests", "cc", "wave_emu.h"),
        os.path.join(CSRC, "grdma_h2_kernels.h"), os.path.join(CSRC, "grdma_h2_fast.h"),
        os.path.join(CSRC, "grdma_devfn.h"), os.path.join(CSRC, "grdma_dev.h")]

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ as host compiler")

SERVER, FIRST, STEP, NO_STEP, PAIRS = 1, 2, 4, 8, 16
VARIANTS = {"bytewise": NO_STEP, "boundary": STEP, "boundary+pairs": STEP | PAIRS, "pairs": NO_STEP | PAIRS}


@pytest.fixture(scope="module")
def lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS):
        subprocess.check_call([CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unused-value",
                               "-I" + os.path.join(ROOT, "tests", "cc"), "-I" + os.path.join(ROOT, "tests", "cc", "emu_include"),
                               SRCS[0], "-o", SO])
    L = C.CDLL(SO)
    L.h2_emu_deframe.restype = C.c_int64
    L.h2_emu_deframe.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p, C.c_uint64,
                                 C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64,
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    return L


def emu(L, chunks, flags, streams=(), gap_rng=None):
    arena, table = bytearray(), []
    for s in chunks:
        arena += (b"\xee" * gap_rng.randrange(1, 16)) if gap_rng is not None else bytes((-len(arena)) % 16)
        table += [len(arena), len(s)]
        arena += s
    cap = 8 * len(chunks) + 4096
    ev = (C.c_uint32 * (6 * cap))()
    err, st = C.c_int(0), (C.c_uint64 * 4)()
    ids = (C.c_uint32 * max(1, len(streams)))(*streams)
    tb = (C.c_uint64 * max(1, len(table)))(*table)
    n = L.h2_emu_deframe(flags, 16384, ids, len(streams), bytes(arena), len(arena), tb, len(chunks), ev, cap,
                         C.byref(err), st)
    assert n >= 0
    return err.value, [tuple(ev[6 * i:6 * i + 6]) for i in range(n)], dict(bulk_steps=st[0], bulk_frames=st[1],
                                                                             boundary_steps=st[2], parsed=st[3])


def oracle(chunks, prefix, streams=()):
    p = pyorc.H2Parser(expect_client_prefix=prefix)
    for s in streams:
        assert p.open_stream(s) == 0
    out = []
    for i, s in enumerate(chunks):
        rc, ev = p.feed(s)
        out += [(k, a, b, c, d, i) for k, a, b, c, d in ev]
        if rc:
            return rc, out
    return 0, out


PRE = [PREFACE + frame(4, 0, 0), frame(1, 4, 1, b"\x82")]


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("shape", ["sender", "receiver"])
def test_streaming_shapes_under_the_emulator(lib, shape, variant):
    sizes = [1 << 20, 16384 * 3 - 5, 40000, 16384 - 5, 7, 16384 * 70 + 123, 1, 300000, 5, 2, 16379, 16384 * 130]
    tx = sender_slices(sizes)
    chunks = PRE + (tx if shape == "sender" else receiver_slices(tx))
    rc, exp = oracle(chunks, True)
    assert rc == 0
    rng = random.Random(3)
    for gaps in (None, rng):
        err, ev, st = emu(lib, chunks, SERVER | FIRST | VARIANTS[variant], gap_rng=gaps)
        assert err == 0 and st["parsed"] == len(chunks)
        assert ev == exp
        frames = sum(1 for e in exp if e[0] == 1 and e[1] == 0)
        assert st["bulk_frames"] > frames // 2, st                # the bulk step carried the steady state
        # 285 / 283 frames of these messages: 13 bulk steps at 32 frames per step, 9 at 64
        assert st["bulk_steps"] == (9 if "pairs" in variant else 13), st
        assert (st["boundary_steps"] > 0) == ("boundary" in variant)


@pytest.mark.parametrize("variant", ["boundary", "boundary+pairs"])
def test_cut_slices_and_interleaved_streams_under_the_emulator(lib, variant):
    rng = random.Random(8)
    for trial in range(6):
        parts = [PREFACE + frame(4, 0, 0)] + [frame(1, 4, sid, b"\x82\x86") for sid in (1, 3)]
        body = []
        for sid in (1, 3, 1):
            body += sender_slices([rng.choice([1, 5, 9, 100, 16379, 16384, 20000, 70000, 16384 * 40]) for _ in range(rng.randrange(1, 4))],
                                  sid=sid, end_stream=False, seed=trial)
            if rng.random() < 0.5:
                body.append(frame(6, 0, 0, bytes(8)))
            if rng.random() < 0.3:
                body.append(frame(0, 0, 9, b"\0\0\0\0\1x"))  # DATA for a stream that is not in the map
        if trial % 2:
            body = receiver_slices(body)
        if trial % 3 == 2:
            cut = []
            for s_ in body:
                if len(s_) > 2 and rng.random() < 0.2:
                    k = rng.randrange(1, len(s_))
                    cut += [s_[:k], s_[k:]]
                else:
                    cut.append(s_)
            body = cut
        chunks = parts + body
        rc, exp = oracle(chunks, True)
        assert rc == 0
        err, ev, _ = emu(lib, chunks, SERVER | FIRST | VARIANTS[variant], gap_rng=rng if trial % 2 else None)
        assert err == 0 and ev == exp, trial


def test_reference_vectors_under_the_emulator(lib):
    """The byte vectors of the reference's bad_client tests (tests/golden/h2_bad_client.json), whole and cut at
    random places, through the emulated kernel with every variant."""
    vecs = json.load(open(os.path.join(ROOT, "tests", "golden", "h2_bad_client.json")))["vectors"]
    rng = random.Random(5)
    for vec in vecs:
        data = bytes.fromhex(vec["hex"])
        for cuts in ([], sorted(rng.sample(range(1, len(data)), min(12, len(data) - 1)))):
            bounds = [0] + cuts + [len(data)]
            chunks = [data[a:b] for a, b in zip(bounds, bounds[1:])]
            rc, exp = oracle(chunks, True)
            for v in VARIANTS.values():
                err, ev, _ = emu(lib, chunks, SERVER | FIRST | v, gap_rng=rng)
                assert err == rc == 0 and ev == exp, vec["name"]
