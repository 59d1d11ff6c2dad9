"""The zero-copy send's record pricing (grpc-rdma_amd/csrc/grdma_zc_core.h, the loop body of k_tx_plan_zc) on
the CPU: the function the kernel calls, run over a host model of the device side (tests/cc/zc_core_host.cc) and
compared call by call with the oracle's SendZerocopy (itself pinned against the reference-built ring codec in
tests/test_oracle_vs_ref.py)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libzc_core_host.so")
SRCS = [os.path.join(ROOT, "tests", "cc", "zc_core_host.cc"),
        os.path.join(ROOT, "grpc-rdma_amd", "csrc", "grdma_zc_core.h"),
        os.path.join(ROOT, "grpc-rdma_amd", "csrc", "grdma_dev.h"),
        os.path.join(ROOT, "oracle", "grdma_oracle.c"), os.path.join(ROOT, "oracle", "grdma_oracle.h")]


@pytest.fixture(scope="module")
def lib():
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS):
        obj = os.path.join(BUILD, "zc_core_oracle.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-c", SRCS[3], "-o", obj])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fPIC", "-shared", SRCS[0], obj, "-o", SO])
    L = C.CDLL(SO)
    L.zc_core_sequence.restype = C.c_int
    L.zc_core_sequence.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    return L


CONFIGS = [(64, 100, 128), (256, 5, 64), (1024, 4, 2048), (4096, 30, 8192), (4096, 3, 64), (65536, 8, 4096),
           (1 << 20, 30, 1 << 21), (128, 7, 256)]


@pytest.mark.parametrize("R,sge,Z", CONFIGS)
def test_zero_copy_pricing_matches_the_oracle(lib, R, sge, Z):
    tot = [0, 0, 0]
    for seed in range(1, 41):
        why = C.create_string_buffer(256)
        rec, zrec, wraps = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib.zc_core_sequence(R, sge, Z, seed * 7919, 60, why, 256, C.byref(rec), C.byref(zrec), C.byref(wraps))
        assert rc == 0, why.value.decode()
        tot = [tot[0] + rec.value, tot[1] + zrec.value, tot[2] + wraps.value]
    # the sequences really went through both kinds of record (sge 3 can never carry a zero-copy record)
    assert tot[0] > 200 and (tot[1] > 50 or sge < 4)
    if R <= 4096:
        assert tot[2] > 0, "no record ever wrapped around the ring end"
