"""Self-test of the wave emulator (tests/cc/wave_emu.h): cross-lane primitives, the DPP prefix sums and the block-wide
scan of csrc/grdma_devfn.h against closed forms, LDS hand-off across waves behind a barrier, lanes that leave early,
and the diagnostic for a cross-lane operation reached from two places by the lanes of one wave."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SO = os.path.join(ROOT, "oracle", "_build", "libwave_emu_selftest.so")
SRCS = [os.path.join(ROOT, "tests", "cc", f) for f in ("wave_emu_selftest.cc", "wave_emu.h", "hip_api_emu.h")] + \
       [os.path.join(ROOT, "grpc-rdma_amd", "csrc", "grdma_devfn.h")]

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ as host compiler")


@pytest.fixture(scope="module")
def so():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRCS):
        subprocess.check_call([CLANG, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unused-value",
                               "-I" + os.path.join(ROOT, "tests", "cc"), "-I" + os.path.join(ROOT, "tests", "cc", "emu_include"),
                               SRCS[0], "-o", SO])
    return SO


def test_primitives_scans_and_handoffs(so):
    L = C.CDLL(so)
    L.emu_selftest.argtypes = [C.c_int]
    assert L.emu_selftest(0) == 0


def test_divergent_cross_lane_operation_is_reported(so):
    p = subprocess.run([sys.executable, "-c", "import ctypes; ctypes.CDLL(%r).emu_selftest(99)" % so],
                       capture_output=True, text=True)
    assert p.returncode != 0
    assert "is at the cross-lane operation called from" in p.stderr
