import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)  # shared helper modules of the tests (h2_helpers)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # A GPU test never needs minutes.  If a kernel ever wedged, end the run (pytest-timeout's thread method exits the
    # process, which also releases the device) instead of sitting in a synchronize until the caller's limit.
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    """Initialised HIP data plane; fails loudly when there is no device."""
    import grpc_rdma_amd as g
    g.init(0)
    # Which library is this suite testing?  The product (grpc-rdma_amd/libgrdma_amd.so, built in-tree) -- unless the
    # run SAYS it is the emulated one (tests/test_emu_gpu_suite.py, tools/emu_site.sh set GRDMA_TEST_ALLOW_EMU=1 next
    # to GRDMA_LIB_PATH).  A stray GRDMA_LIB_PATH must not turn the hardware suite into an emulation run.
    product = os.path.realpath(os.path.join(ROOT, "grpc-rdma_amd", "libgrdma_amd.so"))
    mapped = set()
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split()[-1]
            if "libgrdma" in path:
                mapped.add(os.path.realpath(path))
    if os.environ.get("GRDMA_TEST_ALLOW_EMU") != "1":
        assert mapped == {product}, "the gpu suite must load the in-tree product library only, found %s" % sorted(mapped)
    return g
