"""GPU: the grpc_endpoint mirror (include/grdma_endpoint.hpp over the C ABI) under the
reference's endpoint conformance test shape (test/core/iomgr/endpoint_tests.cc:341-355),
run as a C++ binary so the test reads like the reference's own."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cc", "endpoint_conformance")


def run(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([HARNESS] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_multiple_shutdown_and_half_close(gpu):
    out = run("multiple_shutdown")
    assert "multiple_shutdown_test: ok" in out and "half_close_test: ok" in out
    assert "write_after_peer_exit_test: ok" in out


@pytest.mark.parametrize("num_bytes,write_size,slice_size,shutdown", [
    (10000000, 100000, 8192, 0),   # endpoint_tests.cc:345 as is
    (100000, 10000, 1, 0),         # :346 scaled 10x down (1-byte slices: one ring record each)
    (1000000, 100000, 1, 1),       # :347 scaled 100x down, shutdown right after the first read
])
def test_read_and_write(gpu, num_bytes, write_size, slice_size, shutdown):
    assert ": ok" in run(num_bytes, write_size, slice_size, shutdown)


def test_write_equals_slice_sweep(gpu):
    out = run("sweep", 1, 1000)
    assert out.count(": ok") >= 25


def test_small_ring_forces_partial_writes(gpu):
    """64 KiB ring: every 100 kB write needs several rdma_flush retries through the
    writable edge (notify_on_write), and the credit protocol keeps cycling."""
    assert ": ok" in run(2000000, 100000, 8192, 0, env={"GRPC_RDMA_RING_BUFFER_SIZE_KB": "64"})


@pytest.mark.parametrize("bpev", [0, 1], ids=["RDMA_BP", "RDMA_BPEV"])
def test_pollset_with_64_connections(gpu, bpev):
    """128 endpoints (64 client / server pairs) in ONE pollset, driven only by
    grdma_pollset_work(): seeded message sizes, server echoes, clients check the i % 256
    pattern.  RDMA_BP busy-polls (one k_poll launch per pass over all 128 fds); RDMA_BPEV
    busy-polls 200 us, then sleeps in epoll_wait on the pairs' wakeup fds, which the background
    poller thread signals."""
    out = run("pollset", 64, 3, bpev, env={"GRPC_RDMA_RING_BUFFER_SIZE_KB": "256"})
    assert ": ok" in out

