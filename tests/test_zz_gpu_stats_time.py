"""GPU: the GRPCProfiler mirror records where rdma_bp_posix.cc and pair.cc place their scopes (the endpoint
conformance harness with GRDMA_PROFILE=1).  Host-side statistics are covered by tests/test_stats_time.py."""
import pytest

from tests.test_gpu_endpoint_conformance import run

pytestmark = pytest.mark.gpu


def test_profile_table_carries_the_reference_op_names(gpu):
    """GRDMA_PROFILE=1: the harness records into profiler slot 0 and prints the table of
    grpc_stats_time_print; the endpoint mirror and the pair record under the names rdma_bp_posix.cc and
    pair.cc use (stats_time.h:11-44)."""
    out = run(1000000, 100000, 8192, 0, env={"GRDMA_PROFILE": "1"})
    assert ": ok" in out and "Profiling Result" in out and "Slot: 0" in out
    rows = {ln.split("|")[1].strip(): int(ln.split("|")[2]) for ln in out.splitlines()
            if ln.startswith("| ") and not ln.startswith("| Name")}
    for name in ("TRANSPORT_WRITE", "TRANSPORT_FLUSH", "TRANSPORT_READ", "TRANSPORT_HANDLE_READ",
                 "TRANSPORT_CONTINUE_READ", "TRANSPORT_DO_READ", "PAIR_SEND", "PAIR_RECV"):
        assert rows.get(name, 0) > 0, (name, rows)
    # (writes that arrive while a Send is in flight share the send buffer that waits and go out as ONE Send, round 5)
    assert rows["TRANSPORT_WRITE"] >= 10 and rows["PAIR_SEND"] >= 1
    # every write a Send of its own, as the reference's rdma_flush: GRPC_RDMA_HIP_COALESCE=0
    out = run(1000000, 100000, 8192, 0, env={"GRDMA_PROFILE": "1", "GRPC_RDMA_HIP_COALESCE": "0"})
    rows = {ln.split("|")[1].strip(): int(ln.split("|")[2]) for ln in out.splitlines()
            if ln.startswith("| ") and not ln.startswith("| Name")}
    assert rows["TRANSPORT_WRITE"] >= 10 and rows["PAIR_SEND"] >= rows["TRANSPORT_WRITE"]
