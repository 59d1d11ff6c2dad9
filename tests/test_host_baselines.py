"""The host-side baselines bench.py quotes next to the GPU numbers (tcp_baseline, BASELINE.md's
"reference TCP endpoint timed on the same box's host cores") run and report what bench.py reads:
oracle/tcp_floor.c (raw sendmsg / recvmsg floor of the TCP platform, tcp_posix.cc) and
oracle/grpcio_loopback.py (a real gRPC stack over loop-back).  Sizes are tiny here: this checks
the plumbing, the bench takes the measurements."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR = os.path.join(ROOT, "oracle", "_build", "tcp_floor")


def _run(cmd, timeout=60):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0 and ("connect" in r.stderr or "bind" in r.stderr or "socket" in r.stderr):
        pytest.skip("no loop-back networking in this sandbox: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_tcp_floor_stream_and_pingpong(built):
    assert os.path.exists(FLOOR), "oracle/Makefile builds it (make -C oracle oracle)"
    s = _run([FLOOR, "stream", "50", str(1 << 20)])
    assert s["mode"] == "stream" and s["msgs"] == 50 and s["iov_per_sendmsg"] == 130 and s["GiBps"] > 0 and s["threads"] == 2
    p = _run([FLOOR, "pingpong", "2000", "80"])
    assert p["mode"] == "pingpong" and p["iters"] == 2000 and 0 < p["p50_us"] <= p["p99_us"]


def test_grpcio_loopback_stream_and_unary():
    pytest.importorskip("grpc")
    script = os.path.join(ROOT, "oracle", "grpcio_loopback.py")
    try:
        s = _run([sys.executable, script, "stream", "0.5", str(1 << 20)], timeout=120)
        u = _run([sys.executable, script, "unary", "0.5", "66"], timeout=120)
    except subprocess.TimeoutExpired:
        pytest.skip("grpcio could not reach its own loop-back server in this sandbox")
    assert s["mode"] == "stream" and s["msgs"] >= 1 and s["GiBps"] > 0 and s["nproc"] >= 1
    assert u["mode"] == "unary" and u["iters"] >= 1 and u["p50_us"] > 0


def test_bench_helpers_parse_helper_output(tmp_path):
    """bench.run_json keeps the bench line alive when a helper is missing or fails."""
    sys.path.insert(0, ROOT)
    import bench
    assert "error" in bench.run_json([str(tmp_path / "does_not_exist")], 5)
    ok = bench.run_json([sys.executable, "-c", "print('noise'); print('{\"GiBps\": 1.5}')"], 20)
    assert ok == {"GiBps": 1.5}
    bad = bench.run_json([sys.executable, "-c", "import sys; sys.exit(3)"], 20)
    assert "error" in bad and "rc 3" in bad["error"]
