"""The GPU parity tests on the CPU: the product sources compiled over the wave emulator and the HIP API stand-in
(tests/cc/wave_emu.h, tests/cc/hip_api_emu.h, tests/cc/build_emu.sh -> oracle/_build/libgrdma_emu.so), loaded in
place of libgrdma_amd.so (GRDMA_LIB_PATH), and the `-m gpu` tests run against it in a child pytest.

What runs here: the deframer (all of tests/test_gpu_h2.py but the link-engine pipeline: the boundary step, 64 frames
per bulk step, frame -> job -> deframe as a HIP graph), the zero-copy send (tests/test_zz_gpu_zerocopy.py: k_tx_plan_zc + k_copy + the host API), and the
pair protocol on random operation sequences and the reference-generated golden traces (k_tx_plan, k_copy, k_rx_plan,
k_rx_apply, k_poll), and the receive planner's multi-record drains.  What the emulator cannot run is deselected:
resident kernels of more than one workgroup per launch.  The streaming jobs (HIP graphs of kernel
nodes, run node by node) work too but take minutes: GRDMA_LIB_PATH=oracle/_build/libgrdma_emu.so python -m pytest
tests/test_gpu_stream_job.py -m gpu -k r256k.

This checks kernel LOGIC when no GPU is at hand; the GPU runs stay the reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
EMU_SO = os.path.join(ROOT, "oracle", "_build", "libgrdma_emu.so")

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ as host compiler")


@pytest.fixture(scope="module")
def emu_lib(built):
    srcs = []
    for d in (os.path.join(ROOT, "grpc-rdma_amd", "csrc"), os.path.join(ROOT, "tests", "cc"), os.path.join(ROOT, "include")):
        srcs += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".cc", ".h", ".hpp", ".sh", ".inc"))]
    if not os.path.exists(EMU_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMU_SO) for s in srcs):
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "cc", "build_emu.sh")], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return EMU_SO


def run_gpu_tests(emu_lib, args, min_passed):
    env = dict(os.environ, GRDMA_LIB_PATH=emu_lib, GRDMA_TEST_NEW="1", GRDMA_TEST_ALLOW_EMU="1")
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--timeout", "600"] + args
    last = ""
    for attempt in range(2):  # (the staging waves of the deframer are real threads: one retry on a scheduling hiccup)
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        last = (p.stdout + p.stderr)[-3000:]
        if p.returncode == 0:
            tail = p.stdout.strip().splitlines()[-1]
            n = int(tail.split(" passed")[0].split()[-1])
            assert n >= min_passed, tail
            return
    raise AssertionError(last)


def test_deframer_gpu_tests_under_the_emulator(emu_lib):
    run_gpu_tests(emu_lib, ["tests/test_gpu_h2.py", "tests/test_zz_gpu_h2_boundary.py"], 32)


def test_chunked_deframer_gpu_tests_under_the_emulator(emu_lib):
    """The deframer over chunks (k_h2_deframe_chunks / k_h2_merge_or_deframe): merged lists, empty chunks, lists the
    chain verification declines; all of tests/test_zz_gpu_h2_chunks.py but the bench-sized pipe (256 MiB per step)."""
    run_gpu_tests(emu_lib, ["tests/test_zz_gpu_h2_chunks.py", "-n", "4", "-k", "not bench_configuration"], 10)


def test_zero_copy_gpu_tests_under_the_emulator(emu_lib):
    run_gpu_tests(emu_lib, ["tests/test_zz_gpu_zerocopy.py"], 10)


def test_steady_state_planner_and_pair_pool_gpu_tests_under_the_emulator(emu_lib):
    """Round 3: the job kernels (k_rx_plan_job / k_tx_plan_job: steady-state bodies with the general
    planners behind them, csrc/grdma_rx_fast.h, grdma_tx_fast.h) on periodic streams cut by max_sge, sequential and
    pipelined graphs against the oracle's rounds; the paired graph at a credit-limited ring against the oracle driven with
    the credit one round late; the PairPool on recycled memory."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_stream_job.py", "tests/test_gpu_pair_pool.py", "-n", "4",
                            "-k", "(fast_planner and sge130) or pool or (credit_limited and r256k)"], 8)


def test_planner_pair_of_many_workgroups_and_bidirectional_job_under_the_emulator(emu_lib):
    """Round 4: k_plan_pair_mw (csrc/grdma_rx_multi.h, grdma_tx_multi.h: the drain plan and the send plan of a round laid
    out by sixteen four-wave workgroups each from closed forms over the record pattern / a search over the slice index,
    nothing exchanged but the arrival word) on drains of up to 3001 records that span workgroups, wrap the ring and
    cross the credit threshold; the bidirectional job (both directions of one pair in every launch), paired schedule."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_stream_job.py", "-n", "4",
                            "-k", "(several_workgroups and r8m_sge3001) or (bidirectional and paired)"], 4)


def test_several_sends_per_plan_and_promised_credit_under_the_emulator(emu_lib):
    """Round 4, second half: a round's plan holding two consecutive Sends priced one after the other, or the Sends of a
    round folded into one cut of the slice table's index (csrc/grdma_tx_multi.h) -- on the paired and on the sequential
    schedule, at the reference's max_sge of 30 --, and the paired schedule with the promised
    credit (k_plan_pair_mw: the Send waits for the drain plan of its launch) against the oracle's plain rounds at rings
    every round fills."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_stream_job.py", "-n", "4",
                            "-k", "(two_sends and r16m_small and staged) or (many_sends and r32m_sge30x8 and staged) or "
                                  "(paired_schedule_with_promised and (r256k_sge30 or r1m_sge64x2))"], 5)


def test_round6_parity_cases_under_the_emulator(emu_lib):
    """Round 6: one workgroup of the multi-workgroup drain plan declining while its neighbours accept (a record that
    changes payload inside its encoded size: the general planner rewrites the plan in the same launch, the verdict
    counter shows the mix); the promised-credit hand-over through the plan's 64-bit word (here the drain's workgroups run
    first, so the promise is always kept -- the wait that runs out is provoked by the test's knob); the wire of a round
    inside the planner pair's launch against the same job with a wire launch of its own, both against the oracle; the
    reference's mixed message sizes at max_sge 30 on that schedule against the oracle's plain rounds; rounds of more than
    4096 records without a period (the second half of the size table, the period search that is due).
    (BASELINE configs[3] bidirectional, every link against the oracle, takes two minutes here: MI355X only --
    GRDMA_LIB_PATH=oracle/_build/libgrdma_emu.so python -m pytest tests/test_gpu_stream_job.py -m gpu -k "config3 and pairs2".)"""
    run_gpu_tests(emu_lib, ["tests/test_gpu_stream_job.py", "-n", "4",
                            "-k", "(one_drain_workgroup and r8m and staged) or (runs_out and r256k) or (wire_inside and r256k) or "
                                  "mixed_message_sizes or (without_a_period and x2_small and staged)"], 7)


def test_concurrent_writer_and_poller_gpu_tests_under_the_emulator(emu_lib):
    """Records landing header-first / footer-last from a second thread while the receiver polls and reads; the
    background poller thread (one k_poll launch per pass, eventfd wakeups)."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_concurrent_writer.py", "tests/test_gpu_poller.py"], 5)


def test_latency_engine_gpu_tests_under_the_emulator(emu_lib):
    """k_engine, the resident kernel behind bench.py's RTT leg: under emulation it runs in a thread of its own and
    serves the mailbox like on the device."""
    run_gpu_tests(emu_lib, ["tests/test_zz_gpu_latency_engine.py"], 3)


def test_arrival_triggered_reads_under_the_emulator(emu_lib):
    """Round 5: k_watch -- the watcher workgroup that carries out a pair's standing read order when bytes land in its
    ring (a second resident thread here), its single-wave drain (rxw_fast) and the plan body behind it, on unary
    ping-pongs, random sequences over staged / direct / fine-grained / wrapping rings, an ordered wire, an engine that
    is stopped and started: bytes, state, histories and rings against the oracle (tests/test_zzz_gpu_watch_read.py)."""
    run_gpu_tests(emu_lib, ["tests/test_zzz_gpu_watch_read.py", "-n", "4"], 13)


def test_nic_wire_back_end_over_the_verbs_stand_in(emu_lib):
    """Round 5: csrc/grdma_wire_verbs.cc -- memory registration (the ring through the dma-buf call), queue-pair bring-up,
    the <= 2 chained RDMA WRITEs of a Send, the 16-byte status write, completion reaping -- compiled against
    oracle/fakeverbs and driven by the reference-made endpoint traces: results step by step, write requests and ring
    image after every Send against the oracle (tests/test_zz_gpu_wire_verbs.py; skipped on the GPU box, whose library
    is built without <infiniband/verbs.h>)."""
    run_gpu_tests(emu_lib, ["tests/test_zz_gpu_wire_verbs.py", "-n", "4"], 5)


def test_two_process_gpu_tests_under_the_emulator(emu_lib):
    """Round 6: tests/test_gpu_two_process.py with BOTH processes emulated -- the emulator keeps fine-grained device
    memory in anonymous shared-memory files and its HIP IPC handles name them (tests/cc/hip_api_emu.h), so the server
    process and the client process map each other's ring and connection block as two processes on one GPU do.  The
    bootstrap over the inherited socket (48-byte Address + memory handles, rdma_bp_posix.cc:640-692, 763-784), the
    endpoint conformance grid echoed across the process boundary on a staged wire, and GRDMA_WIRE_DIRECT encoding
    straight into the mapped peer ring (the cross-GPU pair's code path): the world-2 stand-in of the CPU suite."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_two_process.py", "-n", "4", "-k",
                            "(echo_between_two_processes and not gpus) or connect_checks or bootstrap_path"], 8)


def test_pair_protocol_gpu_tests_under_the_emulator(emu_lib):
    """All of tests/test_gpu_pair_parity.py: random operation sequences in the four wire / memory modes, the golden
    traces, batched polling, the multi-record drains of k_rx_plan (chain walker, one-lane-per-record replay, bulk tier
    with its period predictor), latency mode (single-launch small sends, express drain), the unary ping-pong."""
    run_gpu_tests(emu_lib, ["tests/test_gpu_pair_parity.py", "-n", "4"], 55)


def test_endpoint_vtable_tools_under_the_emulator(emu_lib, tmp_path):
    """tools/endpoint_pingpong (unary round trips through grpc_endpoint_write / _read: the blocking ABI, the resident
    engine, the engine with standing reads carried out by a watcher) and tools/endpoint_stream in latency mode, against the emulated library
    (the binaries link libgrdma_amd.so: a link of that name in front of their run path); both check the bytes."""
    import json
    os.symlink(emu_lib, str(tmp_path / "libgrdma_amd.so"))
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               GRPC_RDMA_RING_BUFFER_SIZE_KB="4096")
    pp = os.path.join(ROOT, "tools", "endpoint_pingpong")
    for mode in (0, 1, 2):
        p = subprocess.run([pp, "20", "64", str(mode)], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-500:]
        r = json.loads(p.stdout.strip().splitlines()[-1])
        # (mode 2: every completion of the 27 round trips came from a watcher workgroup -- a second resident thread here)
        assert r["checked"] and r["mode"] == mode and r["watch_hits"] == (2 * (20 + 7) if mode == 2 else 0), r
    # a message larger than the inline command and the fast lane: [14 B][3000 B], the pointer path of the engine
    p = subprocess.run([pp, "10", "3000", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["watch_hits"] > 0, p.stderr[-500:]
    es = os.path.join(ROOT, "tools", "endpoint_stream")
    p = subprocess.run([es, "5", str(1 << 20), "1", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["checked"] and r["latency_mode"] and r["endpoint_bytes"] > 5 << 20
    # the launch-chain path with the endpoint's send buffers: write k + 1 is queued behind the Sends of write k
    # (grdma_endpoint_write_queue) and promoted when write k has gone out whole; a 256 KiB ring makes some writes come up
    # short, which skips the queued chain on the device and submits it again the ordinary way -- same bytes either way
    for ring_kb, want in (("4096", "promoted"), ("256", "skipped")):
        # (GRDMA_WRITE_QUEUE_ALWAYS: queue even when the state line says the ring has no room for both writes)
        p = subprocess.run([es, "12", str(1 << 20), "1", "0", "2"],
                           env=dict(env, GRPC_RDMA_RING_BUFFER_SIZE_KB=ring_kb, GRDMA_WRITE_QUEUE_ALWAYS="1"),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-500:]
        r = json.loads(p.stdout.strip().splitlines()[-1])
        queued, promoted, skipped = r["writes_queued"]
        assert r["checked"] and not r["latency_mode"] and queued >= 1 and promoted + skipped <= queued, r
        assert (promoted if want == "promoted" else skipped) >= 1, r
    # round 5: the send buffer that waits COALESCES -- writes that arrive while a Send is in flight share it (up to the
    # buffer, sixteen Sends, a quarter of the ring: three 1 MiB messages at a 64 MiB ring) and go out as one chain;
    # GRPC_RDMA_HIP_COALESCE=0 gives every write a chain of its own.  Same bytes either way.
    chains = {}
    for co in ("1", "0"):
        p = subprocess.run([es, "24", str(1 << 20), "1", "0", "2"],
                           env=dict(env, GRPC_RDMA_RING_BUFFER_SIZE_KB="65536", GRPC_RDMA_HIP_COALESCE=co),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-500:]
        r = json.loads(p.stdout.strip().splitlines()[-1])
        assert r["checked"] and r["endpoint_bytes"] > 24 << 20, r
        chains[co] = r["writes_queued"][0]
    assert 1 <= chains["1"] <= 12 and chains["0"] >= chains["1"] + 4, chains
    # the same, randomised (ENDPOINT_STREAM_SEED: writes of 2 .. 130 slices, a reader that stalls now and then)
    for seed, ring_kb, always in ((1, "1024", "1"), (3, "4096", "0")):
        p = subprocess.run([es, "24", str(1 << 20), "1", "0", "2"],
                           env=dict(env, GRPC_RDMA_RING_BUFFER_SIZE_KB=ring_kb, GRDMA_WRITE_QUEUE_ALWAYS=always,
                                    ENDPOINT_STREAM_SEED=str(seed)), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-500:]
        r = json.loads(p.stdout.strip().splitlines()[-1])
        queued, promoted, skipped = r["writes_queued"]
        assert r["checked"] and queued >= 4 and promoted + skipped <= queued and promoted + skipped >= 4, r


def test_endpoint_conformance_under_the_emulator(emu_lib, tmp_path):
    """tests/cc/endpoint_conformance -- the reference's endpoint test shape over the vtable mirror, i.e. over the
    endpoint logic shared with the drop-in for the gRPC tree (include/grdma_endpoint_impl.hpp): shutdown / half close,
    a small ring that parks every write on credit, and a pollset of 16 endpoints in both platforms (RDMA_BPEV sleeps
    in epoll_wait and is woken by the poller threads).  Asynchronous Sends and drains, receive windows, zero-copy
    slices: all of it runs here, on the emulated device."""
    os.symlink(emu_lib, str(tmp_path / "libgrdma_amd.so"))
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    h = os.path.join(ROOT, "tests", "cc", "endpoint_conformance")

    def run(args, ring_kb):
        p = subprocess.run([h] + [str(a) for a in args], env=dict(env, GRPC_RDMA_RING_BUFFER_SIZE_KB=str(ring_kb)),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-1000:]
        return p.stdout
    out = run(["multiple_shutdown"], 4096)
    assert "multiple_shutdown_test: ok" in out and "half_close_test: ok" in out and "write_after_peer_exit_test: ok" in out
    assert ": ok" in run([300000, 100000, 8192, 0], 64)
    out = run(["pollset", 8, 2, 0], 256)
    assert ": ok" in out and "device polls 0" in out   # the busy-poll pass is host loads only
    out = run(["pollset", 8, 2, 1, 2], 256)             # two threads inside pollset_work
    assert ": ok" in out and "epoll waits 0" not in out
