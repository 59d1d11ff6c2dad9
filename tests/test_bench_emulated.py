"""bench.py --gpus 2, launched the way the driver launches it -- `python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W` -- and run END TO END on
the CPU: GRDMA_BENCH_EMULATED=1 puts the product sources compiled over the wave emulator (oracle/_build/libgrdma_emu.so)
in the library's place, CPU tensors and gloo in place of HBM tensors and RCCL.  No second GPU has been available to any
round (SCALE_r0*.json: skipped), so this is where the N > 1 path of the script executes: one process per rank, every
rank its own connections (no data-path collective), the barriers and the max-over-ranks of the contract, the
BASELINE configs[3] legs (several connections per rank, one-directional and BIDIRECTIONAL, verified), and the
configs[4] fan-out: rank 0 ingests one stream, its delivered arena goes through ONE grouped send / recv step, and the
ranks' shares -- checksummed where they lie, position-weighted, summed over the ranks -- must be the framed stream.
What it cannot say is how fast anything is: the printed line's `data` says so, `value` is not a measurement.
UNMEASURED ON HARDWARE: DESIGN.md section 7."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "oracle", "_build", "libgrdma_emu.so")
SMALL = ["--msgs", "2", "--payload", "20000", "--ring-kb", "256", "--max-sge", "30", "--conns", "2", "--conn-msgs", "2",
         "--fanout-msgs", "2", "--no-rtt", "--no-cpu-baseline", "--no-tcp-baseline", "--reps", "1"]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def emu_env(built):
    if not os.path.exists(EMU_SO):
        pytest.skip("oracle/_build/libgrdma_emu.so not built (needs the ROCm clang++ as host compiler)")
    return dict(os.environ, GRDMA_BENCH_EMULATED="1", GRDMA_LIB_PATH=EMU_SO, GRDMA_TEST_ALLOW_EMU="1",
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_bench_two_ranks_as_the_driver_launches_it(emu_env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + SMALL
    p = subprocess.run(cmd, cwd=ROOT, env=emu_env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line: %r" % lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "EMULATED" in d["data"] and d["verified"] is True
    assert d["config"]["workload"].startswith("client-streaming") and len(d["config"]["repetitions_ms_per_step"]) == 1
    assert d["roofline"]["bound"] == "hbm" and d["cpu_baseline"] is None
    # BASELINE configs[3]: --conns connections per rank, one direction and both directions of every pair
    assert d["conns2_64KiB_bidi_verified"] is True and d["value_conns2_64KiB_bidi"] >= 0 and "value_conns2_64KiB_ring4096" in d
    # BASELINE configs[4]: the fan-out, checked by checksum over what the ranks hold afterwards
    assert d.get("fanout_error") is None, d.get("fanout_error")
    assert d["fanout_bytes_ok"] is True and d["fanout_checksum_ok"] is True, d.get("fanout_checksum")
    cs = d["fanout_checksum"]
    assert [cs["byte_sum"], cs["position_weighted_sum"]] == cs["expected"] and cs["byte_sum"] > 0


def test_bench_refuses_a_dry_run_of_the_product_library(emu_env):
    """GRDMA_BENCH_EMULATED=1 without the emulated library named: the script stops instead of emulating anything about a
    real device (and without the variable it needs a GPU: torch.cuda.set_device fails loudly on this box)."""
    env = dict(emu_env)
    env.pop("GRDMA_LIB_PATH")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"] + SMALL, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "refusing to dry-run" in (p.stdout + p.stderr)
