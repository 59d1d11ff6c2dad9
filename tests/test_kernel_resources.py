"""What the planner pair's kernel may cost the launcher: no scratch segment.

k_plan_pair_mw holds five bodies (two drain planners, two send planners, the wire mover) at one wave per SIMD and the full
register file; round 6 met a version of it whose drain body indexed a register array through a loop variable -- 72 bytes of
scratch per lane -- and every launch of the kernel paid 1-2 us for it (the reference-default-knob leg 68 -> 59 GiB/s,
DESIGN.md section 2.7).  The compiler says what it allocated (-Rpass-analysis=kernel-resource-usage); this pins it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def test_the_planner_pairs_kernel_has_no_scratch_segment(tmp_path):
    src = os.path.join(ROOT, "grpc-rdma_amd", "csrc", "grdma_rx_plan.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
           "--cuda-device-only", "-c", src, "-o", str(tmp_path / "rx_plan.co"), "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    text = p.stderr
    at = text.find("k_plan_pair_mw")
    assert at >= 0, "no resource remarks for k_plan_pair_mw"
    block = text[at:at + 4000]
    got = {k: int(v) for k, v in re.findall(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)", block)[:12]}
    print(got)
    assert got.get("ScratchSize") == 0, got
    assert got.get("VGPRs Spill") == 0, got
    assert got.get("LDS Size", 1 << 30) <= 160 * 1024, got
