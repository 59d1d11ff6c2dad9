#!/bin/bash
# Round 4: the reference's default knobs (4 MiB ring, max_sge 30) on the paired schedule with the Sends of a round
# folded into one pricing -- parity on hardware, then the leg next to the burst schedule's.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-knobs}; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_stream_job.py -m gpu -q -x -k "sends" > $out/pytest.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -3 $out/pytest.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, json, time
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-tcp-baseline", "--no-rtt", "--no-small-ring", "--steps", "8", "--warmup", "2"]
import bench
# reuse bench's main but only the legs of interest: run the full extra legs would take minutes -- call measure() via a tiny shim
PY
Q="--no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 8 --warmup 2 --conns 1"
timeout 900 python bench.py $Q > $out/bench.json 2> $out/bench.err < /dev/null
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
for k in d:
    if "ring4096" in k or k in ("value", "value_mixed_sizes"):
        print(k, json.dumps(d[k])[:200])
PY
