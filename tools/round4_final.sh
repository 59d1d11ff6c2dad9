#!/bin/bash
# Round 4 record: the gpu suite, the full bench line, the planner pair's phase stamps, the unary round trip's split,
# the profile sets -> gpurun_out/{gate4,profiles_new,prof_job}.  What is kept is copied into profiles/r04_*.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/gate4
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1 < /dev/null
rc=$?; tail -3 $out/pytest.log; echo "pytest rc=$rc"
timeout 700 python bench.py > $out/bench_full.json 2> $out/bench_full.err < /dev/null
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/gate4/bench_full.json").read().strip().splitlines()[-1])
    for k in d:
        if k.startswith("value") or k in ("ms_per_step","repetitions","roofline","conn_setup_us","rtt_p50_us","rtt_armed_read_p50_us","verified","with_h2_stages","kernels","rtt_endpoint_vtable_us") or k.endswith("_error"):
            print(k, json.dumps(d[k])[:420])
except Exception as e:
    print("parse failed", e)
PY
tail -5 $out/bench_full.err | grep -v amdgpu.ids
timeout 100 python tools/mw_phases.py > $out/mw_phases.txt 2>&1 < /dev/null; tail -6 $out/mw_phases.txt
timeout 100 python tools/rtt_probe.py 20000 > $out/rtt_probe.txt 2>&1 < /dev/null; timeout 100 python tools/rtt_probe.py 20000 armed >> $out/rtt_probe.txt 2>&1 < /dev/null; grep -v amdgpu $out/rtt_probe.txt
./tools/trip_probe > $out/trip_probe.txt 2>&1; head -3 $out/trip_probe.txt
timeout 500 bash tools/prof_all.sh < /dev/null 2>&1 | tail -30
timeout 200 bash tools/prof_job.sh 40 < /dev/null 2>&1 | tail -44
