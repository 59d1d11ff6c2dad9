#!/bin/bash
# Round 3 record: the gpu suite (gate), the full bench line, planner phases, the profile set of tools/prof_all.sh
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final3
rm -rf $out; mkdir -p $out
cd $R
timeout 600 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
rc=$?; tail -3 $out/pytest.log; echo "pytest rc=$rc"
timeout 500 python bench.py > $out/bench_full.json 2> $out/bench_full.err < /dev/null
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/final3/bench_full.json").read().strip().splitlines()[-1])
    for k in d:
        if k.startswith("value") or k in ("ms_per_step","repetitions","roofline","conn_setup_us","rtt_p50_us","rtt_armed_read_p50_us","verified","with_h2_stages","kernels"):
            print(k, json.dumps(d[k])[:420])
except Exception as e:
    print("parse failed", e)
PY
timeout 60 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null; tail -3 $out/phases.log
timeout 300 bash tools/prof_all.sh < /dev/null 2>&1 | tail -30
