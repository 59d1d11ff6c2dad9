#!/bin/bash
# the gpu suite (gate) and the full bench line -> gpurun_out/gate3/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/gate3
rm -rf $out; mkdir -p $out
cd $R
timeout 700 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
rc=$?; tail -3 $out/pytest.log; echo "pytest rc=$rc"
timeout 600 python bench.py > $out/bench_full.json 2> $out/bench_full.err < /dev/null
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/gate3/bench_full.json").read().strip().splitlines()[-1])
    for k in d:
        if k.startswith("value") or k in ("ms_per_step","repetitions","roofline","conn_setup_us","rtt_p50_us","rtt_armed_read_p50_us","verified","with_h2_stages","kernels","rtt_endpoint_vtable_us") or k.endswith("_error"):
            print(k, json.dumps(d[k])[:420])
except Exception as e:
    print("parse failed", e)
PY
tail -5 $out/bench_full.err | grep -v amdgpu.ids
