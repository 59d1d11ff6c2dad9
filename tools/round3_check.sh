#!/bin/bash
# after the planner micro-optimisations: parity, headline variants, phases, kernel timeline, then the full bench line
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/check3
rm -rf $out; mkdir -p $out
cd $R
timeout 300 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_link_engine.py -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5"
run() { tag=$1; shift; timeout 120 env "$@" $B $EXTRA > $out/$tag.log 2> $out/$tag.err < /dev/null; echo "$tag rc=$? $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*\|"rx_plan": {[^}]*}\|"tx_plan": {[^}]*}' $out/$tag.log | tr '\n' ' ')"; grep -v amdgpu.ids $out/$tag.err | tail -2; }
EXTRA=""
run pairjob X=1
run nopair GRDMA_PAIR_JOB=0
EXTRA="--pipeline 0"
run sequential X=1
EXTRA="--wire direct"
run direct X=1
timeout 60 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null
tail -3 $out/phases.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o t -- $B --steps 6 --warmup 2 --no-verify --pipeline 0 > $out/tr.stdout 2>&1
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $f 30 > $out/timeline_seq.txt; tail -14 $out/timeline_seq.txt
rm -rf $out/tr
cd $R
timeout 400 python bench.py > $out/bench_full.json 2> $out/bench_full.err < /dev/null
echo "full bench rc=$?"
python - <<'PY'
import json,sys
try:
    d=json.loads(open("gpurun_out/check3/bench_full.json").read().strip().splitlines()[-1])
    keys=[k for k in d if k.startswith("value") or k.startswith("rtt") or k in ("ms_per_step","cpu_baseline","roofline")]
    for k in keys: print(k, json.dumps(d[k])[:300])
except Exception as e:
    print("parse failed", e)
PY
