#!/bin/bash
# Round 4: two consecutive Sends per round in one plan (grdma_stream_job_set_sends) -- parity on hardware, then the
# headline leg with 2 and with 1 Send per round, staged and direct.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-sends}; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_stream_job.py -m gpu -q -x > $out/pytest_job.log 2>&1 < /dev/null; echo "stream-job tests rc=$?"; tail -3 $out/pytest_job.log
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for v in 2 1; do
  for w in staged direct; do
    timeout 300 python bench.py --wire $w --sends $v $Q > $out/bench_${w}_s$v.json 2> $out/bench_${w}_s$v.err < /dev/null
    echo "sends=$v $w: $(python - <<PY
import json
try:
    d=json.loads(open('$out/bench_${w}_s$v.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['value'], 'rounds', d['config']['rounds_per_step'], {k:(v['launches'],v['us_per_launch']) for k,v in r.get('schedule_kernels',{}).items()}, 'frac', r['frac'], 'step', r['step_level']['frac'], 'verified', d['verified'])
except Exception as e:
    print('failed', e, open('$out/bench_${w}_s$v.err').read()[-600:])
PY
)"
  done
done
