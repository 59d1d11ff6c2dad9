#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-sends2}; rm -rf $out; mkdir -p $out
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for cfg in "262144 2" "262144 1" "524288 2"; do
  set -- $cfg
  for w in staged direct; do
    timeout 300 python bench.py --wire $w --ring-kb $1 --sends $2 $Q > $out/bench_${w}_r$1_s$2.json 2> $out/bench_${w}_r$1_s$2.err < /dev/null
    echo "ring $1 KiB sends=$2 $w: $(python - <<PY
import json
try:
    d=json.loads(open('$out/bench_${w}_r$1_s$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(d['value'], 'rounds', d['config']['rounds_per_step'], {k:(v['launches'],v['us_per_launch']) for k,v in r.get('schedule_kernels',{}).items()}, 'frac', r['frac'], 'step', r['step_level']['frac'], 'dominant', r.get('dominant_by_time',{}).get('kernel'), r.get('dominant_by_time',{}).get('share_of_kernel_time'), 'verified', d['verified'])
except Exception as e:
    print('failed', e, open('$out/bench_${w}_r$1_s$2.err').read()[-600:])
PY
)"
  done
done
