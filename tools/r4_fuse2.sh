#!/bin/bash
# Round 4: phase stamps of the fused round launch (k_round_xag) next to the paired schedule, then the headline A/B.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-fuse2}; rm -rf $out; mkdir -p $out
echo "== GRDMA_JOB_FUSE_ROUND=1"; GRDMA_JOB_FUSE_ROUND=1 timeout 200 python tools/mw_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/phases_fr1.txt
echo "== paired schedule"; timeout 200 python tools/mw_phases.py 2>&1 | grep -v amdgpu.ids | tee $out/phases_fr0.txt
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for v in 1 0; do
  for w in staged direct; do
    GRDMA_JOB_FUSE_ROUND=$v timeout 200 python bench.py --wire $w $Q > $out/bench_${w}_fr$v.json 2> $out/bench_${w}_fr$v.err < /dev/null
    echo "FUSE_ROUND=$v $w: $(python - <<PY
import json
d=json.loads(open('$out/bench_${w}_fr$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], {k:(v['launches'],v['us_per_launch']) for k,v in r['schedule_kernels'].items()}, 'frac', r['frac'], 'step', r['step_level']['frac'])
PY
)"
  done
done
