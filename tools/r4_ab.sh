#!/bin/bash
# A/B on ONE box: the library of the commit before (grpc-rdma_amd/_exp/libgrdma_prev.so, built by hand) against the
# tree's, paired schedule: phase stamps and the headline leg, twice each, interleaved.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-ab}; rm -rf $out; mkdir -p $out
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for rep in 1 2; do
  for v in prev tree; do
    if [ $v = prev ]; then export GRDMA_LIB_PATH=$R/grpc-rdma_amd/_exp/libgrdma_prev.so; else unset GRDMA_LIB_PATH; fi
    echo "== $v ($rep)"; timeout 200 python tools/mw_phases.py 2>&1 | grep "us per launch\|drain plan" | cut -c1-230
    timeout 200 python bench.py --wire staged $Q 2>/dev/null > $out/b_${v}_$rep.json
    python - <<PY
import json
d=json.loads(open('$out/b_${v}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], {k:(v['launches'],v['us_per_launch']) for k,v in r['schedule_kernels'].items()})
PY
  done
done
