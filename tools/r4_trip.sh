#!/bin/bash
# Round 4 work trips: stream-job parity on hardware, then the headline leg staged / direct with the new planner kernels
# against the round-3 ones (GRDMA_RX_MULTI=0), then the per-launch split of the timed schedule.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-trip}; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_stream_job.py -m gpu -q -x > $out/pytest_job.log 2>&1 < /dev/null; echo "stream-job tests rc=$?"; tail -3 $out/pytest_job.log
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for v in 1 0; do
  for w in staged direct; do
    GRDMA_RX_MULTI=$v timeout 200 python bench.py --wire $w $Q > $out/bench_${w}_mw$v.json 2> $out/bench_${w}_mw$v.err < /dev/null
    echo "RX_MULTI=$v $w: $(grep -o '"value": [0-9.]*' $out/bench_${w}_mw$v.json | head -1)  $(grep -o '"schedule_kernels": {[^}]*}' $out/bench_${w}_mw$v.json | head -1)"
  done
done
timeout 100 python tools/plan_phases.py 2>&1 | tail -3 | tee $out/phases.txt
