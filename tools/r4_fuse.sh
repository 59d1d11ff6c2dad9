#!/bin/bash
# Round 4: the fused round kernel (GRDMA_JOB_FUSE_ROUND=1: k_round_xag) -- stream-job parity on hardware with it, then
# the headline leg staged / direct with and without it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-fuse}; rm -rf $out; mkdir -p $out
GRDMA_JOB_FUSE_ROUND=1 timeout 900 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_link_engine.py -m gpu -q -x > $out/pytest_job.log 2>&1 < /dev/null; echo "stream-job tests (fused round) rc=$?"; tail -3 $out/pytest_job.log
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for v in 1 0; do
  for w in staged direct; do
    GRDMA_JOB_FUSE_ROUND=$v timeout 200 python bench.py --wire $w $Q > $out/bench_${w}_fr$v.json 2> $out/bench_${w}_fr$v.err < /dev/null
    echo "FUSE_ROUND=$v $w: $(grep -o '"value": [0-9.]*' $out/bench_${w}_fr$v.json | head -1)  $(grep -o '"schedule_kernels": {[^}]*}' $out/bench_${w}_fr$v.json | head -1) $(tail -2 $out/bench_${w}_fr$v.err | tr '\n' ' ' | cut -c1-300)"
  done
done
