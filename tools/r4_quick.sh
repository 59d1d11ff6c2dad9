#!/bin/bash
# Quick trip: stream-job / h2 / link-engine parity on hardware, then the headline leg staged / direct.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${1:-quick}; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_h2.py tests/test_zz_gpu_h2_chunks.py tests/test_gpu_link_engine.py -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -3 $out/pytest.log
Q="--no-extra-legs --no-cpu-baseline --no-tcp-baseline --no-rtt --no-small-ring --steps 20 --warmup 3"
for w in staged direct; do timeout 300 python bench.py --wire $w $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$w', d['value'], d['ms_per_step'], {k:(v['launches'],v['us_per_launch']) for k,v in r['schedule_kernels'].items()}, 'frac', r['frac'], 'step', r['step_level']['frac'], r.get('dominant_by_time'))"; done
