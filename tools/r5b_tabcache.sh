#!/bin/bash
# the read-state table cache of rxm_body: stream-job parity tests, planner phases, A/B against the library before it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_bench_configs.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 100 python tools/mw_phases.py 2>&1 | grep -v amdgpu | grep "drain plan\|graph step\|per launch" | cut -c1-260
run() { lib=$1
  env ${lib:+GRDMA_LIB_PATH=$R/grpc-rdma_amd/variants/$lib} GRDMA_TEST_ALLOW_EMU=1 timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 20 --reps 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; sk = r['schedule_kernels']
        print('%-14s value %.1f  ms/step %.4f  frac %.4f  copy launch %.2f us  wire %.2f  pair %.2f  verified %s' % ('${lib:-new}', d['value'], d['ms_per_step'], r['frac'], r['us_per_launch'], sk['wire']['us_per_launch'], sk['plan_pair']['us_per_launch'], d['verified']))
"
}
for rep in 1 2 3 4 5 6; do run lib_prev.so; run ""; done
