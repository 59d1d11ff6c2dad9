#!/bin/bash
# copy-kernel variants of the headline leg, interleaved on ONE box (grpc-rdma_amd/variants/*.so, GRDMA_LIB_PATH)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5bvar
rm -rf $out; mkdir -p $out
run() { lib=$1
  env ${lib:+GRDMA_LIB_PATH=$R/grpc-rdma_amd/variants/$lib} GRDMA_TEST_ALLOW_EMU=1 timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 20 --reps 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; sk = r['schedule_kernels']
        print('%-14s value %.1f  ms/step %.4f  frac %.4f  copy launch %.2f us  wire %.2f  pair %.2f  verified %s' % ('${lib:-base}', d['value'], d['ms_per_step'], r['frac'], r['us_per_launch'], sk['wire']['us_per_launch'], sk['plan_pair']['us_per_launch'], d['verified']))
"
}
for rep in 1 2 3 4 5 6; do
  run ""; run lib_active.so
done 2>&1 | tee $out/variants.txt
