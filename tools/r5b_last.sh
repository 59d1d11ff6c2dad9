#!/bin/bash
# last GPU seconds of round 5: the stream-job parity tests on HEAD (table-cache slot hashing), one headline bench run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
timeout 200 python -m pytest tests/test_gpu_stream_job.py -m gpu -x -q -p no:cacheprovider -n 4 2>&1 | tail -2
timeout 100 python bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; print('value %.1f ms/step %.4f frac %.4f pair %.2f verified %s' % (d['value'], d['ms_per_step'], r['frac'], r['schedule_kernels']['plan_pair']['us_per_launch'], d['verified']))
"
python - <<'PY'
import ctypes as C, sys
sys.path.insert(0, '.')
import grpc_rdma_amd as g
lib = g.load(); out = (C.c_uint64 * 2)(); print('ok')
PY
