#!/bin/bash
# copy grid (GRDMA_COPY_BLOCKS) of the headline leg, alternating on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
run() { cb=$1
  GRDMA_COPY_BLOCKS=$cb timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 20 --reps 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; sk = r['schedule_kernels']
        print('blocks %-6s value %.1f  ms/step %.4f  frac %.4f  copy launch %.2f us  wire %.2f  gather %.2f verified %s' % ('$cb', d['value'], d['ms_per_step'], r['frac'], r['us_per_launch'], sk['wire']['us_per_launch'], sk['gather']['us_per_launch'], d['verified']))
"
}
for rep in 1 2 3 4; do for cb in 0 1024 1536 2048; do run $cb; done; done
