#!/bin/bash
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rtt --no-small-ring --conns 1"
run() { echo "== $ENVX $*"; timeout 300 env $ENVX $B "$@" 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print(d['value'], d['config']['rounds_per_step'], d['verified'])
except Exception as e:
    print('ERR', l[-600:])
"; }
for r in 131072 524288; do
ENVX="X=1" run --ring-kb $r
ENVX="GRDMA_PIPE_VARIANT=0" run --ring-kb $r --pipeline 1
ENVX="GRDMA_PIPE_VARIANT=1" run --ring-kb $r --pipeline 1
ENVX="GRDMA_PIPE_VARIANT=2" run --ring-kb $r --pipeline 1
done
ENVX="X=1" run --ring-kb 65536 --wire direct
