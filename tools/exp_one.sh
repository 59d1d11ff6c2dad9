#!/bin/bash
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rtt --no-small-ring --no-extra-legs --conns 1"
run() { echo "== $ENVX $*"; timeout 300 env $ENVX $B "$@" 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l)
    print(d['value'], d['config']['rounds_per_step'], d['verified'], {k: v['us_per_launch'] for k, v in d['kernels'].items()})
except Exception as e:
    print('ERR', l[-600:])
"; }
ENVX="X=1" run
ENVX="X=1" run --pipeline 0
