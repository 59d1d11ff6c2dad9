#!/bin/bash
# bench headline at a few max_sge values (does a round that ends on a payload record plan faster?)
for m in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --no-tcp-baseline --max-sge $m 2>&1 | tail -1 > /tmp/sge_$m.json
  python - <<PY
import json
d = json.load(open("/tmp/sge_$m.json"))
print($m, d["value"], d["ms_per_step"], d["config"]["rounds_per_step"], {k: v["us_per_launch"] for k, v in d["kernels"].items()})
PY
done
