#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5k}
rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_gpu_two_process.py -m gpu -x -q -s -p no:cacheprovider -k "pingpong or fine_grained" > $out/pytest_2proc.log 2>&1 < /dev/null
echo "two-process rc=$?"; grep -a "ok client\|passed\|failed\|Error\|error" $out/pytest_2proc.log | tail -8
echo "== bench rtt legs"
timeout 300 python bench.py --rtt-only --rtt-iters 200000 2>/dev/null | tail -1 | tee $out/rtt_only.json | cut -c1-1500
timeout 100 python bench.py --rtt-commands-only --rtt-iters 20000 2>/dev/null | tail -1 | tee $out/rtt_commands.json | cut -c1-400
