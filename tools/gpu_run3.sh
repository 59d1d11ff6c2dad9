#!/bin/bash
# Round 3, first trip of the session: gpu tests (gate), bench with the limit-driven schedule and with the old one, planner phases.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/run3
rm -rf $out; mkdir -p $out
cd $R
timeout 400 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs > $out/bench_deep.log 2> $out/bench_deep.err < /dev/null
echo "bench deep rc=$?"
GRDMA_JOB_SCHEDULE=pair timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs > $out/bench_pair.log 2> $out/bench_pair.err < /dev/null
echo "bench pair rc=$?"
for f in deep pair; do grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*\|"value_wire_direct": [0-9.]*\|"value_sequential": [0-9.]*' $out/bench_$f.log | tr '\n' ' '; echo; done
timeout 60 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null
tail -3 $out/phases.log
