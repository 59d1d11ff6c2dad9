#!/bin/bash
# One GPU trip at the end of a round: the gpu test suite (gate), the full bench line, the planner
# phase stamps, then the profile set of tools/prof_all.sh.  -> gpurun_out/check/, gpurun_out/profiles_new/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/check
rm -rf $out; mkdir -p $out
cd $R
timeout 150 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
rc=$?
tail -3 $out/pytest.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout 100 python bench.py > $out/bench.log 2> $out/bench.err < /dev/null
echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"rx_plan": {[^}]*}\|"value_ring4096_sge30": [0-9.]*\|"value_mixed_sizes": [0-9.]*\|"rtt_p50_us": [0-9.]*' $out/bench.log | head -5
timeout 30 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null
tail -1 $out/phases.log
timeout 45 bash tools/prof_all.sh < /dev/null 2>&1 | grep -A3 "== ring256m"
