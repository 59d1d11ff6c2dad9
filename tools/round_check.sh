#!/bin/bash
# One GPU trip: the gpu test suite, kernel stats at the bench default, then the full bench line.
# Stops after the tests if they fail.  -> gpurun_out/check/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/check
rm -rf $out; mkdir -p $out
cd $R
timeout 150 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
rc=$?
tail -3 $out/pytest.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- \
  python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 10 \
  > $out/stats.stdout 2>&1 < /dev/null
f=$(find $out/stats -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $out/ring128m_kernel_stats.csv; head -8 "$f"; fi
grep '^{"metric"' $out/stats.stdout | tail -1 > $out/ring128m_under_rocprof.json
rm -rf $out/stats
cd $R
timeout 230 python bench.py > $out/bench.log 2> $out/bench.err < /dev/null
echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"rx_plan": {[^}]*}\|"value_ring4096_sge30": [0-9.]*\|"value_mixed_sizes": [0-9.]*\|"rtt_p50_us": [0-9.]*' $out/bench.log | head
