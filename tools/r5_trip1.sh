#!/bin/bash
# Round 5, first GPU trip: the watcher workgroups (k_watch) on hardware -- parity tests, then where a round trip goes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5a}
rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_zzz_gpu_watch_read.py tests/test_zz_gpu_latency_engine.py tests/test_zzz_gpu_armed_read.py -m gpu -x -q -p no:cacheprovider > $out/pytest_watch.log 2>&1 < /dev/null
echo "watch tests rc=$?"; tail -5 $out/pytest_watch.log
for w in 1 4; do
  echo "== rtt, watched reads, $w watcher workgroup(s)"
  GRDMA_ENGINE_WATCHERS=$w timeout 120 python tools/rtt_probe.py 20000 watch 2>&1 | tee $out/rtt_watch_w$w.txt
done
echo "== rtt, unarmed"
timeout 120 python tools/rtt_probe.py 20000 2>&1 | tee $out/rtt_unarmed.txt
echo "== rtt, chained (round 4)"
timeout 120 python tools/rtt_probe.py 20000 chain 2>&1 | tee $out/rtt_chain.txt
