#!/bin/bash
# quick GPU trip: the watched ping-pong's split (tools/rtt_probe.py), optionally the latency tests first (TESTS=1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5rtt}
rm -rf $out; mkdir -p $out
if [ "${TESTS:-0}" = "1" ]; then
  timeout 600 python -m pytest tests/test_zzz_gpu_watch_read.py tests/test_zz_gpu_latency_engine.py tests/test_zzz_gpu_armed_read.py -m gpu -x -q -p no:cacheprovider > $out/pytest_watch.log 2>&1 < /dev/null
  echo "watch tests rc=$?"; tail -5 $out/pytest_watch.log
fi
for w in ${WATCHERS:-4}; do
  echo "== rtt, watched reads, $w watcher workgroup(s)"
  GRDMA_ENGINE_WATCHERS=$w timeout 120 python tools/rtt_probe.py ${ITERS:-20000} watch 2>&1 | grep -v amdgpu.ids | tee $out/rtt_watch_w$w.txt
  echo "== the same with the phase stamps"
  GRDMA_ENGINE_WATCHERS=$w timeout 120 python tools/rtt_probe.py ${ITERS:-20000} watch prof 2>&1 | grep -v amdgpu.ids | tee $out/rtt_watch_w${w}_prof.txt
done
