#!/bin/bash
# GPU trip: the endpoint vtable's unary round trip (tools/endpoint_pingpong) in its modes, the endpoint conformance tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5vt}
rm -rf $out; mkdir -p $out
for m in 1 2; do
  echo "== vtable ping-pong, mode $m"; timeout 120 tools/endpoint_pingpong ${ITERS:-20000} 64 $m 2>&1 | grep -v amdgpu.ids | tee $out/pp_mode$m.json
done
echo "== mode 2, 1 KiB and 3000 B payloads"; timeout 120 tools/endpoint_pingpong 5000 1000 2 2>&1 | grep -v amdgpu.ids; timeout 120 tools/endpoint_pingpong 5000 3000 2 2>&1 | grep -v amdgpu.ids
if [ "${TESTS:-0}" = "1" ]; then
  timeout 600 python -m pytest tests/test_gpu_endpoint_conformance.py tests/test_zzz_gpu_watch_read.py tests/test_adapter_trace.py -m gpu -x -q -p no:cacheprovider > $out/pytest.log 2>&1 < /dev/null
  echo "tests rc=$?"; tail -3 $out/pytest.log
fi
