#!/bin/bash
# second session of round 5, first trip: gate on HEAD (after the removals), host-copy probe, vtable stream diagnosis, full bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b1
rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1 < /dev/null
echo "gpu suite rc=$?"; tail -4 $out/pytest_gpu.log
echo "== host copy probe"; timeout 120 tools/hostcopy_probe 2>&1 | grep -v amdgpu.ids | tee $out/hostcopy.txt
export ENDPOINT_STREAM_PROFILE=1
TRIP=r5b1/vts timeout 900 bash tools/r5_vtstream.sh 2>&1 | tee $out/vtstream.txt
unset ENDPOINT_STREAM_PROFILE
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    keys=["value","ms_per_step","value_index_rebuilt_every_step","value_wire_direct","value_with_h2","value_mixed_sizes","value_ring4096_sge30","value_conns32_64KiB_ring4096","value_conns32_64KiB_bidi","value_endpoint_vtable","value_endpoint_vtable_ring4096","rtt_p50_us","rtt_p95_us"]
    print({k:d.get(k) for k in keys})
    print("roofline", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("step_level"))
    print("vtable rtt", d.get("rtt_endpoint_vtable_us"))
except Exception as e:
    print("no bench line:", e); print(open("$out/bench.err").read()[-1500:])
PY
