#!/bin/bash
# the whole GPU suite, then the default bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5full}
rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1 < /dev/null
echo "gpu suite rc=$?"; tail -6 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    keys=["value","ms_per_step","value_index_rebuilt_every_step","value_direct_wire","value_with_h2","value_mixed_sizes","value_ring4096_sge30","value_conns32_64KiB_ring4096","value_conns32_64KiB_bidi","conns32_64KiB_bidi_error","value_endpoint_vtable","rtt_p50_us","rtt_p95_us","rtt_read_commands_p50_us","rtt_chained_read_p50_us"]
    print({k:d.get(k) for k in keys})
    print("roofline", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("step_level"))
    print("vtable rtt", d.get("rtt_endpoint_vtable_us"))
    print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_ring4096_sge30",{}).get("value"))
except Exception as e:
    print("no bench line:", e); print(open("$out/bench.err").read()[-1500:])
PY
