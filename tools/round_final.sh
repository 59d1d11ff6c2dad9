#!/bin/bash
# One trip for a round's record: gpu suite (gate), profile set (kernel stats at the headline and at the reference's default
# knobs, PMC passes -> summary tied to the sources), the full bench line (reads that summary), the job's timeline, the
# planner phases, the vtable stream's kernel trace, the zero-copy memory probe.  -> gpurun_out/final/ ; usage: round_final.sh r06
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RND=${1:-r06}
cd $R
out=$R/gpurun_out/final
rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q -rs -p no:cacheprovider > $out/pytest_gpu.log 2>&1 < /dev/null
rc=$?; echo "gpu suite rc=$rc"; tail -12 $out/pytest_gpu.log
export GRAFT_REPO_ROOT=$R
timeout 900 bash tools/prof_all.sh < /dev/null 2>&1 | grep -A6 "== ring"
cp $R/gpurun_out/profiles_new/pmc_ring256m_summary.json $R/profiles/${RND}_pmc_ring256m_summary.json 2>/dev/null
mkdir -p $out/profiles_new; cp -r $R/gpurun_out/profiles_new/* $out/profiles_new/ 2>/dev/null
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    keys=["value","ms_per_step","value_msgs256_per_step","value_index_rebuilt_every_step","value_wire_direct","value_with_h2","value_mixed_sizes","value_ring4096_sge30","value_conns32_64KiB_ring4096","value_conns32_64KiB_bidi","value_endpoint_vtable","value_endpoint_vtable_ring4096","value_endpoint_vtable_frac_of_pcie_ceiling","rtt_p50_us","rtt_p95_us","rtt_p99_us","fanout_checksum_ok"]
    print({k:d.get(k) for k in keys})
    r=d.get("roofline",{}); print("roofline", r.get("frac"), r.get("us_per_launch"), r.get("traffic"), r.get("traffic_source","")[:120], r.get("step_level"), r.get("dominant_by_time"))
    print("vtable rtt", d.get("rtt_endpoint_vtable_us")); print("pcie", d.get("pcie_ceiling"))
    print("cpu", {k:(v or {}).get("value") for k,v in d.items() if k.startswith("cpu_baseline")})
    print({k:v for k,v in d.items() if "error" in k})
except Exception as e:
    print("no bench line:", e); print(open("$out/bench.err").read()[-1500:])
PY
timeout 300 bash tools/prof_job.sh 48 > $out/prof_job.txt 2>&1; cp $R/gpurun_out/prof_job/job_timeline.txt $out/ 2>/dev/null; cp $R/gpurun_out/prof_job/job_kernel_stats.csv $out/ 2>/dev/null; tail -5 $out/prof_job.txt
(echo "# default knobs"; MW_SENDS=64 MW_PROMISE=1 timeout 120 python tools/mw_phases.py 4096 30; echo "# headline"; timeout 120 python tools/mw_phases.py) 2>&1 | grep -v "amdgpu\|fused round" > $out/plan_phases.txt
timeout 200 python tools/rtt_probe.py 20000 watch prof > $out/rtt_probe.txt 2>&1; tail -12 $out/rtt_probe.txt | cut -c1-250
tools/zc_mem_probe > $out/zc_mem_probe.txt 2>&1; cat $out/zc_mem_probe.txt
export GRPC_PLATFORM_TYPE=RDMA_BP GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o t -- $R/tools/endpoint_stream 512 1048576 1 0 2 > $out/vtable_stdout.txt 2>&1
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_kernel_stats.csv
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t 48 200 > $out/vtable_timeline.txt 2>&1; head -30 $out/vtable_timeline.txt
rm -rf $out/tr
ENDPOINT_STREAM_PROFILE=1 $R/tools/endpoint_stream 1024 1048576 1 0 2 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $out/vtable_profile.txt
