#!/bin/bash
# single-kernel planners (fast body + general planner in one launch): parity, bench variants, timeline
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/run7
rm -rf $out; mkdir -p $out
cd $R
timeout 300 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_link_engine.py -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
echo "pytest rc=$?"; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5"
run() { tag=$1; shift; timeout 120 env "$@" $B $EXTRA > $out/$tag.log 2> $out/$tag.err < /dev/null; echo "$tag rc=$? $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*\|"rx_plan": {[^}]*}\|"tx_plan": {[^}]*}' $out/$tag.log | tr '\n' ' ')"; grep -v amdgpu.ids $out/$tag.err | tail -2; }
EXTRA=""
run deep X=1
run deep_cb512 GRDMA_COPY_BLOCKS=512
run pairsched GRDMA_JOB_SCHEDULE=pair
EXTRA="--pipeline 0"
run sequential X=1
timeout 60 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null
tail -4 $out/phases.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o t -- $B --steps 6 --warmup 2 --no-verify --pipeline 0 > $out/tr.stdout 2>&1
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $f 30 > $out/timeline_seq.txt; tail -16 $out/timeline_seq.txt
rm -rf $out/tr
