#!/bin/bash
# fast planners on hardware: parity tests of the streaming job, then bench variants
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/run5
rm -rf $out; mkdir -p $out
cd $R
timeout 300 python -m pytest tests/test_gpu_stream_job.py tests/test_gpu_link_engine.py -m gpu -q -x > $out/pytest.log 2>&1 < /dev/null
echo "pytest rc=$?"; tail -5 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5"
run() { tag=$1; shift; timeout 120 env "$@" $B $EXTRA > $out/$tag.log 2> $out/$tag.err < /dev/null; echo "$tag rc=$? $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*\|"rx_plan": {[^}]*}\|"tx_plan": {[^}]*}' $out/$tag.log | tr '\n' ' ')"; tail -2 $out/$tag.err; }
EXTRA=""
run fast_deep X=1
run fast_deep_cb512 GRDMA_COPY_BLOCKS=512
run rxfast_only GRDMA_TX_FAST=0
run txfast_only GRDMA_RX_FAST=0
run nofast_pair GRDMA_TX_FAST=0 GRDMA_RX_FAST=0 GRDMA_JOB_SCHEDULE=pair
EXTRA="--pipeline 0"
run fast_sequential X=1
EXTRA="--launch streams"
run fast_streams X=1
timeout 60 python tools/plan_phases.py > $out/phases.log 2>&1 < /dev/null
tail -3 $out/phases.log
