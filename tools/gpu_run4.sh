#!/bin/bash
# schedule experiments: CU-mask mapping, eager streams vs graph, reserved-CU planners, smaller copy grids
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/run4
rm -rf $out; mkdir -p $out
cd $R
timeout 60 tools/cumask_probe > $out/cumask.txt 2>&1; echo "probe rc=$?"; cat $out/cumask.txt
B="python bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5"
run() { tag=$1; shift; timeout 120 env "$@" $B $EXTRA > $out/$tag.log 2> $out/$tag.err < /dev/null; echo "$tag rc=$? $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*' $out/$tag.log | tr '\n' ' ')"; }
EXTRA=""
run graph_pair GRDMA_JOB_SCHEDULE=pair
run graph_deep GRDMA_JOB_SCHEDULE=deep
run graph_deep_cb512 GRDMA_JOB_SCHEDULE=deep GRDMA_COPY_BLOCKS=512
run graph_pair_cb512 GRDMA_JOB_SCHEDULE=pair GRDMA_COPY_BLOCKS=512
run graph_pair_cb1536 GRDMA_JOB_SCHEDULE=pair GRDMA_COPY_BLOCKS=1536
EXTRA="--launch streams"
run streams_pair GRDMA_JOB_SCHEDULE=pair
run streams_deep GRDMA_JOB_SCHEDULE=deep
run masked8 GRDMA_JOB_CUMASK=8
run masked16 GRDMA_JOB_CUMASK=16
run masked32 GRDMA_JOB_CUMASK=32
run masked16_cb512 GRDMA_JOB_CUMASK=16 GRDMA_COPY_BLOCKS=512
