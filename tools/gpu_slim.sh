#!/bin/bash
# experiment: planner workgroups small enough to be resident BESIDE the copy kernels' workgroups (256 threads, no general
# planner in the kernel: grpc-rdma_amd/variants/libgrdma_amd_slim.so, built with -DGRDMA_SLIM_PLANNERS) -- do the planners
# of round t then overlap the copies of round t + 1 in the limit-driven schedule?
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/slim
rm -rf $out; mkdir -p $out
cd $R
export GRDMA_LIB_PATH=$R/grpc-rdma_amd/variants/libgrdma_amd_slim.so GRDMA_SLIM_AFTER=2
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5"
run() { tag=$1; shift; timeout 120 env "$@" $B $EXTRA > $out/$tag.log 2> $out/$tag.err < /dev/null; echo "$tag rc=$? $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"verified": [a-z]*' $out/$tag.log | tr '\n' ' ')"; grep -v amdgpu.ids $out/$tag.err | tail -2; }
EXTRA=""
run deep_cb768 X=1
run deep_cb512 GRDMA_COPY_BLOCKS=512
run deep_cb640 GRDMA_COPY_BLOCKS=640
run deep_cb384 GRDMA_COPY_BLOCKS=384
EXTRA="--pipeline 0"
run seq_cb512 GRDMA_COPY_BLOCKS=512
EXTRA="--launch streams"
run streams_cb512 GRDMA_COPY_BLOCKS=512
run streams_cb768 X=1
