#!/bin/bash
# first hardware run of the round: the endpoint tools at several knobs, then the GPU test suite
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export GRPC_PLATFORM_TYPE=RDMA_BP
run() { echo "== $*" >> gpurun_out/run1.log; timeout 120 "$@" >> gpurun_out/run1.log 2>&1; echo "rc=$?" >> gpurun_out/run1.log; }
: > gpurun_out/run1.log
for cfg in "131072 4095 0" "131072 4095 65536" "4096 30 0" "4096 30 65536" "4096 4095 65536" "131072 30 65536"; do
  set -- $cfg
  export GRPC_RDMA_RING_BUFFER_SIZE_KB=$1 GRPC_RDMA_MAX_SGE=$2 GRPC_RDMA_HIP_REGISTER_MIN=$3
  run tools/endpoint_stream 512 1048576 1 0
done
export GRPC_RDMA_RING_BUFFER_SIZE_KB=4096 GRPC_RDMA_MAX_SGE=30 GRPC_RDMA_HIP_REGISTER_MIN=0
run tools/endpoint_stream 64 1048576 1 1
for m in 0 1 2; do run tools/endpoint_pingpong 5000 64 $m; done
unset GRPC_RDMA_RING_BUFFER_SIZE_KB GRPC_RDMA_MAX_SGE GRPC_RDMA_HIP_REGISTER_MIN
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/run1.log
tail -5 gpurun_out/gputests1.log >> gpurun_out/run1.log
cat gpurun_out/run1.log
