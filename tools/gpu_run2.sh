#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export GRPC_PLATFORM_TYPE=RDMA_BP
L=gpurun_out/run2.log
: > $L
run() { echo "== [$GRPC_RDMA_RING_BUFFER_SIZE_KB/$GRPC_RDMA_MAX_SGE/$GRPC_RDMA_HIP_WIRE/reg$GRPC_RDMA_HIP_REGISTER_MIN] $*" >> $L; timeout 120 "$@" >> $L 2>&1; echo "rc=$?" >> $L; }
for cfg in "131072 4095 direct 0" "131072 4095 direct 4096" "131072 4095 staged 0" "4096 30 direct 0" "4096 30 direct 4096" "4096 30 staged 0" "4096 4095 direct 0" "4096 4095 direct 4096" "131072 30 direct 4096"; do
  set -- $cfg
  export GRPC_RDMA_RING_BUFFER_SIZE_KB=$1 GRPC_RDMA_MAX_SGE=$2 GRPC_RDMA_HIP_WIRE=$3 GRPC_RDMA_HIP_REGISTER_MIN=$4
  run tools/endpoint_stream 1024 1048576 1 0 2
done
export GRPC_RDMA_RING_BUFFER_SIZE_KB=131072 GRPC_RDMA_MAX_SGE=4095 GRPC_RDMA_HIP_WIRE=direct GRPC_RDMA_HIP_REGISTER_MIN=4096
run tools/endpoint_stream 1024 1048576 0 0 2
run tools/endpoint_stream 1024 1048576 1 0 1
export GRPC_RDMA_RING_BUFFER_SIZE_KB=4096 GRPC_RDMA_MAX_SGE=30 GRPC_RDMA_HIP_REGISTER_MIN=0
run tools/endpoint_stream 64 1048576 1 1 2
for m in 0 1 2; do run tools/endpoint_pingpong 20000 64 $m; done
export GRPC_RDMA_HIP_WIRE=staged
for m in 0 2; do run tools/endpoint_pingpong 20000 64 $m; done
unset GRPC_RDMA_RING_BUFFER_SIZE_KB GRPC_RDMA_MAX_SGE GRPC_RDMA_HIP_REGISTER_MIN GRPC_RDMA_HIP_WIRE
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputests2.log 2>&1
echo "pytest rc=$?" >> $L
tail -15 gpurun_out/gputests2.log >> $L
timeout 300 python bench.py --steps 10 --warmup 3 --no-extra-legs > gpurun_out/bench2.json 2> gpurun_out/bench2.err
echo "bench rc=$?" >> $L
cat $L
