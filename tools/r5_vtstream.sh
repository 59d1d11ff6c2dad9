#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/${TRIP:-r5vts}
rm -rf $out; mkdir -p $out
echo "== pcie"; timeout 120 python tools/pcie_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/pcie.json
export GRPC_PLATFORM_TYPE=RDMA_BP
for ring in 262144 4096; do
  echo "== endpoint_stream ring $ring KiB"
  GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring timeout 120 tools/endpoint_stream 1024 1048576 1 0 2 2>&1 | grep -v amdgpu.ids | tee $out/es_ring$ring.json | cut -c1-400
  echo "   registered host slices (no bounce copy)"
  GRPC_RDMA_HIP_REGISTER_MIN=4096 GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring timeout 120 tools/endpoint_stream 1024 1048576 1 0 2 2>&1 | grep -v amdgpu.ids | tee $out/es_ring${ring}_reg.json | cut -c1-400
  echo "   unchecked"
  GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring timeout 120 tools/endpoint_stream 1024 1048576 0 0 2 2>&1 | grep -v amdgpu.ids | cut -c1-200
  echo "   one thread"
  GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring timeout 120 tools/endpoint_stream 1024 1048576 1 0 1 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
