#!/bin/bash
# trip 6: PCIe duplex probe (kernel vs copy engine per direction); throttle defaults at the other rings
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b6
rm -rf $out; mkdir -p $out
timeout 120 tools/pcie_duplex_probe 2>&1 | grep -v amdgpu.ids | tee $out/pcie_duplex_probe.txt
export GRPC_PLATFORM_TYPE=RDMA_BP
es() { label=$1; shift
  for rep in 1 2; do
    env "$@" timeout 120 tools/endpoint_stream 1024 1048576 ${CHECK:-1} 0 2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('%-56s %7.2f GiB/s  queued %s' % ('$label', d['GiBps'], d['writes_queued']))
    elif l: print('   ', l[:230])
"
  done
}
{
for ring in 4096 16384 262144; do
  export GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring
  echo "== ring $ring KiB"
  es "no throttle"
  es "rx 64 tx 16" GRDMA_HOST_RX_BLOCKS=64 GRDMA_HOST_TX_BLOCKS=16
  es "rx 128 tx 16" GRDMA_HOST_RX_BLOCKS=128 GRDMA_HOST_TX_BLOCKS=16
  es "rx 64 tx 16 rxm ahead4096" GRDMA_HOST_RX_BLOCKS=64 GRDMA_HOST_TX_BLOCKS=16 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
  es "rx 64 tx 16 ahead4096" GRDMA_HOST_RX_BLOCKS=64 GRDMA_HOST_TX_BLOCKS=16 GRPC_RDMA_HIP_READ_AHEAD=4096
done
} 2>&1 | tee $out/vtable_matrix.txt
