#!/bin/bash
# trip 9: very small scatter grids (the duplex probe: 4 workgroups of paced stores leave the reads alone)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b9
rm -rf $out; mkdir -p $out
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
for ring in 262144 4096; do
  export GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring
  echo "== ring $ring KiB"
  for rx in 2 3 4 5 6 64; do for tx in 16 64; do
    es "rx $rx tx $tx" GRDMA_HOST_RX_BLOCKS=$rx GRDMA_HOST_TX_BLOCKS=$tx
  done; done
  es "rx 4 tx 16 rxm ahead4096" GRDMA_HOST_RX_BLOCKS=4 GRDMA_HOST_TX_BLOCKS=16 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
  es "rx 4 tx 16 ahead4096" GRDMA_HOST_RX_BLOCKS=4 GRDMA_HOST_TX_BLOCKS=16 GRPC_RDMA_HIP_READ_AHEAD=4096
  CHECK=0 es "unchecked rx 4 tx 16" GRDMA_HOST_RX_BLOCKS=4 GRDMA_HOST_TX_BLOCKS=16
done
} 2>&1 | tee $out/vtable_matrix.txt
