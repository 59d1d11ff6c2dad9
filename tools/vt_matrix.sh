#!/bin/bash
# The vtable streaming leg (tools/endpoint_stream: host slices through grpc_endpoint_write / _read, two threads) over a
# matrix of knobs, two runs each.  usage: tools/vt_matrix.sh <out file>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=${1:-/dev/stdout}
run() {  # label, env...
  local label=$1; shift
  for k in 1 2; do
    r=$(env GRPC_PLATFORM_TYPE=RDMA_BP "$@" $R/tools/endpoint_stream 1024 1048576 1 0 2 2>/dev/null | tail -1)
    echo "$label | $(echo "$r" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%6.2f GiB/s  mirror %s queued %s checked %s" % (d["GiBps"], d.get("tx_mirror"), d["writes_queued"], d["checked"]))' 2>/dev/null || echo "FAILED: $r")" >> $out
  done
}
for ring in 262144 4096; do
  echo "== ring $ring KiB" >> $out
  run "mirror off rx64 (round 5)          " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_TX_MIRROR=0
  for rx in 32 64 128 256; do
    run "mirror on  rx$rx                   " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=$rx
  done
  run "mirror on  rx128 rxmulti ahead4096 " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=128 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
  run "mirror on  rx128 8M send buffers   " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=128 GRPC_RDMA_HIP_SEND_BUFFER_KB=8192
done
