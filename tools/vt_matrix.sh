#!/bin/bash
# The vtable streaming leg (tools/endpoint_stream: host slices through grpc_endpoint_write / _read, two threads) over a
# matrix of knobs, two runs each.  usage: tools/vt_matrix.sh <out file>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=${1:-/dev/stdout}
run() {  # label, env...
  local label=$1; shift
  for k in 1 2; do
    r=$(env GRPC_PLATFORM_TYPE=RDMA_BP "$@" $R/tools/endpoint_stream 1024 1048576 1 0 2 2>/dev/null | tail -1)
    echo "$label | $(echo "$r" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%6.2f GiB/s  queued %s checked %s" % (d["GiBps"], d["writes_queued"], d["checked"]))' 2>/dev/null || echo "FAILED: $r")" >> $out
  done
}
for ring in 262144 4096; do
  echo "== ring $ring KiB" >> $out
  run "rx64 tx16 (round 5)        " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring
  for rx in 2 4 8 16 32 64; do
    run "rx$rx acked                 " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=$rx GRDMA_HOST_RX_ACKED=1
  done
  run "rx8 acked tx32              " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=8 GRDMA_HOST_RX_ACKED=1 GRDMA_HOST_TX_BLOCKS=32
  run "rx8 acked tx8               " GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring GRDMA_HOST_RX_BLOCKS=8 GRDMA_HOST_RX_ACKED=1 GRDMA_HOST_TX_BLOCKS=8
done
