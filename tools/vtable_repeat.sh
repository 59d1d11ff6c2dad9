#!/bin/bash
# value_endpoint_vtable's command N times in a row: how often the slow mode shows (1024 x 1 MiB, checked, two threads)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-8}
for i in $(seq $N); do
  GRDMA_PIN_CORES=${2:-1,0} GRPC_RDMA_RING_BUFFER_SIZE_KB=131072 $R/tools/endpoint_stream 1024 1048576 1 0 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['GiBps'], d['numa_pinned'], d['writes_queued'])"
done
