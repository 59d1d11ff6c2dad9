#!/bin/bash
# trip 4: the vtable stream with throttled host-facing copy grids, the multi-workgroup drain planner, a larger read-ahead
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b4
rm -rf $out; mkdir -p $out
export GRPC_PLATFORM_TYPE=RDMA_BP GRPC_RDMA_RING_BUFFER_SIZE_KB=${RING:-262144}
es() { label=$1; shift
  for rep in 1 2; do
    env "$@" timeout 120 tools/endpoint_stream 1024 1048576 ${CHECK:-1} 0 2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('%-50s %7.2f GiB/s  queued %s' % ('$label', d['GiBps'], d['writes_queued']))
    elif l: print('   ', l[:230])
"
  done
}
{
echo "== grids: rx (scatter into the window) x tx (gather from the send buffers); 0 = the HBM grid"
for rx in 0 8 16 32 64 128; do for tx in 0 16 64; do
  es "rx $rx tx $tx" GRDMA_HOST_RX_BLOCKS=$rx GRDMA_HOST_TX_BLOCKS=$tx
done; done
echo "== the same with the multi-workgroup drain planner and 4096 reads per pass"
for rx in 0 16 32 64; do for tx in 0 32; do
  es "rxm ahead4096 rx $rx tx $tx" GRDMA_HOST_RX_BLOCKS=$rx GRDMA_HOST_TX_BLOCKS=$tx GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
done; done
es "rxm ahead1024 rx 32 tx 32" GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRDMA_ENDPOINT_RX_MULTI=1
es "ahead4096 rx 32 tx 32" GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRPC_RDMA_HIP_READ_AHEAD=4096
CHECK=0 es "unchecked rxm ahead4096 rx 32 tx 32" GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
CHECK=0 es "unchecked rx 32 tx 32" GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32
echo "== profiles"
es "profile rxm ahead4096 rx 32 tx 32" ENDPOINT_STREAM_PROFILE=1 GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
es "profile rxm ahead1024 rx 32 tx 32" ENDPOINT_STREAM_PROFILE=1 GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRDMA_ENDPOINT_RX_MULTI=1
} 2>&1 | tee $out/vtable_matrix.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o t -- env GRDMA_HOST_RX_BLOCKS=32 GRDMA_HOST_TX_BLOCKS=32 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096 $R/tools/endpoint_stream 512 1048576 1 0 2 > $out/stdout.txt 2>&1
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_kernel_stats.csv; head -8 "$f"
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t 40 200 > $out/vtable_timeline.txt 2>&1; cat $out/vtable_timeline.txt
rm -rf $out/tr
