#!/bin/bash
# trip 5: the vtable stream with the drain delivered through an HBM mirror + copy engine (GRDMA_ENDPOINT_RX_SDMA)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b5
rm -rf $out; mkdir -p $out
export GRPC_PLATFORM_TYPE=RDMA_BP GRPC_RDMA_RING_BUFFER_SIZE_KB=${RING:-262144}
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
es "baseline (no knobs)"
es "mirror" GRDMA_ENDPOINT_RX_SDMA=1
es "mirror rxm ahead4096" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
es "mirror rxm ahead4096 tx16" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096 GRDMA_HOST_TX_BLOCKS=16
es "mirror rxm ahead4096 tx64" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096 GRDMA_HOST_TX_BLOCKS=64
es "mirror rxm ahead2048" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=2048
es "mirror rxm ahead4096 8M buffers" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096 GRPC_RDMA_HIP_SEND_BUFFER_KB=8192
CHECK=0 es "unchecked mirror rxm ahead4096" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
es "profile mirror rxm ahead4096" ENDPOINT_STREAM_PROFILE=1 GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
echo "== ring 4 MiB"
export GRPC_RDMA_RING_BUFFER_SIZE_KB=4096
es "baseline"
es "mirror" GRDMA_ENDPOINT_RX_SDMA=1
es "mirror rxm ahead4096" GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
export GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
} 2>&1 | tee $out/vtable_matrix.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/tr -o t -- env GRDMA_ENDPOINT_RX_SDMA=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096 $R/tools/endpoint_stream 512 1048576 1 0 2 > $out/stdout.txt 2>&1
grep GiBps $out/stdout.txt | cut -c1-120
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_kernel_stats.csv; head -8 "$f"
t=$(find $out/tr -name '*kernel_trace.csv' | head -1); cp "$t" $out/vtable_kernel_trace.csv
m=$(find $out/tr -name '*memory_copy_trace.csv' | head -1); [ -n "$m" ] && cp "$m" $out/vtable_memory_copy_trace.csv && head -5 "$m"
python $R/tools/timeline.py $t 40 200 > $out/vtable_timeline.txt 2>&1; cat $out/vtable_timeline.txt
rm -rf $out/tr
