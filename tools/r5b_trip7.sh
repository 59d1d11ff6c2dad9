#!/bin/bash
# trip 7: drains delivered through the HBM mirror + an exact-size device-to-host copy (two stages), defaults rx 64 / tx 16
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b7
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
for ring in 262144 16384 4096; do
  export GRPC_RDMA_RING_BUFFER_SIZE_KB=$ring
  echo "== ring $ring KiB"
  es "scatter into the window (mirror off)" GRDMA_ENDPOINT_RX_SDMA=0
  es "mirror (default)"
  es "mirror rxm" GRDMA_ENDPOINT_RX_MULTI=1
  es "mirror rxm ahead4096" GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
  es "mirror ahead4096" GRPC_RDMA_HIP_READ_AHEAD=4096
  es "mirror tx 0" GRDMA_HOST_TX_BLOCKS=0
  es "mirror tx 64" GRDMA_HOST_TX_BLOCKS=64
done
export GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
CHECK=0 es "unchecked mirror"
CHECK=0 es "unchecked mirror rxm ahead4096" GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
es "profile mirror" ENDPOINT_STREAM_PROFILE=1
es "profile mirror rxm ahead4096" ENDPOINT_STREAM_PROFILE=1 GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096
} 2>&1 | tee $out/vtable_matrix.txt
cd /tmp && export TMPDIR=/tmp
for tag in plain rxm; do
  extra=""; [ $tag = rxm ] && extra="GRDMA_ENDPOINT_RX_MULTI=1 GRPC_RDMA_HIP_READ_AHEAD=4096"
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/tr -o t -- env $extra $R/tools/endpoint_stream 512 1048576 1 0 2 > $out/stdout_$tag.txt 2>&1
  grep GiBps $out/stdout_$tag.txt | cut -c1-120
  f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_${tag}_kernel_stats.csv; head -8 "$f"
  t=$(find $out/tr -name '*kernel_trace.csv' | head -1); cp "$t" $out/vtable_${tag}_kernel_trace.csv
  m=$(find $out/tr -name '*memory_copy_trace.csv' | head -1); [ -n "$m" ] && cp "$m" $out/vtable_${tag}_memory_copy_trace.csv
  python $R/tools/timeline.py $t 36 200 > $out/vtable_${tag}_timeline.txt 2>&1; cat $out/vtable_${tag}_timeline.txt
  rm -rf $out/tr
done
cd $R
timeout 600 python -m pytest tests/test_gpu_endpoint_conformance.py tests/test_adapter_trace.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
