#!/bin/bash
# trip 3: kernel trace of the vtable stream with coalescing (who overlaps whom, how long the PCIe-bound kernels take)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b3
rm -rf $out; mkdir -p $out
export GRPC_PLATFORM_TYPE=RDMA_BP GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
cd /tmp && export TMPDIR=/tmp
for tag in co rxm; do
  extra=""; [ $tag = rxm ] && extra="GRDMA_ENDPOINT_RX_MULTI=1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$tag -o t -- env $extra $R/tools/endpoint_stream 512 1048576 1 0 2 > $out/stdout_$tag.txt 2>&1
  grep GiBps $out/stdout_$tag.txt | cut -c1-200
  f=$(find $out/tr_$tag -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_${tag}_kernel_stats.csv; head -10 "$f"
  t=$(find $out/tr_$tag -name '*kernel_trace.csv' | head -1)
  python $R/tools/timeline.py $t 48 200 > $out/vtable_${tag}_timeline.txt 2>&1; cat $out/vtable_${tag}_timeline.txt
  rm -rf $out/tr_$tag
done
cd $R
timeout 300 python -m pytest tests/test_gpu_endpoint_conformance.py -m gpu -x -q -p no:cacheprovider -k streamed 2>&1 | tail -3
