#!/bin/bash
# Kernel trace of tools/endpoint_stream (streaming through the endpoint vtable, host slices): per-kernel stats and the
# timeline of the last writes -> gpurun_out/prof_vtable/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_vtable
rm -rf $out; mkdir -p $out
export GRPC_RDMA_RING_BUFFER_SIZE_KB=${GRPC_RDMA_RING_BUFFER_SIZE_KB:-131072}
GRPC_PLATFORM_TYPE=RDMA_BP $R/tools/endpoint_stream 1024 1048576 1 0 2 | tee $out/plain.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o t -- env GRPC_PLATFORM_TYPE=RDMA_BP $R/tools/endpoint_stream 256 1048576 1 0 2 > $out/stdout.txt 2>&1
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/vtable_kernel_stats.csv; head -12 "$f"
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t 40 > $out/vtable_timeline.txt 2>&1; tail -42 $out/vtable_timeline.txt
rm -rf $out/tr
