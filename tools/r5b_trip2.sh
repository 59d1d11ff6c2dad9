#!/bin/bash
# second session of round 5, trip 2: the vtable stream with coalescing send buffers (matrix), endpoint GPU tests, copy-grid sweep
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
out=$R/gpurun_out/r5b2
rm -rf $out; mkdir -p $out
export GRPC_PLATFORM_TYPE=RDMA_BP
es() { # label, env..., -- args
  label=$1; shift
  for rep in 1 2; do
    env "$@" timeout 120 tools/endpoint_stream 1024 1048576 ${CHECK:-1} 0 2 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('%-44s %7.2f GiB/s  queued %s  checked %s' % ('$label', d['GiBps'], d['writes_queued'], d['checked']))
    elif l: print('   ', l[:200])
"
  done
}
echo "== host copy probe"; timeout 120 tools/hostcopy_probe 2>&1 | grep -v amdgpu.ids | tee $out/hostcopy.txt
{
echo "== ring 256 MiB"
es "coalesce (default)"            GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
es "coalesce off"                  GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRPC_RDMA_HIP_COALESCE=0
es "coalesce + rx multi"           GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRDMA_ENDPOINT_RX_MULTI=1
es "coalesce + 8 MiB send buffers" GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRPC_RDMA_HIP_SEND_BUFFER_KB=8192
es "coalesce + sse2 sum"           GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 ENDPOINT_STREAM_SSE2=1
CHECK=0 es "coalesce, unchecked"           GRPC_RDMA_RING_BUFFER_SIZE_KB=262144
CHECK=0 es "coalesce + rx multi, unchecked" GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRDMA_ENDPOINT_RX_MULTI=1
CHECK=0 es "coalesce off, unchecked"       GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRPC_RDMA_HIP_COALESCE=0
echo "== ring 16 MiB"
es "coalesce (default)"            GRPC_RDMA_RING_BUFFER_SIZE_KB=16384
es "coalesce + rx multi"           GRPC_RDMA_RING_BUFFER_SIZE_KB=16384 GRDMA_ENDPOINT_RX_MULTI=1
echo "== ring 4 MiB"
es "default"                       GRPC_RDMA_RING_BUFFER_SIZE_KB=4096
es "rx multi"                      GRPC_RDMA_RING_BUFFER_SIZE_KB=4096 GRDMA_ENDPOINT_RX_MULTI=1
echo "== profile, ring 256 MiB, coalesce"
GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 ENDPOINT_STREAM_PROFILE=1 timeout 120 tools/endpoint_stream 1024 1048576 1 0 2 2>&1 | grep -v amdgpu.ids | cut -c1-300
GRPC_RDMA_RING_BUFFER_SIZE_KB=262144 GRDMA_ENDPOINT_RX_MULTI=1 ENDPOINT_STREAM_PROFILE=1 timeout 120 tools/endpoint_stream 1024 1048576 1 0 2 2>&1 | grep -v amdgpu.ids | cut -c1-300
} 2>&1 | tee $out/vtable_matrix.txt
timeout 600 python -m pytest tests/test_gpu_endpoint_conformance.py tests/test_adapter_trace.py -m gpu -x -q -p no:cacheprovider > $out/pytest_endpoint.log 2>&1 < /dev/null
echo "endpoint tests rc=$?"; tail -3 $out/pytest_endpoint.log
echo "== copy grid sweep (headline leg only)"
for cb in 0 512 768 1024 1536 3072; do
  GRDMA_COPY_BLOCKS=$cb timeout 200 python bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); r = d['roofline']; print('GRDMA_COPY_BLOCKS=$cb value %.1f  ms/step %.4f  frac %.4f  us/launch %.2f' % (d['value'], d['ms_per_step'], r['frac'], r['us_per_launch']))
"
done 2>&1 | tee $out/copy_blocks_sweep.txt
