cd $GRAFT_REPO_ROOT
E="GRPC_PLATFORM_TYPE=RDMA_BP"
for v in 0 1; do for r in 131072 4096; do
  for i in 1 2; do env $E GRPC_RDMA_RING_BUFFER_SIZE_KB=$r GRDMA_ENDPOINT_RX_MULTI=$v ./tools/endpoint_stream 1024 1048576 1 0 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rx_multi=$v ring=$r', d.get('GiBps'), d.get('checked'))"; done
done; done
timeout 300 python -m pytest tests/test_gpu_endpoint_conformance.py tests/test_adapter_trace.py -m gpu -q 2>&1 | tail -2
