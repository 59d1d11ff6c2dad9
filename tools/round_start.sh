#!/bin/bash
# First GPU trip of round 4: what round 3 left unmeasured when its GPU minutes ran out (docs/pending/).
# BEFORE the trip, on the CPU box:   bash tools/round_start.sh prepare
#   builds grpc-rdma_amd/variants/lib_burst_fetch.so = the product with docs/pending/tx_burst_wave_one_fetch.patch applied
#   (built .so files travel to the GPU box).
# ON the GPU box (gpurun -- 'bash tools/round_start.sh'), -> gpurun_out/start/:
#   1. the burst / endpoint-chain parity tests against the patched library
#   2. the reference's default knobs (4 MiB ring, max_sge 30, burst 16), product vs patched: per-launch split
#      (tools/plan_phases.py 4096 30 16) -- k_tx_plan_seq was 21.9 us, k_rx_plan 32.9 of 83 us per round
#   3. the same knobs on the link engine (never timed there): bench.py --schedule engine --ring-kb 4096 --max-sge 30
#   4. the drain's phase stamps at those knobs, both parities (GRDMA_DBG_ODD)
# Keep the patch only if 1 is green and 2 says so.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
V=grpc-rdma_amd/variants/lib_burst_fetch.so
if [ "$1" = "prepare" ]; then
  tmp=$(mktemp -d); cp -r grpc-rdma_amd/csrc $tmp/csrc; mkdir -p $tmp/include; cp include/*.h include/*.hpp $tmp/include/
  (cd $tmp && mkdir -p grpc-rdma_amd && mv csrc grpc-rdma_amd/ && patch -p1 < $R/docs/pending/tx_burst_wave_one_fetch.patch) || exit 1
  mkdir -p grpc-rdma_amd/variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -o $V \
      $tmp/grpc-rdma_amd/csrc/*.hip $tmp/grpc-rdma_amd/csrc/*.cc && echo "built $V"
  rm -rf $tmp
  exit 0
fi
out=$R/gpurun_out/start
rm -rf $out; mkdir -p $out
if [ -f $V ]; then
  GRDMA_LIB_PATH=$R/$V GRDMA_TEST_ALLOW_EMU=1 timeout 300 python -m pytest tests/test_gpu_link_engine.py tests/test_gpu_endpoint_conformance.py \
      tests/test_gpu_pair_parity.py -m gpu -q -x -k "burst or stream or queue or sequence" > $out/pytest_patched.log 2>&1 < /dev/null
  echo "patched library, burst / chain tests rc=$?"; tail -2 $out/pytest_patched.log
fi
echo "== default knobs, product";  timeout 100 python tools/plan_phases.py 4096 30 16 2>&1 | tail -3 | tee $out/phases_product.txt
if [ -f $V ]; then
  echo "== default knobs, patched"; GRDMA_LIB_PATH=$R/$V GRDMA_TEST_ALLOW_EMU=1 timeout 100 python tools/plan_phases.py 4096 30 16 2>&1 | tail -3 | tee $out/phases_patched.txt
fi
echo "== drain stamps, other parity"; GRDMA_DBG_ODD=1 timeout 100 python tools/plan_phases.py 4096 30 16 2>&1 | tail -1 | tee $out/phases_odd.txt
echo "== link engine at the default knobs"
timeout 120 python bench.py --schedule engine --ring-kb 4096 --max-sge 30 --no-extra-legs --no-cpu-baseline --no-tcp-baseline \
    --no-rtt --no-small-ring --steps 4 --warmup 1 --reps 1 > $out/engine_r4m.json 2> $out/engine_r4m.err < /dev/null
echo "engine rc=$?"; grep -o '"value": [0-9.]*\|"schedule": "[^"]*"' $out/engine_r4m.json | head -3
