#!/bin/bash
# First GPU trip of a round: the paths written after the previous round's GPU budget ran out.
#   1. tests/cc/gpu_quickcheck        (seconds) boundary step / bulk pairs vs the verified deframer,
#                                      zero-copy send vs the oracle, with the deframer's tick counters; the armed read
#                                      through the latency engine vs the oracle (prints plain / armed RTT p50)
#   2. the newest GPU tests           tests/test_zz_gpu_zerocopy.py, tests/test_zz_gpu_h2_boundary.py (so far run
#                                      against the emulated library only)
#   3. bench.py --no-rtt              value_with_h2 with and without the boundary step, and the same
#                                      with GRDMA_H2_BULK_PAIRS=1
# -> gpurun_out/start/.  When 1 and 2 pass: consider GRDMA_H2_BULK_PAIRS on by default.
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/start
rm -rf $out; mkdir -p $out
cd $R
timeout 40 ./tests/cc/gpu_quickcheck $out/quickcheck.txt 30
echo "quickcheck rc=$?"
timeout 200 python -m pytest tests/test_zz_gpu_zerocopy.py tests/test_zz_gpu_h2_boundary.py tests/test_zz_gpu_latency_engine.py tests/test_zzz_gpu_armed_read.py -m gpu -q -x > $out/pytest_new.log 2>&1 < /dev/null
echo "new tests rc=$?"; tail -3 $out/pytest_new.log
timeout 150 python bench.py --no-rtt > $out/bench.log 2> $out/bench.err < /dev/null
echo "bench rc=$?"
grep -o '"value": [0-9.]*\|"value_with_h2[a-z_]*": [0-9.]*\|"deframe_us": [0-9]*' $out/bench.log | head -8
BENCH_H2=1 GRDMA_H2_BULK_PAIRS=1 timeout 150 python bench.py --no-rtt --no-extra-legs > $out/bench_pairs.log 2> $out/bench_pairs.err < /dev/null
grep -o '"value_with_h2[a-z_]*": [0-9.]*\|"deframe_us": [0-9]*' $out/bench_pairs.log | head -4
