#!/bin/bash
# rocprofv3 kernel trace of a bench command, then the timeline of its last kernels
tag=${1:-trace}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-small-ring --no-rtt --conns 1 --no-verify "$@" > $out/bench_stdout.log 2>&1
tail -1 $out/bench_stdout.log | cut -c1-300
f=$(find $out -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/timeline.py $f 50
