#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command -> gpurun_out/prof_<tag>/
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 "$@" > $out/bench_stdout.log 2>&1
ls -R $out | head -30
f=$(find $out -name '*kernel_stats.csv' | head -1)
echo "== $f"; head -20 "$f"
