#!/bin/bash
# Kernel trace of the with-h2 leg alone (bench.py --h2-only): per-kernel stats and a timeline of one step -> gpurun_out/prof_h2/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_h2
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o t -- env BENCH_H2_DEFAULT_LEG_ONLY=1 python $R/bench.py --h2-only --steps 10 --warmup 3 > $out/stdout.txt 2>&1
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/h2_kernel_stats.csv; head -14 "$f"
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t 60 > $out/h2_timeline.txt 2>&1; tail -70 $out/h2_timeline.txt
rm -rf $out/tr
