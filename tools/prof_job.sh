#!/bin/bash
# Kernel trace of the headline leg alone: per-kernel stats and the timeline of the last step -> gpurun_out/prof_job/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_job
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr -o t -- python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 6 --warmup 2 --no-verify --reps 1 > $out/stdout.txt 2>&1
f=$(find $out/tr -name '*kernel_stats.csv' | head -1); cp "$f" $out/job_kernel_stats.csv; head -12 "$f" | cut -c1-200
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t ${1:-40} > $out/job_timeline.txt 2>&1; tail -${1:-40} $out/job_timeline.txt
rm -rf $out/tr
