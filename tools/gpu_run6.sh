#!/bin/bash
# kernel timeline (rocprofv3 kernel trace) of the sequential and the limit-driven schedule with the fast planners
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/run6
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 6 --warmup 2 --no-verify"
for v in seq deep; do
  extra=""; [ $v = seq ] && extra="--pipeline 0"
  rocprofv3 --kernel-trace --output-format csv -d $out/$v -o t -- $B $extra > $out/$v.stdout 2>&1
  f=$(find $out/$v -name '*kernel_trace.csv' | head -1)
  python $R/tools/timeline.py $f 80 > $out/timeline_$v.txt
  echo "== $v"; tail -40 $out/timeline_$v.txt
  rm -rf $out/$v
done
