R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof4m
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o t -- python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --no-fanout --conns 1 --steps 3 --warmup 1 --no-verify --reps 1 --msgs 256 --leg-msgs 256 --ring-kb 4096 --max-sge 30 --sends 64 --promise > $out/stdout.txt 2>&1
t=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python $R/tools/timeline.py $t 60 > $out/timeline.txt 2>&1; tail -60 $out/timeline.txt
rm -rf $out/tr
