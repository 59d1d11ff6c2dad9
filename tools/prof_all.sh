#!/bin/bash
# Round profile set: kernel stats at the bench default and at the 4 MiB ring (reference default knobs), HBM traffic
# (FETCH_SIZE / WRITE_SIZE, one counter per pass), SQ wave-time split.  -> gpurun_out/profiles_new/
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profiles_new
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1"
stats() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -o bench -- $B "$@" > $out/stats_$tag.stdout 2>&1
  f=$(find $out/stats_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $out/${tag}_kernel_stats.csv; grep '^{"metric"' $out/stats_$tag.stdout | tail -1 > $out/${tag}_under_rocprof.json
  echo "== $tag"; head -8 "$f"; }
stats ring256m --steps 10
stats ring4m --steps 6 --ring-kb 4096 --max-sge 30 --sends 64 --promise
mkdir -p $out/pmc
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  d=$out/pmc/$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o pmc -- $B --no-verify --steps 4 --warmup 1 > $d.stdout 2>&1
done
python $R/tools/pmc_summary.py $out/pmc $out/pmc_ring256m_summary.json "rocprofv3 --pmc <CTRS> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-tcp-baseline --no-verify --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 4 --warmup 1 (one counter family per pass; tools/prof_all.sh)"
rm -rf $out/stats_ring256m $out/stats_ring4m $out/pmc $out/*.stdout
ls -la $out
