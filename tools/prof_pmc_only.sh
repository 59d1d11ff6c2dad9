#!/bin/bash
# The counter passes of tools/prof_all.sh alone -> gpurun_out/profiles_new/pmc_ring256m_summary.json
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/profiles_new
mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  d=$out/pmc/$(echo $ctr | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o pmc -- $B --no-verify --steps 4 --warmup 1 > $d.stdout 2>&1
done
python $R/tools/pmc_summary.py $out/pmc $out/pmc_ring256m_summary.json "rocprofv3 --pmc <CTRS> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-tcp-baseline --no-verify --no-small-ring --no-rtt --no-extra-legs --conns 1 --steps 4 --warmup 1 (one counter family per pass; tools/prof_all.sh)"
rm -rf $out/pmc $out/*.stdout
