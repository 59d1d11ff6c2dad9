#!/bin/bash
# the two HBM-traffic counter passes alone (FETCH_SIZE, WRITE_SIZE), two steps each: what ties roofline.traffic to HEAD's sources
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/pmc_min
rm -rf $out; mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-small-ring --no-rtt --no-extra-legs --conns 1 --reps 1"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 40 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc/$ctr -o pmc -- $B --no-verify --steps 2 --warmup 1 > $out/$ctr.stdout 2>&1
done
python $R/tools/pmc_summary.py $out/pmc $out/pmc_ring256m_summary.json "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-tcp-baseline --no-verify --no-small-ring --no-rtt --no-extra-legs --conns 1 --reps 1 --steps 2 --warmup 1 (one counter per pass; tools/r5b_pmc_min.sh: the traffic counters alone, collected from HEAD's sources in the round's last GPU seconds; the SQ wave-time split of the full set was collected two commits earlier: profiles/r05_pmc_ring256m_summary_with_sq_split.json)" | cut -c1-300
rm -rf $out/pmc
