#!/bin/bash
# PMC passes (one counter family per run, as MI355X_MICROARCH.md prescribes) for the bench command.
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-verify --no-small-ring --no-rtt --no-extra-legs --conns 1 "$@" > $out/$ctr.stdout.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE","WRITE_SIZE"):
    files=glob.glob("$out/%s/**/*counter_collection.csv"%ctr, recursive=True)
    agg=collections.defaultdict(lambda:[0,0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name")==ctr:
                k=row["Kernel_Name"].split("(")[0]
                agg[k][0]+=1; agg[k][1]+=float(row["Counter_Value"])
    print(ctr, {k:(n, round(v/n,1)) for k,(n,v) in agg.items() if n})
PY
