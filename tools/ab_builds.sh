#!/bin/bash
# Two builds of libgrdma_amd.so alternating on ONE box: how the "x -> y GiB/s, three alternations" figures of round 6 were
# taken (boxes of the pool differ by a few per cent; two runs on two leases do not compare).
#   here (no GPU):   tools/ab_builds.sh prepare <git-rev>      # build/ab/lib_a.so = <git-rev>, build/ab/lib_b.so = the working tree
#   on the GPU box:  gpurun -- 'bash tools/ab_builds.sh run [bench.py arguments ...]'
# `run` alternates the two libraries three times through GRDMA_LIB_PATH and prints `value` and ms_per_step of each run;
# default arguments: the reference's default knobs on one connection (value_ring4096_sge30's command).
# (A library older than the Python binding lacks its newer symbols and fails to load: compare revisions of one API.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
prepare)
  rev=${2:?git revision}
  mkdir -p $R/build/ab
  python -c "import __graft_entry__ as g; g._build_lib(g.LIB, g._verbs_flags())"
  cp $R/grpc-rdma_amd/libgrdma_amd.so $R/build/ab/lib_b.so
  rm -rf /tmp/grdma_ab_wt; git -C $R worktree add -f /tmp/grdma_ab_wt $rev > /dev/null
  (cd /tmp/grdma_ab_wt && python -c "import __graft_entry__ as g; g._build_lib(g.LIB, g._verbs_flags())")
  cp /tmp/grdma_ab_wt/grpc-rdma_amd/libgrdma_amd.so $R/build/ab/lib_a.so
  git -C $R worktree remove --force /tmp/grdma_ab_wt
  ls -la $R/build/ab ;;
run)
  shift
  args=${*:---no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --no-fanout --conns 1 --steps 20 --warmup 3 --no-verify --reps 3 --msgs 256 --leg-msgs 256 --ring-kb 4096 --max-sge 30 --sends 64 --promise}
  cd $R
  for i in 1 2 3; do for v in a b; do
    echo -n "lib_$v "
    GRDMA_LIB_PATH=$R/build/ab/lib_$v.so python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('repetitions_ms_per_step'))"
  done; done ;;
*) echo "usage: $0 prepare <git-rev> | run [bench.py arguments]"; exit 2 ;;
esac
