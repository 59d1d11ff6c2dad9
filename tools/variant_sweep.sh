#!/bin/bash
# Headline leg under every library variant in grpc-rdma_amd/variants (built with other -D settings, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGRDMA_PLAN_LD=0 -DGRDMA_PLAN_ST=16 -o grpc-rdma_amd/variants/lib_ld0_st16.so grpc-rdma_amd/csrc/*.hip grpc-rdma_amd/csrc/*.cc
# the cache policies of the plan tiles' loads / stores, csrc/grdma_devfn.h): value, ms per step,
# kernel times.  usage (on the GPU box): bash tools/variant_sweep.sh [extra bench args]
R=$GRAFT_REPO_ROOT
cd $R
B="python $R/bench.py --no-cpu-baseline --no-tcp-baseline --no-rtt --no-extra-legs --no-small-ring --conns 1 --steps 20 --warmup 5 --reps 1 $*"
one() { tag=$1; shift; env "$@" $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('%-16s value %8.3f  ms %.4f  %s  verified %s' % ('$tag', d['value'], d['ms_per_step'], ' '.join('%s %.2f' % (n, v['us_per_launch']) for n, v in k.items()), d.get('verified')))
"; }
one product X=1
for f in grpc-rdma_amd/variants/*.so; do one $(basename $f .so) GRDMA_LIB_PATH=$R/$f GRDMA_TEST_ALLOW_EMU=1; done
