#!/bin/bash
# Round 3 record: the gpu suite (gate), the full bench line, planner phases, the profile sets
# (tools/prof_all.sh, prof_h2.sh, prof_vtable.sh) -> gpurun_out/{gate3,profiles_new,prof_h2,prof_vtable}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/round3_gate.sh
timeout 60 python tools/plan_phases.py > gpurun_out/gate3/phases.log 2>&1 < /dev/null; tail -3 gpurun_out/gate3/phases.log
timeout 400 bash tools/prof_all.sh < /dev/null 2>&1 | tail -24
timeout 200 bash tools/prof_h2.sh < /dev/null 2>&1 | grep "k_h2" | tail -12
GRPC_RDMA_RING_BUFFER_SIZE_KB=131072 timeout 200 bash tools/prof_vtable.sh < /dev/null 2>&1 | head -12
