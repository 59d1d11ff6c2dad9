#!/bin/bash
# TEST INFRASTRUCTURE: the product sources built for gfx950 exactly as __graft_entry__.build() builds them, except that
# the NIC wire back end (csrc/grdma_wire_verbs.cc) is compiled IN -- against the verbs stand-in of oracle/fakeverbs, whose
# fabric here moves the bytes of an RDMA WRITE with the copy engine into the registered HBM ring, in address order, the
# last eight bytes last (-DFAKEVERBS_HIP).  -> oracle/_build/libgrdma_amd_fakeverbs.so: what tests/test_zz_gpu_wire_verbs.py
# loads on the MI355X (in a child pytest, GRDMA_LIB_PATH), so that registration through the ring's dma-buf, queue-pair
# bring-up, the Send's <= 2 chained writes, the status write and completion reaping run against the real k_tx_* / k_rx_*
# kernels.  The product library (grpc-rdma_amd/libgrdma_amd.so) is unchanged by this: no HCA, no verbs header, no NIC wire.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
python3 - <<PY
import os, sys
sys.path.insert(0, "$R")
import __graft_entry__ as g
fv = os.path.join(g.ROOT, "oracle", "fakeverbs")
out = os.path.join(g.ROOT, "oracle", "_build", "libgrdma_amd_fakeverbs.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
wire = os.path.join(g.CSRC, "grdma_wire_verbs.cc")
g._build_lib(out, ([], []), variant={wire: ["-DGRDMA_WITH_VERBS", "-DGRDMA_VERBS_HAVE_DMABUF", "-I" + fv],
                                     os.path.join(fv, "fakeverbs.cc"): ["-DFAKEVERBS_HIP", "-I" + fv]})
print("built", out)
PY
