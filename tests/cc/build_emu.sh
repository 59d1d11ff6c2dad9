#!/bin/bash
# TEST INFRASTRUCTURE: the product sources compiled for the CPU over the wave emulator and the HIP API stand-in
# (tests/cc/wave_emu.h, hip_api_emu.h) -> oracle/_build/libgrdma_emu.so.  Load it with
# GRDMA_LIB_PATH=oracle/_build/libgrdma_emu.so to run the parity tests without a GPU (tests/test_emu_pair.py).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=$R/oracle/_build
mkdir -p $OUT/emu_obj
FLAGS="-O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -pthread -Wno-unused-value -Wno-unknown-attributes -Wno-ignored-attributes -I$R/tests/cc -I$R/tests/cc/emu_include"
objs=""
pids=""
for f in grdma_kernels.hip grdma_rx_plan.hip grdma_tx_fast.hip grdma_zc.hip grdma_h2.hip grdma_pair.hip grdma_host.cc grdma_endpoint.cc grdma_stats_time.cc; do
  o=$OUT/emu_obj/${f%.*}.o
  rm -f $o
  $CXX $FLAGS -x c++ -c $R/grpc-rdma_amd/csrc/$f -o $o &
  pids="$pids $!"
  objs="$objs $o"
done
# the NIC wire back end (csrc/grdma_wire_verbs.cc) against the verbs stand-in of oracle/fakeverbs: the build and GPU
# images have no <infiniband/verbs.h>, so this is where that file's logic is compiled for real and run
rm -f $OUT/emu_obj/grdma_wire_verbs.o $OUT/emu_obj/fakeverbs.o
$CXX $FLAGS -I$R/oracle/fakeverbs -DGRDMA_WITH_VERBS -DGRDMA_VERBS_HAVE_DMABUF -c $R/grpc-rdma_amd/csrc/grdma_wire_verbs.cc -o $OUT/emu_obj/grdma_wire_verbs.o &
pids="$pids $!"
$CXX $FLAGS -I$R/oracle/fakeverbs -c $R/oracle/fakeverbs/fakeverbs.cc -o $OUT/emu_obj/fakeverbs.o &
pids="$pids $!"
objs="$objs $OUT/emu_obj/grdma_wire_verbs.o $OUT/emu_obj/fakeverbs.o"
$CXX $FLAGS -c $R/tests/cc/emu_segv.cc -o $OUT/emu_obj/segv.o &
pids="$pids $!"
# (a compile that fails must fail the build: a stale library would otherwise be tested in its place)
for p in $pids; do wait $p || { echo "build_emu.sh: a source failed to compile" >&2; rm -f $OUT/libgrdma_emu.so; exit 1; }; done
$CXX -shared -pthread -o $OUT/libgrdma_emu.so $objs $OUT/emu_obj/segv.o -rdynamic
echo "built $OUT/libgrdma_emu.so"
