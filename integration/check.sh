#!/bin/sh
# Syntax-check integration/rdma_hip_posix.cc against the REFERENCE's own headers (gRPC 1.38 +
# RR-Compound under $REF), with stand-ins for abseil and libibverbs from integration/shim.
# Nothing is linked and nothing of the reference is copied; exit status 0 = the adapter still
# matches the reference's endpoint / event-engine / slice interfaces.
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
if [ ! -f "$REF/src/core/lib/iomgr/rdma_bp_posix.h" ]; then
  echo "reference tree absent ($REF): nothing to check against" >&2
  exit 77
fi
exec ${CXX:-g++} -std=c++17 -fsyntax-only -Wall -Wno-unused-function -DGRPC_USE_IBVERBS \
  -I"$HERE/shim" -isystem "$REF" -isystem "$REF/include" -I"$HERE/../include" "$HERE/rdma_hip_posix.cc"
