#!/bin/sh
# Syntax-check the drop-in against the REFERENCE's own sources (gRPC 1.38 + RR-Compound under $REF), with stand-ins
# for abseil / HdrHistogram / libibverbs from integration/shim and the ibverbs facade of this build
# (integration/ibverbs_facade: PairPollable / Poller / Config with the reference's names over libgrdma_amd.so)
# FIRST on the include path:
#   1. integration/rdma_hip_posix.cc                              -- the endpoint a maintainer drops into src/core/lib/iomgr/
#   2. $REF/src/core/lib/iomgr/ev_epollex_rdma_bpev_linux.cc      -- the reference's event engines, UNMODIFIED: they cast the
#   3. $REF/src/core/lib/iomgr/ev_epollex_rdma_bp_linux.cc           fd's arg to PairPollable* and call HasMessage() /
#                                                                    HasPendingWrites() / get_status() / get_wakeup_fd() on it
# Nothing is linked and nothing of the reference is copied; exit status 0 = the adapter, the facade and the
# reference's endpoint / event-engine / slice interfaces still fit together.
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
if [ ! -f "$REF/src/core/lib/iomgr/rdma_bp_posix.h" ]; then
  echo "reference tree absent ($REF): nothing to check against" >&2
  exit 77
fi
CXX=${CXX:-g++}
FLAGS="-std=c++17 -fsyntax-only -Wall -Wno-unused-function -Wno-unused-variable -Wno-sign-compare -Wno-switch -DGRPC_USE_IBVERBS \
  -I$HERE/ibverbs_facade -I$HERE/shim -isystem $REF -isystem $REF/include -I$HERE/../include"
rc=0
for f in "$HERE/rdma_hip_posix.cc" "$REF/src/core/lib/iomgr/ev_epollex_rdma_bpev_linux.cc" \
         "$REF/src/core/lib/iomgr/ev_epollex_rdma_bp_linux.cc"; do
  if $CXX $FLAGS "$f"; then echo "ok: $f"; else echo "FAILED: $f" >&2; rc=1; fi
done
exit $rc
