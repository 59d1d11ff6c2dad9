#!/bin/bash
# Runs one `-m gpu` test against the emulated library (tests/cc/build_emu.sh -> oracle/_build/libgrdma_emu.so) and, if the
# wave emulator aborts because two lanes of a wave met at different cross-lane operations -- or a kernel faults --,
# symbolizes both call sites.  usage: tools/emu_site.sh <pytest node id>
# More knobs of the emulator (tests/cc/wave_emu.h, hip_api_emu.h):
#   EMU_TRACE=1        per-lane (tag, value) logs from emu::trace() calls added to a kernel while debugging; the first
#                      entry where a lane differs from lane 0 is printed when the wave ends or aborts
#   EMU_SEGV_TRACE=1   faulting address + backtrace on SIGSEGV inside the library
#   EMU_GUARD_ALLOC=1  every device allocation ends in front of an inaccessible page (overruns fault at once)
cd "$(dirname "$0")/.."
out=$(GRDMA_LIB_PATH=$PWD/oracle/_build/libgrdma_emu.so GRDMA_TEST_ALLOW_EMU=1 EMU_SEGV_TRACE=1 timeout 900 python -m pytest "$1" -m gpu -x -q -s -p no:faulthandler 2>&1)
echo "$out" | grep -E "^wave_emu|passed|failed|^emu: SIGSEGV|^E  " | head -8
base_line=$(echo "$out" | grep -E "libgrdma_emu.so\(\+0x" | head -4)
for a in $(echo "$out" | grep -oE "libgrdma_emu.so\(\+0x[0-9a-f]+" | sed 's/.*+//' | head -4); do
  /opt/rocm/lib/llvm/bin/llvm-symbolizer --inlining -e oracle/_build/libgrdma_emu.so $a | grep -B1 "csrc/" | grep -v "^--" | head -8
  echo "--"
done
# the other lane's site (absolute address): compute offset from the first symbolized frame
l=$(echo "$out" | grep "^wave_emu: lane" | head -1)
if [ -n "$l" ]; then
  a1=$(echo "$l" | grep -oE "called from 0x[0-9a-f]+" | sed 's/called from //')
  a2=$(echo "$l" | grep -oE "the one from 0x[0-9a-f]+" | sed 's/the one from //')
  f=$(echo "$out" | grep -E "libgrdma_emu.so\(\+0x" | sed -n 2p)
  off=$(echo "$f" | grep -oE "\+0x[0-9a-f]+" | tr -d '+'); abs=$(echo "$f" | grep -oE "\[0x[0-9a-f]+\]" | tr -d '[]')
  base=$((abs - off))
  printf "other lane site: "; /opt/rocm/lib/llvm/bin/llvm-symbolizer --inlining -e oracle/_build/libgrdma_emu.so $(printf "0x%x" $((a2 - base - 1))) | grep "csrc/" | head -3
fi
