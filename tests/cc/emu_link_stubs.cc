// TEST INFRASTRUCTURE.  libgrdma_emu.so leaves out the persistent link engine (csrc/grdma_link.hip: 512
// workgroups of 256 threads that talk to each other while resident -- not something a one-workgroup-at-a-time
// emulator can run); its entry points report "not supported".
#include "wave_emu.h"
struct lk_ctl;
extern "C" {
hipError_t grdma_launch_link(lk_ctl* const*, uint32_t, uint32_t, uint64_t, hipStream_t) { return hipErrorNotSupported; }
uint32_t grdma_link_resident_blocks(void) { return 0; }
}
