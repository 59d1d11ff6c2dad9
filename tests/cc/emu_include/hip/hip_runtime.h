// TEST INFRASTRUCTURE: what <hip/hip_runtime.h> is to a kernel source compiled for the wave emulator.
#include "../../wave_emu.h"
