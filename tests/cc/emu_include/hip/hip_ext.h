// TEST INFRASTRUCTURE: what <hip/hip_ext.h> is to a host source compiled against the HIP API stand-in
// (tests/cc/hip_api_emu.h): a CU-masked stream is a stream.
#pragma once
