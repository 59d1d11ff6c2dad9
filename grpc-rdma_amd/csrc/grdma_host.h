// Host-only logic (no HIP): env/config parsing and the scalar ring arithmetic
// the control plane needs.  Testable without a GPU.
#ifndef GRDMA_HOST_H
#define GRDMA_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
// GetWritableSize(head, tail), ring_buffer.cc:99-116
uint64_t grdma_host_free_size(uint64_t cap, uint64_t head, uint64_t tail);
uint64_t grdma_host_writable(uint64_t cap, uint64_t head, uint64_t tail);
// GetEncodedSize / CalculateWritableSize, ring_buffer.h:180-189
uint64_t grdma_host_encoded_size(uint64_t payload);
uint64_t grdma_host_calc_writable(uint64_t space);
#ifdef __cplusplus
}
#endif
#endif
