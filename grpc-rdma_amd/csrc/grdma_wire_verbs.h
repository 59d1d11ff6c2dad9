// Internal interface of the NIC wire back end (csrc/grdma_wire_verbs.cc): plain pointers and sizes, so that the file
// needs nothing of the pair's internals and compiles for the CPU suite against the verbs stand-in.
#ifndef GRDMA_WIRE_VERBS_H
#define GRDMA_WIRE_VERBS_H
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "../../include/grdma_amd.h"

// UNSUPPORTED: built without verbs; DEVICE: no (such) RDMA device; SETUP: registration / queue-pair bring-up failed, the
// pair is unusable but nothing was sent; WIRE: a posted write failed or never completed -- the queue pair is dead
enum { GRDMA_VERBS_ERR_UNSUPPORTED = 7, GRDMA_VERBS_ERR_DEVICE = 2, GRDMA_VERBS_ERR_SETUP = 3, GRDMA_VERBS_ERR_WIRE = 4 };

struct grdma_verbs_wire;
bool grdma_verbs_available();
grdma_verbs_wire* grdma_verbs_open(const char* device, int port, int gid_index, void* ring, size_t ring_size, int ring_dmabuf_fd,
                                   void* staging, size_t staging_size, void* status_send, void* status_recv, size_t status_size,
                                   std::string* err);
int grdma_verbs_address_of(const grdma_verbs_wire* w, grdma_verbs_address* out);
int grdma_verbs_connect(grdma_verbs_wire* w, const grdma_verbs_address* peer, std::string* err);
int grdma_verbs_post_data(grdma_verbs_wire* w, const uint64_t wr_off[2], const uint64_t wr_len[2], uint64_t wr_count, std::string* err);
int grdma_verbs_post_status(grdma_verbs_wire* w, std::string* err);
void grdma_verbs_counts(const grdma_verbs_wire* w, uint64_t out[3]);  // data writes posted, status writes posted, completions reaped
void grdma_verbs_close(grdma_verbs_wire* w);
#endif
