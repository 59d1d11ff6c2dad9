/* TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
 *
 * Type-only stand-in for <infiniband/verbs.h>, written from scratch, so that
 * the reference's own src/core/lib/ibverbs/ring_buffer.{h,cc} can be compiled
 * unmodified in a container that has no libibverbs.  Only the declarations
 * that ring_buffer.h / ring_buffer.cc / buffer.h actually name are provided
 * (SURVEY.md section 8c).  Nothing here performs I/O.
 */
#ifndef GRDMA_ORACLE_SHIM_VERBS_H
#define GRDMA_ORACLE_SHIM_VERBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ibv_pd {
  int unused;
};

struct ibv_mr {
  void* addr;
  uint64_t length;
  uint32_t lkey;
  uint32_t rkey;
};

struct ibv_sge {
  uint64_t addr;
  uint32_t length;
  uint32_t lkey;
};

enum ibv_wr_opcode {
  IBV_WR_RDMA_WRITE = 0,
  IBV_WR_SEND_WITH_IMM = 3
};

enum ibv_send_flags { IBV_SEND_SIGNALED = 2 };

struct ibv_send_wr {
  uint64_t wr_id;
  struct ibv_send_wr* next;
  struct ibv_sge* sg_list;
  int num_sge;
  enum ibv_wr_opcode opcode;
  unsigned int send_flags;
  struct {
    struct {
      uint64_t remote_addr;
      uint32_t rkey;
    } rdma;
  } wr;
};

union ibv_gid {
  uint8_t raw[16];
};

#ifdef __cplusplus
}
#endif

#endif /* GRDMA_ORACLE_SHIM_VERBS_H */
