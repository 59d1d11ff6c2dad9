/* SYNTAX-CHECK SHIM (integration/README.md) -- never linked into anything.
 *
 * Type-only stand-in for <infiniband/verbs.h>, written from scratch: the reference's iomgr headers
 * (ev_posix.h -> rdma_sender_receiver.h -> rdma_conn.h / rdma_utils.h) name verbs types, and this
 * container has no libibverbs.  Only declarations; no function here has a body that does I/O.
 */
#ifndef GRDMA_INTEGRATION_SHIM_VERBS_H
#define GRDMA_INTEGRATION_SHIM_VERBS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ibv_context { int unused; };
struct ibv_pd { struct ibv_context* context; };
struct ibv_cq { int unused; };
struct ibv_qp { uint32_t qp_num; };
struct ibv_comp_channel { int fd; };
struct ibv_async_event { int event_type; };
struct ibv_device_attr { int max_qp_wr; int max_sge; int max_cqe; };
struct ibv_port_attr { int state; uint16_t lid; uint8_t link_layer; int active_mtu; };
struct ibv_mr { void* addr; size_t length; uint32_t lkey; uint32_t rkey; };
struct ibv_sge { uint64_t addr; uint32_t length; uint32_t lkey; };
union ibv_gid { uint8_t raw[16]; struct { uint64_t subnet_prefix; uint64_t interface_id; } global; };

enum ibv_access_flags {
  IBV_ACCESS_LOCAL_WRITE = 1,
  IBV_ACCESS_REMOTE_WRITE = 2,
  IBV_ACCESS_REMOTE_READ = 4
};
enum ibv_wr_opcode {
  IBV_WR_RDMA_WRITE = 0,
  IBV_WR_RDMA_WRITE_WITH_IMM = 1,
  IBV_WR_SEND = 2,
  IBV_WR_SEND_WITH_IMM = 3
};
enum ibv_send_flags { IBV_SEND_SIGNALED = 2 };
enum ibv_wc_status { IBV_WC_SUCCESS = 0 };
enum ibv_wc_opcode { IBV_WC_SEND = 0, IBV_WC_RDMA_WRITE = 1, IBV_WC_RECV = 128, IBV_WC_RECV_RDMA_WITH_IMM = 129 };

struct ibv_wc {
  uint64_t wr_id;
  enum ibv_wc_status status;
  enum ibv_wc_opcode opcode;
  uint32_t byte_len;
  uint32_t imm_data;
};
struct ibv_send_wr {
  uint64_t wr_id;
  struct ibv_send_wr* next;
  struct ibv_sge* sg_list;
  int num_sge;
  enum ibv_wr_opcode opcode;
  unsigned int send_flags;
  uint32_t imm_data;
  struct {
    struct {
      uint64_t remote_addr;
      uint32_t rkey;
    } rdma;
  } wr;
};
struct ibv_recv_wr {
  uint64_t wr_id;
  struct ibv_recv_wr* next;
  struct ibv_sge* sg_list;
  int num_sge;
};
struct ibv_qp_cap { uint32_t max_send_wr, max_recv_wr, max_send_sge, max_recv_sge, max_inline_data; };
struct ibv_qp_init_attr {
  void* qp_context;
  struct ibv_cq* send_cq;
  struct ibv_cq* recv_cq;
  struct ibv_qp_cap cap;
  int qp_type;
  int sq_sig_all;
};

struct ibv_mr* ibv_reg_mr(struct ibv_pd* pd, void* addr, size_t length, int access);
int ibv_dereg_mr(struct ibv_mr* mr);

#ifdef __cplusplus
}
#endif

#endif /* GRDMA_INTEGRATION_SHIM_VERBS_H */
