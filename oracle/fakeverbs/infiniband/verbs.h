/* TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
 *
 * A software stand-in for libibverbs, written from scratch: enough of <infiniband/verbs.h> that the reference's own
 * src/core/lib/ibverbs/{pair,ring_buffer,device,memory_region,buffer,address,config}.cc compile UNMODIFIED and two
 * PairPollable objects in one process can be connected to each other.  oracle/fakeverbs/fakeverbs.cc implements the
 * calls over process memory: a queue pair is an entry of a table, IBV_WR_RDMA_WRITE is a gather + memcpy into the
 * (registered) remote address, IBV_WR_SEND_WITH_IMM lands in the first receive posted on the destination queue pair,
 * completions are queued on the completion queue and handed out by ibv_poll_cq.  In-order, immediate, loss-free --
 * what a reliable-connected queue pair promises, minus the wire.  (The type-only oracle/shim/infiniband/verbs.h stays
 * for the library that only builds the reference's ring codec.)
 */
#ifndef GRDMA_ORACLE_FAKEVERBS_H
#define GRDMA_ORACLE_FAKEVERBS_H

#include <errno.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>  /* (the real header brings these in as well; the reference's files rely on it) */

#ifdef __cplusplus
extern "C" {
#endif

struct ibv_device {
  char name[64];
};
struct ibv_context {
  struct ibv_device* device;
};
struct ibv_pd {
  struct ibv_context* context;
};
struct ibv_mr {
  struct ibv_context* context;
  struct ibv_pd* pd;
  void* addr;
  size_t length;
  uint32_t handle;
  uint32_t lkey;
  uint32_t rkey;
};
struct ibv_sge {
  uint64_t addr;
  uint32_t length;
  uint32_t lkey;
};
union ibv_gid {
  uint8_t raw[16];
  struct {
    uint64_t subnet_prefix;
    uint64_t interface_id;
  } global;
};

struct ibv_device_attr {
  int max_sge;
  int max_qp_wr;
  int max_cqe;
};
enum ibv_port_state { IBV_PORT_ACTIVE = 4 };
struct ibv_port_attr {
  enum ibv_port_state state;
  uint16_t lid;
  uint8_t link_layer;
  int active_mtu;
};

enum ibv_access_flags { IBV_ACCESS_LOCAL_WRITE = 1, IBV_ACCESS_REMOTE_WRITE = 2, IBV_ACCESS_REMOTE_READ = 4 };
/* (named by headers of the reference's other RDMA transport that rdma_bp_posix.cc includes; never used here) */
struct ibv_comp_channel { int fd; };
struct ibv_async_event { int event_type; };

enum ibv_qp_type { IBV_QPT_RC = 2 };
enum ibv_qp_state { IBV_QPS_RESET, IBV_QPS_INIT, IBV_QPS_RTR, IBV_QPS_RTS, IBV_QPS_SQD, IBV_QPS_SQE, IBV_QPS_ERR };
enum ibv_mtu { IBV_MTU_256 = 1, IBV_MTU_512 = 2, IBV_MTU_1024 = 3, IBV_MTU_2048 = 4, IBV_MTU_4096 = 5 };
enum ibv_qp_attr_mask {
  IBV_QP_STATE = 1 << 0, IBV_QP_CUR_STATE = 1 << 1, IBV_QP_EN_SQD_ASYNC_NOTIFY = 1 << 2, IBV_QP_ACCESS_FLAGS = 1 << 3,
  IBV_QP_PKEY_INDEX = 1 << 4, IBV_QP_PORT = 1 << 5, IBV_QP_QKEY = 1 << 6, IBV_QP_AV = 1 << 7, IBV_QP_PATH_MTU = 1 << 8,
  IBV_QP_TIMEOUT = 1 << 9, IBV_QP_RETRY_CNT = 1 << 10, IBV_QP_RNR_RETRY = 1 << 11, IBV_QP_RQ_PSN = 1 << 12,
  IBV_QP_MAX_QP_RD_ATOMIC = 1 << 13, IBV_QP_ALT_PATH = 1 << 14, IBV_QP_MIN_RNR_TIMER = 1 << 15, IBV_QP_SQ_PSN = 1 << 16,
  IBV_QP_MAX_DEST_RD_ATOMIC = 1 << 17, IBV_QP_PATH_MIG_STATE = 1 << 18, IBV_QP_CAP = 1 << 19, IBV_QP_DEST_QPN = 1 << 20
};

struct ibv_cq {
  struct ibv_context* context;
  void* cq_context;
  int cqe;
  uint32_t handle;
};
struct ibv_qp_cap {
  uint32_t max_send_wr, max_recv_wr, max_send_sge, max_recv_sge, max_inline_data;
};
struct ibv_qp_init_attr {
  void* qp_context;
  struct ibv_cq* send_cq;
  struct ibv_cq* recv_cq;
  void* srq;
  struct ibv_qp_cap cap;
  enum ibv_qp_type qp_type;
  int sq_sig_all;
};
struct ibv_qp {
  struct ibv_context* context;
  void* qp_context;
  struct ibv_pd* pd;
  struct ibv_cq* send_cq;
  struct ibv_cq* recv_cq;
  uint32_t handle;
  uint32_t qp_num;
  enum ibv_qp_state state;
  enum ibv_qp_type qp_type;
};
struct ibv_global_route {
  union ibv_gid dgid;
  uint32_t flow_label;
  uint8_t sgid_index, hop_limit, traffic_class;
};
struct ibv_ah_attr {
  struct ibv_global_route grh;
  uint16_t dlid;
  uint8_t sl, src_path_bits, static_rate, is_global, port_num;
};
struct ibv_qp_attr {
  enum ibv_qp_state qp_state, cur_qp_state;
  enum ibv_mtu path_mtu;
  int path_mig_state;
  uint32_t qkey, rq_psn, sq_psn, dest_qp_num;
  unsigned int qp_access_flags;
  struct ibv_qp_cap cap;
  struct ibv_ah_attr ah_attr, alt_ah_attr;
  uint16_t pkey_index, alt_pkey_index;
  uint8_t en_sqd_async_notify, sq_draining, max_rd_atomic, max_dest_rd_atomic, min_rnr_timer, port_num, timeout,
      retry_cnt, rnr_retry, alt_port_num, alt_timeout;
};

enum ibv_wr_opcode { IBV_WR_RDMA_WRITE, IBV_WR_RDMA_WRITE_WITH_IMM, IBV_WR_SEND, IBV_WR_SEND_WITH_IMM, IBV_WR_RDMA_READ };
enum ibv_send_flags { IBV_SEND_FENCE = 1, IBV_SEND_SIGNALED = 2, IBV_SEND_SOLICITED = 4, IBV_SEND_INLINE = 8 };
struct ibv_send_wr {
  uint64_t wr_id;
  struct ibv_send_wr* next;
  struct ibv_sge* sg_list;
  int num_sge;
  enum ibv_wr_opcode opcode;
  unsigned int send_flags;
  uint32_t imm_data;
  union {
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
enum ibv_wc_status { IBV_WC_SUCCESS = 0, IBV_WC_LOC_LEN_ERR, IBV_WC_LOC_PROT_ERR, IBV_WC_REM_ACCESS_ERR, IBV_WC_RNR_RETRY_EXC_ERR, IBV_WC_GENERAL_ERR };
enum ibv_wc_opcode { IBV_WC_SEND, IBV_WC_RDMA_WRITE, IBV_WC_RDMA_READ, IBV_WC_RECV = 1 << 7, IBV_WC_RECV_RDMA_WITH_IMM };
struct ibv_wc {
  uint64_t wr_id;
  enum ibv_wc_status status;
  enum ibv_wc_opcode opcode;
  uint32_t vendor_err;
  uint32_t byte_len;
  uint32_t imm_data;
  uint32_t qp_num;
  uint32_t src_qp;
  unsigned int wc_flags;
};

struct ibv_device** ibv_get_device_list(int* num_devices);
void ibv_free_device_list(struct ibv_device** list);
struct ibv_context* ibv_open_device(struct ibv_device* device);
int ibv_close_device(struct ibv_context* context);
int ibv_query_device(struct ibv_context* context, struct ibv_device_attr* attr);
int ibv_query_port(struct ibv_context* context, uint8_t port_num, struct ibv_port_attr* attr);
int ibv_query_gid(struct ibv_context* context, uint8_t port_num, int index, union ibv_gid* gid);
struct ibv_pd* ibv_alloc_pd(struct ibv_context* context);
int ibv_dealloc_pd(struct ibv_pd* pd);
struct ibv_mr* ibv_reg_mr(struct ibv_pd* pd, void* addr, size_t length, int access);
/* (rdma-core >= 34: a memory region over a dma-buf -- how device memory is registered for GPUDirect RDMA.  Here the
 *  "device" is process memory: iova is the address, the descriptor is not looked at.) */
struct ibv_mr* ibv_reg_dmabuf_mr(struct ibv_pd* pd, uint64_t offset, size_t length, uint64_t iova, int fd, int access);
int ibv_dereg_mr(struct ibv_mr* mr);
struct ibv_cq* ibv_create_cq(struct ibv_context* context, int cqe, void* cq_context, void* channel, int comp_vector);
int ibv_destroy_cq(struct ibv_cq* cq);
struct ibv_qp* ibv_create_qp(struct ibv_pd* pd, struct ibv_qp_init_attr* attr);
int ibv_destroy_qp(struct ibv_qp* qp);
int ibv_modify_qp(struct ibv_qp* qp, struct ibv_qp_attr* attr, int attr_mask);
int ibv_query_qp(struct ibv_qp* qp, struct ibv_qp_attr* attr, int attr_mask, struct ibv_qp_init_attr* init_attr);
int ibv_post_send(struct ibv_qp* qp, struct ibv_send_wr* wr, struct ibv_send_wr** bad_wr);
int ibv_post_recv(struct ibv_qp* qp, struct ibv_recv_wr* wr, struct ibv_recv_wr** bad_wr);
int ibv_poll_cq(struct ibv_cq* cq, int num_entries, struct ibv_wc* wc);

#ifdef __cplusplus
}
#endif
#endif  /* GRDMA_ORACLE_FAKEVERBS_H */
