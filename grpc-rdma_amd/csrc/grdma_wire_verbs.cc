// NIC wire back end: the ibverbs side of a pair whose ring a remote HCA writes (GPUDirect RDMA) -- the place of
// PairPollable's verbs calls in the reference:
//   Init()           ibv_reg_mr of send buffer, ring and the two status blocks   src/core/lib/ibverbs/pair.cc:107-119,
//                                                                                 rdma_utils.h:108-160 (MemoryRegion)
//   initQPs/Connect  RESET -> INIT -> RTR -> RTS against the peer's Address       pair.cc:143-262
//   Send()           the <= 2 chained IBV_WR_RDMA_WRITEs of GetWriteRequests      pair.cc:709-734, ring_buffer.cc:261-330
//   waitDataWrites   reaping their completions                                    pair.cc:560-585
//   updateStatus()   the 16-byte credit report as one more RDMA WRITE             pair.cc:624-641, 587-622 (postWrite)
// Here the byte work -- record encoding, the split at the ring's end, credit accounting -- has happened on the device:
// the send planner leaves {remote ring offset, length} x <= 2 in the Send's result block (grdma_tx_result::wr_off /
// wr_len: K2 of SURVEY.md 2.3) and the encoded records in the staging buffer; this file only registers the memory and
// posts what the planner emitted.  The ring is registered through its dma-buf (ibv_reg_dmabuf_mr on the fd of
// grdma_pair_export_ring_dmabuf) when the verbs library offers that call, with ibv_reg_mr otherwise.
//
// Compiled in by -DGRDMA_WITH_VERBS, which __graft_entry__.build() passes together with -libverbs when the build host has
// <infiniband/verbs.h> (and -DGRDMA_VERBS_HAVE_DMABUF when that header declares ibv_reg_dmabuf_mr): the library then
// carries no undefined ibv_* symbol it was not linked for.  The images this repository is built and measured on have
// neither the header nor an HCA: there every entry point reports GRDMA_ERR_UNSUPPORTED, and the logic runs against the
// verbs stand-in of oracle/fakeverbs instead (test infrastructure) -- in the CPU suite as the product sources over the
// wave emulator (tests/cc/build_emu.sh), and on the MI355X as the product sources built for gfx950 with that stand-in's
// "HCA" moving the bytes of an RDMA WRITE into the registered HBM ring in address order (tests/cc/build_fakeverbs_hip.sh
// -> oracle/_build/libgrdma_amd_fakeverbs.so): tests/test_zz_gpu_wire_verbs.py replays the reference-made endpoint
// traces through two pairs connected by that fabric, the real k_tx_* / k_rx_* kernels on either side of it.
#include "grdma_wire_verbs.h"

#include <stdio.h>
#include <string.h>

#ifdef GRDMA_WITH_VERBS
#include <infiniband/verbs.h>
#include <sched.h>

#include <chrono>

struct grdma_verbs_wire {
  ibv_context* ctx = nullptr;
  ibv_pd* pd = nullptr;
  ibv_cq* cq = nullptr;
  ibv_qp* qp = nullptr;
  ibv_mr* mr_ring = nullptr;         // recv_buffers_[kDataBuffer]: remote-writable
  ibv_mr* mr_status_recv = nullptr;  // recv_buffers_[kStatusBuffer]: remote-writable
  ibv_mr* mr_staging = nullptr;      // send_buffers_[kDataBuffer]
  ibv_mr* mr_status_send = nullptr;  // send_buffers_[kStatusBuffer]
  uint8_t* staging = nullptr;
  void* status_send = nullptr;
  int port = 1, gid_index = 0;
  uint32_t max_sge = 30;
  grdma_verbs_address self, peer;
  bool connected = false;
  bool dead = false;                              // a work completion came back in error: the queue pair is in the error state
  uint64_t pending_data = 0, pending_status = 0;  // pending_write_num_data_ / _status_ (pair.h:191-192)
  uint64_t posted_data = 0, posted_status = 0, reaped = 0;
};

namespace {
enum { WR_ID_DATA = 1, WR_ID_STATUS = 2 };  // pair.h:38-39
int failv(std::string* err, int code, const char* what) {
  if (err) *err = what;
  return -code;
}
}  // namespace

bool grdma_verbs_available() { return true; }

grdma_verbs_wire* grdma_verbs_open(const char* device, int port, int gid_index, void* ring, size_t ring_size, int ring_dmabuf_fd,
                                   void* staging, size_t staging_size, void* status_send, void* status_recv, size_t status_size,
                                   std::string* err) {
  int n = 0;
  ibv_device** list = ibv_get_device_list(&n);
  if (!list || n <= 0) {
    failv(err, GRDMA_VERBS_ERR_DEVICE, "ibv_get_device_list: no RDMA device");
    return nullptr;
  }
  ibv_device* dev = list[0];
  if (device && device[0]) {
    dev = nullptr;
    for (int i = 0; i < n; i++)
      if (strcmp(list[i]->name, device) == 0) dev = list[i];
  }
  if (!dev) {
    ibv_free_device_list(list);
    failv(err, GRDMA_VERBS_ERR_DEVICE, "the named RDMA device does not exist (GRPC_RDMA_DEVICE_NAME)");
    return nullptr;
  }
  grdma_verbs_wire* w = new grdma_verbs_wire();
  w->ctx = ibv_open_device(dev);
  ibv_free_device_list(list);
  auto bail = [&](const char* what) -> grdma_verbs_wire* {
    failv(err, GRDMA_VERBS_ERR_SETUP, what);
    grdma_verbs_close(w);
    return nullptr;
  };
  if (!w->ctx) return bail("ibv_open_device failed");
  w->port = port > 0 ? port : 1;
  w->gid_index = gid_index;
  ibv_device_attr da;
  if (ibv_query_device(w->ctx, &da) != 0) return bail("ibv_query_device failed");
  w->max_sge = da.max_sge > 0 ? (uint32_t)da.max_sge : 1;
  w->pd = ibv_alloc_pd(w->ctx);
  if (!w->pd) return bail("ibv_alloc_pd failed");
  // rdma_utils.h:108-160: send buffers local-write, receive buffers remote-write too
  const int acc_send = IBV_ACCESS_LOCAL_WRITE, acc_recv = IBV_ACCESS_LOCAL_WRITE | IBV_ACCESS_REMOTE_WRITE;
#ifdef GRDMA_VERBS_HAVE_DMABUF
  if (ring_dmabuf_fd >= 0) w->mr_ring = ibv_reg_dmabuf_mr(w->pd, 0, ring_size, (uint64_t)ring, ring_dmabuf_fd, acc_recv);
#else
  (void)ring_dmabuf_fd;
#endif
  if (!w->mr_ring) w->mr_ring = ibv_reg_mr(w->pd, ring, ring_size, acc_recv);
  w->mr_status_recv = ibv_reg_mr(w->pd, status_recv, status_size, acc_recv);
  w->mr_staging = ibv_reg_mr(w->pd, staging, staging_size, acc_send);
  w->mr_status_send = ibv_reg_mr(w->pd, status_send, status_size, acc_send);
  if (!w->mr_ring || !w->mr_status_recv || !w->mr_staging || !w->mr_status_send) return bail("memory registration failed (ibv_reg_mr)");
  w->staging = static_cast<uint8_t*>(staging);
  w->status_send = status_send;
  // pair.cc:143-168: one completion queue for both directions, a reliable-connection queue pair
  w->cq = ibv_create_cq(w->ctx, 4096, nullptr, nullptr, 0);
  if (!w->cq) return bail("ibv_create_cq failed");
  ibv_qp_init_attr qa;
  memset(&qa, 0, sizeof(qa));
  qa.send_cq = w->cq;
  qa.recv_cq = w->cq;
  qa.cap.max_send_wr = 4096;
  qa.cap.max_recv_wr = 16;
  qa.cap.max_send_sge = 1;   // (every work request of this wire is ONE contiguous piece of staging: the device has
  qa.cap.max_recv_sge = 1;   //  already gathered the slices -- the reference spends its max_sge entries on that gather)
  qa.qp_type = IBV_QPT_RC;
  w->qp = ibv_create_qp(w->pd, &qa);
  if (!w->qp) return bail("ibv_create_qp failed");
  ibv_port_attr pa;
  if (ibv_query_port(w->ctx, (uint8_t)w->port, &pa) != 0) return bail("ibv_query_port failed");
  memset(&w->self, 0, sizeof(w->self));
  w->self.qpn = w->qp->qp_num;
  w->self.lid = pa.lid;
  ibv_gid gid;
  if (ibv_query_gid(w->ctx, (uint8_t)w->port, gid_index, &gid) == 0) memcpy(w->self.gid, &gid, 16);
  w->self.psn = 0;
  w->self.ring_addr = (uint64_t)ring;
  w->self.ring_rkey = w->mr_ring->rkey;
  w->self.ring_size = ring_size;
  w->self.status_addr = (uint64_t)status_recv;
  w->self.status_rkey = w->mr_status_recv->rkey;
  w->self.status_size = (uint32_t)status_size;
  return w;
}

int grdma_verbs_address_of(const grdma_verbs_wire* w, grdma_verbs_address* out) {
  if (!w || !out) return -GRDMA_VERBS_ERR_SETUP;
  *out = w->self;
  return 0;
}

// RESET -> INIT -> RTR -> RTS (pair.cc:170-262)
int grdma_verbs_connect(grdma_verbs_wire* w, const grdma_verbs_address* peer, std::string* err) {
  if (!w || !peer) return failv(err, GRDMA_VERBS_ERR_SETUP, "null argument");
  if (peer->ring_size != w->self.ring_size) return failv(err, GRDMA_VERBS_ERR_SETUP, "ring sizes differ (pair.cc:149)");
  ibv_qp_attr a;
  memset(&a, 0, sizeof(a));
  a.qp_state = IBV_QPS_INIT;
  a.pkey_index = 0;
  a.port_num = (uint8_t)w->port;
  a.qp_access_flags = IBV_ACCESS_LOCAL_WRITE | IBV_ACCESS_REMOTE_WRITE;
  if (ibv_modify_qp(w->qp, &a, IBV_QP_STATE | IBV_QP_PKEY_INDEX | IBV_QP_PORT | IBV_QP_ACCESS_FLAGS) != 0)
    return failv(err, GRDMA_VERBS_ERR_SETUP, "queue pair: INIT failed");
  memset(&a, 0, sizeof(a));
  a.qp_state = IBV_QPS_RTR;
  a.path_mtu = IBV_MTU_4096;
  a.dest_qp_num = peer->qpn;
  a.rq_psn = peer->psn;
  a.max_dest_rd_atomic = 1;
  a.min_rnr_timer = 12;
  a.ah_attr.is_global = 1;
  memcpy(&a.ah_attr.grh.dgid, peer->gid, 16);
  a.ah_attr.grh.sgid_index = (uint8_t)w->gid_index;
  a.ah_attr.grh.hop_limit = 1;
  a.ah_attr.dlid = peer->lid;
  a.ah_attr.port_num = (uint8_t)w->port;
  if (ibv_modify_qp(w->qp, &a, IBV_QP_STATE | IBV_QP_AV | IBV_QP_PATH_MTU | IBV_QP_DEST_QPN | IBV_QP_RQ_PSN |
                                   IBV_QP_MAX_DEST_RD_ATOMIC | IBV_QP_MIN_RNR_TIMER) != 0)
    return failv(err, GRDMA_VERBS_ERR_SETUP, "queue pair: RTR failed");
  memset(&a, 0, sizeof(a));
  a.qp_state = IBV_QPS_RTS;
  a.sq_psn = w->self.psn;
  a.timeout = 14;
  a.retry_cnt = 7;
  a.rnr_retry = 7;
  a.max_rd_atomic = 1;
  if (ibv_modify_qp(w->qp, &a, IBV_QP_STATE | IBV_QP_TIMEOUT | IBV_QP_RETRY_CNT | IBV_QP_RNR_RETRY | IBV_QP_SQ_PSN |
                                   IBV_QP_MAX_QP_RD_ATOMIC) != 0)
    return failv(err, GRDMA_VERBS_ERR_SETUP, "queue pair: RTS failed");
  w->peer = *peer;
  w->connected = true;
  return 0;
}

namespace {
// waitDataWrites / pollCompletion (pair.cc:560-585, 500-558): every signalled write of this wire is reaped here
// A wire whose queue pair has gone into the error state (a failed completion, a post the HCA refused half-way, a
// completion that never came): nothing posted on it will complete any more, so nothing is waited for either.
int wire_dead(grdma_verbs_wire* w, std::string* err, const char* what) {
  w->dead = true;
  w->pending_data = w->pending_status = 0;
  return failv(err, GRDMA_VERBS_ERR_WIRE, what);
}

int reap(grdma_verbs_wire* w, std::string* err) {
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t idle = 0;
  while (w->pending_data + w->pending_status > 0) {
    ibv_wc wc[16];
    const int n = ibv_poll_cq(w->cq, 16, wc);
    if (n < 0) return wire_dead(w, err, "ibv_poll_cq failed");
    for (int i = 0; i < n; i++) {
      if (wc[i].status != IBV_WC_SUCCESS) {
        char buf[96];
        snprintf(buf, sizeof(buf), "work completion %llu with status %d", (unsigned long long)wc[i].wr_id, (int)wc[i].status);
        return wire_dead(w, err, buf);
      }
      if (wc[i].wr_id == WR_ID_DATA && w->pending_data) w->pending_data--;
      else if (wc[i].wr_id == WR_ID_STATUS && w->pending_status) w->pending_status--;
      w->reaped++;
    }
    if (n > 0) {
      idle = 0;
      continue;
    }
    // (a write of a megabyte takes tens of microseconds on the wire: spin briefly, then let the core go)
    if (++idle > 2000) sched_yield();
    if ((idle & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
      return wire_dead(w, err, "a posted RDMA WRITE did not complete within 10 s");
  }
  return 0;
}

// What a failed ibv_post_send leaves behind: the requests in front of `bad` were taken by the HCA and will complete,
// `bad` and everything behind it were not (ibv_post_send(3)).
uint64_t posted_before(const ibv_send_wr* first, const ibv_send_wr* bad) {
  uint64_t n = 0;
  for (const ibv_send_wr* q = first; q && q != bad; q = q->next) n++;
  return n;
}
}  // namespace

// The Send's <= 2 RDMA WRITEs (pair.cc:709-734): piece k lies at staging + (length of the pieces in front of it) and goes to
// the peer's ring at wr_off[k]; the second one exists when the records wrap at the ring's end (ring_buffer.cc:261-330).
// Chained through wr.next and posted with ONE ibv_post_send, both signalled, reaped before the call returns (the
// staging buffer is reused by the next Send: waitDataWrites, pair.cc:667).
int grdma_verbs_post_data(grdma_verbs_wire* w, const uint64_t wr_off[2], const uint64_t wr_len[2], uint64_t wr_count, std::string* err) {
  if (!w || !w->connected) return failv(err, GRDMA_VERBS_ERR_WIRE, "the wire is not connected");
  if (w->dead) return failv(err, GRDMA_VERBS_ERR_WIRE, "the queue pair is in the error state (an earlier write failed)");
  if (wr_count == 0) return 0;
  if (wr_count > 2) return failv(err, GRDMA_VERBS_ERR_WIRE, "a Send has at most two write requests");
  ibv_send_wr wrs[2];
  ibv_sge sge[2];
  memset(wrs, 0, sizeof(wrs));
  uint64_t staged = 0;
  for (uint64_t k = 0; k < wr_count; k++) {
    if (wr_len[k] == 0 || wr_off[k] + wr_len[k] > w->peer.ring_size || wr_len[k] > 0xFFFFFFFFull)
      return failv(err, GRDMA_VERBS_ERR_WIRE, "write request outside the peer's ring");
    sge[k].addr = (uint64_t)(w->staging + staged);
    sge[k].length = (uint32_t)wr_len[k];
    sge[k].lkey = w->mr_staging->lkey;
    wrs[k].wr_id = WR_ID_DATA;
    wrs[k].sg_list = &sge[k];
    wrs[k].num_sge = 1;
    wrs[k].opcode = IBV_WR_RDMA_WRITE;
    wrs[k].send_flags = IBV_SEND_SIGNALED;
    wrs[k].wr.rdma.remote_addr = w->peer.ring_addr + wr_off[k];
    wrs[k].wr.rdma.rkey = w->peer.ring_rkey;
    wrs[k].next = (k + 1 < wr_count) ? &wrs[k + 1] : nullptr;
    staged += wr_len[k];
  }
  ibv_send_wr* bad = nullptr;
  if (ibv_post_send(w->qp, &wrs[0], &bad) != 0) {
    // (counted only once the HCA has them: what it took in front of `bad` is reaped, then the wire is given up -- a
    //  Send of which one half went out cannot be repaired from here)
    const uint64_t took = posted_before(&wrs[0], bad);
    w->pending_data += took;
    w->posted_data += took;
    std::string ignored;
    (void)reap(w, &ignored);
    return wire_dead(w, err, "ibv_post_send (data) failed");
  }
  w->pending_data += wr_count;
  w->posted_data += wr_count;
  return reap(w, err);
}

// updateStatus() (pair.cc:624-641): the 16 bytes of status_send -- remote_head as the drain's planner left it -- into
// the peer's status_recv.
int grdma_verbs_post_status(grdma_verbs_wire* w, std::string* err) {
  if (!w || !w->connected) return failv(err, GRDMA_VERBS_ERR_WIRE, "the wire is not connected");
  if (w->dead) return failv(err, GRDMA_VERBS_ERR_WIRE, "the queue pair is in the error state (an earlier write failed)");
  ibv_sge sge;
  sge.addr = (uint64_t)w->status_send;
  sge.length = w->peer.status_size;
  sge.lkey = w->mr_status_send->lkey;
  ibv_send_wr wr;
  memset(&wr, 0, sizeof(wr));
  wr.wr_id = WR_ID_STATUS;
  wr.sg_list = &sge;
  wr.num_sge = 1;
  wr.opcode = IBV_WR_RDMA_WRITE;
  wr.send_flags = IBV_SEND_SIGNALED;
  wr.wr.rdma.remote_addr = w->peer.status_addr;
  wr.wr.rdma.rkey = w->peer.status_rkey;
  ibv_send_wr* bad = nullptr;
  if (ibv_post_send(w->qp, &wr, &bad) != 0) return wire_dead(w, err, "ibv_post_send (status) failed");
  w->pending_status++;
  w->posted_status++;
  return reap(w, err);
}

void grdma_verbs_counts(const grdma_verbs_wire* w, uint64_t out[3]) {
  out[0] = w ? w->posted_data : 0;
  out[1] = w ? w->posted_status : 0;
  out[2] = w ? w->reaped : 0;
}

void grdma_verbs_close(grdma_verbs_wire* w) {
  if (!w) return;
  if (w->qp) ibv_destroy_qp(w->qp);
  if (w->cq) ibv_destroy_cq(w->cq);
  if (w->mr_ring) ibv_dereg_mr(w->mr_ring);
  if (w->mr_status_recv) ibv_dereg_mr(w->mr_status_recv);
  if (w->mr_staging) ibv_dereg_mr(w->mr_staging);
  if (w->mr_status_send) ibv_dereg_mr(w->mr_status_send);
  if (w->pd) ibv_dealloc_pd(w->pd);
  if (w->ctx) ibv_close_device(w->ctx);
  delete w;
}

#else  // ---- built without -DGRDMA_WITH_VERBS (no <infiniband/verbs.h> on this build host) ------------------------------------------------------------

struct grdma_verbs_wire {};
bool grdma_verbs_available() { return false; }
grdma_verbs_wire* grdma_verbs_open(const char*, int, int, void*, size_t, int, void*, size_t, void*, void*, size_t, std::string* err) {
  if (err) *err = "built without GRDMA_WITH_VERBS (no <infiniband/verbs.h> on the build host): no NIC wire in this library";
  return nullptr;
}
int grdma_verbs_address_of(const grdma_verbs_wire*, grdma_verbs_address*) { return -GRDMA_VERBS_ERR_UNSUPPORTED; }
int grdma_verbs_connect(grdma_verbs_wire*, const grdma_verbs_address*, std::string*) { return -GRDMA_VERBS_ERR_UNSUPPORTED; }
int grdma_verbs_post_data(grdma_verbs_wire*, const uint64_t[2], const uint64_t[2], uint64_t, std::string*) { return -GRDMA_VERBS_ERR_UNSUPPORTED; }
int grdma_verbs_post_status(grdma_verbs_wire*, std::string*) { return -GRDMA_VERBS_ERR_UNSUPPORTED; }
void grdma_verbs_counts(const grdma_verbs_wire*, uint64_t out[3]) { out[0] = out[1] = out[2] = 0; }
void grdma_verbs_close(grdma_verbs_wire*) {}

#endif
