// TEST INFRASTRUCTURE ONLY -- part of oracle/, never linked into the product.
//
// The calls of oracle/fakeverbs/infiniband/verbs.h over process memory (see the header).  One device, one port.  Every
// call takes one process-wide lock: the reference's pair code is driven from one or two threads here (two while two
// pairs connect to each other), throughput does not matter.  Checks that a real HCA would make and the reference's code
// relies on are kept: a work request must name a registered region (lkey / rkey) that covers it, a queue pair must be
// ready to send, a SEND needs a posted receive; violations complete with an error status, as on hardware.
//
// -DFAKEVERBS_HIP (tests/cc/build_fakeverbs_hip.sh, the MI355X box): the registered regions are DEVICE memory -- an HBM
// ring, the staging buffer, the status words in the connection block -- and the fabric's "HCA" moves the bytes of a work
// request with the copy engine (hipMemcpyAsync on a stream of its own, never the null stream: a resident kernel of the
// library may be running): an RDMA WRITE lands in increasing address order, its last eight bytes -- the footer of the
// last record, ring_buffer.h:84-104 -- in a second copy behind everything else, and has completed (the stream is
// waited for) before its completion is queued, as on hardware.
#include <infiniband/verbs.h>

#ifdef FAKEVERBS_HIP
#include <hip/hip_runtime.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>

namespace {
struct recv_slot {
  uint64_t wr_id;
  uint64_t addr;
  uint32_t length;
  uint32_t lkey;
};
struct qp_state {
  ibv_qp* qp = nullptr;
  uint32_t dest_qp_num = 0;
  std::deque<recv_slot> recvs;
};
struct fabric {
  std::mutex mu;
  ibv_device dev;
  ibv_device* list[2] = {nullptr, nullptr};
  uint32_t next_key = 0x1000, next_qpn = 0x100, next_handle = 1;
  std::map<uint32_t, ibv_mr*> mrs;             // by lkey (== rkey)
  std::map<uint32_t, qp_state> qps;            // by qp_num
  std::map<ibv_cq*, std::deque<ibv_wc>> cqs;
  int fail_writes = 0;
  fabric() {
    memset(&dev, 0, sizeof(dev));
    snprintf(dev.name, sizeof(dev.name), "fakeverbs0");
    list[0] = &dev;
  }
};
fabric& F() {
  static fabric f;
  return f;
}
bool covers(uint32_t key, uint64_t addr, uint64_t len) {
  auto it = F().mrs.find(key);
  if (it == F().mrs.end()) return false;
  const uint64_t base = (uint64_t)it->second->addr;
  return addr >= base && addr + len <= base + it->second->length;
}
#ifdef FAKEVERBS_HIP
hipStream_t fabric_stream() {
  static hipStream_t s = nullptr;
  if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
  return s;
}
// one scatter-gather entry of a write: in address order, and -- `last` -- its final eight bytes behind the rest
bool fabric_copy(uint8_t* dst, const void* src, size_t n, bool last) {
  hipStream_t s = fabric_stream();
  if (!s) return false;
  const size_t tail = (last && n > 8) ? 8 : 0;
  if (n - tail && hipMemcpyAsync(dst, src, n - tail, hipMemcpyDefault, s) != hipSuccess) return false;
  if (tail && hipMemcpyAsync(dst + n - tail, static_cast<const uint8_t*>(src) + n - tail, tail, hipMemcpyDefault, s) != hipSuccess)
    return false;
  return true;
}
bool fabric_flush() { return hipStreamSynchronize(fabric_stream()) == hipSuccess; }
#else
bool fabric_copy(uint8_t* dst, const void* src, size_t n, bool) {
  memcpy(dst, src, n);
  return true;
}
bool fabric_flush() { return true; }
#endif
void complete(ibv_cq* cq, uint64_t wr_id, ibv_wc_status st, ibv_wc_opcode op, uint32_t bytes, uint32_t imm, uint32_t qpn) {
  ibv_wc wc;
  memset(&wc, 0, sizeof(wc));
  wc.wr_id = wr_id;
  wc.status = st;
  wc.opcode = op;
  wc.byte_len = bytes;
  wc.imm_data = imm;
  wc.qp_num = qpn;
  F().cqs[cq].push_back(wc);
}
}  // namespace

extern "C" {

ibv_device** ibv_get_device_list(int* n) {
  if (n) *n = 1;
  return F().list;
}
void ibv_free_device_list(ibv_device**) {}
ibv_context* ibv_open_device(ibv_device* d) {
  ibv_context* c = new ibv_context;
  c->device = d;
  return c;
}
int ibv_close_device(ibv_context* c) {
  delete c;
  return 0;
}
int ibv_query_device(ibv_context*, ibv_device_attr* a) {
  memset(a, 0, sizeof(*a));
  const char* e = getenv("FAKEVERBS_MAX_SGE");
  a->max_sge = e ? atoi(e) : 30;
  a->max_qp_wr = 16384;
  a->max_cqe = 1 << 20;
  return 0;
}
int ibv_query_port(ibv_context*, uint8_t, ibv_port_attr* a) {
  memset(a, 0, sizeof(*a));
  a->state = IBV_PORT_ACTIVE;
  a->lid = 1;
  return 0;
}
int ibv_query_gid(ibv_context*, uint8_t, int, ibv_gid* gid) {
  memset(gid, 0, sizeof(*gid));
  gid->global.subnet_prefix = 0xfe80000000000000ull;
  gid->global.interface_id = 0x0123456789abcdefull;
  return 0;
}
ibv_pd* ibv_alloc_pd(ibv_context* c) {
  ibv_pd* pd = new ibv_pd;
  pd->context = c;
  return pd;
}
int ibv_dealloc_pd(ibv_pd* pd) {
  delete pd;
  return 0;
}
ibv_mr* ibv_reg_mr(ibv_pd* pd, void* addr, size_t length, int) {
  std::lock_guard<std::mutex> lk(F().mu);
  ibv_mr* mr = new ibv_mr;
  memset(mr, 0, sizeof(*mr));
  mr->context = pd ? pd->context : nullptr;
  mr->pd = pd;
  mr->addr = addr;
  mr->length = length;
  mr->handle = F().next_handle++;
  mr->lkey = mr->rkey = F().next_key++;
  F().mrs[mr->lkey] = mr;
  return mr;
}
ibv_mr* ibv_reg_dmabuf_mr(ibv_pd* pd, uint64_t offset, size_t length, uint64_t iova, int, int access) {
  return ibv_reg_mr(pd, reinterpret_cast<void*>(iova + offset), length, access);
}
int ibv_dereg_mr(ibv_mr* mr) {
  std::lock_guard<std::mutex> lk(F().mu);
  F().mrs.erase(mr->lkey);
  delete mr;
  return 0;
}
ibv_cq* ibv_create_cq(ibv_context* c, int cqe, void* cq_context, void*, int) {
  std::lock_guard<std::mutex> lk(F().mu);
  ibv_cq* cq = new ibv_cq;
  cq->context = c;
  cq->cq_context = cq_context;
  cq->cqe = cqe;
  cq->handle = F().next_handle++;
  F().cqs[cq];
  return cq;
}
int ibv_destroy_cq(ibv_cq* cq) {
  std::lock_guard<std::mutex> lk(F().mu);
  F().cqs.erase(cq);
  delete cq;
  return 0;
}
ibv_qp* ibv_create_qp(ibv_pd* pd, ibv_qp_init_attr* a) {
  std::lock_guard<std::mutex> lk(F().mu);
  ibv_qp* qp = new ibv_qp;
  memset(qp, 0, sizeof(*qp));
  qp->context = pd ? pd->context : nullptr;
  qp->qp_context = a->qp_context;
  qp->pd = pd;
  qp->send_cq = a->send_cq;
  qp->recv_cq = a->recv_cq;
  qp->handle = F().next_handle++;
  qp->qp_num = F().next_qpn++;
  qp->state = IBV_QPS_RESET;
  qp->qp_type = a->qp_type;
  F().qps[qp->qp_num].qp = qp;
  return qp;
}
int ibv_destroy_qp(ibv_qp* qp) {
  std::lock_guard<std::mutex> lk(F().mu);
  F().qps.erase(qp->qp_num);
  delete qp;
  return 0;
}
int ibv_modify_qp(ibv_qp* qp, ibv_qp_attr* a, int mask) {
  std::lock_guard<std::mutex> lk(F().mu);
  auto it = F().qps.find(qp->qp_num);
  if (it == F().qps.end()) return 22;
  if (mask & IBV_QP_STATE) {
    qp->state = a->qp_state;
    if (a->qp_state == IBV_QPS_RESET) it->second.recvs.clear();
  }
  if (mask & IBV_QP_DEST_QPN) it->second.dest_qp_num = a->dest_qp_num;
  return 0;
}
int ibv_query_qp(ibv_qp* qp, ibv_qp_attr* a, int, ibv_qp_init_attr* ia) {
  std::lock_guard<std::mutex> lk(F().mu);
  memset(a, 0, sizeof(*a));
  if (ia) memset(ia, 0, sizeof(*ia));
  a->qp_state = a->cur_qp_state = qp->state;
  return 0;
}
int ibv_post_recv(ibv_qp* qp, ibv_recv_wr* wr, ibv_recv_wr** bad) {
  std::lock_guard<std::mutex> lk(F().mu);
  auto it = F().qps.find(qp->qp_num);
  for (; wr; wr = wr->next) {
    if (it == F().qps.end() || wr->num_sge != 1) {
      if (bad) *bad = wr;
      return 22;
    }
    it->second.recvs.push_back({wr->wr_id, wr->sg_list[0].addr, wr->sg_list[0].length, wr->sg_list[0].lkey});
  }
  return 0;
}
int ibv_post_send(ibv_qp* qp, ibv_send_wr* wr, ibv_send_wr** bad) {
  std::lock_guard<std::mutex> lk(F().mu);
  auto me = F().qps.find(qp->qp_num);
  for (; wr; wr = wr->next) {
    if (me == F().qps.end() || qp->state != IBV_QPS_RTS) {
      if (bad) *bad = wr;
      return 22;
    }
    const bool signaled = (wr->send_flags & IBV_SEND_SIGNALED) != 0;
    uint64_t total = 0;
    bool local_ok = true;
    for (int i = 0; i < wr->num_sge; i++) {
      total += wr->sg_list[i].length;
      local_ok = local_ok && covers(wr->sg_list[i].lkey, wr->sg_list[i].addr, wr->sg_list[i].length);
    }
    auto peer = F().qps.find(me->second.dest_qp_num);
    if (wr->opcode == IBV_WR_RDMA_WRITE) {
      ibv_wc_status st = IBV_WC_SUCCESS;
      if (!local_ok) st = IBV_WC_LOC_PROT_ERR;
      else if (peer == F().qps.end() || !covers(wr->wr.rdma.rkey, wr->wr.rdma.remote_addr, total)) st = IBV_WC_REM_ACCESS_ERR;
      else if (F().fail_writes > 0) {   // (fakeverbs_fail_next_writes: what a peer that deregistered its ring looks like)
        F().fail_writes--;
        st = IBV_WC_REM_ACCESS_ERR;
        qp->state = IBV_QPS_ERR;
      }
      if (st == IBV_WC_SUCCESS) {
        uint8_t* dst = reinterpret_cast<uint8_t*>(wr->wr.rdma.remote_addr);
        bool ok = true;
        for (int i = 0; i < wr->num_sge; i++) {
          ok = ok && fabric_copy(dst, reinterpret_cast<const void*>(wr->sg_list[i].addr), wr->sg_list[i].length, i + 1 == wr->num_sge);
          dst += wr->sg_list[i].length;
        }
        if (!ok || !fabric_flush()) st = IBV_WC_GENERAL_ERR;
      }
      if (signaled || st != IBV_WC_SUCCESS) complete(qp->send_cq, wr->wr_id, st, IBV_WC_RDMA_WRITE, (uint32_t)total, 0, qp->qp_num);
    } else if (wr->opcode == IBV_WR_SEND_WITH_IMM || wr->opcode == IBV_WR_SEND) {
      ibv_wc_status st = IBV_WC_SUCCESS;
      if (!local_ok) st = IBV_WC_LOC_PROT_ERR;
      else if (peer == F().qps.end() || peer->second.recvs.empty()) st = IBV_WC_RNR_RETRY_EXC_ERR;
      if (st == IBV_WC_SUCCESS) {
        const recv_slot r = peer->second.recvs.front();
        if (total > r.length || !covers(r.lkey, r.addr, total)) {
          st = IBV_WC_REM_ACCESS_ERR;
        } else {
          peer->second.recvs.pop_front();
          uint8_t* dst = reinterpret_cast<uint8_t*>(r.addr);
          for (int i = 0; i < wr->num_sge; i++) {
            memcpy(dst, reinterpret_cast<const void*>(wr->sg_list[i].addr), wr->sg_list[i].length);
            dst += wr->sg_list[i].length;
          }
          complete(peer->second.qp->recv_cq, r.wr_id, IBV_WC_SUCCESS, IBV_WC_RECV, (uint32_t)total, wr->imm_data,
                   peer->second.qp->qp_num);
        }
      }
      if (signaled || st != IBV_WC_SUCCESS) complete(qp->send_cq, wr->wr_id, st, IBV_WC_SEND, (uint32_t)total, 0, qp->qp_num);
    } else {
      if (bad) *bad = wr;
      return 95;
    }
  }
  return 0;
}
// test hook: the next n RDMA WRITEs complete with a remote access error and leave their queue pair in the error state
void fakeverbs_fail_next_writes(int n) {
  std::lock_guard<std::mutex> lk(F().mu);
  F().fail_writes = n;
}
int ibv_poll_cq(ibv_cq* cq, int n, ibv_wc* wc) {
  std::lock_guard<std::mutex> lk(F().mu);
  auto it = F().cqs.find(cq);
  if (it == F().cqs.end()) return -1;
  int k = 0;
  while (k < n && !it->second.empty()) {
    wc[k++] = it->second.front();
    it->second.pop_front();
  }
  return k;
}

}  // extern "C"
