// Host-only half of the C ABI: GRPC_PLATFORM_TYPE selection
// (src/core/lib/iomgr/iomgr_internal.cc:37-62), the GRPC_RDMA_* knobs
// (src/core/lib/ibverbs/config.cc:45-115) and scalar ring arithmetic.
#include "grdma_host.h"

#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../include/grdma_amd.h"

extern "C" {

uint64_t grdma_host_free_size(uint64_t cap, uint64_t head, uint64_t tail) {
  uint64_t occupied = (tail + cap - head) & (cap - 1);
  return cap - occupied;
}

uint64_t grdma_host_writable(uint64_t cap, uint64_t head, uint64_t tail) {
  uint64_t remaining = grdma_host_free_size(cap, head, tail);
  return remaining > 24 ? remaining - 24 : 0;
}

uint64_t grdma_host_encoded_size(uint64_t payload) { return 16 + ((payload + 7) & ~7ull); }

uint64_t grdma_host_calc_writable(uint64_t space) {
  return space > 24 ? ((space - 24) & ~7ull) : 0;
}

int grdma_parse_platform(const char* value) {
  // unset -> TCP (iomgr_internal.cc:41-43); exact, case-sensitive names (:45-56)
  if (value == nullptr) return GRDMA_IOMGR_TCP;
  if (strcmp(value, "TCP") == 0) return GRDMA_IOMGR_TCP;
  if (strcmp(value, "RDMA_BP") == 0) return GRDMA_IOMGR_RDMA_BP;
  if (strcmp(value, "RDMA_BPEV") == 0) return GRDMA_IOMGR_RDMA_BPEV;
  if (strcmp(value, "RDMA_EVENT") == 0) return GRDMA_IOMGR_RDMA_EVENT;
  return -GRDMA_ERR_CONFIG;  // the reference logs and exit(1)s here (:57-58)
}

int grdma_determine_platform(void) { return grdma_parse_platform(getenv("GRPC_PLATFORM_TYPE")); }

static bool env_int(const char* name, long long* out) {
  const char* s = getenv(name);
  if (s == nullptr) return false;
  *out = atoll(s);  // the reference uses atoi/atoll without validation
  return true;
}

int grdma_config_from_env(grdma_config* c) {
  if (c == nullptr) return -GRDMA_ERR_INVALID;
  memset(c, 0, sizeof(*c));
  long long v;
  const char* s = getenv("GRPC_RDMA_DEVICE_NAME");
  if (s != nullptr) strncpy(c->device_name, s, sizeof(c->device_name) - 1);
  c->port_num = env_int("GRPC_RDMA_PORT_NUM", &v) ? (int32_t)v : 1;
  c->gid_index = env_int("GRPC_RDMA_GID_INDEX", &v) ? (int32_t)v : 0;
  c->poller_thread_num = 1;
  if (env_int("GRPC_RDMA_POLLER_THREAD_NUM", &v)) {
    if (v <= 0) return -GRDMA_ERR_CONFIG;  // GPR_ASSERT(poller_thread_num_ > 0)
    c->poller_thread_num = (int32_t)v;
  }
  c->busy_polling_timeout_us = 500;
  if (env_int("GRPC_RDMA_BUSY_POLLING_TIMEOUT_US", &v)) {
    if (v < 0) return -GRDMA_ERR_CONFIG;
    c->busy_polling_timeout_us = (int32_t)v;
  }
  c->poller_sleep_timeout_ms = 1000;
  if (env_int("GRPC_RDMA_POLLER_SLEEP_TIMEOUT_MS", &v)) {
    if (v < 0) return -GRDMA_ERR_CONFIG;
    c->poller_sleep_timeout_ms = (int32_t)v;
  }
  c->ring_buffer_size_kb = 4 * 1024;
  c->zerocopy_buffer_size_kb = 32 * 1024;
  if (env_int("GRPC_RDMA_RING_BUFFER_SIZE_KB", &v)) {
    if (v <= 0) return -GRDMA_ERR_CONFIG;
    uint64_t bytes = (uint64_t)v * 1024;
    if (bytes & (bytes - 1)) return -GRDMA_ERR_CONFIG;  // asserted at ring_buffer.cc:22
    c->ring_buffer_size_kb = (uint32_t)v;
    // config.cc:100-106 reads the zero-copy buffer size from the SAME variable
    // (Appendix A.10); kept so both knobs behave as in the reference.
    c->zerocopy_buffer_size_kb = (uint32_t)v;
  }
  c->zerocopy_threshold_kb = std::numeric_limits<uint32_t>::max();
  if (env_int("GRPC_RDMA_ZEROCOPY_THRESHOLD_KB", &v)) c->zerocopy_threshold_kb = (uint32_t)v;
  c->max_sge = 30;  // mlx5 reports 30; the loop-back wire has no SGE limit of its own
  if (env_int("GRPC_RDMA_MAX_SGE", &v)) {
    if (v <= 0) return -GRDMA_ERR_CONFIG;
    c->max_sge = (int32_t)v;
  }
  c->hip_wire_direct = 1;
  if (const char* w = getenv("GRPC_RDMA_HIP_WIRE")) {
    if (!strcmp(w, "staged")) c->hip_wire_direct = 0;
    else if (!strcmp(w, "direct")) c->hip_wire_direct = 1;
    else return -GRDMA_ERR_CONFIG;
  }
  c->hip_register_min = env_int("GRPC_RDMA_HIP_REGISTER_MIN", &v) && v > 0 ? (uint32_t)v : 0;
  c->hip_pair_pool_mb = env_int("GRPC_RDMA_HIP_PAIR_POOL_MB", &v) && v >= 0 ? (uint32_t)v : 4096;
  c->hip_device = 0;
  if (env_int("GRPC_RDMA_HIP_DEVICE", &v)) c->hip_device = (int32_t)v;
  else if (env_int("LOCAL_RANK", &v)) c->hip_device = (int32_t)v;
  return 0;
}

}  // extern "C"
