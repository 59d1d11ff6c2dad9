// src/core/lib/ibverbs/config.h over libgrdma_amd.so: Config::Get() and its getters (config.h / config.cc:45-115),
// filled by grdma_config_from_env -- the same GRPC_RDMA_* variables with the same defaults.
#ifndef GRPC_SRC_CORE_LIB_IBVERBS_CONFIG_H
#define GRPC_SRC_CORE_LIB_IBVERBS_CONFIG_H
#ifdef GRPC_USE_IBVERBS
#include <string>

#include "grdma_amd.h"

namespace grpc_core {
namespace ibverbs {

class Config {
  Config() { grdma_config_from_env(&cfg_); }

 public:
  static Config& Get() {
    static Config inst;
    return inst;
  }
  std::string get_device_name() const { return cfg_.device_name; }
  int get_port_num() const { return cfg_.port_num; }
  int get_gid_index() const { return cfg_.gid_index; }
  int get_poller_thread_num() const { return cfg_.poller_thread_num; }
  int get_busy_polling_timeout_us() const { return cfg_.busy_polling_timeout_us; }
  int get_poller_sleep_timeout_ms() const { return cfg_.poller_sleep_timeout_ms; }
  uint32_t get_ring_buffer_size_kb() const { return cfg_.ring_buffer_size_kb; }
  uint32_t get_zerocopy_buffer_size_kb() const { return cfg_.zerocopy_buffer_size_kb; }
  uint32_t get_zerocopy_threshold_kb() const { return cfg_.zerocopy_threshold_kb; }
  int get_max_sge() const { return cfg_.max_sge; }
  int get_hip_device() const { return cfg_.hip_device; }

 private:
  grdma_config cfg_;
};

}  // namespace ibverbs
}  // namespace grpc_core
#endif
#endif  // GRPC_SRC_CORE_LIB_IBVERBS_CONFIG_H
