// RAII + exception bridge between the host C++ classes and the C ABI of
// libkompass_hip.so.  A failed call becomes the C++ exception type the
// reference would throw (std::invalid_argument / std::out_of_range /
// std::runtime_error); nothing here computes anything.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>

#include "kompass_hip.h"

namespace Kompass {
namespace hip {

inline void check(int rc) {
  if (rc == KC_OK) return;
  const std::string msg = kc_last_error();
  switch (rc) {
    case KC_ERR_INVALID: throw std::invalid_argument(msg);
    case KC_ERR_RANGE: throw std::out_of_range(msg);
    default: throw std::runtime_error(msg);
  }
}

struct DwaDeleter {
  void operator()(kc_dwa *p) const { kc_dwa_destroy(p); }
};
struct MapperDeleter {
  void operator()(kc_mapper *p) const { kc_mapper_destroy(p); }
};
struct CloudDeleter {
  void operator()(kc_cloud *p) const { kc_cloud_destroy(p); }
};
using CloudHandle = std::unique_ptr<kc_cloud, CloudDeleter>;
using DwaHandle = std::shared_ptr<kc_dwa>;
using MapperHandle = std::unique_ptr<kc_mapper, MapperDeleter>;

inline DwaHandle makeDwa(const kc_dwa_params &p) {
  kc_dwa *raw = nullptr;
  check(kc_dwa_create(&p, &raw));
  return DwaHandle(raw, DwaDeleter());
}

inline CloudHandle makeCloud(size_t max_bytes, size_t max_bins) {
  kc_cloud *raw = nullptr;
  check(kc_cloud_create(max_bytes, max_bins, 0, &raw));
  return CloudHandle(raw);
}

}  // namespace hip
}  // namespace Kompass
