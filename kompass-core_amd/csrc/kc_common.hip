// Error string, device probing and key helpers of libkompass_hip.so.
#include "kc_internal.h"

namespace kc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace kc

extern "C" {

const char *kc_last_error(void) { return kc::g_err; }

int kc_abi_version(void) { return KC_ABI_VERSION; }

int kc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

float kc_key_cost(int64_t key) {
  return kc::sortable_float(static_cast<int32_t>(key >> 32));
}
int64_t kc_key_index(int64_t key) {
  return static_cast<int64_t>(static_cast<uint32_t>(key & 0xFFFFFFFFll));
}
int64_t kc_key_pack(float cost, int64_t index) {
  if (!(cost < 3.402823466e+38f)) return kc::KEY_NONE;
  return kc::key_pack(cost, static_cast<uint32_t>(index));
}

}  // extern "C"
