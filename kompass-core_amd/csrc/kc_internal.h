// Internal helpers shared by the HIP translation units of libkompass_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kompass_hip.h"

namespace kc {

// ---- error plumbing: nothing throws across the C ABI ----------------------
void set_error(const char *fmt, ...);

#define KC_FAIL(code, ...)        \
  do {                            \
    ::kc::set_error(__VA_ARGS__); \
    return (code);                \
  } while (0)

#define KC_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) {                                                 \
      ::kc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                      __FILE__, __LINE__);                                  \
      return KC_ERR_HIP;                                                    \
    }                                                                       \
  } while (0)

#define KC_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != KC_OK) return _rc; \
  } while (0)

// ---- grow-only device buffer (reference grows its buffers the same way:
// cost_evaluator_gpu.cpp:248-271, 314-333) ----------------------------------
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return KC_OK;
    if (p) {
      hipError_t e = hipFree(p);
      (void)e;
      p = nullptr;
      cap = 0;
    }
    size_t want = n + n / 4 + 16;
    KC_HIP(hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)));
    cap = want;
    return KC_OK;
  }
  void release() {
    if (p) {
      hipError_t e = hipFree(p);
      (void)e;
    }
    p = nullptr;
    cap = 0;
  }
};

// pinned host staging buffer (async H2D / D2H without a hidden sync)
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return KC_OK;
    if (p) {
      hipError_t e = hipHostFree(p);
      (void)e;
      p = nullptr;
      cap = 0;
    }
    size_t want = n + n / 4 + 16;
    KC_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), want * sizeof(T),
                         hipHostMallocDefault));
    cap = want;
    return KC_OK;
  }
  void release() {
    if (p) {
      hipError_t e = hipHostFree(p);
      (void)e;
    }
    p = nullptr;
    cap = 0;
  }
};

// ---- per-kernel HIP-event timing on the launch stream ----------------------
struct Timing {
  bool enabled = false;
  struct Rec {
    const char *name;
    hipEvent_t a, b;
  };
  std::vector<Rec> pool;  // events are created once and reused
  size_t used = 0;
  // host-side phases of the same cycle (wall clock), reported as "host:<name>"
  struct HostRec {
    const char *name;
    double ms;
  };
  std::vector<HostRec> host;
  std::chrono::steady_clock::time_point t_mark;
  void begin_cycle() {
    used = 0;
    host.clear();
    if (enabled) t_mark = std::chrono::steady_clock::now();
  }
  // closes the host phase that started at the previous mark
  void mark(const char *name) {
    if (!enabled) return;
    const auto now = std::chrono::steady_clock::now();
    host.push_back({name, std::chrono::duration<double, std::milli>(now - t_mark).count()});
    t_mark = now;
  }
  int start(const char *name, hipStream_t s) {
    if (!enabled) return KC_OK;
    if (used == pool.size()) {
      Rec r{name, nullptr, nullptr};
      KC_HIP(hipEventCreate(&r.a));
      KC_HIP(hipEventCreate(&r.b));
      pool.push_back(r);
    }
    pool[used].name = name;
    KC_HIP(hipEventRecord(pool[used].a, s));
    return KC_OK;
  }
  int stop(hipStream_t s) {
    if (!enabled) return KC_OK;
    KC_HIP(hipEventRecord(pool[used].b, s));
    used++;
    return KC_OK;
  }
  int get(const char **names, float *ms, size_t cap, size_t *count) {
    size_t n = 0;
    for (size_t i = 0; i < used && n < cap; ++i) {
      KC_HIP(hipEventSynchronize(pool[i].b));
      float t = 0.f;
      KC_HIP(hipEventElapsedTime(&t, pool[i].a, pool[i].b));
      if (names) names[n] = pool[i].name;
      if (ms) ms[n] = t;
      n++;
    }
    for (size_t i = 0; i < host.size() && n < cap; ++i) {
      if (names) names[n] = host[i].name;
      if (ms) ms[n] = static_cast<float>(host[i].ms);
      n++;
    }
    if (count) *count = n;
    return KC_OK;
  }
  void release() {
    for (auto &r : pool) {
      hipError_t e = hipEventDestroy(r.a);
      e = hipEventDestroy(r.b);
      (void)e;
    }
    pool.clear();
    used = 0;
  }
};

// ---- correctly rounded f32 divide / sqrt on the device ----------------------
// HIP's __fsqrt_rn lowers to the *native* (approximate) sqrt unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined; the plain operators are IEEE
// correctly rounded under -fhip-fp32-correctly-rounded-divide-sqrt (on here).
__host__ __device__ inline float div_rn(float a, float b) { return a / b; }
__host__ __device__ inline float sqrt_rn(float a) { return __builtin_sqrtf(a); }
__host__ __device__ inline double dsqrt_rn(double a) { return __builtin_sqrt(a); }

// ---- packed (cost, index) key: signed-comparable int64 ---------------------
// LowestCost::combine (datatypes/trajectory.h:630-636): lower cost wins, ties
// go to the lower index.  key = (sortable_i32(cost) << 32) | u32 index; signed
// int64 comparison reproduces that order.  KEY_NONE = nothing beats FLT_MAX.
constexpr int64_t KEY_NONE = INT64_MAX;

__host__ __device__ inline int32_t float_sortable(float f) {
  f = f + 0.0f;  // -0.0 -> +0.0 so that both compare equal like floats do
  int32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = __float_as_int(f);
#else
  std::memcpy(&b, &f, 4);
#endif
  return b >= 0 ? b : (b ^ 0x7FFFFFFF);
}
__host__ __device__ inline float sortable_float(int32_t s) {
  int32_t b = s >= 0 ? s : (s ^ 0x7FFFFFFF);
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __int_as_float(b);
#else
  std::memcpy(&f, &b, 4);
#endif
  return f;
}
__host__ __device__ inline int64_t key_pack(float cost, uint32_t index) {
  const uint64_t hi = static_cast<uint64_t>(
      static_cast<uint32_t>(float_sortable(cost)));
  return static_cast<int64_t>((hi << 32) | static_cast<uint64_t>(index));
}

// ---- checksum of the pinned result record ----------------------------------
// The device writes the record as separate 8-byte stores without fences; the
// host accepts it when the sequence word matches and this word over (key,
// packed counts, sequence, row word) does.  Each input goes through a multiply
// / shift mix and a different rotation, so a record that mixes words of two
// cycles cannot pass the way it could with a plain XOR (equal differences
// cancelling).
__host__ __device__ inline unsigned long long rec_mix(unsigned long long x) {
  x *= 0x9E3779B97F4A7C15ull;
  return x ^ (x >> 29);
}
__host__ __device__ inline unsigned long long rec_rot(unsigned long long x, int r) {
  return (x << r) | (x >> (64 - r));
}
__host__ __device__ inline long long record_check(long long w0, long long w1, long long seq,
                                                  long long w4 = 0) {
  const unsigned long long c = rec_mix(static_cast<unsigned long long>(w0) + 0x5bd1e9955bd1e995ull) ^
                               rec_rot(rec_mix(static_cast<unsigned long long>(w1) ^ 0x2545F4914F6CDD1Dull), 21) ^
                               rec_rot(rec_mix(static_cast<unsigned long long>(seq)), 42) ^
                               rec_rot(rec_mix(static_cast<unsigned long long>(w4) + 0x9E37ull), 11);
  return static_cast<long long>(c);
}

// internal view of a mapper context for the grid hand-off (kc_dwa.hip)
struct MapperView {
  const int *grid;
  int H, W, c0, c1;
  float res;
  hipStream_t stream;
  int device;
};
int mapper_view(kc_mapper *m, MapperView *out);

// kc_comm.hip: all-reduce (min, or sum) of int64 words on a stream; send == recv is allowed
int comm_allreduce_i64(kc_comm *m, const long long *send, long long *recv, size_t count, bool sum,
                       hipStream_t stream);
int comm_world(const kc_comm *m);
int comm_rank(const kc_comm *m);
int comm_device(const kc_comm *m);

}  // namespace kc
