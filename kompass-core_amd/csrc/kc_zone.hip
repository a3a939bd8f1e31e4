// CriticalZoneChecker on gfx950.
//
// Reference: utils/critical_zone_check.cpp (CPU semantics; the SYCL variant
// critical_zone_check_gpu.cpp computes in float with its own math functions and
// is not what this reproduces).  The preset (trig of the scan angles, the
// forward / backward index sets) is host work with the host libm, once per
// checker.  A check is one small kernel: one lane per preset index, the
// reference's expression per lane, and a minimum over the lanes -- the loop of
// the reference returns 0 at the first critical beam and otherwise the smallest
// slow-down factor, which is the minimum of the per-beam values (0 for a
// critical beam), independent of the order.  Factors are non-negative floats,
// so the minimum is an unsigned atomic min on their bits.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "kc_hostmath.h"
#include "kc_internal.h"
#include "kompass_hip.h"

namespace kc {

struct ZoneArgs {
  const double *ranges;
  const float *cos_a, *sin_a;
  const int *idx;
  int n_idx;
  float R[3][3], t[3];  // sensor_tf_body_
  double robot_radius;
  float critical_distance, slowdown_distance;
  unsigned int *factor_bits;  // armed with bits(1.0f)
};

__global__ __launch_bounds__(256) void zone_check_kernel(ZoneArgs a) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.n_idx) return;
  const int i = a.idx[k];
  // critical_zone_check.cpp:98-113
  const float x = static_cast<float>(a.ranges[i] * static_cast<double>(a.cos_a[i]));
  const float y = static_cast<float>(a.ranges[i] * static_cast<double>(a.sin_a[i]));
  const float z = 0.0f;
  // Isometry3f * Vector3f: t + (R0*x + (R1*y + R2*z)) (kc_hostmath.h: Rigid3f::apply)
  const float px = a.t[0] + (a.R[0][0] * x + (a.R[0][1] * y + a.R[0][2] * z));
  const float py = a.t[1] + (a.R[1][0] * x + (a.R[1][1] * y + a.R[1][2] * z));
  // std::sqrt(std::pow(float, 2) + std::pow(float, 2)): double, stored as float
  const double dx = static_cast<double>(px), dy = static_cast<double>(py);
  const float converted = static_cast<float>(kc::dsqrt_rn(dy * dy + dx * dx));
  const float distance = static_cast<float>(static_cast<double>(converted) - a.robot_radius);
  float f = 1.0f;
  if (distance <= a.critical_distance) {
    f = 0.0f;
  } else if (distance <= a.slowdown_distance) {
    f = kc::div_rn(distance - a.critical_distance, a.slowdown_distance - a.critical_distance);
    if (!(f < 1.0f)) f = 1.0f;  // std::min(1.0f, f): f replaces 1 only when smaller (NaN never)
  }
  if (f < 1.0f) atomicMin(a.factor_bits, __float_as_uint(f < 0.0f ? 0.0f : f));
}

}  // namespace kc

using namespace kc;

struct kc_zone {
  int device = 0;
  hipStream_t stream = nullptr;
  size_t n = 0;
  double robot_radius = 0.0;
  float min_height = 0.f, max_height = 0.f, range_max = 0.f;
  float critical_distance = 0.f, slowdown_distance = 0.f;
  hm::Rigid3f tf{};
  std::vector<int> fwd, bwd;
  DevBuf<float> d_cos, d_sin;
  DevBuf<int> d_fwd, d_bwd;
  DevBuf<double> d_ranges;
  DevBuf<unsigned int> d_factor;
  PinBuf<double> h_ranges;
  PinBuf<unsigned int> h_factor;
  kc_cloud *cloud = nullptr;
  std::vector<double> cloud_ranges;
};

extern "C" {

int kc_zone_create(int shape, const float *dims, int ndims, const float sensor_pos[3],
                   const float sensor_rot_xyzw[4], float critical_angle,
                   float critical_distance, float slowdown_distance, const double *angles,
                   size_t n, float min_height, float max_height, float range_max, int device,
                   kc_zone **out) {
  if (!out || !dims || !sensor_pos || !sensor_rot_xyzw || (n && !angles))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (slowdown_distance <= critical_distance)  // critical_zone_check.cpp:52-56
    KC_FAIL(KC_ERR_INVALID, "SlowDown distance must be greater than the Critical distance!");
  double radius;
  if (shape == KC_CYLINDER && ndims >= 2) {
    radius = dims[0];
  } else if (shape == KC_BOX && ndims >= 3) {
    radius = std::sqrt(std::pow(static_cast<double>(dims[0]), 2) +
                       std::pow(static_cast<double>(dims[1]), 2)) / 2;
  } else if (shape == KC_SPHERE && ndims >= 1) {
    radius = dims[0];
  } else {
    KC_FAIL(KC_ERR_INVALID, "Invalid robot geometry type");
  }
  int ndev = 0;
  KC_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    KC_FAIL(KC_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
  auto *z = new kc_zone();
  z->device = device;
  z->n = n;
  z->robot_radius = radius;
  z->min_height = min_height;
  z->max_height = max_height;
  z->range_max = range_max;
  z->critical_distance = critical_distance;
  z->slowdown_distance = slowdown_distance;
  auto fail = [&](int rc) {
    kc_zone_destroy(z);
    return rc;
  };
  // Eigen::Quaternionf(Vector4f): coefficients in (x, y, z, w) order
  const hm::Quat q{sensor_rot_xyzw[3], sensor_rot_xyzw[0], sensor_rot_xyzw[1], sensor_rot_xyzw[2]};
  z->tf = hm::Rigid3f::from_quat(q, sensor_pos);
  // :46-48
  const float angle_rad = static_cast<float>(static_cast<double>(critical_angle) * M_PI / 180.0);
  double half = std::fmod(static_cast<double>(angle_rad / 2) + M_PI, 2 * M_PI);  // angles.h:21-29
  if (half < 0) half += 2 * M_PI;
  half -= M_PI;
  const float crit = static_cast<float>(half);
  std::vector<float> cs(n), sn(n);
  for (size_t i = 0; i < n; ++i) {  // preset, :60-83
    cs[i] = static_cast<float>(std::cos(angles[i]));
    sn[i] = static_cast<float>(std::sin(angles[i]));
    float p[3];
    z->tf.apply(cs[i], sn[i], 0.0f, p);
    const float abs_theta = std::fabs(::atan2f(p[1], p[0]));
    if (abs_theta <= crit) z->fwd.push_back(static_cast<int>(i));
    if (static_cast<double>(abs_theta) >= M_PI - static_cast<double>(crit))
      z->bwd.push_back(static_cast<int>(i));
  }
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&z->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("HIP stream creation failed on device %d", device);
    return fail(KC_ERR_HIP);
  }
  const size_t m = std::max<size_t>(n, 1);
  int rc;
  if ((rc = z->d_cos.reserve(m)) || (rc = z->d_sin.reserve(m)) || (rc = z->d_fwd.reserve(m)) ||
      (rc = z->d_bwd.reserve(m)) || (rc = z->d_ranges.reserve(m)) || (rc = z->h_ranges.reserve(m)) ||
      (rc = z->d_factor.reserve(1)) || (rc = z->h_factor.reserve(1)))
    return fail(rc);
  if (n) {
    if (hipMemcpy(z->d_cos.p, cs.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(z->d_sin.p, sn.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        (!z->fwd.empty() && hipMemcpy(z->d_fwd.p, z->fwd.data(), z->fwd.size() * sizeof(int),
                                      hipMemcpyHostToDevice) != hipSuccess) ||
        (!z->bwd.empty() && hipMemcpy(z->d_bwd.p, z->bwd.data(), z->bwd.size() * sizeof(int),
                                      hipMemcpyHostToDevice) != hipSuccess)) {
      set_error("preset upload failed");
      return fail(KC_ERR_HIP);
    }
  }
  *out = z;
  return KC_OK;
}

void kc_zone_destroy(kc_zone *z) {
  if (!z) return;
  hipError_t e = hipSetDevice(z->device);
  if (z->stream) {
    e = hipStreamSynchronize(z->stream);
    e = hipStreamDestroy(z->stream);
  }
  (void)e;
  if (z->cloud) kc_cloud_destroy(z->cloud);
  z->d_cos.release();
  z->d_sin.release();
  z->d_fwd.release();
  z->d_bwd.release();
  z->d_ranges.release();
  z->d_factor.release();
  z->h_ranges.release();
  z->h_factor.release();
  delete z;
}

int kc_zone_check(kc_zone *z, const double *ranges, size_t n, int forward, float *factor_out) {
  if (!z || !factor_out || (n && !ranges)) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (n < z->n)
    KC_FAIL(KC_ERR_RANGE, "%zu ranges for a checker preset with %zu angles", n, z->n);
  const std::vector<int> &idx = forward ? z->fwd : z->bwd;
  *factor_out = 1.0f;
  if (idx.empty()) return KC_OK;
  KC_HIP(hipSetDevice(z->device));
  hipStream_t s = z->stream;
  std::memcpy(z->h_ranges.p, ranges, z->n * sizeof(double));
  const float one = 1.0f;
  std::memcpy(z->h_factor.p, &one, sizeof(float));
  KC_HIP(hipMemcpyAsync(z->d_ranges.p, z->h_ranges.p, z->n * sizeof(double), hipMemcpyHostToDevice, s));
  KC_HIP(hipMemcpyAsync(z->d_factor.p, z->h_factor.p, sizeof(unsigned int), hipMemcpyHostToDevice, s));
  ZoneArgs a{};
  a.ranges = z->d_ranges.p;
  a.cos_a = z->d_cos.p;
  a.sin_a = z->d_sin.p;
  a.idx = forward ? z->d_fwd.p : z->d_bwd.p;
  a.n_idx = static_cast<int>(idx.size());
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) a.R[r][c] = z->tf.R[r][c];
    a.t[r] = z->tf.t[r];
  }
  a.robot_radius = z->robot_radius;
  a.critical_distance = z->critical_distance;
  a.slowdown_distance = z->slowdown_distance;
  a.factor_bits = z->d_factor.p;
  hipLaunchKernelGGL(zone_check_kernel, dim3((a.n_idx + 255) / 256), dim3(256), 0, s, a);
  KC_HIP(hipGetLastError());
  KC_HIP(hipMemcpyAsync(z->h_factor.p, z->d_factor.p, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  KC_HIP(hipStreamSynchronize(s));
  std::memcpy(factor_out, z->h_factor.p, sizeof(float));
  return KC_OK;
}

int kc_zone_check_cloud(kc_zone *z, const int8_t *data, size_t nbytes, int point_step,
                        int row_step, int height, int width, int x_offset, int y_offset,
                        int z_offset, int forward, float *factor_out) {
  return kc_zone_check_cloud_typed(z, data, nbytes, point_step, row_step, height, width, x_offset, y_offset, z_offset,
                                   KC_FIELD_FLOAT32, forward, factor_out);
}

int kc_zone_check_cloud_typed(kc_zone *z, const int8_t *data, size_t nbytes, int point_step,
                              int row_step, int height, int width, int x_offset, int y_offset,
                              int z_offset, int field_type, int forward, float *factor_out) {
  if (!z || !factor_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *factor_out = 1.0f;
  if (z->n == 0) return KC_OK;
  if (!z->cloud) KC_TRY(kc_cloud_create(std::max<size_t>(nbytes, 1 << 16), z->n, z->device, &z->cloud));
  z->cloud_ranges.resize(z->n);
  size_t bins = 0;
  // critical_zone_check.cpp:124-129: num_bins overload over the preset angle count
  KC_TRY(kc_cloud_to_laserscan_typed(z->cloud, data, nbytes, 0, point_step, row_step, height, width,
                                     x_offset, y_offset, z_offset, field_type, static_cast<double>(z->range_max),
                                     static_cast<double>(z->min_height), static_cast<double>(z->max_height),
                                     0.0, static_cast<int>(z->n), z->cloud_ranges.data(), nullptr, z->n,
                                     &bins));
  return kc_zone_check(z, z->cloud_ranges.data(), z->n, forward, factor_out);
}

int kc_zone_indices(kc_zone *z, int forward, int64_t *out, size_t cap, size_t *count) {
  if (!z || !count) KC_FAIL(KC_ERR_INVALID, "null argument");
  const std::vector<int> &idx = forward ? z->fwd : z->bwd;
  *count = idx.size();
  for (size_t i = 0; i < idx.size() && i < cap && out; ++i) out[i] = idx[i];
  return KC_OK;
}

}  // extern "C"
