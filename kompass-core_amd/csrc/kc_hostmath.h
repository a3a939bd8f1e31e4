// Host-side scalar math of the controller hot path: the float rigid transforms
// the reference builds with Eigen (utils/transformation.h:9-41) and the
// dynamic-window velocity lattice (trajectory_sampler.cpp:181-275, 328-372).
// Runs once per cycle on O(1)..O(N) data; the batch work is in the kernels.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>

#include "kompass_hip.h"

namespace kc {
namespace hm {

// 3-term float reduction in Eigen's fixed-size order: a + (b + c)
inline float add3(float a, float b, float c) { return a + (b + c); }

struct Quat {
  float w, x, y, z;
};

struct Rigid3f {  // Eigen::Isometry3f: linear part + translation
  float R[3][3];
  float t[3];

  static Rigid3f identity() {
    Rigid3f T{};
    T.R[0][0] = T.R[1][1] = T.R[2][2] = 1.0f;
    return T;
  }
  // Eigen QuaternionBase::toRotationMatrix
  static void rotation_of(const Quat &q, float R[3][3]) {
    const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0][0] = 1.0f - (tyy + tzz);
    R[0][1] = txy - twz;
    R[0][2] = txz + twy;
    R[1][0] = txy + twz;
    R[1][1] = 1.0f - (txx + tzz);
    R[1][2] = tyz - twx;
    R[2][0] = txz - twy;
    R[2][1] = tyz + twx;
    R[2][2] = 1.0f - (txx + tyy);
  }
  // Quaternionf(Matrix3f)
  static Quat quat_of(const float R[3][3]) {
    Quat q{};
    float t = add3(R[0][0], R[1][1], R[2][2]);
    if (t > 0.0f) {
      t = std::sqrt(t + 1.0f);
      q.w = 0.5f * t;
      t = 0.5f / t;
      q.x = (R[2][1] - R[1][2]) * t;
      q.y = (R[0][2] - R[2][0]) * t;
      q.z = (R[1][0] - R[0][1]) * t;
    } else {
      int i = 0;
      if (R[1][1] > R[0][0]) i = 1;
      if (R[2][2] > R[i][i]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      float v[3];
      t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0f);
      v[i] = 0.5f * t;
      t = 0.5f / t;
      q.w = (R[k][j] - R[j][k]) * t;
      v[j] = (R[j][i] + R[i][j]) * t;
      v[k] = (R[k][i] + R[i][k]) * t;
      q.x = v[0];
      q.y = v[1];
      q.z = v[2];
    }
    return q;
  }
  // getTransformation(Quaternionf, Vector3f), transformation.h:19-33
  static Rigid3f from_quat(const Quat &q, const float t[3]) {
    Rigid3f T;
    rotation_of(q, T.R);
    T.t[0] = t[0];
    T.t[1] = t[1];
    T.t[2] = t[2];
    return T;
  }
  // getTransformation(Matrix3f, Vector3f): goes through Quaternionf(matrix)
  static Rigid3f from_rotation(const float R[3][3], const float t[3]) {
    return from_quat(quat_of(R), t);
  }
  // eulerToRotationMatrix(0, 0, yaw) then getTransformation(rotation, (x,y,0))
  // -- collision_check.cpp:125-135 and transformation.h:35-41
  static Rigid3f from_pose2d(double x, double y, double yaw) {
    const float ha = 0.5f * static_cast<float>(yaw);
    Quat qz{std::cos(ha), 0.0f, 0.0f, std::sin(ha)};
    float R[3][3];
    rotation_of(qz, R);
    const float t[3] = {static_cast<float>(x), static_cast<float>(y), 0.0f};
    return from_rotation(R, t);
  }
  // Isometry3f * Isometry3f
  Rigid3f operator*(const Rigid3f &B) const {
    Rigid3f C;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j)
        C.R[i][j] =
            add3(R[i][0] * B.R[0][j], R[i][1] * B.R[1][j], R[i][2] * B.R[2][j]);
      C.t[i] =
          add3(R[i][0] * B.t[0], R[i][1] * B.t[1], R[i][2] * B.t[2]) + t[i];
    }
    return C;
  }
  // Isometry3f * Vector3f
  void apply(float px, float py, float pz, float out[3]) const {
    for (int i = 0; i < 3; ++i)
      out[i] = t[i] + add3(R[i][0] * px, R[i][1] * py, R[i][2] * pz);
  }
  bool planar() const {
    const float eps = 1e-6f;
    return std::fabs(R[0][2]) < eps && std::fabs(R[1][2]) < eps &&
           std::fabs(R[2][0]) < eps && std::fabs(R[2][1]) < eps &&
           R[2][2] > 0.0f;
  }
};

// trajectory.h:19-29
inline void linear_sample_split(int ctr_type, int max_lin, int &vx_n,
                                int &vy_n) {
  auto odd = [](int n) { return (n % 2 == 0) ? n + 1 : n; };
  if (ctr_type == KC_OMNI) {
    vx_n = odd(std::max(3, max_lin * 3 / 4));
    vy_n = odd(std::max(3, max_lin * 1 / 4));
  } else {
    vx_n = odd(std::max(3, max_lin));
    vy_n = 1;
  }
}

constexpr double kMinVel = 0.01;  // utils/trajectory_sampler.h:13-15

// The sample list as the device sees it: the DISTINCT values of each axis as small tables and, per
// sample, an index into each (the reference's lattices are products of a few axis values; an explicit
// list is de-duplicated the same way).  A controller draws a new window every cycle: the values
// change, the index pattern rarely does -- `signature` (window lattices only, != 0) says when it
// cannot have changed, so that neither the host nor the device copy of the indices is rebuilt.
struct VelocityLattice {
  std::vector<double> vx_values, vy_values, omega_values;
  std::vector<uint16_t> ix, iy;  // per sample: index into vx_values / vy_values
  std::vector<int32_t> row;      // per sample: index into omega_values (= row of the host's trig table)
  uint64_t signature = 0;

  void clear() {
    vx_values.clear();
    vy_values.clear();
    omega_values.clear();
    ix.clear();
    iy.clear();
    row.clear();
    signature = 0;
  }
  size_t size() const { return row.size(); }
  double vx(size_t i) const { return vx_values[ix[i]]; }
  double vy(size_t i) const { return vy_values[iy[i]]; }
  double omega(size_t i) const { return omega_values[static_cast<size_t>(row[i])]; }
  void push(uint16_t a, uint16_t b, int32_t r) {
    ix.push_back(a);
    iy.push_back(b);
    row.push_back(r);
  }
};

inline uint64_t lattice_mix(uint64_t h, uint64_t v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h * 0xff51afd7ed558ccdull;
}

// UpdateReachableVelocityRange + lattice loops; the (vx, omega) lattice shares
// one omega axis, so trig rows are assigned here without any de-duplication.
// `out` may hold the previous window: when the index pattern of this one is the
// same (same signature) only the value tables are rewritten.
inline void build_window_lattice(int ctr_type, const kc_limits &L, double cvx,
                                 double cvy, double com, double dt,
                                 int max_lin, int max_ang,
                                 VelocityLattice &out) {
  int lin_x, lin_y;
  linear_sample_split(ctr_type, max_lin, lin_x, lin_y);
  const int ang_n = max_ang + 1 - (max_ang % 2);
  double vy_max = L.vy_max, vy_acc = L.vy_acc, vy_dec = L.vy_dec;
  if (ctr_type != KC_OMNI) vy_max = vy_acc = vy_dec = 0.0;

  const double max_vx = std::min(L.vx_max, cvx + L.vx_acc * dt);
  const double min_vx = std::max(-L.vx_max, cvx - L.vx_dec * dt);
  double max_vy = 0.0, min_vy = 0.0;
  if (ctr_type == KC_OMNI) {
    max_vy = std::min(vy_max, cvy + vy_acc * dt);
    min_vy = std::max(-vy_max, cvy - vy_dec * dt);
  }
  const double res_x = std::max((max_vx - min_vx) / (lin_x - 1), 0.001);
  const double res_y =
      (lin_y > 1) ? std::max((max_vy - min_vy) / (lin_y - 1), 0.001) : 0.001;
  const double max_om = std::min(L.omega_max, com + L.omega_acc * dt);
  const double min_om = std::max(-L.omega_max, com - L.omega_dec * dt);
  const double res_om = std::max((max_om - min_om) / (ang_n - 1), 0.001);

  // the axes, by the reference's repeated addition (trajectory_sampler.cpp:207-217, 256-272)
  // (per-thread scratch: this runs once per controller cycle, and three growing vectors were 25 allocations --
  // more than half of the call)
  thread_local std::vector<double> xs, ys, oms;
  xs.clear();
  ys.clear();
  oms.clear();
  for (double v = min_vx; v <= max_vx; v += res_x) xs.push_back(v);
  if (ctr_type == KC_OMNI)
    for (double w = min_vy; w <= max_vy; w += res_y) ys.push_back(w);
  for (double o = min_om; o <= max_om; o += res_om) oms.push_back(o);
  const size_t n_om = oms.size();
  auto small = [](double a) { return std::fabs(a) < kMinVel; };
  // which samples exist depends only on which axis values are "zero" (|v| < kMinVel): the x rows without
  // an omega block, and the (v, w, 0) samples dropped as all-zero (trajectory_sampler.cpp:122-125)
  uint64_t sig = lattice_mix(0x6b6f6d70617373ull, static_cast<uint64_t>(ctr_type == KC_OMNI));
  sig = lattice_mix(sig, xs.size());
  sig = lattice_mix(sig, ys.size());
  sig = lattice_mix(sig, n_om);
  for (size_t i = 0; i < xs.size(); ++i)
    if (small(xs[i])) sig = lattice_mix(sig, 0x100000ull + i);
  for (size_t j = 0; j < ys.size(); ++j)
    if (small(ys[j])) sig = lattice_mix(sig, 0x200000ull + j);
  if (sig == 0) sig = 1;
  const bool same = out.signature == sig && out.vx_values.size() == xs.size() + 0 &&
                    out.omega_values.size() == n_om + (ctr_type == KC_OMNI ? 1 : 0);
  // value tables: vx_values = the x axis; vy_values = [0.0, the y axis]; omega_values = the omega axis
  // [+ 0.0 as the row of the (vx, vy, 0) omni samples]
  if (!same) {
    out.signature = 0;
    if (xs.size() > 65535 || ys.size() + 1 > 65535) {  // (limits far above any sample budget)
      out.clear();
      return;
    }
  }
  out.vx_values.assign(xs.begin(), xs.end());
  out.vy_values.assign(1, 0.0);
  out.vy_values.insert(out.vy_values.end(), ys.begin(), ys.end());
  out.omega_values.assign(oms.begin(), oms.end());
  if (ctr_type == KC_OMNI) out.omega_values.push_back(0.0);
  if (same) return;
  out.signature = sig;
  const int32_t zero_row = static_cast<int32_t>(n_om);  // omni only
  // (sized once, filled by plain loops: this runs whenever an axis value crosses |v| = kMinVel or an axis
  // gains / loses a value -- most cycles of a robot whose velocity moves)
  size_t total = 0;
  for (size_t i = 0; i < xs.size(); ++i) {
    if (ctr_type == KC_OMNI)
      for (size_t j = 0; j < ys.size(); ++j) total += !(small(xs[i]) && small(ys[j]));
    if (!small(xs[i])) total += n_om;
  }
  out.ix.resize(total);
  out.iy.resize(total);
  out.row.resize(total);
  uint16_t *pix = out.ix.data(), *piy = out.iy.data();
  int32_t *prow = out.row.data();
  size_t at = 0, first_block = static_cast<size_t>(-1);
  for (size_t i = 0; i < xs.size(); ++i) {
    if (ctr_type == KC_OMNI)
      for (size_t j = 0; j < ys.size(); ++j)
        if (!(small(xs[i]) && small(ys[j]))) {  // (omega = 0 is small)
          pix[at] = static_cast<uint16_t>(i);
          piy[at] = static_cast<uint16_t>(j + 1);
          prow[at] = zero_row;
          ++at;
        }
    if (!small(xs[i])) {  // (v, 0, every omega) with |v| >= kMinVel: no sample of it is all-zero
      std::fill_n(pix + at, n_om, static_cast<uint16_t>(i));
      std::fill_n(piy + at, n_om, static_cast<uint16_t>(0));
      if (first_block == static_cast<size_t>(-1)) {
        for (size_t r = 0; r < n_om; ++r) prow[at + r] = static_cast<int32_t>(r);
        first_block = at;
      } else {
        std::memcpy(prow + at, prow + first_block, n_om * sizeof(int32_t));
      }
      at += n_om;
    }
  }
}

}  // namespace hm
}  // namespace kc
