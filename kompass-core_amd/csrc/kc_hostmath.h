// Host-side scalar math of the controller hot path: the float rigid transforms
// the reference builds with Eigen (utils/transformation.h:9-41) and the
// dynamic-window velocity lattice (trajectory_sampler.cpp:181-275, 328-372).
// Runs once per cycle on O(1)..O(N) data; the batch work is in the kernels.
#pragma once

#include <cmath>
#include <cstddef>
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

struct VelocityLattice {
  std::vector<double> vx, vy;
  std::vector<int32_t> row;      // index into omega_values
  std::vector<double> omega_values;

  void clear() {
    vx.clear();
    vy.clear();
    row.clear();
    omega_values.clear();
  }
  size_t size() const { return vx.size(); }
  void push(double a, double b, int32_t r) {
    vx.push_back(a);
    vy.push_back(b);
    row.push_back(r);
  }
};

// UpdateReachableVelocityRange + lattice loops; the (vx, omega) lattice shares
// one omega axis, so trig rows are assigned here without any de-duplication.
inline void build_window_lattice(int ctr_type, const kc_limits &L, double cvx,
                                 double cvy, double com, double dt,
                                 int max_lin, int max_ang,
                                 VelocityLattice &out) {
  out.clear();
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

  // the omega axis is identical for every vx row: enumerate it once
  for (double o = min_om; o <= max_om; o += res_om)
    out.omega_values.push_back(o);
  const int32_t n_om = static_cast<int32_t>(out.omega_values.size());
  int32_t zero_row = -1;  // row for the (vx, vy, 0) omni samples
  auto zero_omega_row = [&]() {
    if (zero_row < 0) {
      zero_row = static_cast<int32_t>(out.omega_values.size());
      out.omega_values.push_back(0.0);
    }
    return zero_row;
  };
  auto all_zero = [](double a, double b, double c) {
    return std::fabs(a) < kMinVel && std::fabs(b) < kMinVel &&
           std::fabs(c) < kMinVel;
  };

  // (v, 0, every omega) with |v| >= kMinVel: no sample of such a row is all-zero, so the row is three fills
  // (this runs every control cycle: 8 k samples took 25 us one push_back at a time)
  auto push_omega_row = [&](double v) {
    const size_t at = out.vx.size(), to = at + static_cast<size_t>(n_om);
    out.vx.resize(to, v);
    out.vy.resize(to, 0.0);
    out.row.resize(to);
    for (int32_t r = 0; r < n_om; ++r) out.row[at + static_cast<size_t>(r)] = r;
  };
  if (ctr_type == KC_OMNI) {
    for (double v = min_vx; v <= max_vx; v += res_x) {
      for (double w = min_vy; w <= max_vy; w += res_y)
        if (!all_zero(v, w, 0.0)) out.push(v, w, zero_omega_row());
      if (std::fabs(v) >= kMinVel) push_omega_row(v);
    }
  } else {
    for (double v = min_vx; v <= max_vx; v += res_x)
      if (std::fabs(v) >= kMinVel) push_omega_row(v);
  }
}

}  // namespace hm
}  // namespace kc
