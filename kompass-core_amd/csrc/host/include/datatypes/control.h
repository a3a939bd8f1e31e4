// Control value types of the kompass_cpp surface (reference:
// datatypes/control.h:12, 112-140, 190-247).
#pragma once

#include <cmath>
#include <string>
#include <vector>

#include "kc_linalg.h"

namespace Kompass {
namespace Control {

enum class ControlType { ACKERMANN = 0, DIFFERENTIAL_DRIVE = 1, OMNI = 2 };

class Velocity2D {
 public:
  Velocity2D() = default;
  Velocity2D(double vx, double vy, double omega, double steer_ang = 0.0)
      : v_{vx, vy, omega, steer_ang} {}
  double vx() const { return v_[0]; }
  double vy() const { return v_[1]; }
  double omega() const { return v_[2]; }
  double steer_ang() const { return v_[3]; }
  void setVx(double x) { v_[0] = x; }
  void setVy(double x) { v_[1] = x; }
  void setOmega(double x) { v_[2] = x; }
  void setSteerAng(double x) { v_[3] = x; }
  Velocity2D operator-() const { return Velocity2D(-v_[0], -v_[1], -v_[2]); }

 private:
  double v_[4] = {0.0, 0.0, 0.0, 0.0};
};

struct LinearVelocityControlParams {
  double maxVel, maxAcceleration, maxDeceleration;
  LinearVelocityControlParams(double maxVel = 1.0, double maxAcc = 10.0,
                              double maxDec = 10.0)
      : maxVel(maxVel), maxAcceleration(maxAcc), maxDeceleration(maxDec) {}
};

struct AngularVelocityControlParams {
  double maxAngle, maxOmega, maxAcceleration, maxDeceleration;
  AngularVelocityControlParams(double maxAng = M_PI, double maxOmg = 1.0,
                               double maxAcc = 10.0, double maxDec = 10.0)
      : maxAngle(maxAng), maxOmega(maxOmg), maxAcceleration(maxAcc),
        maxDeceleration(maxDec) {}
};

struct ControlLimitsParams {
  LinearVelocityControlParams velXParams, velYParams;
  AngularVelocityControlParams omegaParams;
  ControlLimitsParams() = default;
  ControlLimitsParams(const LinearVelocityControlParams &x,
                      const LinearVelocityControlParams &y,
                      const AngularVelocityControlParams &w)
      : velXParams(x), velYParams(y), omegaParams(w) {}
};

struct LaserScan {
  std::vector<double> ranges, angles;
  LaserScan(std::vector<double> ranges, std::vector<double> angles)
      : ranges(std::move(ranges)), angles(std::move(angles)) {}
};

// A point cloud the caller already holds as packed (x, y, z) floats -- a numpy (N, 3) float32 array from
// the Python layer: consumed where it lies (the sensor update stores it straight to the device), no
// std::vector<Path::Point> in between.  Same meaning as the vector form (global_frame = true).
struct PointCloudView {
  const float *xyz;
  size_t n;
};

}  // namespace Control
}  // namespace Kompass
