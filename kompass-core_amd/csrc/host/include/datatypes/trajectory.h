// Trajectory containers of the kompass_cpp surface (reference:
// datatypes/trajectory.h).  Sample-major float matrices: one trajectory is one
// contiguous row -- the layout the device kernels read and write.
#pragma once

#include <cfloat>
#include <cmath>
#include <stdexcept>
#include <vector>

#include "datatypes/control.h"
#include "datatypes/path.h"
#include "kc_linalg.h"

namespace Kompass {
namespace Control {

constexpr float DEFAULT_MIN_DIST = FLT_MAX;

// split of the linear sample budget between vx and vy (trajectory.h:19-29)
inline void computeLinearSampleSplit(ControlType t, int maxLinearSamples,
                                     int &vx_n, int &vy_n) {
  auto odd = [](int n) { return n % 2 == 0 ? n + 1 : n; };
  if (t == ControlType::OMNI) {
    vx_n = odd(std::max(3, maxLinearSamples * 3 / 4));
    vy_n = odd(std::max(3, maxLinearSamples * 1 / 4));
  } else {
    vx_n = odd(std::max(3, maxLinearSamples));
    vy_n = 1;
  }
}
// trajectory.h:32-45
inline size_t getNumTrajectories(ControlType t, int maxLinearSamples,
                                 int maxAngularSamples) {
  const int ang = maxAngularSamples + 1 - (maxAngularSamples % 2);
  int vx_n, vy_n;
  computeLinearSampleSplit(t, maxLinearSamples, vx_n, vy_n);
  size_t n = static_cast<size_t>(vx_n) * static_cast<size_t>(ang);
  if (t == ControlType::OMNI) n += static_cast<size_t>(vx_n) * static_cast<size_t>(vy_n);
  return n;
}
// trajectory.h:48-51
inline size_t getNumPointsPerTrajectory(double timeStep, double horizon) {
  return horizon / timeStep;
}

struct TrajectoryVelocities2D {
  Eigen::VectorXf vx, vy, omega;
  size_t numPointsPerTrajectory_ = 0;

  TrajectoryVelocities2D() = default;
  explicit TrajectoryVelocities2D(size_t numPointsPerTrajectory)
      : vx((Eigen::Index)numPointsPerTrajectory - 1),
        vy((Eigen::Index)numPointsPerTrajectory - 1),
        omega((Eigen::Index)numPointsPerTrajectory - 1),
        numPointsPerTrajectory_(numPointsPerTrajectory) {}
  explicit TrajectoryVelocities2D(const std::vector<Velocity2D> &v)
      : TrajectoryVelocities2D(v.size() + 1) {
    for (size_t i = 0; i < v.size(); ++i) add(i, v[i]);
  }
  TrajectoryVelocities2D(const Eigen::VectorXf &vx_, const Eigen::VectorXf &vy_,
                         const Eigen::VectorXf &omega_)
      : vx(vx_), vy(vy_), omega(omega_),
        numPointsPerTrajectory_((size_t)vx_.size() + 1) {}
  void add(size_t i, const Velocity2D &v) {
    vx((Eigen::Index)i) = v.vx();
    vy((Eigen::Index)i) = v.vy();
    omega((Eigen::Index)i) = v.omega();
  }
  void add(size_t i, float a, float b, float c) {
    vx((Eigen::Index)i) = a;
    vy((Eigen::Index)i) = b;
    omega((Eigen::Index)i) = c;
  }
  Velocity2D getIndex(size_t i) const {
    return Velocity2D(vx((Eigen::Index)i), vy((Eigen::Index)i), omega((Eigen::Index)i));
  }
  Velocity2D getFront() const { return getIndex(0); }
  Velocity2D getEnd() const { return getIndex(numPointsPerTrajectory_ - 2); }
};

struct TrajectoryPath {
  Eigen::VectorXf x, y, z;
  size_t numPointsPerTrajectory_ = 0;

  TrajectoryPath() = default;
  explicit TrajectoryPath(size_t n)
      : x((Eigen::Index)n), y((Eigen::Index)n), z((Eigen::Index)n),
        numPointsPerTrajectory_(n) {}
  explicit TrajectoryPath(const Path::Path &p) : TrajectoryPath(p.getSize()) {
    for (size_t i = 0; i < p.getSize(); ++i) add(i, p.getIndex(i));
  }
  TrajectoryPath(const Eigen::VectorXf &x_, const Eigen::VectorXf &y_,
                 const Eigen::VectorXf &z_)
      : x(x_), y(y_), z(z_), numPointsPerTrajectory_((size_t)x_.size()) {}
  void add(size_t i, const Path::Point &p) { add(i, p.x(), p.y(), p.z()); }
  void add(size_t i, float px, float py, float pz = 0) {
    x((Eigen::Index)i) = px;
    y((Eigen::Index)i) = py;
    z((Eigen::Index)i) = pz;
  }
  Path::Point getIndex(size_t i) const {
    return Path::Point(x((Eigen::Index)i), y((Eigen::Index)i), z((Eigen::Index)i));
  }
  Path::Point getFront() const { return getIndex(0); }
  Path::Point getEnd() const { return getIndex(numPointsPerTrajectory_ - 1); }
};

struct Trajectory2D {
  TrajectoryVelocities2D velocities;
  TrajectoryPath path;
  size_t numPointsPerTrajectory_ = 0;

  Trajectory2D() = default;
  explicit Trajectory2D(size_t n)
      : velocities(n), path(n), numPointsPerTrajectory_(n) {}
  Trajectory2D(const TrajectoryVelocities2D &v, const TrajectoryPath &p) {
    if (v.numPointsPerTrajectory_ != p.numPointsPerTrajectory_)
      throw std::invalid_argument(
          "TrajectoryVelocities2D and TrajectoryPath must have the same "
          "numPointsPerTrajectory");
    velocities = v;
    path = p;
    numPointsPerTrajectory_ = v.numPointsPerTrajectory_;
  }
};

struct TrajectoryVelocitySamples2D {
  MatrixXfR vx, vy, omega;  // [maxNumTrajectories x (P-1)]
  size_t maxNumTrajectories_ = 0, numPointsPerTrajectory_ = 0;
  Eigen::Index velocitiesIndex_ = -1;

  TrajectoryVelocitySamples2D() = default;
  TrajectoryVelocitySamples2D(size_t maxN, size_t P)
      : vx((Eigen::Index)maxN, (Eigen::Index)P - 1),
        vy((Eigen::Index)maxN, (Eigen::Index)P - 1),
        omega((Eigen::Index)maxN, (Eigen::Index)P - 1),
        maxNumTrajectories_(maxN), numPointsPerTrajectory_(P) {}
  void push_back(const std::vector<Velocity2D> &v) {
    ++velocitiesIndex_;
    for (size_t i = 0; i + 1 < numPointsPerTrajectory_; ++i) {
      vx(velocitiesIndex_, (Eigen::Index)i) = v[i].vx();
      vy(velocitiesIndex_, (Eigen::Index)i) = v[i].vy();
      omega(velocitiesIndex_, (Eigen::Index)i) = v[i].omega();
    }
  }
  void push_back(const TrajectoryVelocities2D &v) {
    ++velocitiesIndex_;
    for (size_t i = 0; i + 1 < numPointsPerTrajectory_; ++i) {
      vx(velocitiesIndex_, (Eigen::Index)i) = v.vx((Eigen::Index)i);
      vy(velocitiesIndex_, (Eigen::Index)i) = v.vy((Eigen::Index)i);
      omega(velocitiesIndex_, (Eigen::Index)i) = v.omega((Eigen::Index)i);
    }
  }
  size_t size() const { return static_cast<size_t>(velocitiesIndex_ + 1); }
};

struct TrajectoryPathSamples {
  MatrixXfR x, y, z;  // [maxNumTrajectories x P]
  size_t maxNumTrajectories_ = 0, numPointsPerTrajectory_ = 0;
  Eigen::Index pathIndex_ = -1;

  TrajectoryPathSamples() = default;
  TrajectoryPathSamples(size_t maxN, size_t P)
      : x((Eigen::Index)maxN, (Eigen::Index)P), y((Eigen::Index)maxN, (Eigen::Index)P),
        z((Eigen::Index)maxN, (Eigen::Index)P), maxNumTrajectories_(maxN),
        numPointsPerTrajectory_(P) {}
  void push_back(const Path::Path &p) {
    ++pathIndex_;
    for (size_t i = 0; i < numPointsPerTrajectory_; ++i) {
      const Path::Point q = p.getIndex(i);
      x(pathIndex_, (Eigen::Index)i) = q.x();
      y(pathIndex_, (Eigen::Index)i) = q.y();
      z(pathIndex_, (Eigen::Index)i) = q.z();
    }
  }
  void push_back(const TrajectoryPath &p) {
    ++pathIndex_;
    for (size_t i = 0; i < numPointsPerTrajectory_; ++i) {
      x(pathIndex_, (Eigen::Index)i) = p.x((Eigen::Index)i);
      y(pathIndex_, (Eigen::Index)i) = p.y((Eigen::Index)i);
      z(pathIndex_, (Eigen::Index)i) = p.z((Eigen::Index)i);
    }
  }
  size_t size() const { return static_cast<size_t>(pathIndex_ + 1); }
};

struct TrajectorySamples2D {
  TrajectoryVelocitySamples2D velocities;
  TrajectoryPathSamples paths;
  size_t maxNumTrajectories_ = 0, numPointsPerTrajectory_ = 0;

  TrajectorySamples2D() = default;
  TrajectorySamples2D(size_t maxN, size_t P)
      : velocities(maxN, P), paths(maxN, P), maxNumTrajectories_(maxN),
        numPointsPerTrajectory_(P) {}
  TrajectorySamples2D(TrajectoryVelocitySamples2D &v, TrajectoryPathSamples &p) {
    if (v.maxNumTrajectories_ != p.maxNumTrajectories_)
      throw std::invalid_argument(
          "TrajectoryVelocitySamples2D and TrajectoryPathSamples must have "
          "the same numTrajectories");
    if (v.numPointsPerTrajectory_ != p.numPointsPerTrajectory_)
      throw std::invalid_argument(
          "TrajectoryVelocitySamples2D and TrajectoryPathSamples must have "
          "the same numPointsPerTrajectory");
    velocities = v;
    paths = p;
    maxNumTrajectories_ = v.maxNumTrajectories_;
    numPointsPerTrajectory_ = v.numPointsPerTrajectory_;
  }
  template <typename V, typename P>
  void push_back(V &v, P &p) {
    velocities.push_back(v);
    paths.push_back(p);
  }
  Trajectory2D getIndex(Eigen::Index i) const {
    const Eigen::Index nv = (Eigen::Index)numPointsPerTrajectory_ - 1;
    const Eigen::Index np = (Eigen::Index)numPointsPerTrajectory_;
    return Trajectory2D(
        TrajectoryVelocities2D(Eigen::VectorXf(velocities.vx.rowPtr(i), nv),
                               Eigen::VectorXf(velocities.vy.rowPtr(i), nv),
                               Eigen::VectorXf(velocities.omega.rowPtr(i), nv)),
        TrajectoryPath(Eigen::VectorXf(paths.x.rowPtr(i), np),
                       Eigen::VectorXf(paths.y.rowPtr(i), np),
                       Eigen::VectorXf(paths.z.rowPtr(i), np)));
  }
  size_t size() const { return velocities.size(); }
};

struct TrajSearchResult {
  Trajectory2D trajectory;
  bool isTrajFound = false;
  float trajCost = 0.0;
};

// (cost, index) with the lowest-index tie-break the device key reproduces
struct LowestCost {
  float cost;
  Eigen::Index sampleIndex;
  LowestCost(float v = DEFAULT_MIN_DIST, Eigen::Index i = 0) : cost(v), sampleIndex(i) {}
  void combine(float c, Eigen::Index i) {
    if (c < cost || (c == cost && i < sampleIndex)) {
      cost = c;
      sampleIndex = i;
    }
  }
};
inline LowestCost operator+(const LowestCost &a, const LowestCost &b) {
  LowestCost r = a;
  r.combine(b.cost, b.sampleIndex);
  return r;
}

}  // namespace Control
}  // namespace Kompass
