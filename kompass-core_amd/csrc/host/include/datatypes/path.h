// Reference path container of the kompass_cpp surface (reference:
// datatypes/path.h, src/datatypes/path.cpp).  Host-only: the path is prepared
// once per new path; the hot path consumes a View of it on the device.
#pragma once

#include <cmath>
#include <cstddef>
#include <vector>

#include "datatypes/control.h"
#include "kc_linalg.h"

namespace Path {

enum class InterpolationType { LINEAR, CUBIC_SPLINE, HERMITE_SPLINE };

struct State {
  double x, y, yaw, speed;
  State(double poseX = 0.0, double poseY = 0.0, double PoseYaw = 0.0,
        double speedValue = 0.0)
      : x(poseX), y(poseY), yaw(PoseYaw), speed(speedValue) {}
  // one kinematic step (path.h:24-30): the time step is narrowed to float
  void update(const Kompass::Control::Velocity2D &vel, const float timeStep) {
    const double c = std::cos(yaw), s = std::sin(yaw);
    x += (vel.vx() * c - vel.vy() * s) * timeStep;
    y += (vel.vx() * s + vel.vy() * c) * timeStep;
    yaw += vel.omega() * timeStep;
  }
};

typedef Eigen::Vector3f Point;

struct Path {
  // window [start, start + length) of a parent path; the parent must outlive it
  struct View {
    const float *X, *Y, *Z, *Curvature;
    const float *AccumulatedLengths;  // absolute prefix arc lengths
    size_t start_idx_, size_;
    size_t acc_available_;  // entries of the parent's prefix array from start

    View(const Path &parent, size_t start, size_t length);
    size_t getSize() const { return size_; }
    size_t getStartIndex() const { return start_idx_; }
    const float *getXPointer() const { return X; }
    const float *getYPointer() const { return Y; }
    const float *getZPointer() const { return Z; }
    const float *getCurvaturePointer() const { return Curvature; }
    const float *getAccumulatedLengthsPointer() const {
      return AccumulatedLengths;
    }
    Point getIndex(size_t i) const { return Point(X[i], Y[i], Z[i]); }
    double getCurvature(size_t i) const { return Curvature[i]; }
    float totalSegmentLength() const;
  };

  Path(const std::vector<Point> &points = {});
  Path(const Eigen::VectorXf &x, const Eigen::VectorXf &y,
       const Eigen::VectorXf &z);

  Eigen::VectorXf getX() const { return Eigen::VectorXf(X_.data(), (Eigen::Index)size_); }
  Eigen::VectorXf getY() const { return Eigen::VectorXf(Y_.data(), (Eigen::Index)size_); }
  Eigen::VectorXf getZ() const { return Eigen::VectorXf(Z_.data(), (Eigen::Index)size_); }
  size_t getSize() const { return size_; }
  Point getEnd() const { return getIndex(size_ - 1); }
  Point getStart() const { return getIndex(0); }
  Point getIndex(size_t i) const { return Point(X_[i], Y_[i], Z_[i]); }
  double getCurvature(size_t i) const { return i >= size_ ? 0.0 : K_[i]; }
  float getDistanceAtIndex(size_t i) const {
    return i >= acc_.size() ? 0.0f : acc_[i];
  }
  const float *getAccumulatedLengthsPointer() const { return acc_.data(); }
  size_t getAccumulatedLengthsSize() const { return acc_.size(); }

  static float distance(const Point &a, const Point &b);
  static float distanceSquared(const Point &a, const Point &b);
  static float distanceSquared(const State &s, const Point &p);

  void resize(size_t n);
  bool endReached(State currentState, double minDist);
  View getPart(size_t start, size_t end) const;
  void pushPoint(const Point &p);
  float getEndOrientation() const;
  float getStartOrientation() const;
  float getOrientation(size_t index) const;
  float totalPathLength() const;
  size_t getNumSegments() const { return segments_.size(); }
  View getSegment(size_t segment_index) const;
  size_t getSegmentSize(size_t segment_index) const;
  size_t getSegmentStartIndex(size_t segment_index) const;
  size_t getSegmentEndIndex(size_t segment_index) const;
  void interpolate(double max_interpolation_point_dist, InterpolationType type);
  void segment(double pathSegmentLength, size_t maxPointsPerSegment);
  Point getSegmentStart(size_t segment_index) const;
  Point getSegmentEnd(size_t segment_index) const;

  const float *xData() const { return X_.data(); }
  const float *yData() const { return Y_.data(); }
  const float *zData() const { return Z_.data(); }
  const float *curvatureData() const { return K_.data(); }
  // changes whenever the points or the accumulated lengths may have changed: lets
  // the cost evaluator keep the path resident on the device between cycles
  unsigned long long serial() const { return serial_; }

 private:
  std::vector<float> X_, Y_, Z_, K_;
  std::vector<size_t> segments_;  // first index of every segment
  size_t size_ = 0;
  float total_length_ = 0.0f;
  std::vector<float> acc_;
  bool interpolated_ = false;
  unsigned long long serial_ = 0;
  void touch();
  void checkSegment(size_t s) const;
};

struct PathPosition {
  size_t index{0};
  size_t segment_index{0};
  double segment_length{-1.0};
  double parallel_distance{0.0};
  double normal_distance{0.0};
  State state;
};

}  // namespace Path
