// Path-follower base of the kompass_cpp surface (reference: controllers/
// follower.{h,cpp}).  Closest-point tracking on the host: serial, O(segment).
#pragma once

#include <cmath>
#include <limits>
#include <memory>

#include "controllers/controller.h"
#include "datatypes/control.h"
#include "datatypes/parameter.h"
#include "datatypes/path.h"

namespace Kompass {
namespace Control {

class Follower : public Controller {
 public:
  class FollowerParameters : public Controller::ControllerParameters {
   public:
    FollowerParameters() : Controller::ControllerParameters() {
      addParameter("max_point_interpolation_distance", Parameter(0.01, 0.0001, 1000.0));
      addParameter("lookahead_distance", Parameter(1.0, 0.0, 1000.0));
      addParameter("speed_regulation_curvature", Parameter(0.5, 0.0, 1.0));
      addParameter("speed_regulation_angular", Parameter(0.5, 0.0, 1.0));
      addParameter("min_speed_regulation_factor", Parameter(0.5, 1e-3, 1.0));
      addParameter("goal_dist_tolerance", Parameter(0.1, 0.001, 1000.0));
      addParameter("path_segment_length", Parameter(1.0, 0.001, 1000.0));
      addParameter("goal_orientation_tolerance", Parameter(0.1, 0.001, 2 * M_PI));
      addParameter("loosing_goal_distance", Parameter(0.5, 0.001, 1000.0));
      addParameter("curvature_horizon_tolerance", Parameter(1.5, 0.5, 1000.0));
    }
  };

  struct Target {
    size_t segment_index{0};
    double position_in_segment{0.0};
    Path::State movement = Path::State();
    bool reverse{false};
    double lookahead{0.0};
    double crosstrack_error{0.0};
    double heading_error{0.0};
  };

  Follower();
  Follower(const FollowerParameters &config);
  void setParams(const FollowerParameters &config);
  virtual ~Follower() = default;

  void setCurrentPath(const Path::Path &path, const bool interpolate = true);
  void clearCurrentPath();
  bool isGoalReached();
  void setInterpolationType(Path::InterpolationType type);
  size_t getCurrentSegmentIndex();
  Target getTrackedTarget() const;

  double getLinearVelocityCmdX() const {
    return std::max(std::min(command_.vx(), limits_.velXParams.maxVel),
                    -limits_.velXParams.maxVel);
  }
  double getLinearVelocityCmdY() const {
    return std::max(std::min(command_.vy(), limits_.velYParams.maxVel),
                    -limits_.velYParams.maxVel);
  }
  double getAngularVelocityCmd() const {
    return std::max(std::min(command_.omega(), limits_.omegaParams.maxOmega),
                    -limits_.omegaParams.maxOmega);
  }
  double getSteeringAngleCmd() const { return command_.steer_ang(); }
  double getPathLength() const { return on_.path->totalPathLength(); }
  bool hasPath() const {
    if (!on_.path || !on_.ready) return false;
    return on_.path->totalPathLength() > 0.0;
  }
  const Path::Path getCurrentPath() const;

 protected:
  // What setParams reads out of the parameter set, once per change
  struct Knobs {
    double slow_in_curves{0.0}, slow_in_turns{0.0}, slowest{0.0};  // speed regulation
    double goal_radius{0.0}, goal_yaw{0.0}, lost_radius{0.0};       // goal tests
    double horizon_tolerance{1.0}, lookahead{0.0};
    double segment_length{0.0}, point_spacing{0.0};                 // how a new path is cut / interpolated
    bool turn_in_place{false}, reverse{false};
  } knob_;
  // The path being followed and the robot's place on it (rewritten by locateOnPath / aimAtTarget every cycle)
  struct Place {
    std::unique_ptr<Path::Path> path;
    std::unique_ptr<Path::PathPosition> nearest = std::make_unique<Path::PathPosition>();
    std::unique_ptr<Target> target = std::make_unique<Target>();
    Path::InterpolationType spline = Path::InterpolationType::LINEAR;
    bool ready{false};                       // a path was set and cut into segments
    size_t segment{0}, last_segment{0}, longest_segment{0};
    double along{0.0};                       // position inside `segment`, 0 .. 1
    double goal_distance{std::numeric_limits<double>::max()};
    double goal_yaw_error{std::numeric_limits<double>::max()};
    bool at_goal{false}, at_yaw{false};
  } on_;
  Control::Velocity2D command_{0.0, 0.0, 0.0};  // what the last cycle decided
  FollowerParameters config = FollowerParameters();

  Path::PathPosition locateOnPath();
  void aimAtTarget();

 private:
  size_t nearestSegment(size_t left, size_t right);
  Path::PathPosition nearestOnSegment(size_t segment_index);
  size_t longestSegment() const;
};

}  // namespace Control
}  // namespace Kompass
