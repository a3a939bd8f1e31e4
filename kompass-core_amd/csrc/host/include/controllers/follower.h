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
    return std::max(std::min(latest_velocity_command_.vx(), ctrlimitsParams.velXParams.maxVel),
                    -ctrlimitsParams.velXParams.maxVel);
  }
  double getLinearVelocityCmdY() const {
    return std::max(std::min(latest_velocity_command_.vy(), ctrlimitsParams.velYParams.maxVel),
                    -ctrlimitsParams.velYParams.maxVel);
  }
  double getAngularVelocityCmd() const {
    return std::max(std::min(latest_velocity_command_.omega(), ctrlimitsParams.omegaParams.maxOmega),
                    -ctrlimitsParams.omegaParams.maxOmega);
  }
  double getSteeringAngleCmd() const { return latest_velocity_command_.steer_ang(); }
  double getPathLength() const { return currentPath->totalPathLength(); }
  bool hasPath() const {
    if (!currentPath || !path_processing_) return false;
    return currentPath->totalPathLength() > 0.0;
  }
  const Path::Path getCurrentPath() const;

 protected:
  double speed_reg_curvature{0.0}, speed_reg_rotation{0.0};
  std::unique_ptr<Path::Path> currentPath = nullptr;
  std::unique_ptr<Path::PathPosition> closestPosition = std::make_unique<Path::PathPosition>();
  double goal_dist_tolerance{0.0}, goal_orientation_tolerance{0.0};
  double loosing_goal_distance{0.0}, curvature_horizon_tolerance_{1.0};
  bool rotate_in_place{false};
  double lookahead_distance{0.0};
  bool enable_reverse_driving{false};
  double path_segment_length_{0.0}, min_speed_regulation_factor{0.0};
  double max_point_interpolation_distance_{0.0};
  size_t max_segment_size_;
  Path::InterpolationType interpolationType = Path::InterpolationType::LINEAR;
  FollowerParameters config = FollowerParameters();

  Path::PathPosition findClosestPathPoint();
  void determineTarget();

  bool path_processing_{false};
  std::unique_ptr<Target> currentTrackedTarget_ = std::make_unique<Target>();
  size_t current_segment_index_{0};
  double current_position_in_segment_{0.0};
  size_t max_segment_index_{0};
  double goal_distance_{std::numeric_limits<double>::max()};
  double goal_orientation_{std::numeric_limits<double>::max()};
  Control::Velocity2D latest_velocity_command_{0.0, 0.0, 0.0};
  bool reached_goal_{false}, reached_yaw_{false};

 private:
  size_t findClosestSegmentIndex(size_t left, size_t right);
  Path::PathPosition findClosestPointOnSegment(size_t segment_index);
  size_t getMaxSegmentSize() const;
};

}  // namespace Control
}  // namespace Kompass
