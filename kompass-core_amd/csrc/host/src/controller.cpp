// Controller / Follower host logic (reference: src/controllers/controller.cpp,
// follower.cpp).  Serial per-cycle bookkeeping; nothing here is batch work.
#include "controllers/controller.h"

#include "controllers/follower.h"
#include "utils/angles.h"
#include "utils/logger.h"

namespace Kompass {
namespace Control {

std::string controlTypeToString(ControlType t) {
  switch (t) {
    case ControlType::ACKERMANN: return "ACKERMANN";
    case ControlType::DIFFERENTIAL_DRIVE: return "DIFFERENTIAL_DRIVE";
    case ControlType::OMNI: return "OMNI";
  }
  return "Unknown";
}

Controller::Controller() : ctrType(), ctrlimitsParams(), maxNumThreads(1) {}
Controller::~Controller() {}

void Controller::setLinearControlLimits(const LinearVelocityControlParams &vx,
                                        const LinearVelocityControlParams &vy) {
  ctrlimitsParams.velXParams = vx;
  ctrlimitsParams.velYParams = vy;
}
void Controller::setAngularControlLimits(const AngularVelocityControlParams &p) {
  ctrlimitsParams.omegaParams = p;
}
void Controller::setControlType(const ControlType &t) { ctrType = t; }
void Controller::setCurrentVelocity(const Velocity2D &v) { currentVel = v; }
void Controller::setCurrentState(const Path::State &s) { currentState = s; }
void Controller::setCurrentState(double x, double y, double yaw, double speed) {
  currentState.x = x;
  currentState.y = y;
  currentState.yaw = yaw;
  currentState.speed = speed;
}
ControlType Controller::getControlType() const { return ctrType; }
Velocity2D Controller::getControl() const { return currentCtr; }

double Controller::restrictVelocityTolimits(double cur, double target, double acc,
                                            double dec, double maxVel, double dt) const {
  double cmd = cur;
  if (cur < target) {
    cmd = std::min(cur + acc * dt, target);
  } else if (cur > target) {
    cmd = std::max(cur - dec * dt, target);
  }
  return std::clamp(cmd, -maxVel, maxVel);
}

// ---------------------------------------------------------------------------
Follower::Follower() : Controller(), config() { setParams(config); }
Follower::Follower(const FollowerParameters &cfg) : Follower() { setParams(cfg); }

void Follower::setParams(const FollowerParameters &cfg) {
  config = cfg;
  lookahead_distance = config.getParameter<double>("lookahead_distance");
  enable_reverse_driving = config.getParameter<bool>("enable_reverse_driving");
  goal_dist_tolerance = config.getParameter<double>("goal_dist_tolerance");
  goal_orientation_tolerance = config.getParameter<double>("goal_orientation_tolerance");
  loosing_goal_distance = config.getParameter<double>("loosing_goal_distance");
  curvature_horizon_tolerance_ = config.getParameter<double>("curvature_horizon_tolerance");
  path_segment_length_ = config.getParameter<double>("path_segment_length");
  max_point_interpolation_distance_ = config.getParameter<double>("max_point_interpolation_distance");
  speed_reg_curvature = config.getParameter<double>("speed_regulation_curvature");
  speed_reg_rotation = config.getParameter<double>("speed_regulation_angular");
  min_speed_regulation_factor = config.getParameter<double>("min_speed_regulation_factor");
  rotate_in_place = ctrType != ControlType::ACKERMANN;
  max_segment_size_ = getMaxSegmentSize();
}

size_t Follower::getMaxSegmentSize() const {
  return config.getParameter<double>("path_segment_length") /
             config.getParameter<double>("max_point_interpolation_distance") +
         1;
}

Follower::Target Follower::getTrackedTarget() const { return *currentTrackedTarget_; }
const Path::Path Follower::getCurrentPath() const { return *currentPath; }
size_t Follower::getCurrentSegmentIndex() { return current_segment_index_; }
void Follower::setInterpolationType(Path::InterpolationType t) { interpolationType = t; }

void Follower::clearCurrentPath() {
  currentPath.reset();
  reached_goal_ = true;
  reached_yaw_ = true;
  path_processing_ = false;
}

void Follower::setCurrentPath(const Path::Path &path, const bool interpolate) {
  currentPath = std::make_unique<Path::Path>(path);
  if (interpolate)
    currentPath->interpolate(max_point_interpolation_distance_, interpolationType);
  currentPath->segment(path_segment_length_, max_segment_size_);
  max_segment_index_ = currentPath->getNumSegments() - 1;
  path_processing_ = true;
  current_segment_index_ = 0;
  current_position_in_segment_ = 0.0;
  goal_distance_ = std::numeric_limits<double>::max();
  goal_orientation_ = currentPath->getEndOrientation();
  reached_goal_ = false;
  reached_yaw_ = false;
}

bool Follower::isGoalReached() {
  if (!path_processing_) return true;
  const Path::Point goal = currentPath->getEnd();
  const double d = std::hypot(currentState.x - goal.x(), currentState.y - goal.y());
  const bool end_reached = d <= goal_dist_tolerance;
  bool loosing = false;
  if ((current_segment_index_ + 1) >= max_segment_index_) {
    if (d < goal_distance_) {
      goal_distance_ = d;
    } else if (std::abs(d - goal_distance_) > loosing_goal_distance) {
      LOG_DEBUG("Already Reached the Goal, Ending Action\n");
      loosing = true;
    }
  }
  if (end_reached || loosing) {
    path_processing_ = false;
    reached_goal_ = true;
  }
  return reached_goal_;
}

Path::PathPosition Follower::findClosestPathPoint() {
  current_segment_index_ = findClosestSegmentIndex(0, max_segment_index_);
  return findClosestPointOnSegment(current_segment_index_);
}

size_t Follower::findClosestSegmentIndex(size_t left, size_t right) {
  if (left == right) return left;
  const size_t mid = (left + right) / 2;
  const float dl = Path::Path::distanceSquared(currentState, currentPath->getSegmentStart(left));
  const float dr = Path::Path::distanceSquared(currentState, currentPath->getSegmentStart(right));
  if (mid == right || mid == left) return dl <= dr ? left : right;
  return dl <= dr ? findClosestSegmentIndex(left, mid) : findClosestSegmentIndex(mid, right);
}

Path::PathPosition Follower::findClosestPointOnSegment(size_t seg) {
  const Path::Path::View view = currentPath->getSegment(seg);
  const size_t first = currentPath->getSegmentStartIndex(seg);
  const Path::Point a = currentPath->getSegmentStart(seg), b = currentPath->getSegmentEnd(seg);
  const double seg_heading = std::atan2(b.y() - a.y(), b.x() - a.x());  // float overload
  double best = std::numeric_limits<float>::max();
  Path::State closest;
  size_t closest_k = 0;
  double pos = 0.0;
  for (size_t k = 0; k < view.getSize(); ++k) {
    const Path::Point p = view.getIndex(k);
    const double d2 = Path::Path::distanceSquared(currentState, p);
    if (d2 <= best) {  // last minimum wins, as in the reference
      best = d2;
      closest = Path::State(p.x(), p.y(), seg_heading);
      closest_k = k;
      pos = view.getSize() > 1 ? static_cast<double>(k) / (view.getSize() - 1) : 1.0;
    }
  }
  Path::PathPosition out;
  out.index = closest_k + first;
  out.segment_index = seg;
  out.segment_length = pos;
  out.state = closest;
  out.normal_distance = std::sqrt(best);
  const double vx = currentState.x - closest.x, vy = currentState.y - closest.y;
  const double cross = std::cos(closest.yaw) * vy - std::sin(closest.yaw) * vx;
  out.parallel_distance = cross > 0 ? out.normal_distance : -out.normal_distance;
  return out;
}

void Follower::determineTarget() {
  currentTrackedTarget_ = std::make_unique<Target>();
  const bool research =
      closestPosition->segment_length <= 0.0 ||
      closestPosition->index >= currentPath->getSegmentEndIndex(current_segment_index_) ||
      closestPosition->segment_length >= 0.9;
  *closestPosition = research ? findClosestPathPoint()
                              : findClosestPointOnSegment(closestPosition->segment_index);
  currentTrackedTarget_->segment_index = current_segment_index_;
  currentTrackedTarget_->position_in_segment = closestPosition->segment_length;
  currentTrackedTarget_->movement = closestPosition->state;
  currentTrackedTarget_->lookahead = lookahead_distance;
  currentTrackedTarget_->heading_error =
      Angle::normalizeToMinusPiPlusPi(currentTrackedTarget_->movement.yaw - currentState.yaw);
  currentTrackedTarget_->crosstrack_error = closestPosition->parallel_distance;
  currentTrackedTarget_->reverse = false;
}

}  // namespace Control
}  // namespace Kompass
