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

Controller::Controller() : drive_(), limits_(), host_threads_(1) {}
Controller::~Controller() {}

void Controller::setLinearControlLimits(const LinearVelocityControlParams &vx,
                                        const LinearVelocityControlParams &vy) {
  limits_.velXParams = vx;
  limits_.velYParams = vy;
}
void Controller::setAngularControlLimits(const AngularVelocityControlParams &p) {
  limits_.omegaParams = p;
}
void Controller::setControlType(const ControlType &t) { drive_ = t; }
void Controller::setCurrentVelocity(const Velocity2D &v) { velocity_ = v; }
void Controller::setCurrentState(const Path::State &s) { pose_ = s; }
void Controller::setCurrentState(double x, double y, double yaw, double speed) {
  pose_.x = x;
  pose_.y = y;
  pose_.yaw = yaw;
  pose_.speed = speed;
}
ControlType Controller::getControlType() const { return drive_; }
Velocity2D Controller::getControl() const { return last_command_; }

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
  knob_.lookahead = config.getParameter<double>("lookahead_distance");
  knob_.reverse = config.getParameter<bool>("enable_reverse_driving");
  knob_.goal_radius = config.getParameter<double>("goal_dist_tolerance");
  knob_.goal_yaw = config.getParameter<double>("goal_orientation_tolerance");
  knob_.lost_radius = config.getParameter<double>("loosing_goal_distance");
  knob_.horizon_tolerance = config.getParameter<double>("curvature_horizon_tolerance");
  knob_.segment_length = config.getParameter<double>("path_segment_length");
  knob_.point_spacing = config.getParameter<double>("max_point_interpolation_distance");
  knob_.slow_in_curves = config.getParameter<double>("speed_regulation_curvature");
  knob_.slow_in_turns = config.getParameter<double>("speed_regulation_angular");
  knob_.slowest = config.getParameter<double>("min_speed_regulation_factor");
  knob_.turn_in_place = drive_ != ControlType::ACKERMANN;
  on_.longest_segment = longestSegment();
}

size_t Follower::longestSegment() const {
  return config.getParameter<double>("path_segment_length") /
             config.getParameter<double>("max_point_interpolation_distance") +
         1;
}

Follower::Target Follower::getTrackedTarget() const { return *on_.target; }
const Path::Path Follower::getCurrentPath() const { return *on_.path; }
size_t Follower::getCurrentSegmentIndex() { return on_.segment; }
void Follower::setInterpolationType(Path::InterpolationType t) { on_.spline = t; }

void Follower::clearCurrentPath() {
  on_.path.reset();
  on_.at_goal = true;
  on_.at_yaw = true;
  on_.ready = false;
}

void Follower::setCurrentPath(const Path::Path &path, const bool interpolate) {
  on_.path = std::make_unique<Path::Path>(path);
  if (interpolate)
    on_.path->interpolate(knob_.point_spacing, on_.spline);
  on_.path->segment(knob_.segment_length, on_.longest_segment);
  on_.last_segment = on_.path->getNumSegments() - 1;
  on_.ready = true;
  on_.segment = 0;
  on_.along = 0.0;
  on_.goal_distance = std::numeric_limits<double>::max();
  on_.goal_yaw_error = on_.path->getEndOrientation();
  on_.at_goal = false;
  on_.at_yaw = false;
}

bool Follower::isGoalReached() {
  if (!on_.ready) return true;
  const Path::Point goal = on_.path->getEnd();
  const double d = std::hypot(pose_.x - goal.x(), pose_.y - goal.y());
  const bool end_reached = d <= knob_.goal_radius;
  bool loosing = false;
  if ((on_.segment + 1) >= on_.last_segment) {
    if (d < on_.goal_distance) {
      on_.goal_distance = d;
    } else if (std::abs(d - on_.goal_distance) > knob_.lost_radius) {
      LOG_DEBUG("Already Reached the Goal, Ending Action\n");
      loosing = true;
    }
  }
  if (end_reached || loosing) {
    on_.ready = false;
    on_.at_goal = true;
  }
  return on_.at_goal;
}

Path::PathPosition Follower::locateOnPath() {
  on_.segment = nearestSegment(0, on_.last_segment);
  return nearestOnSegment(on_.segment);
}

size_t Follower::nearestSegment(size_t left, size_t right) {
  if (left == right) return left;
  const size_t mid = (left + right) / 2;
  const float dl = Path::Path::distanceSquared(pose_, on_.path->getSegmentStart(left));
  const float dr = Path::Path::distanceSquared(pose_, on_.path->getSegmentStart(right));
  if (mid == right || mid == left) return dl <= dr ? left : right;
  return dl <= dr ? nearestSegment(left, mid) : nearestSegment(mid, right);
}

Path::PathPosition Follower::nearestOnSegment(size_t seg) {
  const Path::Path::View view = on_.path->getSegment(seg);
  const size_t first = on_.path->getSegmentStartIndex(seg);
  const Path::Point a = on_.path->getSegmentStart(seg), b = on_.path->getSegmentEnd(seg);
  const double seg_heading = std::atan2(b.y() - a.y(), b.x() - a.x());  // float overload
  double best = std::numeric_limits<float>::max();
  Path::State closest;
  size_t closest_k = 0;
  double pos = 0.0;
  for (size_t k = 0; k < view.getSize(); ++k) {
    const Path::Point p = view.getIndex(k);
    const double d2 = Path::Path::distanceSquared(pose_, p);
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
  const double vx = pose_.x - closest.x, vy = pose_.y - closest.y;
  const double cross = std::cos(closest.yaw) * vy - std::sin(closest.yaw) * vx;
  out.parallel_distance = cross > 0 ? out.normal_distance : -out.normal_distance;
  return out;
}

void Follower::aimAtTarget() {
  on_.target = std::make_unique<Target>();
  const bool research =
      on_.nearest->segment_length <= 0.0 ||
      on_.nearest->index >= on_.path->getSegmentEndIndex(on_.segment) ||
      on_.nearest->segment_length >= 0.9;
  *on_.nearest = research ? locateOnPath()
                              : nearestOnSegment(on_.nearest->segment_index);
  on_.target->segment_index = on_.segment;
  on_.target->position_in_segment = on_.nearest->segment_length;
  on_.target->movement = on_.nearest->state;
  on_.target->lookahead = knob_.lookahead;
  on_.target->heading_error =
      Angle::normalizeToMinusPiPlusPi(on_.target->movement.yaw - pose_.yaw);
  on_.target->crosstrack_error = on_.nearest->parallel_distance;
  on_.target->reverse = false;
}

}  // namespace Control
}  // namespace Kompass
