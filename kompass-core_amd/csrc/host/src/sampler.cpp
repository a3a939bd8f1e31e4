// CollisionChecker + TrajectorySampler host side (reference: src/utils/
// collision_check.cpp, trajectory_sampler.cpp).  Everything batch-shaped is a
// call into libkompass_hip.so.
#include "utils/trajectory_sampler.h"

#include <cmath>
#include <cstring>

namespace Kompass {

namespace {
kc_dwa_params baseParams(CollisionChecker::ShapeType shape, const std::vector<float> &dims,
                         const Eigen::Vector3f &spos, const Eigen::Quaternionf &srot,
                         double res) {
  kc_dwa_params p;
  std::memset(&p, 0, sizeof(p));
  p.shape = static_cast<int>(shape);
  p.ndims = static_cast<int>(std::min<size_t>(dims.size(), 3));
  for (int i = 0; i < p.ndims; ++i) p.dims[i] = dims[static_cast<size_t>(i)];
  for (int i = 0; i < 3; ++i) p.sensor_pos[i] = spos(i);
  p.sensor_rot_xyzw[0] = srot.x();
  p.sensor_rot_xyzw[1] = srot.y();
  p.sensor_rot_xyzw[2] = srot.z();
  p.sensor_rot_xyzw[3] = srot.w();
  p.octree_res = res;
  p.time_step = 0.1;
  p.max_samples = 1;
  p.max_points = 2;
  p.max_segment = 16;
  p.max_obstacles = 1024;
  p.acc_limits[0] = p.acc_limits[1] = p.acc_limits[2] = 1.0f;
  p.device = 0;
  return p;
}
void shapeExtents(CollisionChecker::ShapeType shape, const std::vector<float> &d,
                  double &radius, double &height) {
  // collision_check.cpp:38-58
  switch (shape) {
    case CollisionChecker::ShapeType::CYLINDER:
      radius = d.at(0);
      height = d.at(1);
      break;
    case CollisionChecker::ShapeType::BOX:
      height = d.at(2);
      radius = std::sqrt(std::pow(d.at(0), 2) + std::pow(d.at(1), 2)) / 2;
      break;
    case CollisionChecker::ShapeType::SPHERE:
      radius = d.at(0);
      height = 2 * d.at(0);
      break;
    default:
      throw std::invalid_argument("Invalid robot geometry type");
  }
}
kc_state toKc(const Path::State &s) { return kc_state{s.x, s.y, s.yaw, s.speed}; }
}  // namespace

// ---------------------------------------------------------------------------
CollisionChecker::CollisionChecker(const ShapeType shape, const std::vector<float> &dims,
                                   const Eigen::Vector3f &spos,
                                   const Eigen::Quaternionf &srot, const double res)
    : voxel_(res) {
  shapeExtents(shape, dims, body_radius_, body_height_);
  ctx_ = hip::makeDwa(baseParams(shape, dims, spos, srot, res));
}
CollisionChecker::CollisionChecker(hip::DwaHandle ctx, ShapeType shape,
                                   const std::vector<float> &dims, double res)
    : ctx_(std::move(ctx)), voxel_(res) {
  shapeExtents(shape, dims, body_radius_, body_height_);
}

void CollisionChecker::resetOctreeResolution(const double r) {
  if (r != voxel_) {
    voxel_ = r;
    hip::check(kc_dwa_set_resolution(ctx_.get(), r));
  }
}
float CollisionChecker::getRadius() const { return static_cast<float>(body_radius_); }
void CollisionChecker::updateState(const Path::State s) { pose_ = s; }
void CollisionChecker::updateState(const double x, const double y, const double yaw) {
  pose_ = Path::State(x, y, yaw);
}
void CollisionChecker::updateSensorData(const Control::LaserScan &scan, const bool) {
  const kc_state st = toKc(pose_);
  hip::check(kc_dwa_set_scan(ctx_.get(), &st, scan.ranges.data(), scan.angles.data(),
                             std::min(scan.ranges.size(), scan.angles.size()), maxSensorRange));
}
void CollisionChecker::updateSensorData(const std::vector<Path::Point> &cloud,
                                        const bool global_frame) {
  // Path::Point is three packed floats: the list goes to the device as it lies
  static_assert(sizeof(Path::Point) == 3 * sizeof(float), "Path::Point must be packed (x, y, z)");
  updateSensorData(Control::PointCloudView{cloud.empty() ? nullptr : cloud.data()->data(), cloud.size()}, global_frame);
}
// (collision_check.h:119-131: world-frame lists -- what the controllers pass -- or sensor-frame lists)
void CollisionChecker::updateSensorData(const Control::PointCloudView &cloud, const bool global_frame) {
  const kc_state st = toKc(pose_);
  hip::check(global_frame ? kc_dwa_set_points(ctx_.get(), &st, cloud.xyz, cloud.n, maxSensorRange)
                          : kc_dwa_set_points_sensor_frame(ctx_.get(), &st, cloud.xyz, cloud.n, maxSensorRange));
}
void CollisionChecker::updateSensorData(const Mapping::LocalMapper &mapper, const bool) {
  const kc_state st = toKc(pose_);
  hip::check(kc_dwa_set_grid_from_mapper(ctx_.get(), &st, mapper.hipContext(), maxSensorRange));
}
std::vector<bool> CollisionChecker::checkCollisions(const std::vector<Path::State> &states) {
  const size_t n = states.size();
  std::vector<double> x(n), y(n), yaw(n);
  for (size_t i = 0; i < n; ++i) {
    x[i] = states[i].x;
    y[i] = states[i].y;
    yaw[i] = states[i].yaw;
  }
  std::vector<uint8_t> hit(n);
  hip::check(kc_dwa_check_poses(ctx_.get(), x.data(), y.data(), yaw.data(), n, hit.data()));
  return std::vector<bool>(hit.begin(), hit.end());
}
bool CollisionChecker::checkCollisions(const Path::State s) {
  return checkCollisions(std::vector<Path::State>{s})[0];
}
bool CollisionChecker::checkCollisions() { return checkCollisions(pose_); }
bool CollisionChecker::checkCollisions(const std::vector<double> &ranges,
                                       const std::vector<double> &angles, double) {
  updateSensorData(Control::LaserScan(ranges, angles));
  return checkCollisions();
}

namespace Control {

// ---------------------------------------------------------------------------
TrajectorySampler::TrajectorySampler(
    ControlLimitsParams controlLimits, ControlType controlType, double timeStep,
    double predictionHorizon, double controlHorizon, int maxLinearSamples,
    int maxAngularSamples, const CollisionChecker::ShapeType robotShapeType,
    const std::vector<float> robotDimensions, const Eigen::Vector3f &spos,
    const Eigen::Quaternionf &srot, const double octreeRes, const int maxNumThreads) {
  limits_ = controlLimits;
  drive_ = controlType;
  time_step_ = timeStep;
  max_time_ = base_max_time_ = predictionHorizon;
  control_time_ = controlHorizon;
  lin_samples_max_ = maxLinearSamples;
  ang_samples_max_raw_ = maxAngularSamples;
  host_threads_ = maxNumThreads;
  init(robotShapeType, robotDimensions, spos, srot, octreeRes);
}

TrajectorySampler::TrajectorySampler(
    TrajectorySamplerParameters config, ControlLimitsParams controlLimits,
    ControlType controlType, const CollisionChecker::ShapeType robotShapeType,
    const std::vector<float> robotDimensions, const Eigen::Vector3f &spos,
    const Eigen::Quaternionf &srot, const int maxNumThreads) {
  limits_ = controlLimits;
  drive_ = controlType;
  time_step_ = config.getParameter<double>("time_step");
  max_time_ = base_max_time_ = config.getParameter<double>("prediction_horizon");
  control_time_ = config.getParameter<double>("control_horizon");
  lin_samples_max_ = config.getParameter<int>("max_linear_samples");
  ang_samples_max_raw_ = config.getParameter<int>("max_angular_samples");
  host_threads_ = maxNumThreads;
  drop_samples_ = config.getParameter<bool>("drop_samples");
  init(robotShapeType, robotDimensions, spos, srot,
       config.getParameter<double>("octree_map_resolution"));
  setSampleDroppingMode(drop_samples_);
}

void TrajectorySampler::init(const CollisionChecker::ShapeType shape,
                             const std::vector<float> &dims, const Eigen::Vector3f &spos,
                             const Eigen::Quaternionf &srot, double octreeRes) {
  const int ang = ang_samples_max_raw_ + 1 - (ang_samples_max_raw_ % 2);
  numPointsPerTrajectory = getNumPointsPerTrajectory(time_step_, max_time_);
  numTrajectories = getNumTrajectories(drive_, lin_samples_max_, ang);
  kc_dwa_params p = baseParams(shape, dims, spos, srot, octreeRes);
  p.time_step = time_step_;
  p.max_samples = numTrajectories + 8;
  p.max_points = std::max<size_t>(numPointsPerTrajectory, 2);
  p.max_segment = 512;
  p.acc_limits[0] = static_cast<float>(limits_.velXParams.maxAcceleration);
  p.acc_limits[1] = static_cast<float>(limits_.velYParams.maxAcceleration);
  p.acc_limits[2] = static_cast<float>(limits_.omegaParams.maxAcceleration);
  ctx_ = hip::makeDwa(p);
  checker_ = std::make_unique<CollisionChecker>(ctx_, shape, dims, octreeRes);
  if (drive_ != ControlType::OMNI)  // trajectory_sampler.cpp:51-54
    limits_.velYParams = LinearVelocityControlParams(0.0, 0.0, 0.0);
}

void TrajectorySampler::updateState(const Path::State &s) { checker_->updateState(s); }
void TrajectorySampler::setSampleDroppingMode(const bool drop) {
  // trajectory_sampler.cpp:103-105 / :157-168.  numCtrlPoints_ is control_horizon / time_step as size_t (:88,
  // the config-object constructor; the explicit-argument constructor of the reference leaves the member
  // uninitialised -- reference quirk Q3 -- and gets the same definition here)
  drop_samples_ = drop;
  hip::check(kc_dwa_set_option(ctx_.get(), "num_ctrl_points",
                               static_cast<double>(static_cast<size_t>(control_time_ / time_step_))));
  hip::check(kc_dwa_set_option(ctx_.get(), "drop_samples", drop ? 1.0 : 0.0));
}
void TrajectorySampler::resetOctreeResolution(const double r) {
  checker_->resetOctreeResolution(r);
}
float TrajectorySampler::getRobotRadius() const { return checker_->getRadius(); }

void TrajectorySampler::setPredictionHorizon(double horizon) {
  const double min_h = 2.0 * time_step_;
  horizon = std::max(horizon, min_h);
  horizon = std::min(horizon, base_max_time_);
  max_time_ = horizon;
  numPointsPerTrajectory = getNumPointsPerTrajectory(time_step_, max_time_);
}

size_t TrajectorySampler::launch(const Velocity2D &vel, const Path::State &pose) {
  const size_t n = sampleWindow(vel);
  const kc_state st = toKc(pose);
  hip::check(kc_dwa_rollout(ctx_.get(), &st, numPointsPerTrajectory));
  return n;
}

size_t TrajectorySampler::sampleWindow(const Velocity2D &vel, bool host_copy) {
  kc_limits L;
  L.vx_max = limits_.velXParams.maxVel;
  L.vx_acc = limits_.velXParams.maxAcceleration;
  L.vx_dec = limits_.velXParams.maxDeceleration;
  L.vy_max = limits_.velYParams.maxVel;
  L.vy_acc = limits_.velYParams.maxAcceleration;
  L.vy_dec = limits_.velYParams.maxDeceleration;
  L.omega_max_angle = limits_.omegaParams.maxAngle;
  L.omega_max = limits_.omegaParams.maxOmega;
  L.omega_acc = limits_.omegaParams.maxAcceleration;
  L.omega_dec = limits_.omegaParams.maxDeceleration;
  size_t n = 0;
  if (host_copy) {  // collect() reads the lattice back sample by sample; the device cycle does not
    last_vx_.resize(numTrajectories + 8);
    last_vy_.resize(numTrajectories + 8);
    last_omega_.resize(numTrajectories + 8);
  }
  hip::check(kc_dwa_sample_window(ctx_.get(), static_cast<int>(drive_), &L, vel.vx(), vel.vy(),
                                  vel.omega(), lin_samples_max_, ang_samples_max_raw_, &n,
                                  host_copy ? last_vx_.data() : nullptr, host_copy ? last_vy_.data() : nullptr,
                                  host_copy ? last_omega_.data() : nullptr, host_copy ? last_vx_.size() : 0));
  return n;
}

size_t TrajectorySampler::rolloutOnDevice(const Velocity2D &vel, const Path::State &pose,
                                          const LaserScan &scan, float max_range) {
  checker_->maxSensorRange = max_range;
  checker_->updateState(pose);
  checker_->updateSensorData(scan);
  return launch(vel, pose);
}
size_t TrajectorySampler::rolloutOnDevice(const Velocity2D &vel, const Path::State &pose,
                                          const std::vector<Path::Point> &cloud, float max_range) {
  checker_->maxSensorRange = max_range;
  checker_->updateState(pose);
  checker_->updateSensorData(cloud);
  return launch(vel, pose);
}

size_t TrajectorySampler::rolloutOnDevice(const Velocity2D &vel, const Path::State &pose,
                                          const Mapping::LocalMapper &mapper, float max_range) {
  checker_->maxSensorRange = max_range;
  checker_->updateState(pose);
  checker_->updateSensorData(mapper);
  return launch(vel, pose);
}

std::unique_ptr<TrajectorySamples2D> TrajectorySampler::collect() {
  const size_t P = numPointsPerTrajectory;
  auto out = std::make_unique<TrajectorySamples2D>(numTrajectories, P);
  size_t rows = 0;
  hip::check(kc_dwa_get_samples(ctx_.get(), nullptr, nullptr, nullptr, nullptr, 0, &rows));
  if (rows > numTrajectories)
    throw std::out_of_range("more admissible samples than numTrajectories");
  std::vector<int32_t> raw(rows ? rows : 1);
  hip::check(kc_dwa_get_samples(ctx_.get(), out->paths.x.data(), out->paths.y.data(), raw.data(),
                                nullptr, rows, &rows));
  out->paths.z.fill(0.0f);
  // drop_samples = false: a frozen sample's profile is zero from its freeze step on (trajectory_sampler.cpp:160-163)
  std::vector<int32_t> frozen(rows ? rows : 1, 0);
  if (!drop_samples_) hip::check(kc_dwa_get_freeze_steps(ctx_.get(), frozen.data(), rows, &rows));
  for (size_t r = 0; r < rows; ++r) {
    const size_t g = static_cast<size_t>(raw[r]);
    const size_t stop = frozen[r] > 0 ? static_cast<size_t>(frozen[r]) : P;
    for (size_t i = 0; i + 1 < P; ++i) {  // TrajectoryVelocities2D::add: float = double
      const bool z = i >= stop;
      out->velocities.vx((Eigen::Index)r, (Eigen::Index)i) = z ? 0.0f : static_cast<float>(last_vx_[g]);
      out->velocities.vy((Eigen::Index)r, (Eigen::Index)i) = z ? 0.0f : static_cast<float>(last_vy_[g]);
      out->velocities.omega((Eigen::Index)r, (Eigen::Index)i) = z ? 0.0f : static_cast<float>(last_omega_[g]);
    }
  }
  out->paths.pathIndex_ = static_cast<Eigen::Index>(rows) - 1;
  out->velocities.velocitiesIndex_ = static_cast<Eigen::Index>(rows) - 1;
  return out;
}

std::unique_ptr<TrajectorySamples2D>
TrajectorySampler::generateTrajectories(const Velocity2D &vel, const Path::State &pose,
                                        const LaserScan &scan) {
  rolloutOnDevice(vel, pose, scan, checker_->maxSensorRange);
  return collect();
}
std::unique_ptr<TrajectorySamples2D>
TrajectorySampler::generateTrajectories(const Velocity2D &vel, const Path::State &pose,
                                        const std::vector<Path::Point> &cloud) {
  rolloutOnDevice(vel, pose, cloud, checker_->maxSensorRange);
  return collect();
}

std::unique_ptr<TrajectorySamples2D>
TrajectorySampler::generateTrajectories(const Velocity2D &vel, const Path::State &pose,
                                        const Mapping::LocalMapper &mapper) {
  rolloutOnDevice(vel, pose, mapper, checker_->maxSensorRange);
  return collect();
}

// single un-checked sample (trajectory_sampler.cpp:409-445): one trajectory,
// used only by the rotate-in-place shortcut -- not batch work
Trajectory2D TrajectorySampler::generateSingleSampleFromVel(const Velocity2D &vel,
                                                            const Path::State &pose) {
  Path::State s = pose;
  Trajectory2D t(numPointsPerTrajectory);
  t.path.add(0, s.x, s.y);
  const bool rotate_then_move = drive_ == ControlType::DIFFERENTIAL_DRIVE;
  for (size_t i = 0; i + 1 < numPointsPerTrajectory; ++i) {
    if (rotate_then_move && std::abs(vel.vx()) > MIN_VEL && std::abs(vel.omega()) > MIN_VEL) {
      Velocity2D tmp = vel;
      tmp.setVx(0.0);
      s.update(tmp, time_step_);
      t.path.add(i + 1, s.x, s.y);
      t.velocities.add(i, vel);
      tmp.setVx(vel.vx());
      tmp.setOmega(0.0);
      s.update(tmp, time_step_);
      t.path.add(i + 1, s.x, s.y);
      t.velocities.add(i, vel);
      i++;
      if (i + 1 >= numPointsPerTrajectory) break;
    }
    s.update(vel, time_step_);
    t.path.add(i + 1, s.x, s.y);
    t.velocities.add(i, vel);
  }
  return t;
}

}  // namespace Control
}  // namespace Kompass
