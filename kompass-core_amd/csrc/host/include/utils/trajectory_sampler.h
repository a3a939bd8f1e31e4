// Velocity-space trajectory sampler of the kompass_cpp surface (reference:
// utils/trajectory_sampler.{h,cpp}).  The dynamic window and the sample
// lattice are built on the host in the reference's order; roll-out and the
// collision gate run on the device through the C ABI.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <memory>
#include <vector>

#include "datatypes/control.h"
#include "datatypes/parameter.h"
#include "datatypes/path.h"
#include "datatypes/trajectory.h"
#include "utils/collision_check.h"

#ifndef MIN_VEL
#define MIN_VEL 0.01
#endif

namespace Kompass {
namespace Control {

class TrajectorySampler {
 public:
  class TrajectorySamplerParameters : public Parameters {
   public:
    TrajectorySamplerParameters() : Parameters() {
      addParameter("time_step", Parameter(0.1, 0.001, 1000.0,
                   "Time step in the trajectory points/control generation [sec]"));
      addParameter("prediction_horizon", Parameter(1.0, 0.001, 1000.0,
                   "Future time horizon for the trajectory sampling prediction [sec]"));
      addParameter("control_horizon", Parameter(1.0, 0.001, 1000.0,
                   "Future time horizon for applying the control [sec]"));
      addParameter("max_linear_samples", Parameter(10, 1, 1000,
                   "Maximum number of samples for the linear velocity controls"));
      addParameter("max_angular_samples", Parameter(10, 1, 1000,
                   "Maximum number of samples for the angular velocity controls"));
      addParameter("octree_map_resolution", Parameter(0.1, 0.0, 1000.0,
                   "Resolution of the built-in Octree map used for collision checkings [m]"));
      addParameter("drop_samples", Parameter(true,
                   "Drops the samples with collisions (the only mode of this build)"));
    }
  };

  TrajectorySampler(ControlLimitsParams controlLimits, ControlType controlType,
                    double timeStep, double predictionHorizon,
                    double controlHorizon, int maxLinearSamples,
                    int maxAngularSamples,
                    const CollisionChecker::ShapeType robotShapeType,
                    const std::vector<float> robotDimensions,
                    const Eigen::Vector3f &sensor_position_body,
                    const Eigen::Quaternionf &sensor_rotation_body,
                    const double octreeRes, const int maxNumThreads = 1);

  TrajectorySampler(TrajectorySamplerParameters config,
                    ControlLimitsParams controlLimits, ControlType controlType,
                    const CollisionChecker::ShapeType robotShapeType,
                    const std::vector<float> robotDimensions,
                    const Eigen::Vector3f &sensor_position_body,
                    const Eigen::Quaternionf &sensor_rotation_body,
                    const int maxNumThreads = 1);
  ~TrajectorySampler() = default;

  void updateState(const Path::State &current_state);
  void setSampleDroppingMode(const bool drop_samples);

  std::unique_ptr<TrajectorySamples2D>
  generateTrajectories(const Velocity2D &current_vel,
                       const Path::State &current_pose, const LaserScan &scan);
  std::unique_ptr<TrajectorySamples2D>
  generateTrajectories(const Velocity2D &current_vel,
                       const Path::State &current_pose,
                       const std::vector<Path::Point> &cloud);

  std::unique_ptr<TrajectorySamples2D>
  generateTrajectories(const Velocity2D &current_vel, const Path::State &current_pose,
                       const Mapping::LocalMapper &mapper);

  void resetOctreeResolution(const double resolution);
  float getRobotRadius() const;
  Trajectory2D generateSingleSampleFromVel(const Velocity2D &vel,
                                           const Path::State &pose = Path::State());
  template <typename T>
  bool checkStatesFeasibility(const std::vector<Path::State> &states,
                              const T &sensor_points) {
    checker_->updateSensorData(sensor_points);
    for (bool hit : checker_->checkCollisions(states))
      if (hit) return true;
    return false;
  }
  void setPredictionHorizon(double horizon);
  double getBasePredictionHorizon() const { return base_max_time_; }

  size_t numTrajectories;
  size_t numPointsPerTrajectory;

  // ---- device-resident fast path (used by DWA; no host materialisation) ----
  // A1 + sensor upload + roll-out/collision launch; returns samples generated
  size_t rolloutOnDevice(const Velocity2D &current_vel, const Path::State &pose,
                         const LaserScan &scan, float max_sensor_range);
  size_t rolloutOnDevice(const Velocity2D &current_vel, const Path::State &pose,
                         const std::vector<Path::Point> &cloud,
                         float max_sensor_range);
  size_t rolloutOnDevice(const Velocity2D &current_vel, const Path::State &pose,
                         const Mapping::LocalMapper &mapper, float max_sensor_range);
  // the same without the roll-out launch: sensor data + lattice of this cycle are
  // resident afterwards, CostEvaluator::cycleOnDevice runs the whole cycle (one launch)
  template <typename T>
  size_t prepareOnDevice(const Velocity2D &current_vel, const Path::State &pose, const T &sensor_points,
                         float max_sensor_range) {
    checker_->maxSensorRange = max_sensor_range;
    // the lattice first: its upload needs an idle stream (the list is written over the BAR under the
    // kernels' feet otherwise) -- behind the sensor update it would wait for the device-side sensor build
    const size_t n = sampleWindow(current_vel, /*host_copy=*/false);
    checker_->updateState(pose);
    checker_->updateSensorData(sensor_points);
    return n;
  }
  // velocity triple of generated sample `raw` of the last window (host copy of the lattice)
  Velocity2D sampleVelocity(size_t raw) const {
    double vx = 0.0, vy = 0.0, om = 0.0;
    hip::check(kc_dwa_get_sample_velocity(ctx_.get(), static_cast<int64_t>(raw), &vx, &vy, &om));
    return Velocity2D(vx, vy, om);
  }
  double timeStep() const { return time_step_; }
  const hip::DwaHandle &context() const { return ctx_; }
  ControlType controlType() const { return drive_; }
  const ControlLimitsParams &limits() const { return limits_; }

 protected:
  ControlType drive_;
  ControlLimitsParams limits_;
  std::unique_ptr<CollisionChecker> checker_;
  int host_threads_;

 private:
  void init(const CollisionChecker::ShapeType shape,
            const std::vector<float> &dims, const Eigen::Vector3f &spos,
            const Eigen::Quaternionf &srot, double octreeRes);
  size_t sampleWindow(const Velocity2D &current_vel, bool host_copy = true);
  size_t launch(const Velocity2D &current_vel, const Path::State &pose);
  std::unique_ptr<TrajectorySamples2D> collect();
  hip::DwaHandle ctx_;
  double time_step_{0.0}, max_time_{0.0}, base_max_time_{0.0}, control_time_{0.0};
  int lin_samples_max_{0}, ang_samples_max_raw_{0};
  bool drop_samples_{true};
  std::vector<double> last_vx_, last_vy_, last_omega_;
};

}  // namespace Control
}  // namespace Kompass
