// Dynamic Window Approach local planner of the kompass_cpp surface (reference:
// controllers/dwa.{h,cpp}).  The host keeps target tracking, adaptive horizon
// and tracked-segment selection; sampling, roll-out, collision gate, costs and
// argmin run as one device cycle.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <memory>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "controllers/follower.h"
#include "datatypes/trajectory.h"
#include "utils/cost_evaluator.h"
#include "utils/logger.h"
#include "utils/trajectory_sampler.h"

namespace Kompass {
namespace Control {

class DWA : public Follower {
 public:
  DWA(ControlLimitsParams controlLimits, ControlType controlType, double timeStep,
      double predictionHorizon, double controlHorizon, int maxLinearSamples,
      int maxAngularSamples, const CollisionChecker::ShapeType robotShapeType,
      const std::vector<float> robotDimensions,
      const Eigen::Vector3f &sensor_position_body,
      const Eigen::Vector4f &sensor_rotation_body, const double octreeRes,
      CostEvaluator::TrajectoryCostsWeights costWeights, const int host_threads_ = 1);

  DWA(TrajectorySampler::TrajectorySamplerParameters config,
      ControlLimitsParams controlLimits, ControlType controlType,
      const CollisionChecker::ShapeType robotShapeType,
      const std::vector<float> robotDimensions,
      const Eigen::Vector3f &sensor_position_body,
      const Eigen::Vector4f &sensor_rotation_body,
      CostEvaluator::TrajectoryCostsWeights costWeights, const int host_threads_ = 1);
  ~DWA() = default;

  void configure(ControlLimitsParams controlLimits, ControlType controlType,
                 double timeStep, double predictionHorizon, double controlHorizon,
                 int maxLinearSamples, int maxAngularSamples,
                 const CollisionChecker::ShapeType robotShapeType,
                 const std::vector<float> robotDimensions,
                 const Eigen::Vector3f &sensor_position_body,
                 const Eigen::Vector4f &sensor_rotation_body, const double octreeRes,
                 CostEvaluator::TrajectoryCostsWeights costWeights,
                 const int host_threads_ = 1);
  void configure(TrajectorySampler::TrajectorySamplerParameters config,
                 ControlLimitsParams controlLimits, ControlType controlType,
                 const CollisionChecker::ShapeType robotShapeType,
                 const std::vector<float> robotDimensions,
                 const Eigen::Vector3f &sensor_position_body,
                 const Eigen::Vector4f &sensor_rotation_body,
                 CostEvaluator::TrajectoryCostsWeights costWeights,
                 const int host_threads_ = 1);

  void resetOctreeResolution(const double octreeRes);
  void setSensorMaxRange(const float max_range);
  void setCurrentState(const Path::State &position);
  void addCustomCost(double weight, CostEvaluator::CustomCostFunction custom_cost_function);

  template <typename T>
  Controller::Result computeVelocityCommand(const Velocity2D &global_vel, const T &scan_points) {
    TrajSearchResult r = findBestPath(global_vel, scan_points);
    Controller::Result out;
    if (r.isTrajFound) {
      out.status = Controller::Result::Status::COMMAND_FOUND;
      out.velocity_command = r.trajectory.velocities.getFront();
      command_ = out.velocity_command;
    } else {
      out.status = Controller::Result::Status::NO_COMMAND_POSSIBLE;
    }
    return out;
  }

  template <typename T>
  TrajSearchResult computeVelocityCommandsSet(const Velocity2D &global_vel, const T &scan_points) {
    TrajSearchResult r = findBestPath(global_vel, scan_points);
    if (r.isTrajFound) command_ = r.trajectory.velocities.getFront();
    return r;
  }

  // Multi-GPU: one DWA per process / GPU, all fed the same inputs; every
  // cycle each scores its share of the sample lattice -- dealt by trig row, so that a
  // rank evaluates 1 / world of the host's cos / sin table -- and ONE RCCL
  // all-reduce(min) of the exchange record (best key, error word, admissible bitmaps)
  // picks the winner: every rank returns the same command, or every rank fails the cycle.
  // unique_id: KC_COMM_ID_BYTES from kc_comm_unique_id() of one rank.  Collective.
  void enableSharding(int rank, int world, const uint8_t *unique_id, int device = 0, int mode = KC_SHARD_ROWS) {
    kc_comm *raw = nullptr;
    hip::check(kc_comm_create(rank, world, unique_id, device, &raw));
    adoptComm(raw, mode);
  }
  // rehearsal transport for ranks that share a GPU (kc_comm_create_shm)
  void enableShardingShm(int rank, int world, const std::string &name, int device = 0, int mode = KC_SHARD_ROWS) {
    kc_comm *raw = nullptr;
    hip::check(kc_comm_create_shm(rank, world, name.c_str(), device, &raw));
    adoptComm(raw, mode);
  }
  void disableSharding() {
    comm_.reset();
    hip::check(kc_dwa_set_shard_rule(trajCostEvaluator->context().get(), 0, 1, -1));
  }
  // see CostEvaluator::useResidentPath
  void useResidentPath(bool on) { trajCostEvaluator->useResidentPath(on); }

  std::tuple<MatrixXfR, MatrixXfR> getDebuggingSamples() const;
  Control::TrajectorySamples2D getDebuggingSamplesPure() const;

  template <typename T>
  void debugVelocitySearch(const Velocity2D &global_vel, const T &scan_points, const bool &drop_samples) {
    requirePath();
    aimAtTarget();
    trajSampler->setSampleDroppingMode(drop_samples);
    debuggingSamples_ = trajSampler->generateTrajectories(global_vel, pose_, scan_points);
  }

 protected:
  std::unique_ptr<TrajectorySampler> trajSampler;
  std::unique_ptr<CostEvaluator> trajCostEvaluator;

  template <typename T>
  TrajSearchResult findBestPath(const Velocity2D &global_vel, const T &scan_points) {
    requirePath();
    aimAtTarget();
    if (knob_.turn_in_place &&
        std::abs(on_.target->heading_error) > knob_.goal_yaw * 10.0) {
      LOG_DEBUG("Rotating In Place ...");
      auto trajectory = trajSampler->generateSingleSampleFromVel(Velocity2D(
          0.0, 0.0,
          -on_.target->heading_error * limits_.omegaParams.maxOmega / M_PI));
      return TrajSearchResult{trajectory, true, 0.0};
    }
    const auto T0 = std::chrono::steady_clock::now();
    adaptPredictionHorizonToCurvature();
    const auto T1 = std::chrono::steady_clock::now();
    // lattice + sensor data of this cycle onto the device ...
    const size_t generated =
        trajSampler->prepareOnDevice(global_vel, pose_, scan_points, maxLocalRange_);
    const auto T2 = std::chrono::steady_clock::now();
    if (generated == 0) return TrajSearchResult{Trajectory2D(), false, 0.0};
    // ... then ONE device cycle: roll-out + collision gate + costs + argmin against the
    // tracked segment (dwa.h:215-229 of the reference as a single kernel launch)
    trajCostEvaluator->sensorDataResident = true;
    auto tracked = findTrackedPathSegment();
    const auto T3 = std::chrono::steady_clock::now();
    TrajectorySampler *smp = trajSampler.get();
    auto rr = trajCostEvaluator->cycleOnDevice(
        on_.path.get(), tracked, trajSampler->numPointsPerTrajectory, pose_, trajSampler->timeStep(),
        [smp](size_t raw) { return smp->sampleVelocity(raw); }, generated, comm_.get());
    const auto T4 = std::chrono::steady_clock::now();
    static const bool dbg = std::getenv("KC_DEBUG_CLASS") != nullptr;
    if (dbg) {
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      std::fprintf(stderr, "[class] horizon %.1f | prepare %.1f | tracked %.1f | cycleOnDevice %.1f\n", us(T0, T1), us(T1, T2), us(T2, T3), us(T3, T4));
    }
    return rr;
  }

 private:
  double max_forward_distance_ = 0.0;
  int host_threads_;
  std::shared_ptr<kc_comm> comm_;
  void adoptComm(kc_comm *raw, int mode) {
    comm_ = std::shared_ptr<kc_comm>(raw, [](kc_comm *c) { kc_comm_destroy(c); });
    // the context keeps this rank's share of every lattice it is handed from now on
    hip::check(kc_dwa_set_shard_rule(trajCostEvaluator->context().get(), kc_comm_rank(raw), kc_comm_world(raw), mode));
  }
  std::unique_ptr<TrajectorySamples2D> debuggingSamples_ = nullptr;
  float maxLocalRange_ = 10.0;

  void requirePath() const {
    if (!on_.path)
      throw std::invalid_argument(
          "Pointer to global path is NULL. Cannot use DWA local planner "
          "without setting a global path");
  }
  Path::Path::View findTrackedPathSegment();
  void adaptPredictionHorizonToCurvature();
  void initJitCompile();
};

}  // namespace Control
}  // namespace Kompass
