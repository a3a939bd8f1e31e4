// Weighted trajectory cost evaluator of the kompass_cpp surface (reference:
// utils/cost_evaluator.{h,cpp} CPU semantics; the reference's SYCL variant,
// cost_evaluator_gpu.cpp, is the behavioural precedent for running it on a
// device).  All five built-in costs + argmin run in HIP kernels; custom host
// callbacks are added on the host afterwards, like the reference's GPU build.
#pragma once

#include <array>
#include <functional>
#include <memory>
#include <vector>

#include "datatypes/control.h"
#include "datatypes/parameter.h"
#include "datatypes/path.h"
#include "datatypes/trajectory.h"
#include "utils/hip_backend.h"

namespace Kompass {
namespace Control {

class CostEvaluator {
 public:
  class TrajectoryCostsWeights : public Parameters {
   public:
    TrajectoryCostsWeights() : Parameters() {
      addParameter("reference_path_distance_weight", Parameter(1.0, 0.0, 1000.0,
                   "Weight of the cost for the distance between a trajectory sample and the reference global path"));
      addParameter("goal_distance_weight", Parameter(1.0, 0.0, 1000.0,
                   "Weight of the cost for the distance between the end of a trajectory sample and the end goal point"));
      addParameter("obstacles_distance_weight", Parameter(1.0, 0.0, 1000.0,
                   "Weight of the cost for the distance between a trajectory sample and the closest obstacle"));
      addParameter("smoothness_weight", Parameter(1.0, 0.0, 1000.0,
                   "Weight of the cost for the non-smoothness of the trajectory sample"));
      addParameter("jerk_weight", Parameter(1.0, 0.0, 1000.0,
                   "Weight of the cost for the trajectory sample jerk"));
    }
  };

  CostEvaluator(TrajectoryCostsWeights &costsWeights, ControlLimitsParams ctrLimits,
                size_t maxNumTrajectories, size_t numPointsPerTrajectory,
                size_t maxRefPathSegmentSize);
  CostEvaluator(TrajectoryCostsWeights &costsWeights,
                const Eigen::Vector3f &sensor_position_body,
                const Eigen::Quaternionf &sensor_rotation_body,
                ControlLimitsParams ctrLimits, size_t maxNumTrajectories,
                size_t numPointsPerTrajectory, size_t maxRefPathSegmentSize);
  // shares the device context of a sampler (DWA)
  CostEvaluator(TrajectoryCostsWeights &costsWeights, hip::DwaHandle ctx);
  ~CostEvaluator();

  using CustomCostFunction =
      std::function<float(const Trajectory2D &, const Path::Path &)>;
  struct CustomTrajectoryCost {
    double weight;
    CustomCostFunction evaluator_;
    CustomTrajectoryCost(double w, CustomCostFunction f)
        : weight(w), evaluator_(std::move(f)) {}
  };

  TrajSearchResult
  getMinTrajectoryCost(const std::unique_ptr<TrajectorySamples2D> &trajs,
                       const Path::Path *reference_path,
                       const Path::Path::View &tracked_segment);

  // samples already rolled out on this evaluator's device context
  // (TrajectorySampler::rolloutOnDevice); nothing is copied to the host except
  // the winner row.  sample velocities are needed for the result record.
  TrajSearchResult getMinTrajectoryCostOnDevice(
      const Path::Path *reference_path, const Path::Path::View &tracked_segment,
      size_t numPointsPerTrajectory);

  // The whole DWA::findBestPath device part on this evaluator's context, whose
  // sensor data and sample lattice are resident (TrajectorySampler::
  // prepareOnDevice): tracked segment up, then ONE device cycle -- roll-out,
  // collision gate, costs, argmin in a single kernel launch when the tables fit
  // (kc_dwa_cycle) -- and the winner row from the pinned record.  With a
  // communicator: this context's shard of the lattice + one 8-byte all-reduce
  // (kc_dwa_cycle_sharded); the winner's path is then re-rolled on the host from
  // its velocity (same arithmetic, same libm) when another rank owns it.
  // No kernel of a cycle waits for the host (device trig; the host's libm table -- fallback -- is complete
  // before the launch): a failing cycle throws std::runtime_error, there is nothing to repeat.
  TrajSearchResult cycleOnDevice(const Path::Path *reference_path, const Path::Path::View &tracked_segment,
                                 size_t numPointsPerTrajectory, const Path::State &pose, double time_step,
                                 const std::function<Velocity2D(size_t)> &sampleVelocity, size_t n_generated,
                                 kc_comm *comm = nullptr);
  // tracked-segment tables from a device-resident copy of the path (the window moves, a kernel
  // builds the tables) instead of host-built tables; default from KOMPASS_RESIDENT_PATH=1
  void useResidentPath(bool on) { residentPath_ = on; }

  void addCustomCost(double weight, CustomCostFunction custom_cost_function) {
    customTrajCostsPtrs_.push_back(std::make_unique<CustomTrajectoryCost>(
        weight, std::move(custom_cost_function)));
  }

  void setPointScan(const LaserScan &scan, const Path::State &current_state,
                    const float max_sensor_range,
                    const float max_obstacle_cost_range_multiple = 3.0);
  void setPointScan(const std::vector<Path::Point> &cloud,
                    const Path::State &current_state,
                    const float max_sensor_range,
                    const float max_obstacle_cost_range_multiple = 3.0);
  void updateCostWeights(TrajectoryCostsWeights &costsWeights);
  bool hasCustomCosts() const { return !customTrajCostsPtrs_.empty(); }
  const hip::DwaHandle &context() const { return ctx_; }
  // set by a sampler that already uploaded this cycle's sensor data to the
  // shared context (setPointScan then has nothing left to do)
  bool sensorDataResident = false;

 protected:
  std::array<float, 3> accLimits_;
  std::vector<std::unique_ptr<CustomTrajectoryCost>> customTrajCostsPtrs_;

 private:
  void uploadSegment(const Path::Path *reference_path,
                     const Path::Path::View &tracked_segment);
  // (raw_out: the GLOBAL raw index of the returned sample -- what a sharded cycle hands into its exchange)
  TrajSearchResult finishWithCustomCosts(const Path::Path *reference_path, size_t P, int64_t *raw_out = nullptr);
  std::unique_ptr<TrajectoryCostsWeights> costWeights;
  hip::DwaHandle ctx_;
  unsigned long long residentSerial_ = 0;  // Path::serial() of the path resident on the device
  bool residentPath_ = false;
};

}  // namespace Control
}  // namespace Kompass
