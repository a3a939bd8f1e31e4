// laserscan -> occupancy LocalMapper of the kompass_cpp surface (reference:
// mapping/local_mapper.{h,cpp}).  scanToGrid runs on the device with the CPU
// mapper's semantics (same cells, bit for bit), the raw point-cloud overload
// bins the cloud on the device first (M5); scanToGridBaysian and
// getPreviousGridInCurrentPose (M3) run on the device as well, in the
// reference's single-thread beam order.
#pragma once

#include <cstdint>
#include <tuple>
#include <vector>

#include "kc_linalg.h"
#include "utils/hip_backend.h"

namespace Kompass {
namespace Mapping {

enum class OccupancyType { UNEXPLORED = -1, EMPTY = 0, OCCUPIED = 100 };

class LocalMapper {
 public:
  LocalMapper(const int gridHeight, const int gridWidth, const float resolution,
              const Eigen::Vector3f &laserscanPosition,
              const float laserscanOrientation, const bool isPointCloud,
              const int scanSize, const float angleStep, const float maxHeight,
              const float minHeight, const float rangeMax,
              const int maxPointsPerLine, const int maxNumThreads = 1);
  LocalMapper(const int gridHeight, const int gridWidth, const float resolution,
              const Eigen::Vector3f &laserscanPosition,
              const float laserscanOrientation, const bool isPointCloud,
              const int scanSize, const float pPrior, const float pOccupied,
              const float pEmpty, const float rangeSure, const float rangeMax,
              const float wallSize, const float angleStep, const float maxHeight,
              const float minHeight, const int maxPointsPerLine,
              const int maxNumThreads = 1);
  virtual ~LocalMapper() = default;

  Eigen::MatrixXi &scanToGrid(const std::vector<double> &angles,
                              const std::vector<double> &ranges);
  Eigen::MatrixXi &scanToGrid(const std::vector<int8_t> &data, int point_step,
                              int row_step, int height, int width,
                              float x_offset, float y_offset, float z_offset);
  std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
  scanToGridBaysian(const std::vector<double> &angles,
                    const std::vector<double> &ranges);
  std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
  scanToGridBaysian(const std::vector<int8_t> &data, int point_step, int row_step, int height,
                    int width, float x_offset, float y_offset, float z_offset);
  void getPreviousGridInCurrentPose(const Eigen::Vector2f &currentPositionInPreviousPose,
                                    double currentOrientationInPreviousPose);
  // not in the reference, whose previousGridDataProb is only ever warped
  // (local_mapper.cpp:77): read it back / replace it (nullptr = feed the last
  // scan's probabilities back, a device copy)
  Eigen::MatrixXf &previousGridProb();
  // the device context: lets the controller consume the last grid where it
  // lies (DWA::computeVelocityCommand(vel, mapper))
  kc_mapper *hipContext() const { return ctx_.get(); }
  // scan -> grid without the copy to the host (the grid of scanToGrid is NOT
  // refreshed by this call)
  void scanToGridOnDevice(const std::vector<double> &angles, const std::vector<double> &ranges);
  void setPreviousGridProb(const Eigen::MatrixXf *prob);

 protected:
  const int rows_, cols_;
  const float cell_, sensor_yaw_, range_cap_;
  const int line_cap_;
  const Eigen::Vector3f sensor_at_;
  const int bins_;
  const float z_hi_, z_lo_;
  float bin_step_ = 0.0f;
  // inverse sensor model, defaults of the first ctor (local_mapper.h:22-24)
  float p_prior_ = 0.5f, p_free_ = 0.4f, p_hit_ = 0.6f, sure_range_ = 1.0f, wall_ = 0.2f;
  // pointcloud mode (local_mapper.h:38-56): angles i * 2 pi / scanSize
  std::vector<double> bin_angles_, bin_ranges_;
  Eigen::MatrixXi cells_;
  Eigen::MatrixXf belief_, prior_belief_;
  hip::MapperHandle ctx_;
  bool bayes_on_ = false;
  void turnOnBayes();
};

}  // namespace Mapping
}  // namespace Kompass
