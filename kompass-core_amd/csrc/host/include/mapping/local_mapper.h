// laserscan -> occupancy LocalMapper of the kompass_cpp surface (reference:
// mapping/local_mapper.{h,cpp}).  scanToGrid runs on the device with the CPU
// mapper's semantics (same cells, bit for bit), the raw point-cloud overload
// bins the cloud on the device first (M5); the Bayesian update is outside this
// build's scope (M3) and throws.
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
  void getPreviousGridInCurrentPose(const Eigen::Vector2f &currentPositionInPreviousPose,
                                    double currentOrientationInPreviousPose);

 protected:
  const int m_gridHeight, m_gridWidth;
  const float m_resolution, m_laserscanOrientation, m_rangeMax;
  const int m_maxPointsPerLine;
  const Eigen::Vector3f m_laserscanPosition;
  const int m_scanSize;
  const float m_maxHeight, m_minHeight;
  // pointcloud mode (local_mapper.h:38-56): angles i * 2 pi / scanSize
  std::vector<double> initializedAngles, initializedRanges;
  Eigen::MatrixXi gridData;
  hip::MapperHandle ctx_;
};

}  // namespace Mapping
}  // namespace Kompass
