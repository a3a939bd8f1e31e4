// LocalMapper host side (reference: src/mapping/local_mapper.cpp).
#include "mapping/local_mapper.h"

#include <cmath>

#include "utils/pointcloud.h"

#include <stdexcept>

namespace Kompass {
namespace Mapping {

namespace {
hip::MapperHandle makeMapper(int H, int W, float res, const Eigen::Vector3f &pos, float orient,
                             int scanSize) {
  kc_mapper *raw = nullptr;
  const float p[3] = {pos(0), pos(1), pos(2)};
  hip::check(kc_mapper_create(H, W, res, p, orient, static_cast<size_t>(std::max(scanSize, 1)), 0, &raw));
  return hip::MapperHandle(raw);
}
}  // namespace

LocalMapper::LocalMapper(const int H, const int W, const float res, const Eigen::Vector3f &pos,
                         const float orient, const bool isPointCloud, const int scanSize, const float,
                         const float maxHeight, const float minHeight, const float rangeMax,
                         const int maxPointsPerLine, const int)
    : m_gridHeight(H), m_gridWidth(W), m_resolution(res), m_laserscanOrientation(orient),
      m_rangeMax(rangeMax), m_maxPointsPerLine(maxPointsPerLine), m_laserscanPosition(pos),
      m_scanSize(scanSize), m_maxHeight(maxHeight), m_minHeight(minHeight), gridData(H, W),
      ctx_(makeMapper(H, W, res, pos, orient, scanSize)) {
  if (isPointCloud) {
    // local_mapper.h:38-56: the angle step is derived from the scan size so
    // that binning and ray casting see the same grid
    const double derived_step = (2.0 * M_PI) / static_cast<double>(scanSize);
    initializedAngles.resize(std::max(scanSize, 0));
    initializedRanges.resize(std::max(scanSize, 0));
    for (int i = 0; i < scanSize; ++i) initializedAngles[i] = i * derived_step;
  }
}

LocalMapper::LocalMapper(const int H, const int W, const float res, const Eigen::Vector3f &pos,
                         const float orient, const bool isPointCloud, const int scanSize,
                         const float, const float, const float, const float, const float rangeMax,
                         const float, const float angleStep, const float maxHeight,
                         const float minHeight, const int maxPointsPerLine, const int maxNumThreads)
    : LocalMapper(H, W, res, pos, orient, isPointCloud, scanSize, angleStep, maxHeight, minHeight,
                  rangeMax, maxPointsPerLine, maxNumThreads) {}

Eigen::MatrixXi &LocalMapper::scanToGrid(const std::vector<double> &angles,
                                         const std::vector<double> &ranges) {
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid(ctx_.get(), angles.data(), ranges.data(), n, gridData.data()));
  return gridData;
}

// local_mapper.cpp:243-251
Eigen::MatrixXi &LocalMapper::scanToGrid(const std::vector<int8_t> &data, int point_step,
                                         int row_step, int height, int width, float x_offset,
                                         float y_offset, float z_offset) {
  if (initializedAngles.empty())
    throw std::runtime_error("LocalMapper::scanToGrid(raw point cloud): not constructed with is_pointcloud");
  pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, static_cast<int>(x_offset),
                               static_cast<int>(y_offset), static_cast<int>(z_offset), m_rangeMax,
                               m_minHeight, m_maxHeight, m_scanSize, initializedRanges);
  return scanToGrid(initializedAngles, initializedRanges);
}
std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
LocalMapper::scanToGridBaysian(const std::vector<double> &, const std::vector<double> &) {
  throw std::runtime_error(
      "LocalMapper::scanToGridBaysian: the Bayesian update is outside this build's scope "
      "(SURVEY.md 8a M3; unreachable from the reference's Python binding as well)");
}
void LocalMapper::getPreviousGridInCurrentPose(const Eigen::Vector2f &, double) {
  throw std::runtime_error(
      "LocalMapper::getPreviousGridInCurrentPose: outside this build's scope (SURVEY.md 8a M3)");
}

}  // namespace Mapping
}  // namespace Kompass
