// LocalMapper host side (reference: src/mapping/local_mapper.cpp).
#include "mapping/local_mapper.h"

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
                         const float orient, const bool, const int scanSize, const float,
                         const float, const float, const float rangeMax, const int maxPointsPerLine,
                         const int)
    : m_gridHeight(H), m_gridWidth(W), m_resolution(res), m_laserscanOrientation(orient),
      m_rangeMax(rangeMax), m_maxPointsPerLine(maxPointsPerLine), m_laserscanPosition(pos),
      m_scanSize(scanSize), gridData(H, W), ctx_(makeMapper(H, W, res, pos, orient, scanSize)) {}

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

Eigen::MatrixXi &LocalMapper::scanToGrid(const std::vector<int8_t> &, int, int, int, int, float,
                                         float, float) {
  throw std::runtime_error(
      "LocalMapper::scanToGrid(raw point cloud): the pointcloud -> laserscan step is outside "
      "this build's scope (SURVEY.md 8f rank 1)");
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
