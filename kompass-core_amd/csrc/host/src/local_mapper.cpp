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
                         const float orient, const bool isPointCloud, const int scanSize,
                         const float angleStep, const float maxHeight, const float minHeight, const float rangeMax,
                         const int maxPointsPerLine, const int)
    : m_gridHeight(H), m_gridWidth(W), m_resolution(res), m_laserscanOrientation(orient),
      m_rangeMax(rangeMax), m_maxPointsPerLine(maxPointsPerLine), m_laserscanPosition(pos),
      m_scanSize(scanSize), m_maxHeight(maxHeight), m_minHeight(minHeight), gridData(H, W),
      ctx_(makeMapper(H, W, res, pos, orient, scanSize)) {
  m_angleStep = angleStep;
  if (isPointCloud) {
    // local_mapper.h:38-56: the angle step is derived from the scan size so
    // that binning and ray casting see the same grid
    const double derived_step = (2.0 * M_PI) / static_cast<double>(scanSize);
    initializedAngles.resize(std::max(scanSize, 0));
    initializedRanges.resize(std::max(scanSize, 0));
    for (int i = 0; i < scanSize; ++i) initializedAngles[i] = i * derived_step;
  }
}

// local_mapper.h:58-103: the Bayesian ctor; the first ctor's model parameters
// (:22-24) are p_prior 0.5, p_empty 0.4, p_occupied 0.6, range_sure 1.0,
// wall_size 0.2 -- see enableBayes()
LocalMapper::LocalMapper(const int H, const int W, const float res, const Eigen::Vector3f &pos,
                         const float orient, const bool isPointCloud, const int scanSize,
                         const float pPrior, const float pOccupied, const float pEmpty,
                         const float rangeSure, const float rangeMax, const float wallSize,
                         const float angleStep, const float maxHeight, const float minHeight,
                         const int maxPointsPerLine, const int maxNumThreads)
    : LocalMapper(H, W, res, pos, orient, isPointCloud, scanSize, angleStep, maxHeight, minHeight,
                  rangeMax, maxPointsPerLine, maxNumThreads) {
  m_pPrior = pPrior;
  m_pOccupied = pOccupied;
  m_pEmpty = pEmpty;
  m_rangeSure = rangeSure;
  m_wallSize = wallSize;
}

void LocalMapper::enableBayes() {
  if (bayesEnabled_) return;
  const kc_bayes_params p{m_pPrior, m_pOccupied, m_pEmpty, m_rangeSure, m_rangeMax, m_wallSize};
  hip::check(kc_mapper_enable_bayes(ctx_.get(), &p));
  gridDataProb = Eigen::MatrixXf(m_gridHeight, m_gridWidth);
  bayesEnabled_ = true;
}

Eigen::MatrixXi &LocalMapper::scanToGrid(const std::vector<double> &angles,
                                         const std::vector<double> &ranges) {
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid(ctx_.get(), angles.data(), ranges.data(), n, gridData.data()));
  return gridData;
}

void LocalMapper::scanToGridOnDevice(const std::vector<double> &angles,
                                     const std::vector<double> &ranges) {
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid_device(ctx_.get(), angles.data(), ranges.data(), n));
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

// local_mapper.cpp:222-241 (single-thread order: the last beam that crosses a
// cell decides its probability)
std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
LocalMapper::scanToGridBaysian(const std::vector<double> &angles, const std::vector<double> &ranges) {
  enableBayes();
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid_bayes(ctx_.get(), angles.data(), ranges.data(), n, gridData.data(),
                                          gridDataProb.data()));
  return std::tie(gridData, gridDataProb);
}

// local_mapper.cpp:253-269
std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
LocalMapper::scanToGridBaysian(const std::vector<int8_t> &data, int point_step, int row_step,
                               int height, int width, float x_offset, float y_offset,
                               float z_offset) {
  std::vector<double> angles, ranges;
  pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, static_cast<int>(x_offset),
                               static_cast<int>(y_offset), static_cast<int>(z_offset), m_rangeMax,
                               m_minHeight, m_maxHeight, m_angleStep, ranges, angles);
  return scanToGridBaysian(angles, ranges);
}

// local_mapper.cpp:17-78: the previous probability grid is warped in place, on
// the device
void LocalMapper::getPreviousGridInCurrentPose(const Eigen::Vector2f &currentPositionInPreviousPose,
                                               double currentOrientationInPreviousPose) {
  enableBayes();
  const float p[2] = {currentPositionInPreviousPose(0), currentPositionInPreviousPose(1)};
  hip::check(kc_mapper_warp_previous(ctx_.get(), p, currentOrientationInPreviousPose));
}

Eigen::MatrixXf &LocalMapper::previousGridProb() {
  enableBayes();
  previousGridDataProb = Eigen::MatrixXf(m_gridHeight, m_gridWidth);
  hip::check(kc_mapper_get_previous_prob(ctx_.get(), previousGridDataProb.data()));
  return previousGridDataProb;
}

void LocalMapper::setPreviousGridProb(const Eigen::MatrixXf *prob) {
  enableBayes();
  if (prob && (prob->rows() != m_gridHeight || prob->cols() != m_gridWidth))
    throw std::invalid_argument("LocalMapper::setPreviousGridProb: grid must be grid_height x grid_width");
  hip::check(kc_mapper_set_previous_prob(ctx_.get(), prob ? prob->data() : nullptr));
}

}  // namespace Mapping
}  // namespace Kompass
