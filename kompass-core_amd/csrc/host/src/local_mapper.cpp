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
    : rows_(H), cols_(W), cell_(res), sensor_yaw_(orient),
      range_cap_(rangeMax), line_cap_(maxPointsPerLine), sensor_at_(pos),
      bins_(scanSize), z_hi_(maxHeight), z_lo_(minHeight), cells_(H, W),
      ctx_(makeMapper(H, W, res, pos, orient, scanSize)) {
  bin_step_ = angleStep;
  if (isPointCloud) {
    // local_mapper.h:38-56: the angle step is derived from the scan size so
    // that binning and ray casting see the same grid
    const double derived_step = (2.0 * M_PI) / static_cast<double>(scanSize);
    bin_angles_.resize(std::max(scanSize, 0));
    bin_ranges_.resize(std::max(scanSize, 0));
    for (int i = 0; i < scanSize; ++i) bin_angles_[i] = i * derived_step;
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
  p_prior_ = pPrior;
  p_hit_ = pOccupied;
  p_free_ = pEmpty;
  sure_range_ = rangeSure;
  wall_ = wallSize;
}

void LocalMapper::turnOnBayes() {
  if (bayes_on_) return;
  const kc_bayes_params p{p_prior_, p_hit_, p_free_, sure_range_, range_cap_, wall_};
  hip::check(kc_mapper_enable_bayes(ctx_.get(), &p));
  belief_ = Eigen::MatrixXf(rows_, cols_);
  bayes_on_ = true;
}

Eigen::MatrixXi &LocalMapper::scanToGrid(const std::vector<double> &angles,
                                         const std::vector<double> &ranges) {
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid(ctx_.get(), angles.data(), ranges.data(), n, cells_.data()));
  return cells_;
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
  if (bin_angles_.empty())
    throw std::runtime_error("LocalMapper::scanToGrid(raw point cloud): not constructed with is_pointcloud");
  pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, static_cast<int>(x_offset),
                               static_cast<int>(y_offset), static_cast<int>(z_offset), range_cap_,
                               z_lo_, z_hi_, bins_, bin_ranges_);
  return scanToGrid(bin_angles_, bin_ranges_);
}

// local_mapper.cpp:222-241 (single-thread order: the last beam that crosses a
// cell decides its probability)
std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
LocalMapper::scanToGridBaysian(const std::vector<double> &angles, const std::vector<double> &ranges) {
  turnOnBayes();
  const size_t n = std::min(angles.size(), ranges.size());
  hip::check(kc_mapper_scan_to_grid_bayes(ctx_.get(), angles.data(), ranges.data(), n, cells_.data(),
                                          belief_.data()));
  return std::tie(cells_, belief_);
}

// local_mapper.cpp:253-269
std::tuple<Eigen::MatrixXi &, Eigen::MatrixXf &>
LocalMapper::scanToGridBaysian(const std::vector<int8_t> &data, int point_step, int row_step,
                               int height, int width, float x_offset, float y_offset,
                               float z_offset) {
  std::vector<double> angles, ranges;
  pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, static_cast<int>(x_offset),
                               static_cast<int>(y_offset), static_cast<int>(z_offset), range_cap_,
                               z_lo_, z_hi_, bin_step_, ranges, angles);
  return scanToGridBaysian(angles, ranges);
}

// local_mapper.cpp:17-78: the previous probability grid is warped in place, on
// the device
void LocalMapper::getPreviousGridInCurrentPose(const Eigen::Vector2f &currentPositionInPreviousPose,
                                               double currentOrientationInPreviousPose) {
  turnOnBayes();
  const float p[2] = {currentPositionInPreviousPose(0), currentPositionInPreviousPose(1)};
  hip::check(kc_mapper_warp_previous(ctx_.get(), p, currentOrientationInPreviousPose));
}

Eigen::MatrixXf &LocalMapper::previousGridProb() {
  turnOnBayes();
  prior_belief_ = Eigen::MatrixXf(rows_, cols_);
  hip::check(kc_mapper_get_previous_prob(ctx_.get(), prior_belief_.data()));
  return prior_belief_;
}

void LocalMapper::setPreviousGridProb(const Eigen::MatrixXf *prob) {
  turnOnBayes();
  if (prob && (prob->rows() != rows_ || prob->cols() != cols_))
    throw std::invalid_argument("LocalMapper::setPreviousGridProb: grid must be grid_height x grid_width");
  hip::check(kc_mapper_set_previous_prob(ctx_.get(), prob ? prob->data() : nullptr));
}

}  // namespace Mapping
}  // namespace Kompass
