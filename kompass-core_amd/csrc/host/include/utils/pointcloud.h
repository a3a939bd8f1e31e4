// Raw point cloud -> laserscan of the kompass_cpp surface (reference:
// utils/pointcloud.h:116-177, 205-259).  Same signatures; the binning runs on
// the device through the C ABI (kc_cloud_to_laserscan) and returns the same
// doubles as the reference's CPU loop.  PCD file I/O is out of scope.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "utils/hip_backend.h"

namespace Kompass {

namespace detail {
inline kc_cloud *sharedCloud() {
  static hip::CloudHandle ctx = hip::makeCloud(1 << 20, 4096);
  return ctx.get();
}
}  // namespace detail

// angle_step overload (pointcloud.h:116-177)
inline void pointCloudToLaserScanFromRaw(
    const std::vector<int8_t> &data, const int point_step, const int row_step,
    const int height, const int width, const int x_offset, const int y_offset,
    const int z_offset, const double max_range, const double min_z,
    const double max_z, const double angle_step,
    std::vector<double> &ranges_out, std::vector<double> &angles_out) {
  const int num_bins = static_cast<int>(std::ceil(2.0 * M_PI / angle_step));
  if (!(angle_step > 0.0) || num_bins <= 0)
    throw std::invalid_argument("pointCloudToLaserScanFromRaw: angle_step must be positive");
  ranges_out.resize(num_bins);
  angles_out.resize(num_bins);
  size_t bins = 0;
  hip::check(kc_cloud_to_laserscan(detail::sharedCloud(), data.data(), data.size(), 0, point_step,
                                   row_step, height, width, x_offset, y_offset, z_offset, max_range,
                                   min_z, max_z, angle_step, 0, ranges_out.data(), angles_out.data(),
                                   ranges_out.size(), &bins));
}

// num_bins overload (pointcloud.h:205-259)
inline void pointCloudToLaserScanFromRaw(
    const std::vector<int8_t> &data, const int point_step, const int row_step,
    const int height, const int width, const int x_offset, const int y_offset,
    const int z_offset, const double max_range, const double min_z,
    const double max_z, const int num_bins, std::vector<double> &ranges_out) {
  if (num_bins <= 0) {  // ranges_out.assign(num_bins, max_range) of the reference
    ranges_out.clear();
    return;
  }
  ranges_out.resize(num_bins);
  size_t bins = 0;
  hip::check(kc_cloud_to_laserscan(detail::sharedCloud(), data.data(), data.size(), 0, point_step,
                                   row_step, height, width, x_offset, y_offset, z_offset, max_range,
                                   min_z, max_z, 0.0, num_bins, ranges_out.data(), nullptr,
                                   ranges_out.size(), &bins));
}

}  // namespace Kompass
