// LocalMapperGPU of the kompass_cpp surface (reference: mapping/
// local_mapper_gpu.{h,cpp}, SYCL).  Same constructor; scanToGrid returns the
// CPU mapper's cells (the reference's SYCL kernel rasterises differently and
// truncates rays at max_points_per_line; DESIGN.md "M4 deltas").
#pragma once

#include "mapping/local_mapper.h"

namespace Kompass {
namespace Mapping {

class LocalMapperGPU : public LocalMapper {
 public:
  LocalMapperGPU(const int gridHeight, const int gridWidth, const float resolution,
                 const Eigen::Vector3f &laserscanPosition,
                 const float laserscanOrientation, const bool isPointCloud,
                 const int scanSize, const float angleStep, const float maxHeight,
                 const float minHeight, const float rangeMax,
                 const int maxPointsPerLine = 32)
      : LocalMapper(gridHeight, gridWidth, resolution, laserscanPosition,
                    laserscanOrientation, isPointCloud, scanSize, angleStep,
                    maxHeight, minHeight, rangeMax, maxPointsPerLine) {}
};

}  // namespace Mapping
}  // namespace Kompass
