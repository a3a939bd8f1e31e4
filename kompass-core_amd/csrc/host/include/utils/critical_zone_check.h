// CriticalZoneChecker of the kompass_cpp surface (reference:
// utils/critical_zone_check.{h,cpp}; critical_zone_check_gpu.{h,cpp} for the
// device-named variant).  Preset and checks go through the C ABI (kc_zone_*);
// the results are those of the reference's CPU loop.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "kc_linalg.h"
#include "utils/collision_check.h"
#include "utils/hip_backend.h"

namespace Kompass {

enum class PointFieldType { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 };

class CriticalZoneChecker {
 public:
  enum class InputType { LASERSCAN, POINTCLOUD };

  CriticalZoneChecker(InputType input_type, const CollisionChecker::ShapeType robot_shape_type,
                      const std::vector<float> &robot_dimensions,
                      const Eigen::Vector3f &sensor_position_body,
                      const Eigen::Vector4f &sensor_rotation_body, const float critical_angle,
                      const float critical_distance, const float slowdown_distance,
                      const std::vector<double> &angles, const float min_height,
                      const float max_height, const float range_max)
      : input_type_(input_type) {
    const float pos[3] = {sensor_position_body(0), sensor_position_body(1), sensor_position_body(2)};
    const float rot[4] = {sensor_rotation_body(0), sensor_rotation_body(1), sensor_rotation_body(2),
                          sensor_rotation_body(3)};
    kc_zone *raw = nullptr;
    hip::check(kc_zone_create(static_cast<int>(robot_shape_type), robot_dimensions.data(),
                              static_cast<int>(robot_dimensions.size()), pos, rot, critical_angle,
                              critical_distance, slowdown_distance, angles.data(), angles.size(),
                              min_height, max_height, range_max, 0, &raw));
    ctx_.reset(raw, [](kc_zone *p) { kc_zone_destroy(p); });
  }
  virtual ~CriticalZoneChecker() = default;

  float check(const std::vector<double> &ranges, const bool forward) {
    float f = 1.0f;
    hip::check(kc_zone_check(ctx_.get(), ranges.data(), ranges.size(), forward ? 1 : 0, &f));
    return f;
  }
  float check(const std::vector<int8_t> &data, int point_step, int row_step, int height, int width,
              int x_offset, int y_offset, int z_offset, const bool forward) {
    float f = 1.0f;
    hip::check(kc_zone_check_cloud_typed(ctx_.get(), data.data(), data.size(), point_step, row_step, height,
                                         width, x_offset, y_offset, z_offset, static_cast<int>(field_type_),
                                         forward ? 1 : 0, &f));
    return f;
  }

 protected:
  PointFieldType field_type_ = PointFieldType::FLOAT32;
  InputType input_type_;
  std::shared_ptr<kc_zone> ctx_;
};

// critical_zone_check_gpu.h:36-53: same surface plus the datatype of the cloud's x / y / z fields, decoded as
// load_and_cast_val does (utils/pointcloud.h:49-87)
class CriticalZoneCheckerGPU : public CriticalZoneChecker {
 public:
  CriticalZoneCheckerGPU(InputType input_type, const CollisionChecker::ShapeType robot_shape_type,
                         const std::vector<float> &robot_dimensions,
                         const Eigen::Vector3f &sensor_position_body,
                         const Eigen::Vector4f &sensor_rotation_body, const float critical_angle,
                         const float critical_distance, const float slowdown_distance,
                         const std::vector<double> &angles, const float min_height,
                         const float max_height, const float range_max,
                         const PointFieldType cloud_field_type = PointFieldType::FLOAT32)
      : CriticalZoneChecker(input_type, robot_shape_type, robot_dimensions, sensor_position_body,
                            sensor_rotation_body, critical_angle, critical_distance,
                            slowdown_distance, angles, min_height, max_height, range_max) {
    field_type_ = cloud_field_type;
  }
};

}  // namespace Kompass
