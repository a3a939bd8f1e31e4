// Robot-shape vs sensor-data collision checker of the kompass_cpp surface
// (reference: utils/collision_check.{h,cpp}, FCL + octomap).  Here the octree
// is a voxel-key set and the shape test is analytic, both evaluated on the
// device (DESIGN.md "A4").  Same public methods, same ShapeType enum.
#pragma once

#include "mapping/local_mapper.h"

#include <memory>
#include <vector>

#include "datatypes/control.h"
#include "datatypes/path.h"
#include "kc_linalg.h"
#include "utils/hip_backend.h"

namespace Kompass {

class CollisionChecker {
 public:
  enum class ShapeType { CYLINDER, BOX, SPHERE };

  // body shape + where the sensor sits on it; `voxel` = edge of the occupancy voxels (the reference's octree resolution)
  CollisionChecker(const ShapeType shape, const std::vector<float> &dims, const Eigen::Vector3f &sensor_at,
                   const Eigen::Quaternionf &sensor_facing, const double voxel = 0.01);
  // shares the device context of a sampler / controller
  CollisionChecker(hip::DwaHandle ctx, ShapeType shape, const std::vector<float> &dims, double voxel);
  ~CollisionChecker() = default;

  void resetOctreeResolution(const double voxel);
  void updateState(const Path::State pose);
  void updateState(const double x, const double y, const double yaw);

  // LaserScan (sensor frame) or point list (world frame when global_frame)
  void updateSensorData(const Control::LaserScan &scan, const bool global_frame = true);
  void updateSensorData(const std::vector<Path::Point> &cloud, const bool global_frame = true);
  void updateSensorData(const Control::PointCloudView &cloud, const bool global_frame = true);
  // the OCCUPIED cells of the mapper's device-resident grid as the point list (no host round trip)
  void updateSensorData(const Mapping::LocalMapper &mapper, const bool global_frame = true);

  bool checkCollisions();
  bool checkCollisions(const Path::State pose);
  bool checkCollisions(const std::vector<double> &ranges, const std::vector<double> &angles, double height = 0.1);
  // batch form used by TrajectorySampler::checkStatesFeasibility
  std::vector<bool> checkCollisions(const std::vector<Path::State> &poses);
  float getRadius() const;
  const hip::DwaHandle &context() const { return ctx_; }
  float maxSensorRange = 10.0f;  // forwarded to the obstacle-cost side

 protected:
  double body_height_{1.0}, body_radius_{0.0};

 private:
  hip::DwaHandle ctx_;
  Path::State pose_;
  double voxel_{0.01};
};

}  // namespace Kompass
