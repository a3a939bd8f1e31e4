// Python module `kompass_cpp` (pybind11; the reference uses nanobind, which is
// not installable offline -- SURVEY.md 8b).  Re-exposes the subset of the
// reference module that kompass_core.control.dwa / mapping.local_mapper and
// the named tests use, with the same submodule layout, class names, argument
// names and defaults (reference: src/kompass_cpp/bindings/*.cpp).
#include <cstring>
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "controllers/dwa.h"
#include "mapping/local_mapper_gpu.h"
#include "utils/logger.h"
#include "utils/critical_zone_check.h"
#include "utils/pointcloud.h"

namespace py = pybind11;
using namespace Kompass;

namespace {

using FArr = py::array_t<float, py::array::c_style | py::array::forcecast>;

Eigen::Vector3f vec3(const py::object &o) {
  auto a = py::cast<std::vector<float>>(o);
  if (a.size() != 3) throw std::invalid_argument("expected 3 values");
  return Eigen::Vector3f(a[0], a[1], a[2]);
}
Eigen::Vector4f vec4(const py::object &o) {
  auto a = py::cast<std::vector<float>>(o);
  if (a.size() != 4) throw std::invalid_argument("expected 4 values");
  return Eigen::Vector4f(a[0], a[1], a[2], a[3]);
}
std::vector<Path::Point> points(const py::object &o) {
  // list of 3-sequences or an (N,3) array
  FArr a = py::cast<FArr>(o);
  if (a.ndim() != 2 || a.shape(1) != 3) throw std::invalid_argument("expected an (N, 3) array of points");
  std::vector<Path::Point> out(static_cast<size_t>(a.shape(0)));
  auto r = a.unchecked<2>();
  for (py::ssize_t i = 0; i < a.shape(0); ++i) out[(size_t)i] = Path::Point(r(i, 0), r(i, 1), r(i, 2));
  return out;
}
using DArr = py::array_t<double, py::array::c_style | py::array::forcecast>;
// a 1-D float64 array as the std::vector<double> the reference signatures take: one memcpy (the element-wise
// sequence caster costs ~70 ns per beam)
std::vector<double> dvec(const DArr &a) {
  if (a.ndim() != 1) throw std::invalid_argument("expected a 1-D array");
  return std::vector<double>(a.data(), a.data() + a.size());
}
py::array_t<float> view1(const Eigen::VectorXf &v, py::handle owner) {
  return py::array_t<float>({(py::ssize_t)v.size()}, {(py::ssize_t)sizeof(float)}, v.data(), owner);
}
FArr copy1(const Eigen::VectorXf &v) {
  FArr a(v.size());
  std::memcpy(a.mutable_data(), v.data(), sizeof(float) * (size_t)v.size());
  return a;
}
Eigen::VectorXf toVec(const FArr &a) { return Eigen::VectorXf(a.data(), (Eigen::Index)a.size()); }

void fromDict(Parameters &p, const py::dict &d) {
  for (auto item : d) {
    const std::string name = py::cast<std::string>(item.first);
    auto it = p.parameters.find(name);
    if (it == p.parameters.end()) continue;
    try {
      py::handle v = item.second;
      if (py::isinstance<py::bool_>(v)) it->second.setValue(py::cast<bool>(v));
      else if (py::isinstance<py::float_>(v)) it->second.setValue(py::cast<double>(v));
      else if (py::isinstance<py::str>(v)) it->second.setValue(py::cast<std::string>(v));
      else if (py::isinstance<py::int_>(v)) it->second.setValue(py::cast<int>(v));
    } catch (const std::exception &e) {
      throw std::runtime_error(e.what());
    }
  }
}

}  // namespace

PYBIND11_MODULE(kompass_cpp, m) {
  m.doc() = "Algorithms for robot path tracking and control (MI355X / HIP build of the hot path)";

  // ---------------------------------------------------------------- types
  auto t = m.def_submodule("types", "KOMPASS CPP data types module");
  py::enum_<Path::InterpolationType>(t, "PathInterpolationType")
      .value("LINEAR", Path::InterpolationType::LINEAR)
      .value("CUBIC_SPLINE", Path::InterpolationType::CUBIC_SPLINE)
      .value("HERMITE_SPLINE", Path::InterpolationType::HERMITE_SPLINE)
      .export_values();

  py::class_<Path::State>(t, "State")
      .def(py::init<double, double, double, double>(), py::arg("x") = 0.0, py::arg("y") = 0.0,
           py::arg("yaw") = 0.0, py::arg("speed") = 0.0)
      .def_readwrite("x", &Path::State::x)
      .def_readwrite("y", &Path::State::y)
      .def_readwrite("yaw", &Path::State::yaw)
      .def_readwrite("speed", &Path::State::speed);

  py::class_<Path::PathPosition>(t, "PathPosition")
      .def(py::init<>())
      .def_readwrite("segment_index", &Path::PathPosition::segment_index)
      .def_readwrite("segment_length", &Path::PathPosition::segment_length)
      .def_readwrite("parallel_distance", &Path::PathPosition::parallel_distance)
      .def_readwrite("normal_distance", &Path::PathPosition::normal_distance);

  py::class_<Path::Path>(t, "Path")
      .def(py::init([](const py::object &pts) { return Path::Path(points(pts)); }), py::arg("points"))
      .def("reached_end", &Path::Path::endReached)
      .def("get_total_length", &Path::Path::totalPathLength)
      .def("size", &Path::Path::getSize)
      .def("getIndex", [](const Path::Path &p, size_t i) {
             const Path::Point q = p.getIndex(i);
             FArr a(3);
             a.mutable_at(0) = q.x(); a.mutable_at(1) = q.y(); a.mutable_at(2) = q.z();
             return a;
           }, py::arg("index"))
      .def("x", [](const Path::Path &p) { return copy1(p.getX()); })
      .def("y", [](const Path::Path &p) { return copy1(p.getY()); });

  py::class_<Control::Velocity2D>(t, "Velocity2D")
      .def(py::init([](float vx, float vy, float omega, float steer) {
             return Control::Velocity2D(vx, vy, omega, steer);
           }), py::arg("vx") = 0.0, py::arg("vy") = 0.0, py::arg("omega") = 0.0, py::arg("steer_ang") = 0.0)
      .def_property("vx", &Control::Velocity2D::vx, &Control::Velocity2D::setVx)
      .def_property("vy", &Control::Velocity2D::vy, &Control::Velocity2D::setVy)
      .def_property("omega", &Control::Velocity2D::omega, &Control::Velocity2D::setOmega)
      .def_property("steer_ang", &Control::Velocity2D::steer_ang, &Control::Velocity2D::setSteerAng)
      .def("__str__", [](const Control::Velocity2D &v) {
        return "{" + std::to_string(v.vx()) + ", " + std::to_string(v.vy()) + ", " + std::to_string(v.omega()) + "})";
      });

  py::class_<Control::TrajectoryVelocities2D>(t, "TrajectoryVelocities2D")
      .def(py::init<>())
      .def(py::init([](size_t length) { return Control::TrajectoryVelocities2D(length + 1); }), py::arg("length"))
      .def(py::init<const std::vector<Control::Velocity2D> &>(), py::arg("velocities"))
      .def(py::init([](const FArr &vx, const FArr &vy, const FArr &om) {
             return Control::TrajectoryVelocities2D(toVec(vx), toVec(vy), toVec(om));
           }), py::arg("vx"), py::arg("vy"), py::arg("omega"))
      .def_property_readonly("vx", [](py::object self) {
             return view1(self.cast<const Control::TrajectoryVelocities2D &>().vx, self); })
      .def_property_readonly("vy", [](py::object self) {
             return view1(self.cast<const Control::TrajectoryVelocities2D &>().vy, self); })
      .def_property_readonly("omega", [](py::object self) {
             return view1(self.cast<const Control::TrajectoryVelocities2D &>().omega, self); })
      .def_property("length",
           [](const Control::TrajectoryVelocities2D &v) {
             return v.numPointsPerTrajectory_ > 0 ? v.numPointsPerTrajectory_ - 1 : 0; },
           [](Control::TrajectoryVelocities2D &v, size_t length) { v.numPointsPerTrajectory_ = length + 1; });

  py::class_<Control::TrajectoryPath>(t, "TrajectoryPath")
      .def(py::init<>())
      .def_property_readonly("x", [](py::object self) { return view1(self.cast<const Control::TrajectoryPath &>().x, self); })
      .def_property_readonly("y", [](py::object self) { return view1(self.cast<const Control::TrajectoryPath &>().y, self); })
      .def_property_readonly("z", [](py::object self) { return view1(self.cast<const Control::TrajectoryPath &>().z, self); });

  py::class_<Control::Trajectory2D>(t, "Trajectory")
      .def(py::init<>())
      .def_readonly("velocities", &Control::Trajectory2D::velocities)
      .def_readonly("path", &Control::Trajectory2D::path);

  py::class_<Control::LaserScan>(t, "LaserScan")
      // numpy arrays first (one memcpy each); lists and other sequences fall through to the element-wise form
      .def(py::init([](const py::array_t<double, py::array::c_style | py::array::forcecast> &ranges,
                       const py::array_t<double, py::array::c_style | py::array::forcecast> &angles) {
             if (ranges.ndim() != 1 || angles.ndim() != 1) throw std::invalid_argument("ranges and angles must be 1-D");
             return Control::LaserScan(std::vector<double>(ranges.data(), ranges.data() + ranges.size()),
                                       std::vector<double>(angles.data(), angles.data() + angles.size()));
           }), py::arg("ranges").noconvert(), py::arg("angles").noconvert())
      .def(py::init<std::vector<double>, std::vector<double>>(), py::arg("ranges"), py::arg("angles"))
      .def_readonly("ranges", &Control::LaserScan::ranges)
      .def_readonly("angles", &Control::LaserScan::angles);

  py::enum_<CollisionChecker::ShapeType>(t, "RobotGeometry")
      .value("CYLINDER", CollisionChecker::ShapeType::CYLINDER)
      .value("BOX", CollisionChecker::ShapeType::BOX)
      .value("SPHERE", CollisionChecker::ShapeType::SPHERE)
      .def_static("get", [](const std::string &key) {
        if (key == "CYLINDER") return CollisionChecker::ShapeType::CYLINDER;
        if (key == "BOX") return CollisionChecker::ShapeType::BOX;
        if (key == "SPHERE") return CollisionChecker::ShapeType::SPHERE;
        throw std::runtime_error("Invalid key");
      });

  // ------------------------------------------------------------ configure
  auto cfg = m.def_submodule("configure", "Configuration classes");
  py::class_<Parameters>(cfg, "ConfigParameters").def(py::init<>()).def("from_dict", &fromDict);

  // -------------------------------------------------------------- control
  auto c = m.def_submodule("control", "Control module");
  py::enum_<Control::ControlType>(c, "ControlType")
      .value("ACKERMANN", Control::ControlType::ACKERMANN)
      .value("DIFFERENTIAL_DRIVE", Control::ControlType::DIFFERENTIAL_DRIVE)
      .value("OMNI", Control::ControlType::OMNI);

  py::class_<Control::LinearVelocityControlParams>(c, "LinearVelocityControlParams")
      .def(py::init<double, double, double>(), py::arg("max_vel") = 0.0, py::arg("max_acc") = 0.0,
           py::arg("max_decel") = 0.0)
      .def_readwrite("max_vel", &Control::LinearVelocityControlParams::maxVel)
      .def_readwrite("max_acc", &Control::LinearVelocityControlParams::maxAcceleration)
      .def_readwrite("max_decel", &Control::LinearVelocityControlParams::maxDeceleration);

  py::class_<Control::AngularVelocityControlParams>(c, "AngularVelocityControlParams")
      .def(py::init<double, double, double, double>(), py::arg("max_ang") = M_PI, py::arg("max_omega") = 0.0,
           py::arg("max_acc") = 0.0, py::arg("max_decel") = 0.0)
      .def_readwrite("max_steer_ang", &Control::AngularVelocityControlParams::maxAngle)
      .def_readwrite("max_omega", &Control::AngularVelocityControlParams::maxOmega)
      .def_readwrite("max_acc", &Control::AngularVelocityControlParams::maxAcceleration)
      .def_readwrite("max_decel", &Control::AngularVelocityControlParams::maxDeceleration);

  py::class_<Control::ControlLimitsParams>(c, "ControlLimitsParams")
      .def(py::init<>())
      .def(py::init([](const Control::LinearVelocityControlParams &x, const Control::LinearVelocityControlParams &y,
                       const Control::AngularVelocityControlParams &w) {
             return Control::ControlLimitsParams(x, y, w);
           }), py::arg("vel_x_ctr_params") = Control::LinearVelocityControlParams(),
           py::arg("vel_y_ctr_params") = Control::LinearVelocityControlParams(),
           py::arg("omega_ctr_params") = Control::AngularVelocityControlParams())
      .def_readwrite("linear_x_limits", &Control::ControlLimitsParams::velXParams)
      .def_readwrite("linear_y_limits", &Control::ControlLimitsParams::velYParams)
      .def_readwrite("angular_limits", &Control::ControlLimitsParams::omegaParams);

  py::class_<Control::Controller>(c, "Controller")
      .def(py::init<>())
      .def("set_linear_ctr_limits", &Control::Controller::setLinearControlLimits)
      .def("set_angular_ctr_limits", &Control::Controller::setAngularControlLimits)
      .def("set_ctr_type", &Control::Controller::setControlType)
      .def("set_current_velocity", &Control::Controller::setCurrentVelocity)
      .def("set_current_state", py::overload_cast<const Path::State &>(&Control::Controller::setCurrentState))
      .def("set_current_state", py::overload_cast<double, double, double, double>(&Control::Controller::setCurrentState))
      .def("get_ctr_type", &Control::Controller::getControlType)
      .def("get_control", &Control::Controller::getControl);

  py::class_<Control::Controller::ControllerParameters, Parameters>(c, "ControllerParameters").def(py::init<>());
  py::class_<Control::Follower::FollowerParameters, Control::Controller::ControllerParameters>(c, "FollowerParameters")
      .def(py::init<>());

  py::class_<Control::Follower::Target>(c, "FollowingTarget")
      .def(py::init<>())
      .def_readwrite("segment_index", &Control::Follower::Target::segment_index)
      .def_readwrite("position_in_segment", &Control::Follower::Target::position_in_segment)
      .def_readwrite("movement", &Control::Follower::Target::movement)
      .def_readwrite("reverse", &Control::Follower::Target::reverse)
      .def_readwrite("lookahead", &Control::Follower::Target::lookahead)
      .def_readwrite("crosstrack_error", &Control::Follower::Target::crosstrack_error)
      .def_readwrite("heading_error", &Control::Follower::Target::heading_error);

  py::class_<Control::Follower, Control::Controller>(c, "Follower")
      .def(py::init<>())
      .def(py::init<Control::Follower::FollowerParameters>())
      .def("set_interpolation_type", &Control::Follower::setInterpolationType)
      .def("set_current_path", &Control::Follower::setCurrentPath, py::arg("path"), py::arg("interpolate") = true)
      .def("clear_current_path", &Control::Follower::clearCurrentPath)
      .def("is_goal_reached", &Control::Follower::isGoalReached)
      .def("get_vx_cmd", &Control::Follower::getLinearVelocityCmdX)
      .def("get_vy_cmd", &Control::Follower::getLinearVelocityCmdY)
      .def("get_omega_cmd", &Control::Follower::getAngularVelocityCmd)
      .def("get_steer_cmd", &Control::Follower::getSteeringAngleCmd)
      .def("get_tracked_target", &Control::Follower::getTrackedTarget)
      .def("get_current_path", &Control::Follower::getCurrentPath)
      .def("get_path_length", &Control::Follower::getPathLength)
      .def("has_path", &Control::Follower::hasPath);

  py::enum_<Control::Controller::Result::Status>(c, "FollowingStatus")
      .value("GOAL_REACHED", Control::Controller::Result::Status::GOAL_REACHED)
      .value("LOOSING_GOAL", Control::Controller::Result::Status::LOOSING_GOAL)
      .value("COMMAND_FOUND", Control::Controller::Result::Status::COMMAND_FOUND)
      .value("NO_COMMAND_POSSIBLE", Control::Controller::Result::Status::NO_COMMAND_POSSIBLE);
  py::class_<Control::Controller::Result>(c, "FollowingResult")
      .def(py::init<>())
      .def_readwrite("status", &Control::Controller::Result::status)
      .def_readwrite("velocity_command", &Control::Controller::Result::velocity_command);

  py::class_<Control::TrajSearchResult>(c, "SamplingControlResult")
      .def(py::init<>())
      .def_readwrite("is_found", &Control::TrajSearchResult::isTrajFound)
      .def_readwrite("cost", &Control::TrajSearchResult::trajCost)
      .def_readwrite("trajectory", &Control::TrajSearchResult::trajectory);

  py::class_<Control::CostEvaluator::TrajectoryCostsWeights, Parameters>(c, "TrajectoryCostWeights").def(py::init<>());
  py::class_<Control::TrajectorySampler::TrajectorySamplerParameters, Parameters>(c, "TrajectorySamplerParameters")
      .def(py::init<>());

  using DWA = Control::DWA;
  py::class_<DWA, Control::Follower>(c, "DWA")
      .def(py::init([](Control::ControlLimitsParams lim, Control::ControlType type, double dt, double ph, double ch,
                       int ml, int ma, CollisionChecker::ShapeType shape, std::vector<float> dims,
                       const py::object &spos, const py::object &srot, double res,
                       Control::CostEvaluator::TrajectoryCostsWeights w, int threads) {
             return std::make_unique<DWA>(lim, type, dt, ph, ch, ml, ma, shape, dims, vec3(spos), vec4(srot), res, w, threads);
           }), py::arg("control_limits"), py::arg("control_type"), py::arg("time_step"),
           py::arg("prediction_horizon"), py::arg("control_horizon"), py::arg("max_linear_samples"),
           py::arg("max_angular_samples"), py::arg("robot_shape_type"), py::arg("robot_dimensions"),
           py::arg("sensor_position_robot"), py::arg("sensor_rotation_robot"), py::arg("octree_resolution"),
           py::arg("cost_weights"), py::arg("max_num_threads") = 1)
      .def(py::init([](Control::TrajectorySampler::TrajectorySamplerParameters cfg, Control::ControlLimitsParams lim,
                       Control::ControlType type, CollisionChecker::ShapeType shape, std::vector<float> dims,
                       const py::object &spos, const py::object &srot,
                       Control::CostEvaluator::TrajectoryCostsWeights w, int threads) {
             return std::make_unique<DWA>(cfg, lim, type, shape, dims, vec3(spos), vec4(srot), w, threads);
           }), py::arg("config"), py::arg("control_limits"), py::arg("control_type"), py::arg("robot_shape_type"),
           py::arg("robot_dimensions"), py::arg("sensor_position_robot"), py::arg("sensor_rotation_robot"),
           py::arg("cost_weights"), py::arg("max_num_threads") = 1)
      .def("compute_velocity_commands", [](DWA &d, const Control::Velocity2D &v, const Control::LaserScan &s) {
             return d.computeVelocityCommandsSet<Control::LaserScan>(v, s); })
      // sensor data = the last grid of a LocalMapper, consumed where it lies on
      // the device (not in the reference: SURVEY 8f rank 4)
      .def("compute_velocity_commands", [](DWA &d, const Control::Velocity2D &v, const Mapping::LocalMapper &m) {
             return d.computeVelocityCommandsSet<Mapping::LocalMapper>(v, m); })
      .def("compute_velocity_commands", [](DWA &d, const Control::Velocity2D &v, const py::object &cloud) {
             // an (N, 3) float32 C-contiguous array is consumed where it lies; anything else (lists of
             // tuples, other dtypes) is converted first
             const FArr a = py::cast<FArr>(cloud);
             if (a.ndim() != 2 || a.shape(1) != 3) throw std::invalid_argument("expected an (N, 3) array of points");
             return d.computeVelocityCommandsSet<Control::PointCloudView>(
                 v, Control::PointCloudView{a.data(), static_cast<size_t>(a.shape(0))}); })
      .def("add_custom_cost", &DWA::addCustomCost)
      .def("get_debugging_samples", [](const DWA &d) {
             auto [x, y] = d.getDebuggingSamples();
             FArr ax({(py::ssize_t)x.rows(), (py::ssize_t)x.cols()}), ay({(py::ssize_t)y.rows(), (py::ssize_t)y.cols()});
             std::memcpy(ax.mutable_data(), x.data(), sizeof(float) * (size_t)x.size());
             std::memcpy(ay.mutable_data(), y.data(), sizeof(float) * (size_t)y.size());
             return py::make_tuple(ax, ay);
           })
      .def("debug_velocity_search", [](DWA &d, const Control::Velocity2D &v, const Control::LaserScan &s, bool drop) {
             d.debugVelocitySearch<Control::LaserScan>(v, s, drop); }, py::call_guard<py::gil_scoped_release>())
      .def("debug_velocity_search", [](DWA &d, const Control::Velocity2D &v, const py::object &cloud, bool drop) {
             auto pts = points(cloud);
             py::gil_scoped_release rel;
             d.debugVelocitySearch<std::vector<Path::Point>>(v, pts, drop); })
      .def("set_resolution", &DWA::resetOctreeResolution)
      .def("set_sensor_max_range", &DWA::setSensorMaxRange)
      // additions of this build (no counterpart in the reference: its DWA is one device)
      .def("enable_sharding", [](DWA &d, int rank, int world, const py::bytes &unique_id, int device, bool by_rows) {
             const std::string id = unique_id;
             if (id.size() != KC_COMM_ID_BYTES) throw std::invalid_argument("unique_id must be 128 bytes (comm_unique_id())");
             d.enableSharding(rank, world, reinterpret_cast<const uint8_t *>(id.data()), device,
                              by_rows ? KC_SHARD_ROWS : KC_SHARD_BLOCKS);
           }, py::arg("rank"), py::arg("world"), py::arg("unique_id"), py::arg("device") = 0, py::arg("by_rows") = true,
           "One DWA per process / GPU: shares of the sample lattice (dealt by trig row, or contiguous blocks) + ONE "
           "RCCL all-reduce(min) of the exchange record per cycle")
      .def("enable_sharding_shm", [](DWA &d, int rank, int world, const std::string &name, int device, bool by_rows) {
             d.enableShardingShm(rank, world, name, device, by_rows ? KC_SHARD_ROWS : KC_SHARD_BLOCKS);
           }, py::arg("rank"), py::arg("world"), py::arg("name"), py::arg("device") = 0, py::arg("by_rows") = true,
           "The same over the shared-memory rehearsal transport (ranks that share a GPU)")
      .def("disable_sharding", &DWA::disableSharding)
      .def("use_resident_path", &DWA::useResidentPath, py::arg("on"),
           "Tracked-segment tables from a device-resident copy of the path (saves host time, adds a kernel)");

  // -------------------------------------------------------------- mapping
  auto mp = m.def_submodule("mapping", "Local Mapping module");
  py::enum_<Mapping::OccupancyType>(mp, "OCCUPANCY_TYPE")
      .value("UNEXPLORED", Mapping::OccupancyType::UNEXPLORED)
      .value("EMPTY", Mapping::OccupancyType::EMPTY)
      .value("OCCUPIED", Mapping::OccupancyType::OCCUPIED);

  auto gridView = [](Eigen::MatrixXi &g, py::handle owner) {
    // column-major (H, W) view into the mapper's member matrix, like
    // nanobind's reference_internal on an Eigen::MatrixXi
    return py::array_t<int>({(py::ssize_t)g.rows(), (py::ssize_t)g.cols()},
                            {(py::ssize_t)sizeof(int), (py::ssize_t)(sizeof(int) * g.rows())}, g.data(), owner);
  };

  auto probView = [](Eigen::MatrixXf &g, py::handle owner) {
    return py::array_t<float>({(py::ssize_t)g.rows(), (py::ssize_t)g.cols()},
                              {(py::ssize_t)sizeof(float), (py::ssize_t)(sizeof(float) * g.rows())}, g.data(),
                              owner);
  };

  py::class_<Mapping::LocalMapper>(mp, "LocalMapper")
      .def(py::init([](int H, int W, float res, const py::object &pos, float orient, bool pc, int scan, float step,
                       float maxh, float minh, float rmax, int mppl, int threads) {
             return std::make_unique<Mapping::LocalMapper>(H, W, res, vec3(pos), orient, pc, scan, step, maxh, minh, rmax, mppl, threads);
           }), py::arg("grid_height"), py::arg("grid_width"), py::arg("resolution"), py::arg("laserscan_position"),
           py::arg("laserscan_orientation"), py::arg("is_pointcloud"), py::arg("scan_size"), py::arg("angle_step"),
           py::arg("max_height"), py::arg("min_height"), py::arg("range_max"), py::arg("max_points_per_line") = 32,
           py::arg("max_num_threads") = 1)
      .def(py::init([](int H, int W, float res, const py::object &pos, float orient, bool pc, int scan, float pp,
                       float po, float pe, float rs, float rmax, float wall, float step, float maxh, float minh,
                       int mppl, int threads) {
             return std::make_unique<Mapping::LocalMapper>(H, W, res, vec3(pos), orient, pc, scan, pp, po, pe, rs, rmax, wall, step, maxh, minh, mppl, threads);
           }), py::arg("grid_height"), py::arg("grid_width"), py::arg("resolution"), py::arg("laserscan_position"),
           py::arg("laserscan_orientation"), py::arg("is_pointcloud"), py::arg("scan_size"), py::arg("p_prior"),
           py::arg("p_occupied"), py::arg("p_empty"), py::arg("range_sure"), py::arg("range_max"), py::arg("wall_size"),
           py::arg("angle_step"), py::arg("max_height"), py::arg("min_height"), py::arg("max_points_per_line"),
           py::arg("max_num_threads") = 1)
      .def("scan_to_grid", [gridView](py::object self, const DArr &angles, const DArr &ranges) {
             return gridView(self.cast<Mapping::LocalMapper &>().scanToGrid(dvec(angles), dvec(ranges)), self);
           }, "Convert laser scan data to occupancy grid (float64 arrays: one copy each)", py::arg("angles").noconvert(),
           py::arg("ranges").noconvert())
      .def("scan_to_grid", [gridView](py::object self, const std::vector<double> &angles, const std::vector<double> &ranges) {
             return gridView(self.cast<Mapping::LocalMapper &>().scanToGrid(angles, ranges), self);
           }, "Convert laser scan data to occupancy grid", py::arg("angles"), py::arg("ranges"))
      .def("scan_to_grid_on_device", [](Mapping::LocalMapper &m, const DArr &angles, const DArr &ranges) {
             m.scanToGridOnDevice(dvec(angles), dvec(ranges));
           }, "Scan into the device-resident grid only (float64 arrays: one copy each)", py::arg("angles").noconvert(),
           py::arg("ranges").noconvert())
      .def("scan_to_grid_on_device", &Mapping::LocalMapper::scanToGridOnDevice,
           "Scan into the device-resident grid only (for DWA.compute_velocity_commands(vel, mapper))",
           py::arg("angles"), py::arg("ranges"))
      .def("scan_to_grid", [gridView](py::object self, const std::vector<int8_t> &data, int point_step, int row_step,
                                      int height, int width, float x_offset, float y_offset, float z_offset) {
             return gridView(self.cast<Mapping::LocalMapper &>().scanToGrid(data, point_step, row_step, height, width,
                                                                           x_offset, y_offset, z_offset), self);
           }, "Convert a raw point cloud to occupancy grid", py::arg("data"), py::arg("point_step"),
           py::arg("row_step"), py::arg("height"), py::arg("width"), py::arg("x_offset"), py::arg("y_offset"),
           py::arg("z_offset"))
      // The reference binds scan_to_grid_baysian to scanToGrid
      // (bindings_mapping.cpp:59-75) while its own Python caller unpacks two
      // grids (mapping/local_mapper.py:289-306): bound here to the real
      // scanToGridBaysian, which is what that caller needs.
      .def("scan_to_grid_baysian", [gridView, probView](py::object self, const DArr &angles, const DArr &ranges) {
             auto r = self.cast<Mapping::LocalMapper &>().scanToGridBaysian(dvec(angles), dvec(ranges));
             return py::make_tuple(gridView(std::get<0>(r), self), probView(std::get<1>(r), self));
           }, "Convert laser scan data to occupancy grid, with baysian update (float64 arrays: one copy each)",
           py::arg("angles").noconvert(), py::arg("ranges").noconvert())
      .def("scan_to_grid_baysian", [gridView, probView](py::object self, const std::vector<double> &angles,
                                                        const std::vector<double> &ranges) {
             auto r = self.cast<Mapping::LocalMapper &>().scanToGridBaysian(angles, ranges);
             return py::make_tuple(gridView(std::get<0>(r), self), probView(std::get<1>(r), self));
           }, "Convert laser scan data to occupancy grid, with baysian update", py::arg("angles"), py::arg("ranges"))
      .def("scan_to_grid_baysian", [gridView, probView](py::object self, const std::vector<int8_t> &data,
                                                        int point_step, int row_step, int height, int width,
                                                        float x_offset, float y_offset, float z_offset) {
             auto r = self.cast<Mapping::LocalMapper &>().scanToGridBaysian(data, point_step, row_step, height,
                                                                            width, x_offset, y_offset, z_offset);
             return py::make_tuple(gridView(std::get<0>(r), self), probView(std::get<1>(r), self));
           }, "Convert a raw point cloud to occupancy grid, with baysian update", py::arg("data"),
           py::arg("point_step"), py::arg("row_step"), py::arg("height"), py::arg("width"), py::arg("x_offset"),
           py::arg("y_offset"), py::arg("z_offset"))
      // returns the warped grid (the reference returns None and its Python
      // caller stores the result; `unknown_value`, which that caller passes, is
      // accepted and unused: the fill value is the mapper's p_prior,
      // local_mapper.cpp:41)
      .def("get_previous_grid_in_current_pose", [probView](py::object self, const py::object &pos, double orient,
                                                           const py::object &) {
             auto &m = self.cast<Mapping::LocalMapper &>();
             auto v = py::cast<std::vector<float>>(pos);
             if (v.size() < 2) throw std::invalid_argument("current_position_in_previous_pose needs x and y");
             m.getPreviousGridInCurrentPose(Eigen::Vector2f(v[0], v[1]), orient);
             return probView(m.previousGridProb(), self);
           }, py::arg("current_position_in_previous_pose"), py::arg("current_orientation_in_previous_pose"),
           py::arg("unknown_value") = py::none())
      .def("set_previous_grid", [](Mapping::LocalMapper &m, const py::object &prob) {
             if (prob.is_none()) {
               m.setPreviousGridProb(nullptr);
               return;
             }
             auto a = py::array_t<float, py::array::f_style | py::array::forcecast>::ensure(prob);
             if (!a || a.ndim() != 2) throw std::invalid_argument("previous grid must be a 2-D array");
             Eigen::MatrixXf g(static_cast<int>(a.shape(0)), static_cast<int>(a.shape(1)));
             std::memcpy(g.data(), a.data(), sizeof(float) * static_cast<size_t>(a.size()));
             m.setPreviousGridProb(&g);
           }, "Replace the previous probability grid (None: feed the last scan's probabilities back)",
           py::arg("previous_grid") = py::none());

  py::class_<Mapping::LocalMapperGPU, Mapping::LocalMapper>(mp, "LocalMapperGPU")
      .def(py::init([](int H, int W, float res, const py::object &pos, float orient, bool pc, int scan, float step,
                       float maxh, float minh, float rmax, int mppl) {
             return std::make_unique<Mapping::LocalMapperGPU>(H, W, res, vec3(pos), orient, pc, scan, step, maxh, minh, rmax, mppl);
           }), py::arg("grid_height"), py::arg("grid_width"), py::arg("resolution"), py::arg("laserscan_position"),
           py::arg("laserscan_orientation"), py::arg("is_pointcloud"), py::arg("scan_size"), py::arg("angle_step"),
           py::arg("max_height"), py::arg("min_height"), py::arg("range_max"), py::arg("max_points_per_line") = 32);

  // ----------------------------------------------------------------- utils
  // (bindings_utils.cpp:47-118, bindings_gpu.cpp:40-68; the PCD reader of that
  // submodule is outside this build's scope)
  auto ut = m.def_submodule("utils", "KOMPASS CPP utilities");
  {
    auto czInit = [](auto *tag, CriticalZoneChecker::InputType it, CollisionChecker::ShapeType shape,
                     const std::vector<float> &dims, const py::object &spos, const py::object &srot, float ca,
                     float cd, float sd, const std::vector<double> &angles, float minh, float maxh, float rmax) {
      using T = std::remove_pointer_t<decltype(tag)>;
      return std::make_unique<T>(it, shape, dims, vec3(spos), vec4(srot), ca, cd, sd, angles, minh, maxh, rmax);
    };
    py::class_<CriticalZoneChecker> cz(ut, "CriticalZoneChecker");
    py::enum_<CriticalZoneChecker::InputType>(cz, "InputType")
        .value("LASERSCAN", CriticalZoneChecker::InputType::LASERSCAN)
        .value("POINTCLOUD", CriticalZoneChecker::InputType::POINTCLOUD);
    py::enum_<PointFieldType>(ut, "PointFieldType")
        .value("INT8", PointFieldType::INT8).value("UINT8", PointFieldType::UINT8)
        .value("INT16", PointFieldType::INT16).value("UINT16", PointFieldType::UINT16)
        .value("INT32", PointFieldType::INT32).value("UINT32", PointFieldType::UINT32)
        .value("FLOAT32", PointFieldType::FLOAT32).value("FLOAT64", PointFieldType::FLOAT64);
    cz.def(py::init([czInit](CriticalZoneChecker::InputType it, CollisionChecker::ShapeType shape,
                             const std::vector<float> &dims, const py::object &spos, const py::object &srot,
                             float ca, float cd, float sd, const std::vector<double> &angles, float minh,
                             float maxh, float rmax) {
             return czInit(static_cast<CriticalZoneChecker *>(nullptr), it, shape, dims, spos, srot, ca, cd, sd,
                           angles, minh, maxh, rmax);
           }), py::arg("input_type"), py::arg("robot_shape"), py::arg("robot_dimensions"),
           py::arg("sensor_position_body"), py::arg("sensor_rotation_body"), py::arg("critical_angle"),
           py::arg("critical_distance"), py::arg("slowdown_distance"), py::arg("scan_angles"),
           py::arg("min_height"), py::arg("max_height"), py::arg("range_max"))
        .def("check", py::overload_cast<const std::vector<double> &, bool>(&CriticalZoneChecker::check),
             py::arg("ranges"), py::arg("forward"))
        .def("check", py::overload_cast<const std::vector<int8_t> &, int, int, int, int, int, int, int, bool>(
                          &CriticalZoneChecker::check),
             py::arg("data"), py::arg("point_step"), py::arg("row_step"), py::arg("height"), py::arg("width"),
             py::arg("x_offset"), py::arg("y_offset"), py::arg("z_offset"), py::arg("forward"));
    py::class_<CriticalZoneCheckerGPU, CriticalZoneChecker>(ut, "CriticalZoneCheckerGPU")
        .def(py::init([](CriticalZoneChecker::InputType it, CollisionChecker::ShapeType shape,
                         const std::vector<float> &dims, const py::object &spos, const py::object &srot, float ca,
                         float cd, float sd, const std::vector<double> &angles, float minh, float maxh, float rmax,
                         PointFieldType ft) {
               return std::make_unique<CriticalZoneCheckerGPU>(it, shape, dims, vec3(spos), vec4(srot), ca, cd, sd,
                                                               angles, minh, maxh, rmax, ft);
             }), py::arg("input_type"), py::arg("robot_shape"), py::arg("robot_dimensions"),
             py::arg("sensor_position_body"), py::arg("sensor_rotation_body"), py::arg("critical_angle"),
             py::arg("critical_distance"), py::arg("slowdown_distance"), py::arg("scan_angles"),
             py::arg("min_height"), py::arg("max_height"), py::arg("range_max"),
             py::arg("cloud_field_type") = PointFieldType::FLOAT32);
  }
  ut.def("pointcloud_to_laserscan_from_raw",
         [](const std::vector<int8_t> &data, int point_step, int row_step, int height, int width, int x_offset,
            int y_offset, int z_offset, double max_range, double min_z, double max_z, double angle_step) {
           std::vector<double> ranges_out, angles_out;
           pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, x_offset, y_offset, z_offset,
                                        max_range, min_z, max_z, angle_step, ranges_out, angles_out);
           return std::make_tuple(ranges_out, angles_out);
         },
         py::arg("data"), py::arg("point_step"), py::arg("row_step"), py::arg("height"), py::arg("width"),
         py::arg("x_offset"), py::arg("y_offset"), py::arg("z_offset"), py::arg("max_range"), py::arg("min_z"),
         py::arg("max_z"), py::arg("angle_step"),
         "Converts raw PointCloud2 to ranges and angles using a specific angular step.");
  ut.def("pointcloud_to_laserscan_from_raw",
         [](const std::vector<int8_t> &data, int point_step, int row_step, int height, int width, int x_offset,
            int y_offset, int z_offset, double max_range, double min_z, double max_z, int num_bins) {
           std::vector<double> ranges_out;
           pointCloudToLaserScanFromRaw(data, point_step, row_step, height, width, x_offset, y_offset, z_offset,
                                        max_range, min_z, max_z, num_bins, ranges_out);
           return ranges_out;
         },
         py::arg("data"), py::arg("point_step"), py::arg("row_step"), py::arg("height"), py::arg("width"),
         py::arg("x_offset"), py::arg("y_offset"), py::arg("z_offset"), py::arg("max_range"), py::arg("min_z"),
         py::arg("max_z"), py::arg("num_bins"),
         "Converts raw PointCloud2 to ranges only, using a fixed number of bins.");

  // ---------------------------------------------------------- module level
  py::enum_<LogLevel>(m, "LogLevel")
      .value("DEBUG", LogLevel::DEBUG)
      .value("INFO", LogLevel::INFO)
      .value("WARNING", LogLevel::WARNING)
      .value("WARN", LogLevel::WARNING)
      .value("ERROR", LogLevel::ERROR)
      .export_values();
  m.def("set_log_level", &setLogLevel, "Set the log level");
  m.def("set_log_file", &setLogFile, "Set the log file");
  m.def("comm_unique_id", []() {
    uint8_t id[KC_COMM_ID_BYTES];
    hip::check(kc_comm_unique_id(id));
    return py::bytes(reinterpret_cast<const char *>(id), KC_COMM_ID_BYTES);
  }, "RCCL unique id for DWA.enable_sharding: create on one rank, send to all");
  m.def("set_host_threads", [](int n) { hip::check(kc_set_host_threads(n)); }, py::arg("n"),
        "Threads of the host pool behind the roll-out's libm trig table (default: from the CPUs the process may use)");
  m.def("get_available_accelerators", []() {
    const int n = kc_device_count();
    return n > 0 ? std::string("HIP: ") + std::to_string(n) + " device(s) (gfx950)" : std::string("");
  }, "Get available accelerators");
}
