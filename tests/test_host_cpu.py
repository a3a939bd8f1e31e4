"""CPU-side tests of the host C++ class surface / Python module (no GPU):
path preparation against the oracle, parameter semantics, API surface."""
import json
from pathlib import Path

import numpy as np
import pytest

import kompass_cpp
from oracle import ko

GOLD = Path(__file__).parent / "golden"


def load_path_points():
    d = json.loads((GOLD / "global_path.json").read_text())
    return np.array([[p["pose"]["position"]["x"], p["pose"]["position"]["y"], 0.0] for p in d["poses"]],
                    dtype=np.float32)


def paths_for_test():
    rng = np.random.default_rng(3)
    out = [load_path_points()]
    out.append(np.array([[0, 0, 0], [10, 0, 0]], np.float32))
    xs = np.arange(0, 10.01, 0.5)
    out.append(np.stack([xs, np.zeros_like(xs), np.zeros_like(xs)], 1).astype(np.float32))
    th = np.arange(0, 1.5 * np.pi, 0.1)
    out.append(np.stack([10 * np.cos(th), 10 * np.sin(th), np.zeros_like(th)], 1).astype(np.float32))
    out.append(np.cumsum(rng.random((40, 3)) * [0.7, 0.4, 0.0], axis=0).astype(np.float32))
    return out


@pytest.mark.parametrize("idx", range(5))
def test_follower_path_preparation_matches_oracle(idx):
    """Follower::setCurrentPath (interpolate LINEAR 0.01 + segment 1.0/101) ==
    oracle restatement of path.cpp:167-330, bit for bit."""
    pts = paths_for_test()[idx]
    f = kompass_cpp.control.Follower()
    f.set_current_path(kompass_cpp.types.Path(points=pts))
    got = f.get_current_path()
    want = ko.Path(pts).interpolate(0.01).segment(1.0, 101)
    assert got.size() == want.size
    np.testing.assert_array_equal(got.x().view(np.uint32), want.x.view(np.uint32))
    np.testing.assert_array_equal(got.y().view(np.uint32), want.y.view(np.uint32))
    assert np.float32(got.get_total_length()) == np.float32(want.total_length)
    assert f.has_path()


def test_path_errors():
    with pytest.raises(ValueError):
        kompass_cpp.types.Path(points=np.zeros((1, 3), np.float32))
    f = kompass_cpp.control.Follower()
    f.set_interpolation_type(kompass_cpp.types.PathInterpolationType.CUBIC_SPLINE)
    with pytest.raises(ValueError):
        f.set_current_path(kompass_cpp.types.Path(points=load_path_points()))


def test_parameter_ranges_and_from_dict():
    w = kompass_cpp.control.TrajectoryCostWeights()
    w.from_dict({"goal_distance_weight": 2.5, "jerk_weight": 0, "unknown_key": 7})
    with pytest.raises(RuntimeError):
        w.from_dict({"smoothness_weight": 1e4})  # range [0, 1000]
    with pytest.raises(RuntimeError):
        w.from_dict({"smoothness_weight": -1.0})


def test_module_surface():
    """Names the reference callers use."""
    t, c, m = kompass_cpp.types, kompass_cpp.control, kompass_cpp.mapping
    for n in ["State", "Path", "Velocity2D", "TrajectoryVelocities2D", "TrajectoryPath", "Trajectory",
              "LaserScan", "RobotGeometry", "PathInterpolationType"]:
        assert hasattr(t, n), n
    for n in ["ControlType", "LinearVelocityControlParams", "AngularVelocityControlParams",
              "ControlLimitsParams", "Controller", "Follower", "FollowingTarget", "SamplingControlResult",
              "TrajectoryCostWeights", "DWA"]:
        assert hasattr(c, n), n
    for n in ["OCCUPANCY_TYPE", "LocalMapper", "LocalMapperGPU"]:
        assert hasattr(m, n), n
    for n in ["LogLevel", "set_log_level", "set_log_file", "get_available_accelerators"]:
        assert hasattr(kompass_cpp, n), n
    for meth in ["compute_velocity_commands", "add_custom_cost", "get_debugging_samples",
                 "debug_velocity_search", "set_resolution", "set_current_state", "set_current_path",
                 "clear_current_path", "is_goal_reached", "get_tracked_target", "get_current_path",
                 "has_path", "set_interpolation_type", "get_vx_cmd", "get_vy_cmd", "get_omega_cmd"]:
        assert hasattr(c.DWA, meth), meth
    assert t.RobotGeometry.get("BOX") == t.RobotGeometry.BOX
    assert int(m.OCCUPANCY_TYPE.OCCUPIED) == 100 and int(m.OCCUPANCY_TYPE.UNEXPLORED) == -1
    v = t.Velocity2D(vx=0.5, omega=-0.25)
    assert (v.vx, v.vy, v.omega) == (0.5, 0.0, -0.25)
    tv = t.TrajectoryVelocities2D(np.float32([1, 2]), np.float32([0, 0]), np.float32([3, 4]))
    assert tv.length == 2 and tv.vx.dtype == np.float32 and list(tv.omega) == [3.0, 4.0]


def test_dwa_needs_a_device_no_fallback():
    import kompass_hip as kh

    if kh.device_count() > 0:
        pytest.skip("a HIP device is visible here")
    with pytest.raises(RuntimeError):
        kompass_cpp.control.DWA(
            control_limits=kompass_cpp.control.ControlLimitsParams(),
            control_type=kompass_cpp.control.ControlType.ACKERMANN, time_step=0.1, prediction_horizon=1.0,
            control_horizon=0.2, max_linear_samples=4, max_angular_samples=4,
            robot_shape_type=kompass_cpp.types.RobotGeometry.CYLINDER, robot_dimensions=[0.1, 0.4],
            sensor_position_robot=[0, 0, 0], sensor_rotation_robot=[0, 0, 0, 1], octree_resolution=0.1,
            cost_weights=kompass_cpp.control.TrajectoryCostWeights())
    with pytest.raises(RuntimeError):
        kompass_cpp.mapping.LocalMapperGPU(grid_height=10, grid_width=10, resolution=0.1,
                                           laserscan_position=[0, 0, 0], laserscan_orientation=0.0,
                                           is_pointcloud=False, scan_size=10, angle_step=0.1, max_height=1.0,
                                           min_height=0.0, range_max=5.0)
