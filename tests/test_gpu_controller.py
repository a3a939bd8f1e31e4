"""Closed-loop tests of the drop-in surface on the GPU: kompass_core.control.DWA
(-> kompass_cpp.control.DWA -> C ABI -> HIP) in lockstep with the CPU oracle's
controller restatement.  Mirrors the reference's tests/test_controllers.py::
test_dwa and src/kompass_cpp/tests/dwa_test.cpp scenarios."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_cpp  # noqa: E402
import kompass_hip as kh  # noqa: E402
from kompass_core.control import DWA, DWAConfig, TrajectoryCostsWeights  # noqa: E402
from kompass_core.datatypes import LaserScanData  # noqa: E402
from kompass_core.mapping import LocalMapper, MapConfig  # noqa: E402
from kompass_core.models import (AngularCtrlLimits, LinearCtrlLimits, Robot, RobotCtrlLimits,  # noqa: E402
                                 RobotGeometry, RobotType)
from oracle import ko  # noqa: E402

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert kh.device_count() >= 1, "no HIP device visible"


class _P:  # nav_msgs/Path shaped stand-in, like tests/test_controllers.py:47-80
    def __init__(self, pts):
        mk = lambda x, y: type("Pose", (), {"pose": type("I", (), {"position": type("Pt", (), {"x": x, "y": y})()})()})()
        self.poses = [mk(float(x), float(y)) for x, y in pts]


def ref_path():
    d = json.loads((GOLD / "global_path.json").read_text())
    return [(p["pose"]["position"]["x"], p["pose"]["position"]["y"]) for p in d["poses"]]


CTR = {RobotType.ACKERMANN: ko.ACKERMANN, RobotType.DIFFERENTIAL_DRIVE: ko.DIFFERENTIAL_DRIVE, RobotType.OMNI: ko.OMNI}
SHP = {RobotGeometry.Type.CYLINDER: ko.CYLINDER, RobotGeometry.Type.BOX: ko.BOX, RobotGeometry.Type.SPHERE: ko.SPHERE}


def make_pair(robot_type, geom, dims, vx_lim, om_lim, cfg: DWAConfig, vy_lim=None):
    robot = Robot(robot_type=robot_type, geometry_type=geom, geometry_params=np.array(dims))
    vy = vy_lim or LinearCtrlLimits(max_vel=0.0, max_acc=0.0, max_decel=0.0)
    limits = RobotCtrlLimits(vx_limits=vx_lim, omega_limits=om_lim, vy_limits=vy)
    gpu = DWA(robot=robot, ctrl_limits=limits, config=cfg)
    w = cfg.costs_weights
    cpu = ko.DWA(
        ko.make_limits((vx_lim.max_vel, vx_lim.max_acc, vx_lim.max_decel), (vy.max_vel, vy.max_acc, vy.max_decel),
                       (om_lim.max_steer, om_lim.max_vel, om_lim.max_acc, om_lim.max_decel)),
        CTR[robot_type], cfg.control_time_step, cfg.prediction_horizon * cfg.control_time_step,
        cfg.control_horizon * cfg.control_time_step, cfg.max_linear_samples, cfg.max_angular_samples,
        SHP[geom], dims, tuple(cfg.proximity_sensor_position_to_robot), tuple(cfg.proximity_sensor_rotation_to_robot),
        cfg.octree_resolution,
        ko.make_weights(w.reference_path_distance_weight, w.goal_distance_weight, w.obstacles_distance_weight,
                        w.smoothness_weight, w.jerk_weight))
    return robot, gpu, cpu


def lockstep(robot, gpu, cpu, pts, start, scan=None, cloud=None, max_controls=100, dt=0.1, clearance_to=None):
    """run_control() of tests/test_controllers.py:167-240 with the oracle stepping beside it.
    clearance_to: a point cloud; the smallest distance of the driven positions to it comes back as
    lockstep.min_clearance (dwa_test.cpp:290-295, minDistanceToCloud after every applied control)."""
    lockstep.min_clearance = float("inf")
    gpu.set_path(_P(pts))
    cpu.set_path(np.array([[x, y, 0.0] for x, y in pts], np.float32))
    robot.state.x, robot.state.y, robot.state.yaw = start
    i, cycles = 0, 0
    end = False
    while not end and i < max_controls:
        s = robot.state
        ok = gpu.loop_step(current_state=s, laser_scan=scan, local_map=cloud)
        cpu.set_state(s.x, s.y, s.yaw, s.speed)
        if cpu.is_goal_reached():
            assert not ok
            end = gpu.reached_end()
            break
        # kompass_cpp.types.Velocity2D takes float arguments (bindings_types.cpp:63-67)
        vel = tuple(float(np.float32(v)) for v in (s.vx, s.vy, s.omega))
        o = cpu.compute(vel, scan=(scan.ranges, scan.angles) if scan is not None else None, points=cloud)
        assert ok
        assert gpu.has_result() == bool(o["found"]), f"cycle {cycles}"
        if not o["found"]:
            break
        assert np.float32(gpu.result_cost) == np.float32(o["cost"]), f"cycle {cycles}"
        np.testing.assert_array_equal(np.asarray(gpu.optimal_path().x), o["path_x"])
        np.testing.assert_array_equal(np.asarray(gpu.optimal_path().y), o["path_y"])
        np.testing.assert_array_equal(np.asarray(gpu.control_till_horizon.vx), o["vel"][0])
        np.testing.assert_array_equal(np.asarray(gpu.control_till_horizon.omega), o["vel"][2])
        cycles += 1
        for vx, vy, om in zip(gpu.linear_x_control, gpu.linear_y_control, gpu.angular_control):
            robot.set_control(velocity_x=vx, velocity_y=vy, omega=om)
            robot.get_state(dt=dt)
            if clearance_to is not None:
                d = np.hypot(clearance_to[:, 0] - robot.state.x, clearance_to[:, 1] - robot.state.y).min()
                lockstep.min_clearance = min(lockstep.min_clearance, float(d))
            i += 1
            end = gpu.reached_end()
    return end, i, cycles


def test_dwa_reference_python_scenario():
    """tests/test_controllers.py::test_dwa: Ackermann, global_path.json, L = A = 4,
    10-step horizon, weights path 3 / goal 1, empty 201-beam scan at 20 m."""
    cfg = DWAConfig(max_linear_samples=4, max_angular_samples=4, octree_resolution=0.1,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=3.0, goal_distance_weight=1.0,
                                                         smoothness_weight=0.0, jerk_weight=0.0,
                                                         obstacles_distance_weight=0.0),
                    prediction_horizon=10, control_horizon=2, control_time_step=0.1, max_num_threads=1)
    robot, gpu, cpu = make_pair(RobotType.ACKERMANN, RobotGeometry.Type.CYLINDER, [0.1, 0.4],
                                LinearCtrlLimits(max_vel=1.0, max_acc=5.0, max_decel=10.0),
                                AngularCtrlLimits(max_vel=4.0, max_acc=3.0, max_decel=3.0, max_steer=np.pi), cfg)
    end, n, cycles = lockstep(robot, gpu, cpu, ref_path(), (-0.51731912, 0.0, np.pi / 2), scan=LaserScanData())
    assert end is True and n <= 100 and cycles > 5


def _round_obstacle(x, y, radius, res=0.1):
    pts = []
    r = 0.0
    while r <= radius:
        if r == 0:
            pts.append((x, y, 0.0))
        else:
            th = 0.0
            while th < 2 * math.pi:
                pts.append((x + r * math.cos(th), y + r * math.sin(th), 0.0))
                th += res / r
        r += res
    return np.array(pts, np.float32)


@pytest.mark.parametrize("rtype", [RobotType.ACKERMANN, RobotType.DIFFERENTIAL_DRIVE, RobotType.OMNI])
@pytest.mark.parametrize("with_obstacle", [False, True])
def test_dwa_cpp_scenarios(rtype, with_obstacle):
    """src/kompass_cpp/tests/dwa_test.cpp:161-362: straight path, 3 robot types,
    with / without a round obstacle beside the path (point-cloud input); the
    robot must keep a clearance >= its radius and the GPU controller must agree
    with the oracle every cycle."""
    cfg = DWAConfig(max_linear_samples=11, max_angular_samples=11, octree_resolution=0.1,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=3.0,
                                                         obstacles_distance_weight=1.0, smoothness_weight=0.0,
                                                         jerk_weight=0.0),
                    prediction_horizon=20, control_horizon=2, control_time_step=0.1)
    robot, gpu, cpu = make_pair(rtype, RobotGeometry.Type.CYLINDER, [0.1, 0.4],
                                LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
                                AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0), cfg,
                                vy_lim=LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0))
    pts = [(x, 0.0) for x in np.arange(0.0, 10.01, 0.5)]
    cloud = _round_obstacle(3.0, 0.35, 0.2) if with_obstacle else np.array([[50.0, 50.0, 0.0]], np.float32)
    end, n, cycles = lockstep(robot, gpu, cpu, pts, (0.0, 0.1, 0.0), cloud=cloud, max_controls=400,
                              clearance_to=cloud if with_obstacle else None)
    assert cycles > 10
    assert end is True, f"goal not reached after {n} controls"
    if with_obstacle:  # dwa_test.cpp:355-358: BOOST_TEST(min_clearance >= robotRadius)
        assert lockstep.min_clearance >= 0.1, f"DWA collided with obstacle: clearance {lockstep.min_clearance}"


def test_debug_samples_and_custom_cost():
    cfg = DWAConfig(max_linear_samples=7, max_angular_samples=7, octree_resolution=0.1, prediction_horizon=10,
                    control_horizon=2, control_time_step=0.1,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=1.0,
                                                         obstacles_distance_weight=0.0))
    robot, gpu, cpu = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.BOX, [0.4, 0.3, 0.5],
                                LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
                                AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0), cfg)
    pts = [(x, 0.0) for x in np.arange(0.0, 6.01, 0.5)]
    gpu.set_path(_P(pts))
    cpu.set_path(np.array([[x, y, 0.0] for x, y in pts], np.float32))
    robot.state.x, robot.state.y, robot.state.yaw = 0.0, 0.2, 0.1
    robot.state.vx = 0.4
    cloud = _round_obstacle(0.8, 0.35, 0.2)
    s = robot.state
    assert gpu.loop_step(current_state=s, local_map=cloud, debug=True)
    cpu.set_state(s.x, s.y, s.yaw, s.speed)
    vel = tuple(float(np.float32(v)) for v in (s.vx, s.vy, s.omega))  # Velocity2D(float, ...)
    o = cpu.compute(vel, points=cloud)
    px, py = gpu.planner.get_debugging_samples()
    assert 0 < len(o["samples_x"]) < o["n_generated"]
    np.testing.assert_array_equal(px, o["samples_x"])
    np.testing.assert_array_equal(py, o["samples_y"])
    # custom cost: prefer trajectories ending far to the left (+y); host callback
    calls = []

    def custom(traj, path):
        calls.append(1)
        return float(-traj.path.y[-1])

    gpu.planner.add_custom_cost(5.0, custom)
    assert gpu.loop_step(current_state=s, local_map=cloud)
    assert len(calls) == len(o["samples_x"])
    want = (o["costs"].astype(np.float64) + 5.0 * (-o["samples_y"][:, -1]).astype(np.float32).astype(np.float64)).astype(np.float32)
    k = int(np.argmin(want))
    assert np.float32(gpu.result_cost) == want[k]
    np.testing.assert_array_equal(np.asarray(gpu.optimal_path().y), o["samples_y"][k])


def test_debug_velocity_search_without_sample_dropping():
    """DWA::debugVelocitySearch(vel, data, drop_samples = false) (controllers/dwa.h:146-158): samples that collide
    beyond the control horizon are frozen and kept -- class level against the oracle's restatement of
    trajectory_sampler.cpp:118-179 (numCtrlPoints_ = control_horizon / time_step, :88)."""
    cfg = DWAConfig(max_linear_samples=9, max_angular_samples=9, octree_resolution=0.1, prediction_horizon=40,
                    control_horizon=4, control_time_step=0.1, drop_samples=False,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=1.0,
                                                         obstacles_distance_weight=1.0, smoothness_weight=1.0, jerk_weight=1.0))
    vx_lim = LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0)
    om_lim = AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0)
    robot, gpu, cpu = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4], vx_lim, om_lim, cfg)
    pts = [(x, 0.0) for x in np.arange(0.0, 6.01, 0.5)]
    gpu.set_path(_P(pts))
    robot.state.x, robot.state.y, robot.state.yaw = 0.0, 0.1, 0.05
    robot.state.vx = 0.5
    cloud = _round_obstacle(1.2, 0.1, 0.25)
    s = robot.state
    assert gpu.loop_step(current_state=s, local_map=cloud, debug=True)   # debug -> debug_velocity_search(drop_samples=False)
    px, py = gpu.planner.get_debugging_samples()
    # the oracle: same window, same voxels, both modes
    vel = tuple(float(np.float32(v)) for v in (s.vx, s.vy, s.omega))
    lim = ko.make_limits((vx_lim.max_vel, vx_lim.max_acc, vx_lim.max_decel), (0.0, 0.0, 0.0),
                         (om_lim.max_steer, om_lim.max_vel, om_lim.max_acc, om_lim.max_decel))
    svx, svy, som = ko.sample_velocities(CTR[RobotType.DIFFERENTIAL_DRIVE], lim, vel, 0.1, 9, 9)
    coll = ko.Collision(SHP[RobotGeometry.Type.CYLINDER], [0.1, 0.4], (0, 0, 0), (0, 0, 0, 1), 0.1)
    coll.update_state(s.x, s.y, s.yaw)
    coll.update_points(np.asarray(cloud, np.float32).reshape(-1, 3), True)
    P = int(40 * 0.1 / 0.1)
    kx, ky, kraw, kvel = ko.rollout_mode(coll, (s.x, s.y, s.yaw, 0.0), 0.1, P, svx, svy, som, False, int(0.4 / 0.1))
    dx, dy, draw, _ = ko.rollout_mode(coll, (s.x, s.y, s.yaw, 0.0), 0.1, P, svx, svy, som, True, 0)
    assert len(kraw) > len(draw) > 0            # some samples are kept only because they are frozen
    np.testing.assert_array_equal(np.asarray(px), kx)
    np.testing.assert_array_equal(np.asarray(py), ky)
    # the command of the cycle comes from the same mode (the sampler keeps it, as the reference's does)
    assert gpu.has_result()


def test_local_mapper_frontend():
    """kompass_core.mapping.LocalMapper.update_from_scan (tests/test_local_mapper_
    pytest.py invariants) + cell-exact agreement with the CPU mapper semantics."""
    d = json.loads((GOLD / "laserscan_data.json").read_text())
    scan = LaserScanData(angle_min=d["angle_min"], angle_max=d["angle_max"], angle_increment=d["angle_increment"],
                         range_max=d["range_max"], ranges=np.array(d["ranges"], float),
                         angles=d["angle_min"] + np.arange(len(d["ranges"])) * d["angle_increment"])
    scan.ranges = np.nan_to_num(scan.ranges, posinf=scan.range_max)
    m = LocalMapper(MapConfig(width=8.0, height=6.0, resolution=0.05))
    m.update_from_scan(None, scan)
    g = m.occupancy
    assert g.shape == (120, 160) and g.dtype == np.int32
    assert set(np.unique(g)) <= {-1, 0, 100} and (g == 100).sum() > 0 and (g == 0).sum() > 0
    want = ko.scan_to_grid(120, 160, 0.05, (0, 0, 0), 0.0, scan.angles, np.clip(scan.ranges, 0, 20.0))
    np.testing.assert_array_equal(g, want)
    # out-of-grid scan => (almost) no OCCUPIED cells (test_local_mapper_pytest.py:198-230)
    far = LaserScanData(ranges=np.full(360, 50.0), angles=np.linspace(0, 2 * np.pi, 360, endpoint=False), range_max=60.0)
    m2 = LocalMapper(MapConfig(width=4.0, height=4.0, resolution=0.05, filter_limit=60.0))
    m2.update_from_scan(None, far)
    assert (m2.occupancy == 100).mean() < 0.01


def test_local_mapper_frontend_baysian():
    """MapConfig(baysian_update=True): the module-level scan_to_grid_baysian /
    get_previous_grid_in_current_pose (bound to the real C++ methods here, see
    the binding's comment) against the oracle over a three-pose drive."""
    from kompass_core.mapping import ScanModelConfig
    from kompass_core.models import RobotState
    d = json.loads((GOLD / "laserscan_data.json").read_text())
    ang = d["angle_min"] + np.arange(len(d["ranges"])) * d["angle_increment"]
    rng0 = np.clip(np.nan_to_num(np.array(d["ranges"], float), posinf=d["range_max"]), 0, 20.0)
    sm = ScanModelConfig()
    assert abs(sm.p_empty - 0.1) < 1e-12
    m = LocalMapper(MapConfig(width=8.0, height=6.0, resolution=0.05, baysian_update=True), sm)
    o = ko.BayesMapper(120, 160, 0.05, (0, 0, 0), 0.0, p_prior=sm.p_prior, p_occupied=sm.p_occupied,
                       p_empty=sm.p_empty, range_sure=sm.range_sure, range_max=sm.range_max, wall_size=sm.wall_size)
    poses = [(0.0, 0.0, 0.0), (0.10, 0.02, 0.05), (0.22, 0.01, 0.12)]
    for k, (x, y, yaw) in enumerate(poses):
        scan = LaserScanData(angle_min=d["angle_min"], angle_max=d["angle_max"], angle_increment=d["angle_increment"],
                             range_max=d["range_max"], ranges=rng0 * (1.0 + 0.02 * k), angles=ang)
        if k:
            px, py, pyaw = poses[k - 1]
            c, s = np.cos(pyaw), np.sin(pyaw)
            rel = (c * (x - px) + s * (y - py), -s * (x - px) + c * (y - py))
            want_prev = o.get_previous_grid_in_current_pose(rel, yaw - pyaw)
        m.update_from_scan(RobotState(x=x, y=y, yaw=yaw), scan)
        want_g, want_p = o.scan_to_grid_baysian(ang, np.clip(scan.ranges, 0, 20.0))
        np.testing.assert_array_equal(m.occupancy, want_g)
        np.testing.assert_array_equal(m.scan_occupancy_prob.view(np.uint32), want_p.view(np.uint32))
        if k:
            np.testing.assert_array_equal(np.ascontiguousarray(m.previous_grid_prob_transformed).view(np.uint32),
                                          want_prev.view(np.uint32))
        layer = m.probabilistic_occupancy
        assert set(np.unique(layer)) <= {-1, 0, 100}
        assert ((layer == -1) == (want_p == np.float32(sm.p_prior))).all()
    # kompass_cpp level: feeding the probabilities back (extension, see local_mapper.h)
    import kompass_cpp
    cm = kompass_cpp.mapping.LocalMapper(grid_height=120, grid_width=160, resolution=0.05,
                                         laserscan_position=[0.0, 0.0, 0.0], laserscan_orientation=0.0,
                                         is_pointcloud=False, scan_size=len(ang), p_prior=0.6, p_occupied=0.9,
                                         p_empty=0.1, range_sure=0.1, range_max=20.0, wall_size=0.1, angle_step=0.01,
                                         max_height=10.0, min_height=-10.0, max_points_per_line=600)
    o2 = ko.BayesMapper(120, 160, 0.05, (0, 0, 0), 0.0, 0.6, 0.9, 0.1, 0.1, 20.0, 0.1)
    for k in range(2):
        g, p = cm.scan_to_grid_baysian(angles=list(ang), ranges=list(rng0))
        wg, wp = o2.scan_to_grid_baysian(ang, rng0)
        np.testing.assert_array_equal(np.asarray(p).copy().view(np.uint32), wp.view(np.uint32))
        cm.set_previous_grid(None)
        o2.set_previous(wp)


def test_dwa_consumes_the_mapper_grid_on_the_device():
    """Closed loop with `local_map=<kompass_cpp LocalMapper>`: every cycle the
    scan goes into the device grid (no host copy) and the controller takes its
    OCCUPIED cells from there; the oracle gets the list a host would extract
    from the same grid.  Robot-centred grid, so the world-fixed wall is
    re-scanned from every pose."""
    import kompass_cpp
    cfg = DWAConfig(max_linear_samples=11, max_angular_samples=11, octree_resolution=0.1,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=3.0,
                                                         obstacles_distance_weight=1.0, smoothness_weight=0.0,
                                                         jerk_weight=0.0),
                    prediction_horizon=20, control_horizon=2, control_time_step=0.1)
    robot, gpu, cpu = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4],
                                LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
                                AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0), cfg)
    H = W = 160
    res = 0.1
    ang = np.linspace(-np.pi, np.pi, 720, endpoint=False)
    mapper = kompass_cpp.mapping.LocalMapper(grid_height=H, grid_width=W, resolution=res,
                                             laserscan_position=[0.0, 0.0, 0.0], laserscan_orientation=0.0,
                                             is_pointcloud=False, scan_size=len(ang), angle_step=0.01, max_height=10.0,
                                             min_height=-10.0, range_max=20.0, max_points_per_line=400)
    c0, c1 = H // 2 - 1, W // 2 - 1
    pts = [(x, 0.0) for x in np.arange(0.0, 6.01, 0.5)]
    gpu.set_path(_P(pts))
    cpu.set_path(np.array([[x, y, 0.0] for x, y in pts], np.float32))
    robot.state.x, robot.state.y, robot.state.yaw = 0.0, 0.1, 0.0
    cycles, blocked = 0, 0
    for _ in range(60):
        s = robot.state
        # a round post of radius 0.25 m at (2.5, 0.45) in the world, seen from the robot
        dx, dy = 2.5 - s.x, 0.45 - s.y
        dist, bearing = math.hypot(dx, dy), math.atan2(dy, dx) - s.yaw
        half = math.asin(min(1.0, 0.25 / max(dist, 0.26)))
        rel = (ang - bearing + np.pi) % (2 * np.pi) - np.pi
        rng = np.where(np.abs(rel) < half, max(dist - 0.25, 0.05), 7.5)
        mapper.scan_to_grid_on_device(angles=list(ang), ranges=list(rng))
        ok = gpu.loop_step(current_state=s, local_map=mapper)
        cpu.set_state(s.x, s.y, s.yaw, s.speed)
        if cpu.is_goal_reached():
            assert not ok
            break
        grid = ko.scan_to_grid(H, W, res, (0, 0, 0), 0.0, ang, rng)
        jj, ii = np.nonzero(grid.T == 100)
        cloud = np.zeros((len(ii), 3), np.float32)
        cloud[:, 0] = (ii - c0).astype(np.float32) * np.float32(res)
        cloud[:, 1] = (jj - c1).astype(np.float32) * np.float32(res)
        vel = tuple(float(np.float32(v)) for v in (s.vx, s.vy, s.omega))
        o = cpu.compute(vel, points=cloud)
        assert ok and gpu.has_result() == bool(o["found"]), f"cycle {cycles}"
        if not o["found"]:
            break
        blocked += int(o["n_admissible"] < o["n_generated"])
        assert np.float32(gpu.result_cost) == np.float32(o["cost"]), f"cycle {cycles}"
        np.testing.assert_array_equal(np.asarray(gpu.optimal_path().x), o["path_x"])
        np.testing.assert_array_equal(np.asarray(gpu.optimal_path().y), o["path_y"])
        cycles += 1
        for vx, vy, om in zip(gpu.linear_x_control, gpu.linear_y_control, gpu.angular_control):
            robot.set_control(velocity_x=vx, velocity_y=vy, omega=om)
            robot.get_state(dt=0.1)
    assert cycles > 10 and blocked > 0, "the post must have cost the controller some samples"


def test_closed_loop_with_the_resident_path(tmp_path):
    """KOMPASS_RESIDENT_PATH=1 (read once per process, hence a child): the
    CostEvaluator keeps the reference path on the device and moves a window; the
    closed-loop scenarios must agree with the oracle cycle by cycle as before."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_gpu_controller.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_dwa_cpp_scenarios or test_dwa_reference_python_scenario",
                        "-p", "no:cacheprovider"],
                       env=dict(os.environ, KOMPASS_RESIDENT_PATH="1"), cwd=str(root), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "7 passed" in r.stdout, r.stdout[-500:]


def test_collision_checker_batch_poses():
    """kc_dwa_check_poses == oracle check_at for all shapes (checkStatesFeasibility path)."""
    rng = np.random.default_rng(9)
    cloud = (rng.random((400, 3)) * [6, 6, 0.6] - [3, 3, 0.3]).astype(np.float32)
    for shape, dims in [(kh.CYLINDER, [0.2, 0.5]), (kh.BOX, [0.5, 0.3, 0.4]), (kh.SPHERE, [0.25])]:
        ctx = kh.DwaContext(shape, dims, octree_res=0.1, max_samples=4, max_points=4)
        ctx.set_points((0.5, -0.5, 0.3, 0), cloud)
        x, y, yaw = rng.random(500) * 6 - 3, rng.random(500) * 6 - 3, rng.random(500) * 6.28 - 3.14
        got = ctx.check_poses(x, y, yaw)
        c = ko.Collision(shape, dims, res=0.1)
        c.update_state(0.5, -0.5, 0.3)
        c.update_points(cloud, True)
        want = np.array([c.check_at(a, b, t) for a, b, t in zip(x, y, yaw)])
        assert 0 < want.sum() < 500
        np.testing.assert_array_equal(got, want)


def test_class_level_cycle_is_one_launch_and_shards_through_rccl():
    """kompass_cpp.control.DWA (what kompass_core drives): every compute_velocity_commands is ONE
    kernel launch (`last_cycle_single_launch` of the shared context is not reachable from here, so the
    check is behavioural: results equal the oracle's in the lockstep tests above; here: the sharded
    controller -- world of one rank through the RCCL all-reduce -- gives the same commands as the
    plain one, cycle after cycle, and the pinned-row path feeds the trajectory)."""
    cfg = DWAConfig(max_linear_samples=15, max_angular_samples=15, octree_resolution=0.1,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=3.0,
                                                         obstacles_distance_weight=1.0, smoothness_weight=0.0,
                                                         jerk_weight=0.0),
                    prediction_horizon=20, control_horizon=2, control_time_step=0.1)
    lim = (LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
           AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0))
    robot, plain, cpu = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4], *lim, cfg)
    robot2, shard, _ = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4], *lim, cfg)
    shard._planner.enable_sharding(0, 1, kompass_cpp.comm_unique_id(), 0)
    pts = [(x, 0.0) for x in np.arange(0.0, 10.01, 0.5)]
    cloud = _round_obstacle(3.0, 0.35, 0.2)
    end, n, cycles = lockstep(robot, plain, cpu, pts, (0.0, 0.1, 0.0), cloud=cloud, max_controls=60)
    end2, n2, cycles2 = lockstep(robot2, shard, cpu, pts, (0.0, 0.1, 0.0), cloud=cloud, max_controls=60)
    assert (n, cycles) == (n2, cycles2) and cycles >= 25
    assert (robot.state.x, robot.state.y, robot.state.yaw) == (robot2.state.x, robot2.state.y, robot2.state.yaw)
    shard._planner.disable_sharding()
    shard._planner.use_resident_path(True)
    end3, n3, cycles3 = lockstep(robot2, shard, cpu, pts, (0.0, 0.1, 0.0), cloud=cloud, max_controls=60)
    assert (n3, cycles3) == (n, cycles)


def test_non_planar_sensor_mount_at_the_class_level():
    """A laserscan sensor rotated out of the plane (collision_check.cpp:61-68 accepts any quaternion): the
    octree frame is tilted, the class-level cycle runs the exact 3-D tests of the split roll-out path and
    follows the oracle's restatement step for step; point-cloud input (octree in the world frame,
    collision_check.h:119-131) works with any mount as before."""
    half = 0.2                      # 0.4 rad about the y axis
    tilted = [0.0, math.sin(half), 0.0, math.cos(half)]
    cfg = DWAConfig(max_linear_samples=5, max_angular_samples=5, octree_resolution=0.1, prediction_horizon=10,
                    control_horizon=2, control_time_step=0.1, proximity_sensor_rotation_to_robot=np.array(tilted))
    robot, gpu, cpu = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4],
                                LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
                                AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0), cfg)
    pts = [(x, 0.0) for x in np.arange(0.0, 5.01, 0.5)]
    ang = np.linspace(0.0, 2.0 * np.pi, 360, endpoint=False)
    scan = LaserScanData(ranges=1.5 + 0.5 * np.cos(3 * ang), angles=ang, range_max=10.0)
    end, n, cycles = lockstep(robot, gpu, cpu, pts, (0.0, 0.0, 0.0), scan=scan, max_controls=40)
    assert cycles >= 10
    cloud = _round_obstacle(3.0, 0.35, 0.2)
    assert gpu.loop_step(current_state=robot.state, local_map=cloud) and gpu.has_result()


def test_sensor_input_forms_give_the_same_command():
    """The bindings consume numpy inputs where they lie (an (N, 3) float32 cloud: Control::PointCloudView; float64
    range / angle arrays: one copy each) and convert everything else element by element: a list of tuples, a
    float64 array, a Fortran-ordered array and the float32 array must give the same command, cost and path;
    LaserScan from arrays must equal LaserScan from lists."""
    cfg = DWAConfig(max_linear_samples=11, max_angular_samples=11, octree_resolution=0.1, prediction_horizon=12,
                    control_horizon=2, control_time_step=0.1)
    lim = (LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
           AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0))
    robot, gpu, _ = make_pair(RobotType.DIFFERENTIAL_DRIVE, RobotGeometry.Type.CYLINDER, [0.1, 0.4], *lim, cfg)
    gpu.set_path(_P([(x, 0.0) for x in np.arange(0.0, 6.01, 0.5)]))
    robot.state.x, robot.state.y, robot.state.yaw = 0.0, 0.05, 0.0
    gpu._planner.set_current_state(0.0, 0.05, 0.0, 0.3)
    vel = kompass_cpp.types.Velocity2D(0.3, 0.0, 0.0, 0.0)
    cloud32 = np.ascontiguousarray(_round_obstacle(2.0, 0.4, 0.25), dtype=np.float32)
    forms = [cloud32, [tuple(float(v) for v in p) for p in cloud32], cloud32.astype(np.float64),
             np.asfortranarray(cloud32), cloud32[::1].copy()]
    got = []
    for f in forms:
        r = gpu._planner.compute_velocity_commands(vel, f)
        assert r.is_found
        got.append((np.float32(r.cost), np.array(r.trajectory.path.x), np.array(r.trajectory.velocities.omega)))
    for g in got[1:]:
        assert g[0] == got[0][0]
        np.testing.assert_array_equal(g[1], got[0][1])
        np.testing.assert_array_equal(g[2], got[0][2])
    with pytest.raises(Exception):
        gpu._planner.compute_velocity_commands(vel, np.zeros((5, 2), np.float32))
    # laser scans: arrays and lists
    ang = np.linspace(-1.5, 1.5, 181)
    rng = 2.0 + 0.5 * np.cos(3 * ang)
    a = kompass_cpp.types.LaserScan(ranges=rng, angles=ang)
    b = kompass_cpp.types.LaserScan(ranges=list(map(float, rng)), angles=list(map(float, ang)))
    c = kompass_cpp.types.LaserScan(ranges=rng.astype(np.float32), angles=ang.astype(np.float32))   # converted
    assert list(a.ranges) == list(b.ranges) and list(a.angles) == list(b.angles)
    assert list(c.ranges) == [float(np.float32(v)) for v in rng]
    ra = gpu._planner.compute_velocity_commands(vel, a)
    rb = gpu._planner.compute_velocity_commands(vel, b)
    assert ra.is_found == rb.is_found and (not ra.is_found or np.float32(ra.cost) == np.float32(rb.cost))
