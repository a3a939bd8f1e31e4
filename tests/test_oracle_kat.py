"""Pins the CPU oracle against the reference's own known-answer tests.

Inputs restate src/kompass_cpp/tests/cost_evaluator_test.cpp:217-461 (12 cases,
helpers :34-142) and src/kompass_cpp/tests/collisions_test.cpp:25-77 (3 cases);
expected values live in tests/golden/cost_kat.json.  CPU only (no GPU marker).
"""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import ko

GOLD = json.loads((Path(__file__).parent / "golden" / "cost_kat.json").read_text())


# --- helpers restating cost_evaluator_test.cpp:34-142 ----------------------
def straight_path(length, interp, seg):
    p = ko.Path([[0, 0, 0], [length, 0, 0]])
    p.interpolate(interp)
    p.segment(seg, 10000)
    return p


def circle34_path(R, input_pts, interp, seg):
    pts = []
    max_theta = 3.0 * math.pi / 2.0
    for i in range(input_pts):
        th = (i / (input_pts - 1)) * max_theta
        pts.append([R * math.cos(th), R * math.sin(th), 0.0])
    p = ko.Path(pts)
    p.interpolate(interp)
    p.segment(seg, 10000)
    return p


def solo(name, value=1.0):
    kw = dict(path=0.0, goal=0.0, obstacles=0.0, smoothness=0.0, jerk=0.0)
    kw[name] = value
    return ko.make_weights(**kw)


def eval_cost(weights, ref: ko.Path, seg_idx, path_pts, vels=None, obstacles=None):
    """evalCost(), cost_evaluator_test.cpp:159-178."""
    s0, s1 = ref.segment_range(seg_idx)
    seg = np.stack([ref.x[s0:s1 + 1], ref.y[s0:s1 + 1], ref.z[s0:s1 + 1]], axis=1)
    obs_xy = None
    if obstacles:
        ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), (0, 0, 0, 0), obstacles)
        obs_xy = np.stack([ox, oy], axis=1)
    ci = ko.CostInputs(seg, s0, ref.acc, ref.total_length, obs_xy,
                       max_obstacles_dist=np.float32(30.0) / np.float32(3.0),
                       acc_limits=(1.0, 1.0, 1.0), weights=weights)
    pts = np.asarray(path_pts, np.float32).reshape(-1, 3)
    n = len(pts)
    px, py = pts[:, 0][None, :], pts[:, 1][None, :]
    if vels is None:
        vels = [(0.0, 0.0, 0.0)] * (n - 1)
    v = np.asarray(vels, np.float32).reshape(n - 1, 3)
    vel = [v[:, 0][None, :], v[:, 1][None, :], v[:, 2][None, :]]
    idx, cost, costs = ko.min_trajectory_cost(ci, px, py, vel)
    assert idx == 0, "CostEvaluator did not find a trajectory"
    return cost


def close(a, b, tol):
    if b == 0.0:
        return abs(a) <= 1e-12
    return abs(a - b) <= tol * min(abs(a), abs(b))


def check(name, values):
    g = GOLD["cost"][name]
    for v, e in zip(values, g["expected"]):
        assert close(v, e, g["tol"]), (name, v, e)


def at_endpoint(n, pt):
    return [pt] * n


# --- the 12 cost cases -----------------------------------------------------
def test_goal_cost_on_straight_path():
    ref = straight_path(10.0, 1.0, 5.0)
    assert ref.size == 11
    assert ref.segment_range(0) == (0, 4)
    c = eval_cost(solo("goal"), ref, 0, at_endpoint(5, (4.0, 0.0, 0.0)))
    check("goal_cost_on_straight_path", [c])


def test_goal_cost_arc_remaining_on_curved_path():
    R = 2.0
    ref = circle34_path(R, 60, np.float32(0.05), 20.0)
    total = ref.total_length
    follow_pt = (R * math.cos(0.5), R * math.sin(0.5), 0.0)
    follow = eval_cost(solo("goal"), ref, 0, at_endpoint(5, follow_pt))
    chord = eval_cost(solo("goal"), ref, 0, at_endpoint(5, (1.5, -0.5, 0.0)))
    tol = GOLD["cost"]["goal_cost_arc_remaining_on_curved_path"]["rel_tol"]
    exp_follow = (total - R * 0.5) / total
    exp_chord = 1.0 + math.sqrt(0.5) / total
    assert close(follow, exp_follow, tol)
    assert close(chord, exp_chord, tol)
    assert follow < chord


def test_goal_cost_tie_breaker():
    ref = straight_path(10.0, 1.0, 5.0)
    a = eval_cost(solo("goal"), ref, 0, at_endpoint(5, (4.0, 0.1, 0.0)))
    b = eval_cost(solo("goal"), ref, 0, at_endpoint(5, (4.0, 0.5, 0.0)))
    check("goal_cost_tie_breaker", [a, b])
    assert a < b


def test_path_cost_centered_sample():
    ref = straight_path(10.0, 1.0, 5.0)
    pts = [(float(i), 0.0, 0.0) for i in range(5)]
    check("path_cost_centered_sample", [eval_cost(solo("path"), ref, 0, pts)])


def test_path_cost_constant_lateral_offset():
    ref = straight_path(10.0, 1.0, 5.0)
    d = 0.5
    pts = [(float(i), d, 0.0) for i in range(5)]
    check("path_cost_constant_lateral_offset", [eval_cost(solo("path"), ref, 0, pts)])


ZERO5 = [(0.0, 0.0, 0.0)] * 5


def test_smoothness_cost_constant_velocity():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("smoothness"), ref, 0, ZERO5, [(1.0, 0, 0)] * 4)
    check("smoothness_cost_constant_velocity", [c])


def test_smoothness_cost_single_step_change():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("smoothness"), ref, 0, ZERO5, [(0, 0, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0)])
    check("smoothness_cost_single_step_change", [c])


def test_jerk_cost_constant_acceleration():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("jerk"), ref, 0, ZERO5, [(0.1, 0, 0), (0.2, 0, 0), (0.3, 0, 0), (0.4, 0, 0)])
    # the reference asserts `cost == 0.0f` with a 1e-4 *relative* tolerance;
    # float(0.3)-2*float(0.2)+float(0.1) leaves a ~1e-8 residue whose square
    # (~1e-17) is what any faithful implementation returns.
    assert abs(c) < 1e-12


def test_jerk_cost_known_second_diff():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("jerk"), ref, 0, ZERO5, [(0, 0, 0), (1, 0, 0), (3, 0, 0), (6, 0, 0)])
    check("jerk_cost_known_second_diff", [c])


def test_obstacles_cost_at_max_range():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("obstacles"), ref, 0, ZERO5, obstacles=[(20.0, 0.0, 0.0)])
    check("obstacles_cost_at_max_range", [c])


def test_obstacles_cost_at_zero_distance():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("obstacles"), ref, 0, ZERO5, obstacles=[(0.0, 0.0, 0.0)])
    check("obstacles_cost_at_zero_distance", [c])


def test_obstacles_cost_at_half_range():
    ref = straight_path(10.0, 1.0, 5.0)
    c = eval_cost(solo("obstacles"), ref, 0, ZERO5, obstacles=[(5.0, 0.0, 0.0)])
    check("obstacles_cost_at_half_range", [c])


# --- collisions_test.cpp:11-78 ---------------------------------------------
@pytest.fixture
def fcl_checker():
    # Eigen::Quaternionf{0, 0, 0, 1} is (w, x, y, z) = (0,0,0,1): a half turn
    # about z; coefficient order (x, y, z, w) = (0, 0, 1, 0).
    return ko.Collision(ko.BOX, [0.4, 0.4, 1.0], sensor_pos=(0.0, 0.0, 1.0),
                        sensor_rot_xyzw=(0.0, 0.0, 1.0, 0.0), res=0.1)


def test_collision_scan_far(fcl_checker):
    c = fcl_checker
    c.update_state(0.0, 0.0, 0.0)
    c.update_scan([1.0, 1.0, 1.0], [0.0, 0.1, 0.2])
    assert c.check() is GOLD["collision"]["scan_1m_away"]


def test_collision_scan_touching(fcl_checker):
    c = fcl_checker
    c.update_state(3.0, 5.0, 0.0)
    c.update_scan([0.25, 0.5, 0.5], [0.0, 0.1, 0.2])
    assert c.check() is GOLD["collision"]["scan_quarter_metre_touching"]


def test_collision_cloud(fcl_checker):
    c = fcl_checker
    c.update_state(3.0, 5.0, 0.0)
    c.update_points([[3.1, 5.1, -0.5]], global_frame=True)
    assert c.check() is GOLD["collision"]["cloud_point_inside_box"]
