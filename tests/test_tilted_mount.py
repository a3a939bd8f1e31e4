"""Non-planar sensor mounts with LaserScan input (collision_check.cpp:61-68 takes any quaternion,
collision_check.h:99-117: the octree lives in body_tf * sensor_tf_body): the voxel layer of the scan is tilted
against the upright robot shape.  Exact closed-set 3-D tests (sphere: distance to the cube; box: separating
axes; cylinder: cube clipped to the slab, projected, polygon against the disc), the device against the oracle's
restatement bit for bit.  Parity unpinned by the reference (FCL does this with GJK and the reference holds no
vector for a tilted mount); pinned here by closed-form cases."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

from helpers import assert_cycle_equal, hip_cycle, oracle_cycle  # noqa: E402


def quat(axis, angle):
    ax = np.asarray(axis, float)
    ax = ax / np.linalg.norm(ax)
    s = math.sin(angle / 2)
    return (ax[0] * s, ax[1] * s, ax[2] * s, math.cos(angle / 2))


SHAPES = [(syn.CYLINDER, [0.15, 0.4]), (syn.BOX, [0.4, 0.3, 0.5]), (syn.SPHERE, [0.2])]
MOUNTS = [quat((0, 1, 0), 0.4), quat((1, 0, 0), -0.3), quat((1, 1, 0.3), 0.8), quat((0.2, -1, 2.0), 1.9)]


@pytest.mark.parametrize("shape,dims", SHAPES)
@pytest.mark.parametrize("srot", MOUNTS)
def test_cycle_parity_with_a_tilted_scan_frame(shape, dims, srot):
    inp = syn.make_controller_inputs("cfg1", seed=5)
    inp["robot"] = dict(shape=shape, dims=dims)
    inp["state"] = (1.0, 2.0, 0.4, 0.0)
    inp["seg_xyz"] = inp["seg_xyz"] + np.float32([1.0, 2.0, 0.0])
    ang = np.linspace(0, 2 * math.pi, 720, endpoint=False)
    rng = np.random.default_rng(11)
    ranges = 0.8 + 1.6 * rng.random(720)
    spos = (0.1, -0.05, 0.2)
    o = oracle_cycle(inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    h = hip_cycle(kh, inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    assert_cycle_equal(o, h)


def test_tilt_changes_the_admissible_set():
    """The tilt is really seen: the same scan through a level and through a steeply pitched mount lets different
    samples through (a pitched scan plane leaves the robot's height band a short way out)."""
    inp = syn.make_controller_inputs("cfg1", seed=5)
    inp["robot"] = dict(shape=syn.CYLINDER, dims=[0.15, 0.4])
    ang = np.linspace(0, 2 * math.pi, 720, endpoint=False)
    ranges = np.full(720, 1.0)
    level = oracle_cycle(inp, scan=(ranges, ang), sensor_pos=(0, 0, 0), sensor_rot=(0, 0, 0, 1))
    pitched = oracle_cycle(inp, scan=(ranges, ang), sensor_pos=(0, 0, 0), sensor_rot=quat((0, 1, 0), 0.9))
    assert len(pitched["raw"]) > len(level["raw"])
    h = hip_cycle(kh, inp, scan=(ranges, ang), sensor_pos=(0, 0, 0), sensor_rot=quat((0, 1, 0), 0.9))
    assert_cycle_equal(pitched, h)


@pytest.mark.parametrize("shape,dims", SHAPES)
def test_pose_batches_against_a_tilted_frame(shape, dims):
    """kc_dwa_check_poses (CollisionChecker::checkCollisions for arbitrary poses): a fuzz of poses around the
    scan points, every boolean equal to the oracle's."""
    rng = np.random.default_rng(3)
    ang = np.linspace(0, 2 * math.pi, 500, endpoint=False)
    ranges = 1.0 + 0.8 * rng.random(500)
    for srot in MOUNTS[:3]:
        spos = (0.05, 0.02, 0.1)
        coll = ko.Collision(shape, dims, spos, srot, 0.05)
        st = (0.3, -0.2, 0.6, 0.0)
        coll.update_state(*st[:3])
        coll.update_scan(ranges, ang)
        ctx = kh.DwaContext(shape, dims, spos, srot, 0.05, 0.1, max_samples=16, max_points=8)
        ctx.set_scan(st, ranges, ang, 10.0)
        n = 4000
        x = st[0] + rng.uniform(-2.2, 2.2, n)
        y = st[1] + rng.uniform(-2.2, 2.2, n)
        yaw = rng.uniform(-math.pi, math.pi, n)
        got = ctx.check_poses(x, y, yaw)
        want = np.array([coll.check_at(x[i], y[i], yaw[i]) for i in range(n)], np.uint8)
        assert 0 < want.sum() < n
        np.testing.assert_array_equal(got, want)
        ctx.close()


def test_closed_form_cases_pin_the_tilted_tests():
    """One scan point straight ahead; the mount pitched by 90 degrees about y turns the sensor's x axis into the
    world's -z: the voxel lies BELOW the robot at depth r.  A shape that reaches down to -0.55 m meets it exactly
    when the voxel's face nearest to the robot (depth floor(r / res) res) lies above that."""
    res = 0.1
    srot = quat((0, 1, 0), math.pi / 2)
    for shape, dims, reach in ((syn.CYLINDER, [0.3, 1.1], 0.55), (syn.SPHERE, [0.55], 0.55), (syn.BOX, [0.6, 0.6, 1.1], 0.55)):
        for r, expect in ((0.45, True), (0.55, True), (0.62, False), (0.9, False)):
            # the point at sensor-frame (r, 0, 0) -> world (0, 0, -r) (up to float rounding of the rotation);
            # its voxel spans [floor(r / res) res, + res] along the sensor's x, i.e. the world's -z
            coll = ko.Collision(shape, dims, (0, 0, 0), srot, res)
            coll.update_state(0.0, 0.0, 0.0)
            coll.update_scan(np.array([r]), np.array([0.0]))
            top = math.floor(r / res) * res          # depth of the voxel face nearest to the robot
            assert coll.check_at(0.0, 0.0, 0.0) == (top <= reach + 1e-9) == expect, (shape, r)
            ctx = kh.DwaContext(shape, dims, (0, 0, 0), srot, res, 0.1, max_samples=16, max_points=8)
            ctx.set_scan((0, 0, 0, 0), np.array([r]), np.array([0.0]), 10.0)
            assert bool(ctx.check_poses([0.0], [0.0], [0.0])[0]) == expect
            ctx.close()


def test_a_tilted_scan_wider_than_8192_columns():
    """a fine octree and long ranges put the voxel columns of a tilted scan more than 8192
    cells apart (round 3: KC_ERR_UNSUPPORTED).  The columns beyond 4000 cells of the robot are unreachable for any
    roll-out and are dropped; cycle and pose checks equal the oracle's, which keeps them all."""
    res = 0.01
    srot, spos = quat((0, 1, 0), 0.35), (0.05, 0.0, 0.1)
    inp = syn.make_controller_inputs("cfg1", seed=8)
    inp["octree_res"] = res
    inp["robot"] = dict(shape=syn.CYLINDER, dims=[0.15, 0.4])
    ang = np.linspace(0, 2 * math.pi, 1200, endpoint=False)
    rng = np.random.default_rng(4)
    ranges = 0.7 + 1.4 * rng.random(1200)
    ranges[::9] = 55.0 + 4.0 * rng.random(len(ranges[::9]))      # far returns: +-59 m at 1 cm = 11 800 columns
    o = oracle_cycle(inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    assert 0 < len(o["raw"]) < len(inp["vx"])
    h = hip_cycle(kh, inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    assert_cycle_equal(o, h)
    ctx = h["ctx"]
    st = inp["state"]
    coll = ko.Collision(syn.CYLINDER, [0.15, 0.4], spos, srot, res)
    coll.update_state(*st[:3])
    coll.update_scan(ranges, ang)
    x = st[0] + rng.uniform(-2.0, 2.0, 1500)
    y = st[1] + rng.uniform(-2.0, 2.0, 1500)
    yaw = rng.uniform(-math.pi, math.pi, 1500)
    want = np.array([coll.check_at(x[i], y[i], yaw[i]) for i in range(1500)], np.uint8)
    assert 0 < want.sum() < 1500
    np.testing.assert_array_equal(ctx.check_poses(x, y, yaw), want)
    with pytest.raises(Exception):       # a pose 45 m away: outside what the cropped window can answer for
        ctx.check_poses([st[0] + 45.0], [st[1]], [0.0])
