"""CriticalZoneChecker.

The 14 cases of the reference's tests/critical_zone_test.cpp:12-330 pin the
oracle restatement (CPU) and run against the HIP path through the C ABI and
the kompass_cpp module (GPU); random scans / clouds compare HIP with the oracle
bit for bit."""
import numpy as np
import pytest

from oracle import ko

N = 360
ANGLES = 2.0 * np.pi * np.arange(N) / N  # test.h:55-63 initLaserscan
CYL, BOX, SPH = 0, 1, 2


def set_at(ranges, angle, value):
    """test.h:65-95 setLaserscanAtAngle"""
    a = np.fmod(angle, 2.0 * np.pi)
    if a < 0:
        a += 2.0 * np.pi
    ranges[int(np.argmin(np.abs(ANGLES - a)))] = value


def cloud(points):
    """test.h:97-125: PointXYZ{x, y, z, padding}, 16 bytes"""
    rec = np.zeros((len(points), 4), np.float32)
    if len(points):
        rec[:, :3] = np.asarray(points, np.float32)
    return rec.reshape(-1).view(np.int8)


def laserscan_cases():
    """(ranges, forward, predicate) in the order of critical_zone_test.cpp:37-187"""
    out = []
    r = np.full(N, 10.0)
    for a in (0.0, 0.1, -0.1):
        set_at(r, a, 0.2)
    out.append((r.copy(), True, lambda v: v == 1.0))                       # 1 behind, moving forward
    r = np.full(N, 10.0)
    out.append((r.copy(), True, lambda v: v == 1.0))                       # 2 far
    for a in (np.pi, np.pi + 0.1, np.pi - 0.1):
        set_at(r, a, 0.2)
    out.append((r.copy(), True, lambda v: v == 0.0))                       # 3 front close, forward
    out.append((r.copy(), False, lambda v: v == 1.0))                      # 4 front close, backward
    for a in (0.0, 0.1, -0.1):
        set_at(r, a, 0.2)
    out.append((r.copy(), False, lambda v: v == 0.0))                      # 5 back close, backward
    r = np.full(N, 10.0)
    set_at(r, 0.0, 1.3)
    out.append((r.copy(), False, lambda v: 0.0 < v < 1.0))                 # 6 back slowdown, backward
    out.append((r.copy(), True, lambda v: v == 1.0))                       # 7 back slowdown, forward
    set_at(r, np.pi, 0.7)
    out.append((r.copy(), True, lambda v: 0.0 < v < 1.0))                  # 8 front slowdown, forward
    return out


def cloud_cases():
    """critical_zone_test.cpp:230-330"""
    junk = [(-0.1, -0.1, 3.0), (-0.1, -0.1, -3.0), (0.1, 0.2, 4.0), (0.1, 0.2, -4.0)]
    return [
        ([], True, lambda v: v == 1.0),                                                      # 9 empty
        ([(0.7, 0.0, 0.5)], True, lambda v: v == 0.0),                                       # 10 critical
        ([(0.7, 0.0, 3.0)], True, lambda v: v == 1.0),                                       # 11 too high
        ([(0.95, 0.0, 0.5)], True, lambda v: 0.4 < v < 0.6),                                 # 12 slowdown
        ([(0.95, 0.0, 0.5), (1.0, 1.0, 0.5), (-1.0, -1.0, 0.5)] + junk + [(0.75, 0.0, 0.5)],
         True, lambda v: v == 0.0),                                                          # 13 complex stop
        ([(0.95, 0.0, 0.5), (-0.95, 0.0, 0.5), (1.0, 1.0, 0.5), (-1.0, -1.0, 0.5)] + junk,
         False, lambda v: 0.4 < v < 0.6),                                                    # 14 complex slowdown
    ]


SCAN_ARGS = (CYL, [0.51, 2.0], [0.22, 0.0, 0.4], [0, 0, 0.99, 0.0], 160.0, 0.3, 0.6, ANGLES, 0.1, 2.0, 20.0)
CLOUD_ARGS = (CYL, [0.51, 2.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0], 160.0, 0.3, 0.6, ANGLES, 0.1, 2.0, 20.0)


def test_oracle_reference_cases():
    z = ko.CriticalZone(*SCAN_ARGS)
    for k, (r, fwd, ok) in enumerate(laserscan_cases(), 1):
        assert ok(z.check(r, fwd)), f"laserscan case {k}"
    zc = ko.CriticalZone(*CLOUD_ARGS)
    for k, (pts, fwd, ok) in enumerate(cloud_cases(), 9):
        c = cloud(pts)
        n = len(pts)
        assert ok(zc.check_cloud(c, 16, n * 16, 1, n, 0, 4, 8, fwd)), f"cloud case {k}"


def test_oracle_rejects_bad_distances():
    with pytest.raises(ValueError):
        ko.CriticalZone(CYL, [0.5, 1.0], [0, 0, 0], [0, 0, 0, 1], 90.0, 0.6, 0.6, ANGLES, 0.0, 1.0, 10.0)


@pytest.mark.gpu
def test_gpu_reference_cases_abi_and_module():
    import kompass_hip as kh
    from kompass_cpp.types import RobotGeometry
    from kompass_cpp.utils import CriticalZoneChecker, CriticalZoneCheckerGPU

    z = kh.ZoneContext(*SCAN_ARGS)
    o = ko.CriticalZone(*SCAN_ARGS)
    np.testing.assert_array_equal(z.indices(True), o.indices(True))
    np.testing.assert_array_equal(z.indices(False), o.indices(False))
    for k, (r, fwd, ok) in enumerate(laserscan_cases(), 1):
        v = z.check(r, fwd)
        assert ok(v), f"laserscan case {k}: {v}"
        assert np.float32(v) == np.float32(o.check(r, fwd))
    zc, oc = kh.ZoneContext(*CLOUD_ARGS), ko.CriticalZone(*CLOUD_ARGS)
    for k, (pts, fwd, ok) in enumerate(cloud_cases(), 9):
        c, n = cloud(pts), len(pts)
        v = zc.check_cloud(c, 16, n * 16, 1, n, 0, 4, 8, fwd)
        assert ok(v), f"cloud case {k}: {v}"
        assert np.float32(v) == np.float32(oc.check_cloud(c, 16, n * 16, 1, n, 0, 4, 8, fwd))
    # the reference's Python surface (bindings_utils.cpp:47-73, bindings_gpu.cpp:40-68)
    for cls in (CriticalZoneChecker, CriticalZoneCheckerGPU):
        m = cls(input_type=CriticalZoneChecker.InputType.LASERSCAN, robot_shape=RobotGeometry.CYLINDER,
                robot_dimensions=[0.51, 2.0], sensor_position_body=np.array([0.22, 0.0, 0.4], np.float32),
                sensor_rotation_body=np.array([0, 0, 0.99, 0.0], np.float32), critical_angle=160.0,
                critical_distance=0.3, slowdown_distance=0.6, scan_angles=list(ANGLES), min_height=0.1,
                max_height=2.0, range_max=20.0)
        for k, (r, fwd, ok) in enumerate(laserscan_cases(), 1):
            assert ok(m.check(ranges=list(r), forward=fwd)), f"{cls.__name__} case {k}"
        mc = cls(input_type=CriticalZoneChecker.InputType.POINTCLOUD, robot_shape=RobotGeometry.CYLINDER,
                 robot_dimensions=[0.51, 2.0], sensor_position_body=np.zeros(3, np.float32),
                 sensor_rotation_body=np.array([0, 0, 0, 1.0], np.float32), critical_angle=160.0,
                 critical_distance=0.3, slowdown_distance=0.6, scan_angles=list(ANGLES), min_height=0.1,
                 max_height=2.0, range_max=20.0)
        for k, (pts, fwd, ok) in enumerate(cloud_cases(), 9):
            c, n = cloud(pts), len(pts)
            assert ok(mc.check(data=c, point_step=16, row_step=n * 16, height=1, width=n, x_offset=0, y_offset=4,
                               z_offset=8, forward=fwd)), f"{cls.__name__} cloud case {k}"
    with pytest.raises((ValueError, kh.KompassHipError)):
        kh.ZoneContext(CYL, [0.5, 1.0], [0, 0, 0], [0, 0, 0, 1], 90.0, 0.6, 0.6, ANGLES, 0.0, 1.0, 10.0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dims", [(CYL, [0.3, 1.0]), (BOX, [0.6, 0.4, 1.0]), (SPH, [0.35])])
def test_gpu_random_scans_and_clouds_match_oracle(shape, dims):
    import kompass_hip as kh

    rng = np.random.default_rng(17 + shape)
    n = 720
    angles = np.sort(rng.uniform(0, 2 * np.pi, n))
    yaw = 0.4
    args = (shape, dims, [0.1, -0.05, 0.3], [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)], 120.0, 0.2, 0.9,
            angles, 0.05, 1.5, 8.0)
    z, o = kh.ZoneContext(*args), ko.CriticalZone(*args)
    np.testing.assert_array_equal(z.indices(True), o.indices(True))
    np.testing.assert_array_equal(z.indices(False), o.indices(False))
    seen = set()
    for trial in range(60):
        lo = [0.05, 0.4, 0.7, 1.5][trial % 4]
        r = rng.uniform(lo, lo + 2.0, n)
        if trial % 7 == 0:
            r[rng.integers(0, n, 5)] = np.nan
        for fwd in (True, False):
            want, got = o.check(r, fwd), z.check(r, fwd)
            assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)
            seen.add("stop" if want == 0 else "clear" if want == 1 else "slow")
    assert seen == {"stop", "clear", "slow"}
    for trial in range(10):
        m = 3000
        d = [0.3, 0.8, 1.2, 3.0][trial % 4]
        pts = np.column_stack([rng.uniform(-1, 1, m) * (d + 2), rng.uniform(-1, 1, m) * (d + 2), rng.uniform(-0.5, 2.0, m)])
        pts = pts[np.hypot(pts[:, 0], pts[:, 1]) > d]
        c, k = cloud(pts), len(pts)
        for fwd in (True, False):
            want = o.check_cloud(c, 16, k * 16, 1, k, 0, 4, 8, fwd)
            got = z.check_cloud(c, 16, k * 16, 1, k, 0, 4, 8, fwd)
            assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)
