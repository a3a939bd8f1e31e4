"""Class-level cycle with LaserScan input: kompass_cpp.control.DWA.compute_velocity_commands(vel, LaserScan),
cfg2-sized lattice, scans of 360 / 1440 / 4096 beams (ranges change every call, angles do not)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_cpp
from kompass_cpp.control import (DWA, ControlLimitsParams, LinearVelocityControlParams, AngularVelocityControlParams,
                                 ControlType, TrajectoryCostWeights)
from kompass_cpp.types import Path, Velocity2D, RobotGeometry, LaserScan

lim = ControlLimitsParams(LinearVelocityControlParams(1.0, 2.0, 2.0), LinearVelocityControlParams(0.0, 0.0, 0.0),
                          AngularVelocityControlParams(2.0, 2.0, 3.0, 3.0))
w = TrajectoryCostWeights()
w.from_dict(dict(reference_path_distance_weight=1.0, goal_distance_weight=1.0, obstacles_distance_weight=1.0,
                 smoothness_weight=0.0, jerk_weight=0.0))
d = DWA(lim, ControlType.DIFFERENTIAL_DRIVE, 0.1, 5.0, 0.2, 91, 91, RobotGeometry.get("CYLINDER"), [0.1, 0.4],
        [0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0], 0.05, w, 1)
d.set_current_path(Path([[x, 0.0, 0.0] for x in np.arange(0.0, 12.01, 1.0)]))
for beams in (360, 1440, 4096):
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    base = 4.0 + 1.5 * np.cos(5 * ang)
    ts, tc = [], []
    for i in range(400):
        rng = base + 0.01 * (i % 9)
        d.set_current_state(0.001 * (i % 7), 0.0, 0.0, 0.5)
        t0 = time.perf_counter()
        s = LaserScan(ranges=rng, angles=ang)
        t1 = time.perf_counter()
        r = d.compute_velocity_commands(Velocity2D(0.5, 0.0, 0.001 * (i % 5), 0.0), s)
        t2 = time.perf_counter()
        tc.append(t1 - t0); ts.append(t2 - t1)
    print(f"{beams:5d} beams: LaserScan(...) {np.median(tc[50:]) * 1e6:5.1f} us, compute_velocity_commands p50 {np.median(ts[50:]) * 1e6:6.1f} us, found {r.is_found}")
