"""Class-level cycle with a velocity that wanders (what a closed loop does): kompass_cpp.control.DWA.
compute_velocity_commands over a bounded random walk of the current velocity -- the window lattice changes its index
pattern in every second cycle.  python tools/class_sweep.py [steps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_cpp
import synthetic as syn
from kompass_cpp.control import (DWA, ControlLimitsParams, LinearVelocityControlParams, AngularVelocityControlParams,
                                 ControlType, TrajectoryCostWeights)
from kompass_cpp.types import Path, Velocity2D, RobotGeometry

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lim = ControlLimitsParams(LinearVelocityControlParams(1.0, 2.0, 2.0), LinearVelocityControlParams(0.0, 0.0, 0.0),
                          AngularVelocityControlParams(2.0, 2.0, 3.0, 3.0))
w = TrajectoryCostWeights()
w.from_dict(dict(reference_path_distance_weight=1.0, goal_distance_weight=1.0, obstacles_distance_weight=1.0,
                 smoothness_weight=0.0, jerk_weight=0.0))
d = DWA(lim, ControlType.DIFFERENTIAL_DRIVE, 0.1, 5.0, 0.2, 91, 91, RobotGeometry.get("CYLINDER"), [0.1, 0.4],
        [0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0], 0.05, w, 1)
d.set_current_path(Path([[x, 0.0, 0.0] for x in np.arange(0.0, 12.01, 1.0)]))
pts = syn.scene_points("cfg2", "survey")
rng = np.random.default_rng(1)
vx, om, ts = 0.3, 0.0, []
for i in range(steps + 100):
    vx = float(np.clip(vx + rng.normal(0, 0.02), -0.2, 1.0))
    om = float(np.clip(om + rng.normal(0, 0.05), -1.0, 1.0))
    d.set_current_state(0.001 * (i % 7), 0.0, 0.0, vx)
    t0 = time.perf_counter()
    r = d.compute_velocity_commands(Velocity2D(vx, 0.0, om, 0.0), pts)
    if i >= 100:
        ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print("class-level, wandering velocity: cycles %d | us p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f | above 1.5 x p50: %d"
      % (len(ts), np.percentile(ts, 50), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), ts.mean(),
         int(np.sum(ts > 1.5 * np.percentile(ts, 50)))))
