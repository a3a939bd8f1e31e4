"""Class-level cycle time: kompass_cpp.control.DWA.compute_velocity_commands (lattice of the
current velocity + sensor update + tracked segment + device cycle + winner row), cfg2-sized lattice."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_cpp
import synthetic as syn
from kompass_cpp.control import (DWA, ControlLimitsParams, LinearVelocityControlParams, AngularVelocityControlParams,
                                 ControlType, TrajectoryCostWeights)
from kompass_cpp.types import Path, Velocity2D, RobotGeometry

L, A = (91, 91) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
lim = ControlLimitsParams(LinearVelocityControlParams(1.0, 2.0, 2.0), LinearVelocityControlParams(0.0, 0.0, 0.0),
                          AngularVelocityControlParams(2.0, 2.0, 3.0, 3.0))
w = TrajectoryCostWeights()
w.from_dict(dict(reference_path_distance_weight=1.0, goal_distance_weight=1.0, obstacles_distance_weight=1.0,
                 smoothness_weight=0.0, jerk_weight=0.0))
d = DWA(lim, ControlType.DIFFERENTIAL_DRIVE, 0.1, 5.0, 0.2, L, A, RobotGeometry.get("CYLINDER"), [0.1, 0.4],
        [0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0], 0.05, w, 1)
d.set_current_path(Path([[x, 0.0, 0.0] for x in np.arange(0.0, 12.01, 1.0)]))
for scene in ("survey", "mid", "open"):
    pts = syn.scene_points("cfg2", scene)
    cloud = [tuple(float(v) for v in p) for p in pts]
    ts = []
    for i in range(300):
        d.set_current_state(0.001 * (i % 7), 0.0, 0.0, 0.5)
        t0 = time.perf_counter()
        r = d.compute_velocity_commands(Velocity2D(0.5, 0.0, 0.001 * (i % 5), 0.0), pts)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[50:]) * 1e6
    print(f"{scene:7s} L={L} A={A}: class-level cycle p50 {np.percentile(ts, 50):.1f} us, mean {ts.mean():.1f}, found {r.is_found}")
