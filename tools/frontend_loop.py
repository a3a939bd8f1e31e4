"""A closed loop through the Python front-end (kompass_core.control.DWA.loop_step, the class the reference's users
drive): per-step time of the whole call -- attrs state, path tracking, the C++ controller, the device cycle -- with the
reference's default window (20 x 20 samples) and a cfg2-sized one, point-cloud input.  python tools/frontend_loop.py"""
import os, sys, time, math
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import synthetic as syn
from kompass_core.control import DWA, DWAConfig, TrajectoryCostsWeights
from kompass_core.models import (AngularCtrlLimits, LinearCtrlLimits, Robot, RobotCtrlLimits, RobotGeometry, RobotType)


class _P:
    def __init__(self, pts):
        mk = lambda x, y: type("Pose", (), {"pose": type("I", (), {"position": type("Pt", (), {"x": x, "y": y})()})()})()
        self.poses = [mk(float(x), float(y)) for x, y in pts]


pts = syn.scene_points("cfg2", "survey")
for L, A, P in ((20, 20, 20), (91, 91, 50)):
    cfg = DWAConfig(max_linear_samples=L, max_angular_samples=A, octree_resolution=0.05,
                    costs_weights=TrajectoryCostsWeights(reference_path_distance_weight=1.0, goal_distance_weight=1.0,
                                                         obstacles_distance_weight=1.0, smoothness_weight=0.0, jerk_weight=0.0),
                    prediction_horizon=P, control_horizon=2, control_time_step=0.1)
    robot = Robot(robot_type=RobotType.DIFFERENTIAL_DRIVE, geometry_type=RobotGeometry.Type.CYLINDER, geometry_params=np.array([0.1, 0.4]))
    limits = RobotCtrlLimits(vx_limits=LinearCtrlLimits(max_vel=1.0, max_acc=2.0, max_decel=2.0),
                             omega_limits=AngularCtrlLimits(max_vel=2.0, max_acc=3.0, max_decel=3.0, max_steer=2.0),
                             vy_limits=LinearCtrlLimits(max_vel=0.0, max_acc=0.0, max_decel=0.0))
    ctl = DWA(robot=robot, ctrl_limits=limits, config=cfg)
    ctl.set_path(_P([(x, 0.0) for x in np.arange(0.0, 12.01, 0.5)]))
    robot.state.x, robot.state.y, robot.state.yaw = 0.0, 0.0, 0.0
    ts = []
    for i in range(600):
        t = time.perf_counter()
        ok = ctl.loop_step(current_state=robot.state, laser_scan=None, local_map=pts)
        ts.append(time.perf_counter() - t)
        if not ok or ctl.reached_end():
            break
        for vx, vy, om in zip(ctl.linear_x_control, ctl.linear_y_control, ctl.angular_control):
            robot.set_control(velocity_x=vx * 0.05, velocity_y=vy, omega=om * 0.05)   # (a slow drive: the loop stays inside the scene)
            robot.get_state(dt=0.1)
    ts = np.array(ts[50:]) * 1e6
    print("window %d x %d, horizon %d: %d loop steps, us p50 %.1f p90 %.1f mean %.1f" % (L, A, P, len(ts), np.percentile(ts, 50), np.percentile(ts, 90), ts.mean()))
