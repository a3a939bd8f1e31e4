"""DWA front-end (reference: src/kompass_core/control/dwa.py).  Same config
fields, constructor and loop_step contract; the planner object is
`kompass_cpp.control.DWA` whose cycle runs on the MI355X."""
from __future__ import annotations

import logging
from typing import List, Optional, Union

import numpy as np
from attrs import Factory, define, field, validators

import kompass_cpp
from ..datatypes.laserscan import LaserScanData
from ..models import Robot, RobotCtrlLimits, RobotGeometry, RobotState, RobotType
from ._base_ import FollowerConfig, FollowerTemplate
from ._trajectory_ import TrajectoryCostsWeights


def _rng(lo, hi):
    return [validators.ge(lo), validators.le(hi)]


@define
class DWAConfig(FollowerConfig):
    control_time_step: float = field(default=0.1, validator=_rng(1e-4, 1e6))
    control_horizon: int = field(default=2, validator=_rng(1, 1000))        # steps
    prediction_horizon: int = field(default=10, validator=_rng(1, 1000))    # steps
    max_linear_samples: int = field(default=20, validator=_rng(1, 1e3))
    max_angular_samples: int = field(default=20, validator=_rng(1, 1e3))
    proximity_sensor_position_to_robot: np.ndarray = field(default=np.array([0.0, 0.0, 0.0], dtype=np.float32))
    proximity_sensor_rotation_to_robot: np.ndarray = field(default=np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32))
    octree_resolution: float = field(default=0.1, validator=_rng(1e-9, 1e3))
    costs_weights: TrajectoryCostsWeights = Factory(TrajectoryCostsWeights)
    max_num_threads: int = field(default=1, validator=_rng(1, 1e2))
    drop_samples: bool = field(default=True)

    def __attrs_post_init__(self):
        if self.control_horizon > self.prediction_horizon:
            logging.error("Control horizon cannot exceed the prediction horizon; clamping")
            self.control_horizon = self.prediction_horizon


class DWA(FollowerTemplate):
    """Dynamic Window Approach local planner (sampling, roll-out, collision gate,
    weighted cost, argmin as one device cycle)."""

    def __init__(self, robot: Robot, ctrl_limits: RobotCtrlLimits, config: Optional[DWAConfig] = None,
                 control_time_step: Optional[float] = None, **_):
        self._config = config or DWAConfig()
        config = self._config
        if control_time_step:
            config.control_time_step = control_time_step
        self._got_path = False
        self._planner = kompass_cpp.control.DWA(
            control_limits=ctrl_limits.to_kompass_cpp_lib(),
            control_type=RobotType.to_kompass_cpp_lib(robot.robot_type),
            time_step=config.control_time_step,
            prediction_horizon=config.prediction_horizon * config.control_time_step,
            control_horizon=config.control_horizon * config.control_time_step,
            max_linear_samples=config.max_linear_samples,
            max_angular_samples=config.max_angular_samples,
            robot_shape_type=RobotGeometry.Type.to_kompass_cpp_lib(robot.geometry_type),
            robot_dimensions=[float(v) for v in robot.geometry_params],
            sensor_position_robot=config.proximity_sensor_position_to_robot,
            sensor_rotation_robot=config.proximity_sensor_rotation_to_robot,
            octree_resolution=config.octree_resolution,
            cost_weights=config.costs_weights.to_kompass_cpp(),
            max_num_threads=config.max_num_threads,
        )
        self._result = kompass_cpp.control.SamplingControlResult()
        self._end_of_ctrl_horizon: int = max(config.control_horizon, 1)
        logging.info("DWA PATH CONTROLLER IS READY")

    @property
    def planner(self) -> "kompass_cpp.control.Follower":
        return self._planner

    def loop_step(self, *, current_state: RobotState, laser_scan: Optional[LaserScanData] = None,
                  point_cloud=None, local_map: Optional[np.ndarray] = None,
                  local_map_resolution: Optional[float] = None, debug: bool = False, **_) -> bool:
        if not self._got_path:
            logging.error("Path is not available to DWA controller")
            return False
        self._planner.set_current_state(current_state.x, current_state.y, current_state.yaw, current_state.speed)
        if local_map_resolution:
            self._planner.set_resolution(local_map_resolution)
        if self.reached_end():
            logging.info("End is reached")
            self._result.is_found = False
            return False
        vel = kompass_cpp.types.Velocity2D(vx=current_state.vx, vy=current_state.vy, omega=current_state.omega)
        if isinstance(local_map, kompass_cpp.mapping.LocalMapper):
            # not in the reference: the mapper's last grid is consumed where it
            # lies on the device (OCCUPIED cells -> point list)
            sensor = local_map
        elif local_map is not None:
            sensor = np.asarray(local_map, dtype=np.float32)
        elif laser_scan is not None:
            if len(laser_scan.angles) != len(laser_scan.ranges):
                logging.error("Received incompatible LaserScan data -> Cannot compute control")
                return False
            # (float64 arrays: one copy each into the C++ vectors instead of a Python float per beam)
            sensor = kompass_cpp.types.LaserScan(ranges=np.ascontiguousarray(laser_scan.ranges, dtype=np.float64),
                                                 angles=np.ascontiguousarray(laser_scan.angles, dtype=np.float64))
        elif point_cloud is not None:
            sensor = np.asarray(getattr(point_cloud, "data", point_cloud), dtype=np.float32)
        else:
            logging.error("Cannot compute control without sensor data. Provide 'laser_scan' or 'point_cloud' input")
            return False
        try:
            if debug and not isinstance(sensor, kompass_cpp.mapping.LocalMapper):
                self._planner.debug_velocity_search(vel, sensor, self._config.drop_samples)
            self._result = self._planner.compute_velocity_commands(vel, sensor)
        except Exception as e:  # reference: log and report "no control"
            logging.error(f"Could not find velocity command: {e}")
            return False
        return True

    def has_result(self) -> bool:
        return self._result.is_found

    def logging_info(self) -> str:
        if self._result.is_found:
            return f"DWA Controller found trajectory with cost: {self._result.cost}"
        return "DWA Controller Failed to find a valid trajectory"

    @property
    def control_till_horizon(self):
        return self._result.trajectory.velocities if self._result.is_found else None

    def optimal_path(self):
        return self._result.trajectory.path if self._result.is_found else None

    @property
    def result_cost(self) -> Optional[float]:
        return self._result.cost if self._result.is_found else None

    @property
    def linear_x_control(self) -> Union[List[float], np.ndarray]:
        if self._result.is_found:
            return self.control_till_horizon.vx[: self._end_of_ctrl_horizon]
        return [0.0]

    @property
    def linear_y_control(self) -> Union[List[float], np.ndarray]:
        if self._result.is_found:
            return self.control_till_horizon.vy[: self._end_of_ctrl_horizon]
        return [0.0]

    @property
    def angular_control(self) -> Union[List[float], np.ndarray]:
        if self._result.is_found:
            return self.control_till_horizon.omega[: self._end_of_ctrl_horizon]
        return [0.0]
