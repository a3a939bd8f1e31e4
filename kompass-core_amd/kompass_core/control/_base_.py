"""Follower front-end base (reference: src/kompass_core/control/_base_.py:14-120,
209-390): path hand-over, goal check, tracked state and control getters."""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
from attrs import define, field

import kompass_cpp
from ..models import RobotState


@define
class FollowerConfig:
    # NOTE: the reference never forwards these to the C++ DWA (its constructor
    # takes none), so the C++ follower defaults apply
    max_point_interpolation_distance: float = 0.01
    lookahead_distance: float = 1.0
    goal_dist_tolerance: float = 0.1
    goal_orientation_tolerance: float = 0.1
    path_segment_length: float = 1.0
    loosing_goal_distance: float = 0.1


class FollowerTemplate:
    """Behaviour shared by the path followers; `planner` is the kompass_cpp object."""

    @property
    def planner(self) -> "kompass_cpp.control.Follower":
        raise NotImplementedError

    def reached_end(self) -> bool:
        return self.planner.is_goal_reached()

    def set_path(self, global_path, **_) -> None:
        """global_path: object with `.poses[i].pose.position.{x,y}` (nav_msgs/Path
        shape) or an (N, 2|3) array of points."""
        if hasattr(global_path, "poses"):
            pts = [[p.pose.position.x, p.pose.position.y, 0.0] for p in global_path.poses]
        else:
            a = np.asarray(global_path, dtype=float)
            pts = [[r[0], r[1], 0.0] for r in a]
        if len(pts) < 2:
            self.planner.clear_current_path()
            return
        self.planner.set_current_path(kompass_cpp.types.Path(points=np.asarray(pts, dtype=np.float32)))
        self._got_path = True

    @property
    def path(self) -> bool:
        return self.planner.has_path()

    @path.setter
    def path(self, global_path) -> None:
        self.set_path(global_path=global_path)

    def interpolated_path(self) -> Optional["kompass_cpp.types.Path"]:
        return self.planner.get_current_path()

    def set_interpolation_type(self, interpolation_type) -> None:
        self.planner.set_interpolation_type(interpolation_type)

    @property
    def tracked_state(self) -> Optional[RobotState]:
        if not self.planner.has_path():
            return None
        t = self.planner.get_tracked_target()
        return RobotState(x=t.movement.x, y=t.movement.y, yaw=t.movement.yaw)

    @property
    def distance_error(self) -> float:
        return self.planner.get_tracked_target().crosstrack_error

    @property
    def orientation_error(self) -> float:
        return self.planner.get_tracked_target().heading_error

    @property
    def linear_x_control(self) -> Union[List[float], np.ndarray]:
        return [self.planner.get_vx_cmd()]

    @property
    def linear_y_control(self) -> Union[List[float], np.ndarray]:
        return [self.planner.get_vy_cmd()]

    @property
    def angular_control(self) -> Union[List[float], np.ndarray]:
        return [self.planner.get_omega_cmd()]
