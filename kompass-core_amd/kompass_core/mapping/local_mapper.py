"""LocalMapper front-end (reference: src/kompass_core/mapping/local_mapper.py:
30-347): MapConfig + update_from_scan on top of kompass_cpp.mapping."""
from __future__ import annotations

from typing import Optional

import numpy as np
from attrs import define, field

from ..datatypes.laserscan import LaserScanData
from ..models import RobotState


@define
class MapConfig:
    width: float = 3.0           # metres
    height: float = 3.0          # metres
    resolution: float = 0.05     # metres / cell
    padding: float = 0.0
    filter_limit: float = 20.0   # ranges are clipped to [0, filter_limit]
    baysian_update: bool = False  # local_mapper.py:80


@define
class ScanModelConfig:
    """Inverse sensor model of the Bayesian update (reference:
    datatypes/scan_model.py:40-80, same defaults; p_empty = 1 - p_occupied)."""
    p_prior: float = 0.6
    p_occupied: float = 0.9
    range_sure: float = 0.1
    range_max: float = 20.0
    wall_size: float = 0.1
    angle_step: float = 0.01
    max_height: float = 10.0
    min_height: float = -10.0
    p_empty: float = field(init=False)

    def __attrs_post_init__(self):
        self.p_empty = 1.0 - self.p_occupied


@define
class GridData:
    width: int
    height: int
    occupancy: np.ndarray = field(default=None)
    occupancy_prob: np.ndarray = field(default=None)


class LocalMapper:
    """laserscan -> egocentric occupancy grid (values -1 / 0 / 100, int32)."""

    def __init__(self, config: Optional[MapConfig] = None, scan_model_config=None):
        self.config = config or MapConfig()
        c = self.config
        self.grid_width = int(c.width / c.resolution)
        self.grid_height = int(c.height / c.resolution)
        self.scan_model = scan_model_config or ScanModelConfig()
        self.grid_data = GridData(width=self.grid_width, height=self.grid_height,
                                  occupancy=np.full((self.grid_height, self.grid_width), -1, np.int32),
                                  occupancy_prob=np.full((self.grid_height, self.grid_width), -1, np.int32))
        self._mapper = None
        self._scan_size = 0
        self._previous_state: Optional[RobotState] = None
        self.scan_occupancy_prob = None           # float layer of the last Bayesian scan
        self.previous_grid_prob_transformed = None
        self.processed = False

    def _initialize_mapper(self, scan_size: int, angle_step: float, range_max: float):
        """Prefers the device class like the reference (local_mapper.py:189-222);
        in this build both classes run on the MI355X."""
        c = self.config
        max_points_per_line = int(1.5 * c.filter_limit / c.resolution) + 1
        kw = dict(grid_height=self.grid_height, grid_width=self.grid_width, resolution=c.resolution,
                  laserscan_position=[0.0, 0.0, 0.0], laserscan_orientation=0.0, is_pointcloud=False,
                  scan_size=scan_size, angle_step=angle_step, max_height=10.0, min_height=-10.0,
                  range_max=range_max, max_points_per_line=max_points_per_line)
        if c.baysian_update:
            # the Bayesian ctor exists on LocalMapper only (bindings_mapping.cpp:31-40)
            from kompass_cpp.mapping import LocalMapper as _CppMapper

            sm = self.scan_model
            kw.pop("range_max")
            self._mapper = _CppMapper(**kw, p_prior=sm.p_prior, p_occupied=sm.p_occupied, p_empty=sm.p_empty,
                                      range_sure=sm.range_sure, range_max=sm.range_max, wall_size=sm.wall_size)
            self._scan_size = scan_size
            return
        try:
            from kompass_cpp.mapping import LocalMapperGPU

            self._mapper = LocalMapperGPU(**kw)
        except ImportError:
            from kompass_cpp.mapping import LocalMapper as _CppMapper

            self._mapper = _CppMapper(**kw)
        self._scan_size = scan_size

    def _calculate_grid_shift(self, robot_state: RobotState) -> None:
        """Pose of the current robot frame in the previous one, then the warp of
        the previous probability grid (local_mapper.py:224-247)."""
        prev = self._previous_state
        dx, dy = robot_state.x - prev.x, robot_state.y - prev.y
        c, s = np.cos(prev.yaw), np.sin(prev.yaw)
        position = [c * dx + s * dy, -s * dx + c * dy]
        yaw = robot_state.yaw - prev.yaw
        self.previous_grid_prob_transformed = self._mapper.get_previous_grid_in_current_pose(
            current_position_in_previous_pose=position, current_orientation_in_previous_pose=float(yaw),
            unknown_value=self.scan_model.p_prior)

    def update_from_scan(self, robot_state: Optional[RobotState], laser_scan: LaserScanData) -> None:
        n = len(laser_scan.ranges)
        if self._mapper is None or n != self._scan_size:
            self._initialize_mapper(n, float(laser_scan.angle_increment), float(laser_scan.range_max))
        ranges = np.clip(np.asarray(laser_scan.ranges, dtype=float), 0.0, self.config.filter_limit)
        if self.config.baysian_update:
            # local_mapper.py:286-320
            if self.processed and robot_state is not None and self._previous_state is not None:
                self._calculate_grid_shift(robot_state)
            grid, prob = self._mapper.scan_to_grid_baysian(
                angles=np.ascontiguousarray(laser_scan.angles, dtype=np.float64),
                ranges=np.ascontiguousarray(ranges, dtype=np.float64))
            self.grid_data.occupancy = np.copy(grid)
            self.scan_occupancy_prob = np.copy(prob)
            p_prior = np.float32(self.scan_model.p_prior)
            layer = self.grid_data.occupancy_prob
            layer[prob > p_prior] = 100
            layer[prob == p_prior] = -1
            layer[prob < p_prior] = 0
            if robot_state is not None:
                self._previous_state = RobotState(x=robot_state.x, y=robot_state.y, yaw=robot_state.yaw)
            self.processed = True
            return
        grid = self._mapper.scan_to_grid(angles=np.ascontiguousarray(laser_scan.angles, dtype=np.float64),
                                         ranges=np.ascontiguousarray(ranges, dtype=np.float64))
        self.grid_data.occupancy = np.copy(grid)
        self.processed = True

    @property
    def occupancy(self) -> np.ndarray:
        return self.grid_data.occupancy

    @property
    def probabilistic_occupancy(self) -> np.ndarray:
        return self.grid_data.occupancy_prob
