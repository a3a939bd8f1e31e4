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


@define
class GridData:
    width: int
    height: int
    occupancy: np.ndarray = field(default=None)


class LocalMapper:
    """laserscan -> egocentric occupancy grid (values -1 / 0 / 100, int32)."""

    def __init__(self, config: Optional[MapConfig] = None, scan_model_config=None):
        self.config = config or MapConfig()
        c = self.config
        self.grid_width = int(c.width / c.resolution)
        self.grid_height = int(c.height / c.resolution)
        self.grid_data = GridData(width=self.grid_width, height=self.grid_height,
                                  occupancy=np.full((self.grid_height, self.grid_width), -1, np.int32))
        self._mapper = None
        self._scan_size = 0
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
        try:
            from kompass_cpp.mapping import LocalMapperGPU

            self._mapper = LocalMapperGPU(**kw)
        except ImportError:
            from kompass_cpp.mapping import LocalMapper as _CppMapper

            self._mapper = _CppMapper(**kw)
        self._scan_size = scan_size

    def update_from_scan(self, robot_state: Optional[RobotState], laser_scan: LaserScanData) -> None:
        n = len(laser_scan.ranges)
        if self._mapper is None or n != self._scan_size:
            self._initialize_mapper(n, float(laser_scan.angle_increment), float(laser_scan.range_max))
        ranges = np.clip(np.asarray(laser_scan.ranges, dtype=float), 0.0, self.config.filter_limit)
        grid = self._mapper.scan_to_grid(angles=list(map(float, laser_scan.angles)), ranges=list(map(float, ranges)))
        self.grid_data.occupancy = np.copy(grid)
        self.processed = True

    @property
    def occupancy(self) -> np.ndarray:
        return self.grid_data.occupancy
