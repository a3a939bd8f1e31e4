"""LaserScanData (subset of src/kompass_core/datatypes/laserscan.py:20-80):
defaults give 201 beams over [0, 2 pi] at range_max."""
from __future__ import annotations

import math

import numpy as np
from attrs import define, field


@define
class LaserScanData:
    angle_min: float = 0.0
    angle_max: float = 2 * math.pi
    angle_increment: float = 0.01 * math.pi
    time_increment: float = 1e-3
    scan_time: float = 1e-3
    range_min: float = 0.0
    range_max: float = 20.0
    ranges: np.ndarray = field(default=np.empty(0))
    angles: np.ndarray = field(default=np.empty(0))
    intensities: np.ndarray = field(default=np.empty(0))

    def __attrs_post_init__(self):
        self.ranges = np.asarray(self.ranges, dtype=float)
        self.angles = np.asarray(self.angles, dtype=float)
        if self.angles.size == 0:
            self.angles = np.arange(self.angle_min, self.angle_max + self.angle_increment, self.angle_increment)
        if self.ranges.size == 0:
            self.ranges = np.full(self.angles.size, self.range_max)
        n = min(self.angles.size, self.ranges.size)
        self.angles, self.ranges = self.angles[:n], self.ranges[:n]
