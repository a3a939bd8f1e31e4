"""Robot description helpers (subset of the reference's kompass_core/models.py
needed by the DWA harness, src/kompass_core/models.py:224-268, 656-725,
1138-1420)."""
from __future__ import annotations

import math
from enum import Enum
from typing import Optional

import numpy as np
from attrs import Factory, define, field, validators

import kompass_cpp


class RobotType(Enum):
    ACKERMANN = "ACKERMANN"
    DIFFERENTIAL_DRIVE = "DIFFERENTIAL_DRIVE"
    OMNI = "OMNI"

    @classmethod
    def to_kompass_cpp_lib(cls, value) -> "kompass_cpp.control.ControlType":
        name = value.value if isinstance(value, RobotType) else str(value)
        return getattr(kompass_cpp.control.ControlType, name)


class RobotGeometry:
    class Type(Enum):
        BOX = "BOX"
        CYLINDER = "CYLINDER"
        SPHERE = "SPHERE"

        @classmethod
        def to_kompass_cpp_lib(cls, value) -> "kompass_cpp.types.RobotGeometry":
            name = value.value if isinstance(value, RobotGeometry.Type) else str(value)
            return kompass_cpp.types.RobotGeometry.get(name)

    _NPARAMS = {"BOX": 3, "CYLINDER": 2, "SPHERE": 1}

    @classmethod
    def is_valid_parameters(cls, geometry_type, parameters) -> bool:
        p = np.asarray(parameters, dtype=float)
        return p.size == cls._NPARAMS[geometry_type.value] and bool((p > 0).all())

    @classmethod
    def get_radius(cls, geometry_type, parameters) -> float:
        p = np.asarray(parameters, dtype=float)
        if geometry_type == cls.Type.BOX:
            return float(math.hypot(p[0], p[1]) / 2)
        return float(p[0])


@define(kw_only=True)
class LinearCtrlLimits:
    max_vel: float = field(validator=validators.ge(0.0))
    max_acc: float = field(validator=validators.ge(0.0))
    max_decel: float = field(validator=validators.ge(0.0))
    min_absolute_val: float = field(default=0.01, validator=validators.ge(0.0))


@define(kw_only=True)
class AngularCtrlLimits:
    max_vel: float = field(validator=validators.ge(0.0))
    max_steer: float = field(validator=validators.ge(0.0))
    max_acc: float = field(validator=validators.ge(0.0))
    max_decel: float = field(validator=validators.ge(0.0))
    min_absolute_val: float = field(default=0.01, validator=validators.ge(0.0))


@define(kw_only=True)
class RobotCtrlLimits:
    vx_limits: LinearCtrlLimits = field()
    omega_limits: AngularCtrlLimits = field()
    vy_limits: LinearCtrlLimits = field(default=LinearCtrlLimits(max_vel=0.0, max_acc=0.0, max_decel=0.0))

    @staticmethod
    def _lin(l: LinearCtrlLimits):
        return kompass_cpp.control.LinearVelocityControlParams(max_vel=l.max_vel, max_acc=l.max_acc,
                                                               max_decel=l.max_decel)

    def to_kompass_cpp_lib(self) -> "kompass_cpp.control.ControlLimitsParams":
        w = self.omega_limits
        return kompass_cpp.control.ControlLimitsParams(
            vel_x_ctr_params=self._lin(self.vx_limits),
            vel_y_ctr_params=self._lin(self.vy_limits),
            omega_ctr_params=kompass_cpp.control.AngularVelocityControlParams(
                max_omega=w.max_vel, max_ang=w.max_steer, max_acc=w.max_acc, max_decel=w.max_decel),
        )


@define
class RobotState:
    """Pose + body velocities; `simulate` applies the ideal kinematic model."""
    x: float = 0.0
    y: float = 0.0
    yaw: float = 0.0
    speed: float = 0.0
    vx: float = 0.0
    vy: float = 0.0
    omega: float = 0.0

    def simulate(self, v_x: float, omega: float, dt: float, v_y: float = 0.0) -> None:
        nx = self.x + (v_x * math.cos(self.yaw) - v_y * math.sin(self.yaw)) * dt
        ny = self.y + (v_x * math.sin(self.yaw) + v_y * math.cos(self.yaw)) * dt
        self.speed = math.hypot(nx - self.x, ny - self.y) * (1.0 if v_x >= 0 else -1.0)
        self.x, self.y = nx, ny
        self.yaw = self.yaw + omega * dt
        self.vx, self.vy, self.omega = v_x, v_y, omega


@define(kw_only=True)
class Robot:
    robot_type: RobotType = field()
    geometry_type: RobotGeometry.Type = field()
    geometry_params: np.ndarray = field()
    state: RobotState = field(default=Factory(RobotState))
    _ctrl: tuple = field(init=False, default=(0.0, 0.0, 0.0))

    def __attrs_post_init__(self):
        if not RobotGeometry.is_valid_parameters(self.geometry_type, self.geometry_params):
            raise ValueError(f"invalid geometry parameters {self.geometry_params} for {self.geometry_type}")

    @property
    def radius(self) -> float:
        return RobotGeometry.get_radius(self.geometry_type, self.geometry_params)

    def set_state(self, x: float, y: float, yaw: float, speed: float) -> None:
        self.state = RobotState(x=x, y=y, yaw=yaw, speed=speed)

    def set_control(self, velocity_x: float = 0.0, velocity_y: float = 0.0, omega: float = 0.0) -> None:
        self._ctrl = (float(velocity_x), float(velocity_y), float(omega))

    def get_state(self, dt: float) -> RobotState:
        vx, vy, om = self._ctrl
        self.state.simulate(v_x=vx, v_y=vy, omega=om, dt=dt)
        return self.state
