from ._trajectory_ import TrajectoryCostsWeights  # noqa: F401
from ._base_ import FollowerConfig, FollowerTemplate  # noqa: F401
from .dwa import DWA, DWAConfig  # noqa: F401
from kompass_cpp.types import PathInterpolationType  # noqa: F401
