"""Cost weights of the sampling controllers (reference:
src/kompass_core/control/_trajectory_.py)."""
from attrs import asdict, define, field, validators

import kompass_cpp


def _w(default):
    return field(default=default, validator=[validators.ge(0.0), validators.le(1e3)])


@define
class TrajectoryCostsWeights:
    reference_path_distance_weight: float = _w(3.0)
    goal_distance_weight: float = _w(3.0)
    obstacles_distance_weight: float = _w(1.0)
    smoothness_weight: float = _w(0.0)
    jerk_weight: float = _w(0.0)

    def to_kompass_cpp(self) -> "kompass_cpp.control.TrajectoryCostWeights":
        out = kompass_cpp.control.TrajectoryCostWeights()
        out.from_dict({k: float(v) for k, v in asdict(self).items()})
        return out
