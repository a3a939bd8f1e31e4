"""kompass_core front-end subset for the MI355X build.

Mirrors the Python layer of the reference that sits on the accelerated hot path
(`kompass_core.control.DWA`, `kompass_core.mapping.LocalMapper` and the model /
datatype helpers their harness uses); everything else of kompass_core (other
controllers, vision, OMPL, calibration ...) is out of scope.
"""
import kompass_cpp  # noqa: F401  (the compiled module; fails loudly if not built)

from . import control, datatypes, mapping, models  # noqa: F401
