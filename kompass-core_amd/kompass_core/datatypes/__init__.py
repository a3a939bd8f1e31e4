from .laserscan import LaserScanData  # noqa: F401
