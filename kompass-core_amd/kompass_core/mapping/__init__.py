from .local_mapper import LocalMapper, MapConfig, ScanModelConfig  # noqa: F401
