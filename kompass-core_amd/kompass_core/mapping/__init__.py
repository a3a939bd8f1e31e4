from .local_mapper import LocalMapper, MapConfig  # noqa: F401
