from .stats_subfind import SubFind  # noqa: F401
