from .filters import Filters  # noqa: F401
