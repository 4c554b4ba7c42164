from .skymap import SkyMap, SkyMapWarning  # noqa: F401
from .rayramses import RayRamses, RayRamsesWarning  # noqa: F401
