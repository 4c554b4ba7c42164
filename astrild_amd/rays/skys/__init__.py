from .sky_array import SkyArray, SkyArrayWarning  # noqa: F401
from .sky_utils import SkyUtils  # noqa: F401
