from .bispectrum_3d import Bispectrum3D, Bispectrum3DWarning  # noqa: F401
from .bispectrum_2d import Bispectrum2D, Bispectrum2DWarning  # noqa: F401
