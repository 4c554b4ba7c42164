from .bispectrum_3d import Bispectrum3D, Bispectrum3DWarning  # noqa: F401
