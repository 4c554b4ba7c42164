from .power_spectrum_3d import PowerSpectrum3D, PowerSpectrum3DWarning  # noqa: F401
from .angular_power_spectrum import AngularPowerSpectrum, PowerSpectrum2DWarning  # noqa: F401
