"""``AngularPowerSpectrum.from_array`` with astrild's API
(src/astrild/power_spectra/angular_power_spectrum.py:22-53): the flat-sky C_l of a ``SkyArray`` map.
The 2D R2C transform and the annulus binning run on the GPU (``lensing.flat_power_spectrum``);
lenstools' ``ConvergenceMap.powerSpectrum`` semantics (bins (l_k, l_k+1], half-plane pixel mean,
(angle / npix^2)^2 normalisation) are restated - lenstools is un-vendored, parity unpinned.

Out of scope: ``from_healpix`` / ``from_namaster`` / ``create_healpix`` (HEALPix, NaMaster)."""
from typing import List, Union

import numpy as np

from .. import lensing



def _map_of(skymap, on):
    """The map as it lives: the CUDA tensor of a SkyArray's resident map (no PCIe hop), else the array."""
    dev_fn = getattr(skymap.data, "device", None)
    return dev_fn(on) if dev_fn is not None else skymap.data[on]


class PowerSpectrum2DWarning(BaseException):
    pass


class AngularPowerSpectrum:
    def __init__(self, l: np.array, P: np.array):
        self.ell = l
        self.P = P

    @classmethod
    def from_array(cls, skymap, on: str,
                   multipoles: Union[List[float], np.array] = np.arange(200.0, 50000.0, 200.0)) -> "AngularPowerSpectrum":
        """``skymap``: a SkyArray (``.data[on]``, ``.opening_angle`` in degrees); ``multipoles``: bin edges."""
        l, P = lensing.flat_power_spectrum(_map_of(skymap, on), skymap.opening_angle, np.asarray(multipoles, dtype=np.float64))
        return cls(l, P)
