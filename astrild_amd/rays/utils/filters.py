"""``Filters`` (src/astrild/rays/utils/filters.py): Gaussian low/high-pass on the GPU.

``Filters.gaussian`` wraps lenstools' ``ConvergenceMap.smooth`` in the reference
(filters.py:181-225): sigma_px = sigma * npix / theta; real-space
scipy ``gaussian_filter`` below 500 px, periodic FFT filter from 500 px on."""
import numpy as np

from ... import lensing
from ...device import as_device
from .._units import angle_value


class Filters:
    @staticmethod
    def gaussian(img: np.ndarray, theta, theta_i=None, fwhm_i=None, **kwargs) -> np.ndarray:
        """
        Args:
            img: partial sky-map
            theta: edge length of the field of view [deg or Quantity]
            theta_i: sigma of the Gaussian [arcmin]
            fwhm_i: FWHM of the Gaussian [arcmin or Quantity]
        """
        if theta_i is None and fwhm_i is None:
            raise ValueError("Either theta_i or fwhm_i must be set for smoothing scale.")
        if theta_i is None:
            sigma_i = Filters.fwhm_to_sigma(angle_value(fwhm_i, "arcmin", "arcmin"))
        else:
            sigma_i = angle_value(theta_i, "arcmin", "arcmin")
        npix = len(img)
        sigma_px = (sigma_i / 60.0) * npix / angle_value(theta, "deg", "deg")
        kind = "gaussian" if npix < 500 else "gaussianFFT"      # filters.py:215-224
        t = as_device(np.array(img, dtype=np.float64))
        lensing.smooth_plan(npix).gaussian(t, sigma_px, kind)
        return t.cpu().numpy()

    @staticmethod
    def gaussian_high_pass(img: np.ndarray, theta, theta_i=None, fwhm_i=None) -> np.ndarray:
        """img - gaussian(img)  (filters.py:228-249)."""
        return img - Filters.gaussian(img, theta, theta_i, fwhm_i)

    @staticmethod
    def sigma_to_fwhm(sigma: float) -> float:
        return sigma * (2 * np.sqrt(2 * np.log(2)))

    @staticmethod
    def fwhm_to_sigma(fwhm: float) -> float:
        return fwhm / (2 * np.sqrt(2 * np.log(2)))
