"""``SkyUtils`` (src/astrild/rays/skys/sky_utils.py): unit conversion and
kappa -> deflection / potential, on the GPU.

The reference binds ``libglsg.so`` with ctypes (sky_utils.py:402-435);
``libastrild_hip.so`` exports the same two symbols, and ``_call_alphas`` /
``_call_cal_phi`` below are the same few lines pointed at it."""
import ctypes as ct
from typing import Tuple

import numpy as np

from ... import _lib, lensing
from ...device import as_device
from .._units import angle_value

c_light = lensing.C_LIGHT_KMS


def _call_alphas(kappa: np.ndarray, npix: int, opening_angle: float) -> Tuple[np.ndarray, np.ndarray]:
    """sky_utils.py:411-419 against the HIP library's libglsg-compatible symbol."""
    gls = _lib.lib()
    _kappa = np.array(kappa, dtype=ct.c_double)
    alpha1 = np.array(np.zeros((npix, npix)), dtype=ct.c_double)
    alpha2 = np.array(np.zeros((npix, npix)), dtype=ct.c_double)
    gls.kappa0_to_alphas(_kappa.ctypes.data_as(ct.c_void_p), npix, opening_angle,
                         alpha1.ctypes.data_as(ct.c_void_p), alpha2.ctypes.data_as(ct.c_void_p))
    return alpha1, alpha2


def _call_cal_phi(kappa: np.ndarray, npix: int, opening_angle: float) -> np.ndarray:
    """sky_utils.py:430-435."""
    gls = _lib.lib()
    _kappa = np.array(kappa, dtype=ct.c_double)
    phi = np.array(np.zeros((npix, npix)), dtype=ct.c_double)
    gls.kappa0_to_phi(_kappa.ctypes.data_as(ct.c_void_p), npix, opening_angle, phi.ctypes.data_as(ct.c_void_p))
    return phi


class SkyUtils:
    @staticmethod
    def convert_code_to_phy_units(quantity: str, map_df):
        """RayRamses code units -> physical units, in place on the DataFrame column
        (sky_utils.py:318-339): / c^2 for shear/deflt/kappa_2, / c^3 for isw_rs."""
        if quantity in ["shear_x", "shear_y", "deflt_x", "deflt_y", "kappa_2", "isw_rs"]:
            col = as_device(np.ascontiguousarray(map_df[quantity].values, dtype=np.float64))
            map_df.loc[:, [quantity]] = lensing.convert_code_to_phy_units(quantity, col).cpu().numpy()[:, None]
        return map_df

    @staticmethod
    def convert_convergence_to_deflection_ctypes(kappa: np.ndarray, npix: int, opening_angle
                                                 ) -> Tuple[np.ndarray, np.ndarray]:
        """alpha1, alpha2 in units of opening_angle (sky_utils.py:366-385).
        opening_angle: astropy Quantity or degrees."""
        return _call_alphas(kappa, npix, angle_value(opening_angle, "rad", "deg"))

    @staticmethod
    def convert_convergence_to_potential(kappa: np.ndarray, npix: int, opening_angle) -> np.ndarray:
        return _call_cal_phi(kappa, npix, angle_value(opening_angle, "rad", "deg"))
