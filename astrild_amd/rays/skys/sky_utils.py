"""``SkyUtils`` (src/astrild/rays/skys/sky_utils.py): unit conversion and
kappa -> deflection / potential, on the GPU.

The reference binds ``libglsg.so`` with ctypes (sky_utils.py:402-435);
``libastrild_hip.so`` exports the same two symbols, and ``_call_alphas`` /
``_call_cal_phi`` below are the same few lines pointed at it."""
import ctypes as ct
from typing import Tuple

import numpy as np

from ... import _lib, lensing
from ...device import as_device, to_numpy
from .._resident import is_device_map, to_device
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
    def analytic_Halo_signal_to_SkyArray(halo_idx, halo_cat, extent, direction, suppress, suppression_R,
                                         npix: int, signal: str = "dT") -> np.ndarray:
        """Sum of the analytic NFW stamps of the selected halos (sky_utils.py:79-137): one GPU
        launch over all halos instead of a Python loop (and joblib batches) of numpy stamps."""
        sel = {key: np.asarray(val)[np.asarray(halo_idx)] for key, val in halo_cat.items()}
        return to_numpy(lensing.nfw_paint(sel, extent, direction, suppress, suppression_R, npix, signal))

    @staticmethod
    def add_patch_to_map(limg: np.ndarray, simg: np.ndarray, cen_pix: tuple) -> np.ndarray:
        """sky_utils.py:140-173."""
        big = as_device(np.ascontiguousarray(limg, dtype=np.float64))
        small = as_device(np.ascontiguousarray(simg, dtype=np.float64))
        return to_numpy(lensing.add_patch(big, small, cen_pix))

    @staticmethod
    def _single_stamp(theta_200c, M_200c, c_200c, angu_diam_dist, npix, extent, direction, suppress,
                      suppression_R, signal, vel=(0.0, 0.0)):
        # a stamp of npix pixels centred on itself: the map IS the stamp
        cat = {"r200_deg": [theta_200c], "m200": [M_200c], "c_NFW": [c_200c], "Dc": [angu_diam_dist / 0.6774],
               "theta1_pix": [npix // 2], "theta2_pix": [npix // 2], "theta1_tv": [vel[0]], "theta2_tv": [vel[1]],
               "r200_pix": [(npix - 1) / (2.0 * extent)]}
        out = lensing.nfw_paint(cat, extent, direction, suppress, suppression_R, npix, signal)
        return to_numpy(out)

    @staticmethod
    def NFW_deflection_angle_map(theta_200c, M_200c, c_200c, angu_diam_dist, npix: int = 100, extent: float = 1,
                                 direction=(0,), suppress: bool = False, suppression_R: float = 1) -> np.ndarray:
        """sky_utils.py:214-282 (odd npix: the reference's stamps always are)."""
        assert np.sum(direction) <= 1, "Only 0 and 1 are valid direction indications."
        assert npix % 2 == 1, "stamp maps have an odd number of pixels"
        return SkyUtils._single_stamp(theta_200c, M_200c, c_200c, angu_diam_dist, npix, extent, direction,
                                      suppress, suppression_R, "alpha")

    @staticmethod
    def NFW_temperature_perturbation_map(theta_200c, M_200c, c_200c, vel, angu_diam_dist, npix: int = 100,
                                         extent: float = 1, direction=(0, 1), suppress: bool = False,
                                         suppression_R: float = 1) -> np.ndarray:
        """sky_utils.py:176-211."""
        assert npix % 2 == 1, "stamp maps have an odd number of pixels"
        return SkyUtils._single_stamp(theta_200c, M_200c, c_200c, angu_diam_dist, npix, extent, direction,
                                      suppress, suppression_R, "dT", vel)

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
        if is_device_map(kappa):
            # a map resident in HBM (SkyArray's MapStore): the same transform plan the host-pointer symbol above runs
            # between its two PCIe copies, CUDA tensors in and out
            return lensing.lens_plan(int(npix), angle_value(opening_angle, "rad", "deg")).alphas(to_device(kappa))
        return _call_alphas(kappa, npix, angle_value(opening_angle, "rad", "deg"))

    @staticmethod
    def convert_deflection_to_shear(alpha1: np.ndarray, alpha2: np.ndarray, npix: int, opening_angle
                                    ) -> Tuple[np.ndarray, np.ndarray]:
        """shear1, shear2 from the two deflection components (sky_utils.py:342-362).  The reference's body is marked TODO
        (`coord` is undefined there): np.gradient is taken at the uniform pixel spacing opening_angle / npix in radians
        (opening_angle: astropy Quantity or degrees), the unit convert_convergence_to_deflection_ctypes returns alpha in."""
        h = angle_value(opening_angle, "rad", "deg") / int(npix)
        g1, g2 = lensing.deflection_to_shear(alpha1, alpha2, h)
        if is_device_map(alpha1) and is_device_map(alpha2):
            return g1, g2
        return to_numpy(g1), to_numpy(g2)

    @staticmethod
    def convert_convergence_to_potential(kappa: np.ndarray, npix: int, opening_angle) -> np.ndarray:
        return _call_cal_phi(kappa, npix, angle_value(opening_angle, "rad", "deg"))
