"""``Filters`` (src/astrild/rays/utils/filters.py): Gaussian low/high-pass on the GPU.

``Filters.gaussian`` wraps lenstools' ``ConvergenceMap.smooth`` in the reference
(filters.py:181-225): sigma_px = sigma * npix / theta; real-space
scipy ``gaussian_filter`` below 500 px, periodic FFT filter from 500 px on."""
import numpy as np

from ... import lensing
from .._resident import is_device_map, to_device, to_host
from .._units import angle_value


def _out(t, like):
    """A device result in the form of the input: CUDA tensor for a CUDA tensor (SkyArray's resident maps: no PCIe hop between
    the kernels of a chain), numpy array otherwise."""
    return t if is_device_map(like) else to_host(t)


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
        t = to_device(img).clone() if is_device_map(img) else to_device(np.array(img, dtype=np.float64))     # (smoothed in place)
        lensing.smooth_plan(npix).gaussian(t, sigma_px, kind)
        return _out(t, img)

    @staticmethod
    def gaussian_high_pass(img: np.ndarray, theta, theta_i=None, fwhm_i=None) -> np.ndarray:
        """img - gaussian(img)  (filters.py:228-249)."""
        return img - Filters.gaussian(img, theta, theta_i, fwhm_i)

    @staticmethod
    def _dgd(img, theta, theta_i, direction, order):
        from ... import _lib
        from ...device import ptr, stream
        import torch
        _npix = len(img)
        theta_deg = angle_value(theta, "deg", "deg")
        theta_i_pix = int(np.ceil(_npix * angle_value(theta_i, "deg", "deg") / theta_deg))
        t = to_device(img)
        out = torch.empty_like(t)
        work = torch.empty(2 * t.numel(), dtype=torch.float64, device=t.device)
        # theta_fov_deg / len(dist) with theta_fov_deg = theta * len(dist) / npix  (filters.py:349-354)
        h = (theta_deg * _npix / _npix) / _npix
        _lib.check(_lib.lib().ast_dgd_filter(ptr(t), ptr(out), ptr(work), _npix, float(theta_i_pix), h,
                                            int(direction), order, stream()), "ast_dgd_filter")
        return _out(out, img)

    @staticmethod
    def gaussian_third_derivative(img: np.ndarray, theta, theta_i, direction: int) -> np.ndarray:
        """DGD3 dipole window times the image (filters.py:305-355).  theta, theta_i: Quantities or degrees."""
        return Filters._dgd(img, theta, theta_i, direction, 3)

    @staticmethod
    def gaussian_first_derivative(img: np.ndarray, theta, theta_i, direction: int) -> np.ndarray:
        """First-derivative Gaussian window times the image (filters.py:358-400)."""
        return Filters._dgd(img, theta, theta_i, direction, 1)

    @staticmethod
    def gaussian_field(theta: np.ndarray, sigma) -> np.ndarray:
        """filters.py:403-413 (plain formula; the windows above evaluate it on the GPU)."""
        return np.exp(-theta ** 2 / (2 * sigma ** 2)) / (2 * np.pi * sigma ** 2)

    @staticmethod
    def apodization(img: np.ndarray, theta=None, theta_i=None, r200=None, suppress_radius=None) -> np.ndarray:
        """img * outer(hann, hann)  (filters.py:150-178; the extra arguments are unused there too)."""
        from ... import _lib
        from ...device import ptr, stream
        import torch
        t = to_device(img)
        out = torch.empty_like(t)
        _lib.check(_lib.lib().ast_hann_apodize(ptr(t), ptr(out), len(img), stream()), "ast_hann_apodize")
        return _out(out, img)

    @staticmethod
    def gaussian_third_derivative_convolution(img: np.ndarray, theta, theta_i, direction=1) -> np.ndarray:
        """Third-derivative Gaussians at s/2, s, 2s combined as g1 - g2 + g3, each with scipy's
        ``gaussian_filter(order=3*direction, mode="nearest")`` semantics (filters.py:260-304).
        direction: int or per-axis sequence."""
        from ... import _lib
        from ...device import ptr, stream
        import torch
        _npix = img.shape[0]
        s_pix = int(np.ceil(_npix * angle_value(theta_i, "deg", "deg") / angle_value(theta, "deg", "deg")))
        order = 3 * np.asarray(direction)
        o0, o1 = (int(order), int(order)) if order.ndim == 0 else (int(order[0]), int(order[1]))
        t = to_device(img)
        acc = None
        for fac, sign in ((0.5, 1.0), (1.0, -1.0), (2.0, 1.0)):
            sigma = s_pix * fac
            radius = int(4.0 * sigma + 0.5)
            nwork = t.numel() + 2 * (2 * radius + 1)
            work = torch.empty(nwork, dtype=torch.float64, device=t.device)
            out = torch.empty_like(t)
            _lib.check(_lib.lib().ast_gaussian_filter_order(ptr(t), ptr(out), ptr(work), nwork, _npix, float(sigma),
                                                           o0, o1, 1, stream()), "ast_gaussian_filter_order")
            # gauss_1 - gauss_2 + gauss_3 in that order
            acc = out if acc is None else (acc - out if sign < 0 else acc + out)
        return _out(acc, img)

    @staticmethod
    def gaussian_compensated(img: np.ndarray, theta, theta_i, theta_o) -> np.ndarray:
        """Compensated Gaussian of arxiv:1907.06657 Eq. 16 convolved with the image (filters.py:415-459)."""
        from ... import _lib
        from ...device import ptr, stream
        import torch
        pw = angle_value(theta, "deg", "deg") / img.shape[0]
        t_i = angle_value(theta_i, "deg", "deg") / pw
        t_o = angle_value(theta_o, "deg", "deg") / pw
        t_o_int = int(np.ceil(t_o))
        y, x = np.ogrid[-t_o_int:t_o_int, -t_o_int:t_o_int]
        dist = np.sqrt(x ** 2 + y ** 2)
        xx, x_o = dist / t_i, t_o / t_i
        window = (np.exp(-xx ** 2.0) / (np.pi * t_i ** 2.0)) - ((1.0 - np.exp(-x_o ** 2.0)) / (np.pi * t_o ** 2.0))
        window[t_o < dist] = 0
        t = to_device(img)
        w = to_device(window)
        out = torch.empty_like(t)
        _lib.check(_lib.lib().ast_convolve2d(ptr(t), ptr(w), ptr(out), img.shape[0], w.shape[0], w.shape[1],
                                             stream()), "ast_convolve2d")
        return _out(out, img)

    @staticmethod
    def aperture_photometry(img: np.ndarray, theta, alpha) -> np.ndarray:
        """img minus its mean over the ring alpha < r < sqrt(2) alpha (filters.py:40-73).  Like the
        reference this also updates ``img`` in place when it is a float64 array."""
        from ... import _lib
        from ...device import ptr, stream
        import torch
        _npix = len(img)
        pix_per_deg = _npix / angle_value(theta, "deg", "deg")
        alpha_pix = int(np.ceil(angle_value(alpha, "deg", "deg") * pix_per_deg))
        t = to_device(img)
        work = torch.empty(2048, dtype=torch.float64, device=t.device)
        _lib.check(_lib.lib().ast_aperture_photometry(ptr(t), ptr(t), ptr(work), _npix, float(alpha_pix), stream()),
                   "ast_aperture_photometry")
        if is_device_map(img):                        # a resident float64 map is updated in place, like the array below
            if t.data_ptr() != img.data_ptr():
                img.copy_(t)
            return img
        res = to_host(t)
        if isinstance(img, np.ndarray) and img.dtype == np.float64:
            img[...] = res
            return img
        return res

    @staticmethod
    def tophat_compensated(rad_obj, obj_posx, obj_posy, mapp, alpha, Nbins: int = 10):
        """Compensated top-hat of ONE object on a flat-sky map (filters.py:461-524; best alpha 0.6-0.7): the pixels within
        sqrt(2) alpha rad_obj of (obj_posx, obj_posy) are summed in ``Nbins`` annuli of the scaled radius; the mean of the
        annulus sums inside the filter radius minus the mean of those between it and sqrt(2) times it.  The reference reads
        the number of annuli from an undefined global ``args["Nbins"]``; here it is an argument.  A few hundred pixels
        around one object, evaluated on the host like the reference (not part of the per-map pipeline)."""
        rad_filter = alpha * rad_obj
        extend = np.sqrt(2)
        rad_filter_sqrt2 = int(np.ceil(extend * rad_filter))
        delta_eta = extend / Nbins                       # annulus thickness in units of the filter radius
        pix = np.arange(-rad_filter_sqrt2, rad_filter_sqrt2)
        pix_xx, pix_yy = np.meshgrid(pix, pix)
        pix_dist = np.sqrt(pix_xx ** 2 + pix_yy ** 2) / rad_filter
        eta = (pix_dist / delta_eta).astype(int)
        keep = eta < Nbins
        pix_xx, pix_yy, eta = pix_xx[keep], pix_yy[keep], eta[keep]
        annulus_value = np.zeros(Nbins)
        # (the reference indexes the map with the meshgrid's x offsets on axis 0 and its y offsets on axis 1)
        np.add.at(annulus_value, eta, np.asarray(mapp)[obj_posy + pix_xx, obj_posx + pix_yy])
        middle = int(np.ceil(1 / delta_eta))
        white_hat = np.mean(annulus_value[:middle])      # 0 -> rad_filter
        black_hat = np.mean(annulus_value[middle:])      # rad_filter -> sqrt(2) rad_filter
        return white_hat - black_hat

    @staticmethod
    def sigma_to_fwhm(sigma: float) -> float:
        return sigma * (2 * np.sqrt(2 * np.log(2)))

    @staticmethod
    def fwhm_to_sigma(fwhm: float) -> float:
        return fwhm / (2 * np.sqrt(2 * np.log(2)))
