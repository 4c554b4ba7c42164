"""``SkyArray`` (src/astrild/rays/skys/sky_array.py) for the kappa-map hot path:
unit conversion + reshape, PDF, filters, galaxy shape noise, kappa -> deflection,
crop / division / merge.  ``self.data`` hands out numpy arrays like the reference's dict;
the arithmetic runs on the GPU through libastrild_hip.so, and what a method produces
STAYS in HBM until somebody asks for the array (``rays/_resident.MapStore``): a chain
filter -> kappa->alpha -> shear -> pdf makes one upload instead of a PCIe round trip
per method (4096^2 float64: 134 MB each way, several times any of the kernels).

Not carried over (out of scope): create_cmb (broken in the reference,
sky_array.py:735-739)."""
import copy
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import pandas as pd

from ... import lensing
from ...device import as_device
from .._resident import MapStore, is_device_map, to_host
from ..skyio import SkyIO
from ..utils.filters import Filters
from .sky_utils import SkyUtils


class SkyArrayWarning(BaseException):
    pass


class SkyArray:
    def __init__(self, skymap: np.ndarray, opening_angle: float, quantity: str, dirs: Dict[str, str],
                 map_file: Optional[str] = None):
        self.data = MapStore({"orig": skymap})
        self._npix = skymap.shape[0]
        self._opening_angle = opening_angle
        self.quantity = quantity
        self.dirs = dirs
        self.map_file = map_file

    @classmethod
    def from_file(cls, map_file: str, opening_angle: float, quantity: str, dir_in: str,
                  npix: Optional[int] = None, convert_unit: bool = True) -> "SkyArray":
        assert map_file, "There is no file being pointed at"
        file_extension = map_file.split(".")[-1]
        if file_extension == "h5":
            map_df = pd.read_hdf(map_file, key="df")
            return cls.from_dataframe(map_df, opening_angle, quantity, dir_in, map_file, npix, convert_unit)
        elif file_extension == "npy":
            return cls.from_array(np.load(map_file), opening_angle, quantity, dir_in, map_file)
        raise SkyArrayWarning(f"file type .{file_extension} is not supported")

    @classmethod
    def from_dataframe(cls, map_df: pd.DataFrame, opening_angle: float, quantity: str, dir_in: str,
                       map_file: str, npix: Optional[int] = None, convert_unit: bool = True) -> "SkyArray":
        """sky_array.py:138-164."""
        if convert_unit:
            map_df = SkyUtils.convert_code_to_phy_units(quantity, map_df)
        map_array = SkyIO.transform_RayRamsesOutput_to_NumpyNdarray(map_df[quantity].values)
        return cls.from_array(map_array, opening_angle, quantity, dir_in, map_file)

    @classmethod
    def from_array(cls, map_array: np.ndarray, opening_angle: float, quantity: str, dir_in: str,
                   map_file: Optional[str] = None) -> "SkyArray":
        assert map_array.shape[0] == map_array.shape[1]
        return cls(map_array, opening_angle, quantity, {"sim": dir_in}, map_file)

    @classmethod
    def from_halo_series(cls, halo, npix: int = 100, extent: float = 1, direction: List[int] = [0, 1],
                         suppress: bool = False, suppression_R: float = 1, to: str = "dT") -> "SkyArray":
        """Analytic NFW map of one halo (sky_array.py:189-259)."""
        quantity = {"dT": "rs", "alpha": "alpha"}[to]
        if not (1 in direction and 0 in direction):
            quantity += "_x" if 0 in direction else "_y"
        if to == "dT":
            map_array = SkyUtils.NFW_temperature_perturbation_map(
                halo.r200_deg, halo.m200, halo.c_NFW, [halo.theta1_tv, halo.theta2_tv], halo.Dc, npix=npix,
                extent=extent, direction=direction, suppress=suppress, suppression_R=suppression_R)
        else:
            map_array = SkyUtils.NFW_deflection_angle_map(
                halo.r200_deg, halo.m200, halo.c_NFW, halo.Dc, npix=npix, extent=extent, direction=direction,
                suppress=suppress, suppression_R=suppression_R)
        return cls(map_array, 2 * halo.r200_deg * extent, quantity, dirs=None, map_file=None)

    @classmethod
    def from_halo_dataframe(cls, halo_cat, npix: int = 8192, extent: float = 1, direction: List[int] = [0, 1],
                            suppress: bool = False, suppression_R: float = 1, opening_angle: float = 20.0,
                            ncpus: int = 1, to: str = "dT") -> "SkyArray":
        """Sum of the NFW maps of a halo catalogue (sky_array.py:262-337).  ``ncpus`` is accepted
        for compatibility: every halo is painted by one GPU launch."""
        keys = ["r200_deg", "r200_pix", "m200", "c_NFW", "Dc", "theta1_pix", "theta2_pix"]
        if to == "dT":
            keys += ["theta1_tv", "theta2_tv"]
        halo_dict = {k: np.asarray(halo_cat[k]) for k in keys}
        quantity = {"dT": "rs", "alpha": "alpha"}[to]
        if not (1 in direction and 0 in direction):
            quantity += "_x" if 0 in direction else "_y"
        map_array = SkyUtils.analytic_Halo_signal_to_SkyArray(
            np.arange(len(halo_dict["m200"])), halo_dict, extent, direction, suppress, suppression_R, npix, to)
        map_array = np.nan_to_num(map_array, copy=False, nan=0.0, posinf=0.0, neginf=0.0)
        return cls(map_array, opening_angle, quantity, dirs=None, map_file=None)

    @classmethod
    def from_halo_catalogue_to_temperature_perturbation_map(cls, halo_cat: pd.DataFrame, extent: float = 1,
                                                            direction: List[int] = [0, 1], suppress: bool = False,
                                                            suppression_R: float = 1, npix: int = 8192,
                                                            opening_angle: float = 20.0, ncpus: int = 1) -> "SkyArray":
        """Rees-Sciama / Birkinshaw-Gull / moving-cluster temperature perturbation map dT / T_CMB of a halo catalogue
        (sky_array.py:341-400): the NFW stamps of ``from_halo_dataframe(..., to="dT")`` under the quantity label
        ``isw_rs[_x|_y]``, infinities zeroed.  ``ncpus`` is accepted for compatibility (one GPU launch paints all halos)."""
        keys = ["r200_deg", "r200_pix", "m200", "c_NFW", "Dc", "theta1_pix", "theta2_pix", "theta1_tv", "theta2_tv"]
        halo_dict = {k: np.asarray(halo_cat[k]) for k in keys}
        map_array = SkyUtils.analytic_Halo_signal_to_SkyArray(
            np.arange(len(halo_dict["m200"])), halo_dict, extent, direction, suppress, suppression_R, npix)
        map_array[np.isinf(map_array)] = 0.0
        if 1 in direction and 0 in direction:
            quantity = "isw_rs"
        elif 0 in direction:
            quantity = "isw_rs_x"
        else:
            quantity = "isw_rs_y"
        return cls(map_array, opening_angle, quantity, dirs=None, map_file=None)

    @property
    def npix(self) -> int:
        return self._npix

    @property
    def opening_angle(self) -> float:
        return self._opening_angle

    def pdf(self, nbins: int, of: str = "orig") -> dict:
        """np.histogram(data, bins=nbins, density=True)  (sky_array.py:428-433)."""
        _pdf = {}
        t = self.data.device(of)
        _pdf["values"], _pdf["bins"] = lensing.histogram(t, nbins, density=True)
        return _pdf

    def wl_peak_counts(self, nbins: int, field_conversion: str, of: str = "orig",
                       limits: Optional[tuple] = None) -> pd.DataFrame:
        """Signal peak counts (sky_array.py:435-472): heights of the 8-neighbour local maxima between the map's
        5th and 95th percentile (or ``limits``), histogrammed into ``nbins`` bins; DataFrame(kappa = bin centres,
        counts).  The 3 x 3 stencil over the map and the percentile selection run on the GPU; lenstools'
        ``locatePeaks`` semantics: interior pixels, strictly larger than all 8 neighbours, thresholds[0] <= height
        < thresholds[-1].  ("normalize" subtracts the mean of the map itself; the reference reads a non-existent
        ``self.skymap`` there.)"""
        t = self.data.device(of)
        if limits is None:
            lower_bound, upper_bound = lensing.percentile(t, [5, 95])
        else:
            lower_bound, upper_bound = min(limits), max(limits)
        map_bins = np.arange(lower_bound, upper_bound, (upper_bound - lower_bound) / nbins)
        heights, _ = lensing.peak_find(t)
        if field_conversion == "normalize":
            # the peak heights of (map - mean): same subtraction, same operands (numpy's mean of the host array)
            heights = heights - np.mean(np.ascontiguousarray(self.data[of], dtype=np.float64))
        _kappa = heights[(heights >= map_bins[0]) & (heights < map_bins[-1])]
        _hist, _kappa = np.histogram(_kappa, bins=nbins, density=False)
        _kappa = (_kappa[1:] + _kappa[:-1]) / 2
        return pd.DataFrame(data={"kappa": _kappa, "counts": _hist})

    def resize(self, npix, of: Optional[str] = None, img: Optional[np.ndarray] = None, rtn: bool = False,
               orig_data: str = None) -> Union[np.ndarray, None]:
        """Lower the nr. of pixels of an image (sky_array.py:475-496: ``skimage.transform.resize(img, (npix, npix),
        anti_aliasing=True)``), on the device (``lensing.resize_antialiased``).  The reference never reads
        ``self.data[of]`` (it resizes ``img`` or fails on None, :491); here ``of`` alone selects the stored map."""
        if img is None and of is not None:
            assert of in list(self.data.keys()), "Map does not exist."
            img = self.data.device(of)
        img = self._manage_img_data(img, orig_data)
        img = lensing.resize_antialiased(img, npix)
        if rtn:
            return to_host(img)
        self.data[of] = img

    def crop(self, xlimit, ylimit, of: Optional[str] = None, img: Optional[np.ndarray] = None,
             rtn: bool = False, orig_data: str = None) -> Union[np.ndarray, None]:
        """sky_array.py:498-540."""
        if of:
            assert of in list(self.data.keys()), "Map does not exist."
            img = self.data[of]
        img = self._manage_img_data(img, orig_data)
        xlimit = np.asarray(xlimit)
        ylimit = np.asarray(ylimit)
        if np.diff(xlimit) != np.diff(ylimit):
            raise SkyArrayWarning("The whole class is currently designed for square images.")
        if isinstance(xlimit[0], (float, np.floating)):
            _npix = img.shape[0]
            xlimit = (_npix * xlimit / 100).astype(int)
            ylimit = (_npix * ylimit / 100).astype(int)
        zoom = img[xlimit[0]: xlimit[1], ylimit[0]: ylimit[1]]
        if rtn:
            return zoom
        self.data[of] = zoom
        self._opening_angle = self._opening_angle * abs(np.diff(xlimit)) / self._npix
        self._npix = zoom.shape[0]

    def division(self, ntiles: int, of: Optional[str] = None, img: Optional[np.ndarray] = None,
                 rtn: bool = False, orig_data: str = None) -> Union[List[np.ndarray], None]:
        """sky_array.py:543-580."""
        if of:
            img = self.data[of]
        img = self._manage_img_data(img, orig_data)
        npix = img.shape[0]
        edges = list(np.arange(0, npix, npix / ntiles)) + [npix]
        edges = np.array([edges[idx: idx + 2] for idx in range(len(edges) - 1)]).astype(int)
        tiles = np.asarray([self.crop(xlim, ylim, img=img, rtn=True) for xlim in edges for ylim in edges])
        if rtn:
            return tiles
        self.tiles = tiles
        self._tile_npix = tiles[0].shape[0]
        self._tile_opening_angle = self._opening_angle * self._tile_npix / self._npix

    def merge(self, tiles: np.ndarray, rtn: bool = False) -> Union[np.ndarray, None]:
        """sky_array.py:583-601."""
        ntiles = len(tiles)
        nrows = int(np.sqrt(ntiles))
        img = np.vstack([np.hstack(tiles[r * nrows: (r + 1) * nrows]) for r in range(nrows)])
        if rtn:
            return img
        self.data["merged"] = img

    def substract_mean(self, of: Optional[str] = None, img: Optional[np.ndarray] = None, rtn: bool = False,
                       orig_data: str = None) -> Union[np.ndarray, None]:
        if of:
            assert of in list(self.data.keys()), "Map does not exist."
            img = self.data[of]
        img = self._manage_img_data(img, orig_data)
        img -= np.mean(img)
        if rtn:
            return img
        self.data[of] = img

    def filter(self, filter_dsc: dict, on: Optional[str] = None, img: Optional[np.ndarray] = None,
               rtn: bool = False, orig_data: str = None) -> Union[np.ndarray, None]:
        """Apply the kernels of ``filter_dsc`` in order (sky_array.py:623-662);
        each entry is dispatched by name into :class:`Filters`."""
        if on:
            assert on in list(self.data.keys()), "Map does not exist."
            img = self.data.device(on)               # the kernels of the chain hand CUDA tensors to each other
            map_name = [on]
        else:
            map_name = [""]
        img = self._manage_img_data(img, orig_data)
        for filter_name, args in filter_dsc.items():
            args = dict(args)
            if rtn is False:
                map_name.append(args.pop("abbrev"))
            else:
                args.pop("abbrev", None)
            fct = getattr(Filters, filter_name)
            img = fct(img, self._opening_angle, **args)
        if rtn:
            return to_host(img)
        self.data[("_").join(map_name)] = img

    def create_galaxy_shape_noise(self, std: float, ngal: float, rnd_seed: Optional[int] = None) -> None:
        """sky_array.py:665-690.  sigma_pix is hard-coded to 0.007 in the reference
        (:680; ``std``/``ngal`` are ignored there too).  The draw is numpy's own
        Generator(PCG64(seed)).normal — the same library call the reference makes —
        so a seeded map is bit-identical to astrild's."""
        std_pix = 0.007
        if rnd_seed is None:
            self.data["gsn"] = np.random.normal(loc=0, scale=std_pix, size=[self._npix, self._npix])
        else:
            rg = np.random.Generator(np.random.PCG64(rnd_seed))
            self.data["gsn"] = rg.normal(loc=0, scale=std_pix, size=[self._npix, self._npix])

    def add_galaxy_shape_noise(self, on: str = "orig") -> np.ndarray:
        """sky_array.py:693-706."""
        if "kappa" in self.quantity:
            self.data["orig_gsn"] = lensing.add(self.data.device("orig"), self.data.device("gsn"))
            return self.data["orig_gsn"]
        raise SkyArrayWarning(f"GSN should not be added to {self.quantity}")

    def convert_convergence_to_deflection(self, on: Optional[str] = None, img: Optional[np.ndarray] = None,
                                          npix: Optional[int] = None, opening_angle: Optional[float] = None,
                                          rtn: bool = True, orig_data: str = None
                                          ) -> Tuple[np.ndarray, np.ndarray]:
        """sky_array.py:780-817.  Returns (alpha_2, alpha_1) — the reference's order."""
        assert self.quantity in ["kappa_1", "kappa_2"], "Deflection angle can only be calculated from the kappa map"
        if on:
            img = self.data.device(on) if self.data.resident(on) or not rtn else self.data[on]
        img = self._manage_img_data(img, orig_data)
        if npix is None:
            npix = self._npix
        if opening_angle is None:
            opening_angle = self._opening_angle
        alpha_1, alpha_2 = SkyUtils.convert_convergence_to_deflection_ctypes(img, npix, opening_angle)
        if rtn:
            return to_host(alpha_2), to_host(alpha_1)
        self.data["defltx"] = alpha_2
        self.data["deflty"] = alpha_1

    def convert_deflection_to_shear(self, on: Optional[Tuple[str, str]] = None, img: Optional[Tuple[np.ndarray, np.ndarray]] = None,
                                    rtn: bool = False, orig_data: str = None) -> Tuple[np.ndarray, np.ndarray]:
        """sky_array.py:820-849.  Returns (gamma_2, gamma_1) - the reference's order - or stores them as
        data["gammax"], data["gammay"].  The reference hands ONE array to a two-array function (its body is unfinished);
        here `img` / `on` name the pair (alpha_1, alpha_2), default the stored data["deflty"], data["defltx"]."""
        assert self.quantity in ["alpha"], "Shear can only be calculated from the deflection angle map"
        if on:
            img = (self.data.device(on[0]), self.data.device(on[1]))
        if img is None:
            img = (self.data.device("deflty"), self.data.device("defltx"))
        a1, a2 = (self._manage_img_data(m, orig_data) for m in img)
        gamma_1, gamma_2 = SkyUtils.convert_deflection_to_shear(a1, a2, self._npix, self._opening_angle)
        if rtn:
            return to_host(gamma_2), to_host(gamma_1)
        self.data["gammax"] = gamma_2
        self.data["gammay"] = gamma_1

    @staticmethod
    def _manage_img_data(img: np.ndarray, orig_data: str = None) -> np.ndarray:
        if is_device_map(img):                       # (device operations never write into their input: a copy only on request)
            return img.clone() if orig_data in ("shallow", "deep") else img
        if orig_data == "shallow":
            return copy.copy(img)
        elif orig_data == "deep":
            return copy.deepcopy(img)
        return img
