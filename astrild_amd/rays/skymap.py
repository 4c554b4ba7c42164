"""``SkyMap`` factory (src/astrild/rays/skymap.py:45-172) -> :class:`SkyArray`.

The reference passes ``npix`` where ``SkyArray.from_dataframe`` expects
``opening_angle`` (skymap.py:79-87 vs sky_array.py:139-148); here ``theta`` is the
opening angle, as documented.  HEALPix skies are out of scope."""
from typing import Optional

import numpy as np
import pandas as pd

from .skys.sky_array import SkyArray


class SkyMapWarning(BaseException):
    pass


class SkyMap:
    @staticmethod
    def _check(sky_type):
        assert sky_type in ["array", "healpix"], "The declared 'sky_type' is not known"
        if sky_type == "healpix":
            raise SkyMapWarning("HEALPix skies are not part of the MI355X hot path")

    @staticmethod
    def from_file(npix: int, theta: float, quantity: str, dir_in: str, map_file: Optional[str] = None,
                  convert_unit: bool = True, sky_type: str = "array") -> SkyArray:
        SkyMap._check(sky_type)
        if not map_file:
            raise SkyMapWarning("There is no file being pointed at")
        file_extension = map_file.split(".")[-1]
        if file_extension == "h5":
            map_df = pd.read_hdf(map_file, key="df")
            return SkyArray.from_dataframe(map_df, theta, quantity, dir_in, map_file, npix, convert_unit)
        elif file_extension == "npy":
            return SkyArray.from_array(np.load(map_file), theta, quantity, dir_in, map_file)
        raise SkyMapWarning(f"file type .{file_extension} is not supported")

    @staticmethod
    def from_dataframe(npix: int, theta: float, quantity: str, dir_in: str, map_df: pd.DataFrame,
                       map_file: str, convert_unit: bool = True, sky_type: str = "array") -> SkyArray:
        SkyMap._check(sky_type)
        return SkyArray.from_dataframe(map_df, theta, quantity, dir_in, map_file, npix, convert_unit)

    @staticmethod
    def from_array(map_array: np.ndarray, npix: int, theta: float, quantity: str, dir_in: str,
                   map_file: Optional[str] = None, sky_type: str = "array") -> SkyArray:
        SkyMap._check(sky_type)
        return SkyArray.from_array(map_array, theta, quantity, dir_in, map_file)
