"""``Bispectrum2D.from_skymap`` with astrild's API (src/astrild/bispectra/bispectrum_2d.py:19-60): the equilateral
flat-sky bispectrum of a convergence map.  lenstools' ``ConvergenceMap.bispectrum(l_edges,
configuration="equilateral")`` is evaluated with the FFT estimator on the GPU
(``lensing.flat_bispectrum_equilateral``): ring-filtered inverse transforms and cube sums instead of its
pair enumeration; both average ft(l1) ft(l2) ft(l3) over the closed triangles with all sides in the bin.
lenstools is un-vendored: parity unpinned."""
from typing import List, Union

import numpy as np
import pandas as pd

from .. import lensing
from ..io import IO



def _map_of(skymap, on):
    """The map as it lives: the CUDA tensor of a SkyArray's resident map (no PCIe hop), else the array."""
    dev_fn = getattr(skymap.data, "device", None)
    return dev_fn(on) if dev_fn is not None else skymap.data[on]


class Bispectrum2DWarning(BaseException):
    pass


class Bispectrum2D:
    def __init__(self, l: np.array, B: np.array, skymap, on: str):
        self.l = l
        self.B = B
        self.skymap = skymap
        self.on = on

    @classmethod
    def from_skymap(cls, skymap, on: str, multipoles: Union[List[float], np.array] = np.arange(200.0, 50000.0, 200.0),
                    rtn: bool = False) -> "Bispectrum2D":
        if "kappa" not in skymap.quantity:
            raise Bispectrum2DWarning(f"no bispectrum for quantity {skymap.quantity!r} (the reference handles kappa only)")
        l, B, ntri = lensing.flat_bispectrum_equilateral(_map_of(skymap, on), skymap.opening_angle,
                                                         np.asarray(multipoles, dtype=np.float64))
        out = cls(l, B, skymap, on)
        out.ntri = ntri
        return out

    def to_file(self, dir_out: str, extention: str = "h5") -> None:
        """DataFrame(index=l, columns=["B"]) -> HDF5 key "df" (bispectrum_2d.py:52-66)."""
        df = pd.DataFrame(data=self.B, index=self.l, columns=["B"])
        filename = self._create_filename(dir_out, extention)
        IO._remove_existing_file(filename)
        print(f"Saving results to -> {filename}")
        df.to_hdf(filename, key="df", mode="w")

    def _create_filename(self, dir_out: str, extention: str = "h5") -> str:
        root = getattr(self.skymap, "map_file", None) or "map"
        root = str(root).split("/")[-1].rsplit(".", 1)[0]
        return f"{dir_out}Bl_{self.skymap.quantity}_{self.on}_{root}.{extention}"
