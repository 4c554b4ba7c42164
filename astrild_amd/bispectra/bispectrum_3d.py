"""``Bispectrum3D`` with astrild's API (src/astrild/bispectra/bispectrum_3d.py).

The reference class is a copy of ``PowerSpectrum3D``: ``compute`` and
``_power_spectrum_3d`` return P(k) (bispectrum_3d.py:165-215).  That behaviour is
kept call-for-call (so existing scripts get the same numbers), and the estimator
the docstring cites (:42-44, arXiv:1512.07295 / 1506.02729) is provided as
``_bispectrum_3d`` / ``compute_bispectrum`` on the GPU.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .. import device as dev
from ..power_spectra.power_spectrum_3d import PowerSpectrum3D


class Bispectrum3DWarning(BaseException):
    pass


class Bispectrum3D(PowerSpectrum3D):
    def _bispectrum_3d(self, value_map, shell_width: int = 1, m_min: int = 1, m_max: Optional[int] = None,
                       triangles: Optional[Sequence[Tuple[int, int, int]]] = None, group=None) -> dict:
        """Matter bispectrum B(k1, k2, k3) by FFT triangle counting.

        Shells are [m, m + shell_width) in units of k_F = 2*pi/L on integer |m|;
        ``triangles`` lists shell-index triplets (default: equilateral).  Returns
        dict(k (ntri, 3), B, ntri) with exact integer triangle counts.  ``group``: a torch.distributed
        process group whose ranks all call this with the same grid; the triangle bins are split over them."""
        n = int(self.sim.domain_level)
        m_max = n // 2 if m_max is None else m_max
        edges = list(range(m_min, m_max + 1, shell_width))
        nsh = len(edges) - 1
        if nsh < 1:
            raise Bispectrum3DWarning("no complete shell in the requested range")
        if triangles is None:
            triangles = [(i, i, i) for i in range(nsh)]
        f = dev.as_device(value_map, self.dtype)
        if tuple(f.shape) != (n, n, n):
            raise Bispectrum3DWarning(f"value_map shape {tuple(f.shape)} does not match Nmesh={n}")
        if group is not None:              # triangle bins spread over the ranks of a torch.distributed group
            from ..bispec_shard import bispectrum_sharded
            return bispectrum_sharded(f, self.sim.boxsize, edges, triangles, group=group)
        return dev.bispectrum(f, self.sim.boxsize, edges, triangles)

    def compute_bispectrum(self, quantities: List[str], file_paths: List[str], **kwargs) -> dict:
        """Bispectrum of each gridded file (``.npy`` grid or ``.h5`` particle table)."""
        out = {}
        for path in file_paths:
            out[path] = self._bispectrum_3d(self._read_data(path, quantities), **kwargs)
        return out
