"""``RayRamses.sum_snapshots`` (src/astrild/rays/rayramses.py:151-326): the kappa-map
stack.  Plane selection follows the reference; the summation (and the optional
lensing-kernel re-weighting of ``kappa_2``) runs on the GPU as one streaming
kernel over all selected planes instead of a Python ``+=`` loop.

Deviations from the reference, all of them defects listed in SURVEY.md
Appendix B: ``_get_box_and_ray_nrs`` is called as a method (rayramses.py:186
lacks ``self.``); the re-weighting block is unreachable there (:205-222 sits
behind an unconditional ``raise``) — here it executes when ``z_src_shift`` is
given, with z_next taken from the snapshot table as the dead code intends.
"""
import os
from typing import Optional

import numpy as np
import pandas as pd

from .. import lensing
from ..device import as_device


class RayRamsesWarning(BaseException):
    pass


class PlaneStacker:
    """Shared by RayRamses and SimulationCollection: sum planes column by column."""

    @staticmethod
    def _kernel_function(x: float, x_s: float) -> float:
        """g = (x_s - x) * x / x_s  (rayramses.py:315-326)."""
        return (x_s - x) * x / x_s

    def _comoving(self, z: float) -> float:
        d = self.cosmology.comoving_distance(z)
        return float(d.to_value("Mpc")) if hasattr(d, "to_value") else float(d)

    def _translate_redshift_weight(self, z_near, z_far, z_src, z_src_shift):
        """(numerator, denominator) of rayramses.py:269-312 for one plane."""
        x_far, x_near, x_src = self._comoving(z_far), self._comoving(z_near), self._comoving(z_src)
        x_src_shift = self._comoving(z_far) if z_far > z_src_shift else self._comoving(z_src_shift)
        x_mid = 0.5 * (x_far + x_near)
        return self._kernel_function(x_mid, x_src_shift), self._kernel_function(x_mid, x_src)

    def _translate_redshift(self, quantity, z_near, z_far, z_src, z_src_shift):
        num, den = self._translate_redshift_weight(z_near, z_far, z_src, z_src_shift)
        t = as_device(np.ascontiguousarray(quantity, dtype=np.float64))
        return lensing.kappa_stack([t], [num], [den]).cpu().numpy()

    @staticmethod
    def _stack_columns(frames, columns, weights):
        """frames: list of DataFrames (same index); weights: {column: (num[], den[])} or {}.
        The first frame is the accumulator, like the reference (rayramses.py:224-232)."""
        total = frames[0]
        for column in columns:
            planes = [as_device(np.ascontiguousarray(f[column].values, dtype=np.float64)) for f in frames]
            wn, wd = weights.get(column, (None, None))
            total[column] = lensing.kappa_stack(planes, wn, wd).cpu().numpy()
        return total

    @staticmethod
    def _stack_arrays(arrays, weights=None):
        planes = [as_device(np.ascontiguousarray(a, dtype=np.float64)) for a in arrays]
        wn, wd = weights if weights else (None, None)
        return lensing.kappa_stack(planes, wn, wd).cpu().numpy()


class RayRamses(PlaneStacker):
    """The part of astrild's RayRamses that sits on the hot path.  ``dirs`` needs
    "lc" (light-cone root holding ray_snapshot_info.h5 and the box%d folders)."""

    def __init__(self, dirs: dict, file_dsc: Optional[dict] = None, cosmology=None, ray_info_df=None):
        self.dirs = dict(dirs)
        self.file_dsc = file_dsc or {"root": "Ray_maps", "extension": "h5"}
        self.cosmology = cosmology
        self.ray_info_df = ray_info_df
        self.complete_lc = False

    # -- I/O hooks (pandas HDF5 in the reference; kept separate so tests can feed memory)
    def _load_ray_info(self) -> pd.DataFrame:
        file_name = self.dirs["lc"] + "ray_snapshot_info.h5"
        if not os.path.isfile(file_name):
            raise RayRamsesWarning("The file 'ray_snapshot_info.h5' does note exist")
        return pd.read_hdf(file_name, key="s")

    def _load_ray_map(self, ray_file: str) -> pd.DataFrame:
        return pd.read_hdf(ray_file)

    def sum_snapshots(self, dir_out: str, columns: list, columns_z_shift: list, integration_range: dict,
                      ray_file_root: str = "Ray_maps_output%05d.h5", sim_folder_root: str = "box%d",
                      z_src: float = None, z_src_shift: float = None) -> pd.DataFrame:
        """Add ray-tracing outputs between arbitrary redshifts along the light-cone
        (rayramses.py:151-234).  Returns the summed DataFrame (and writes it like the
        reference when ``dir_out`` is not None)."""
        if self.ray_info_df is None:
            self.ray_info_df = self._load_ray_info()
        sim_folder_root = self.dirs["lc"] + sim_folder_root
        box_ray_nrs = self._get_box_and_ray_nrs(integration_range)

        frames, wnum, wden = [], [], []
        info = self.ray_info_df
        for ii, (box_nr, ray_nr) in enumerate(box_ray_nrs):
            sim_info_df = info.loc[(box_nr, ray_nr)]
            self.dirs["sim"] = sim_folder_root % box_nr + "/"
            ray_map_df = self._load_ray_map(self.dirs["sim"] + ray_file_root % ray_nr)
            frames.append(ray_map_df)
            if z_src_shift is not None:
                # next snapshot along the cone: next row of the table (rayramses.py:206-210)
                z_next = info.iloc[min(ii + 1, len(info) - 1)]["redshift"] if ii + 1 < len(box_ray_nrs) \
                    else sim_info_df["redshift"]
                n, d = self._translate_redshift_weight(sim_info_df["redshift"], z_next, z_src, z_src_shift)
                wnum.append(n)
                wden.append(d)
        if not frames:
            raise RayRamsesWarning("no ray-tracing snapshot in the requested range")
        weights = {}
        if z_src_shift is not None:
            # "only of kappa but not of iswrs" (rayramses.py:311)
            weights = {c: (wnum, wden) for c in (columns_z_shift or ["kappa_2"]) if c in columns}
        ray_df_sum = self._stack_columns(frames, columns, weights)
        if dir_out is not None:
            self._merged_snapshots_to_file(ray_df_sum, dir_out, integration_range)
        return ray_df_sum

    def _get_box_and_ray_nrs(self, integration_range: dict) -> np.ndarray:
        """rayramses.py:237-266."""
        if not integration_range["z"]:
            if integration_range["box"][0] == 0:
                self.complete_lc = True
            elif integration_range["ray"][0] == 0:
                self.ray_info_df = self.ray_info_df[
                    self.ray_info_df.index.get_level_values(0).isin(integration_range["box"])
                ]
                self.complete_lc = False
        else:
            z_range = np.asarray(integration_range["z"])
            self.ray_info_df = self.ray_info_df[
                (z_range.min() < self.ray_info_df["redshift"]) & (self.ray_info_df["redshift"] < z_range.max())
            ]
            self.complete_lc = False
        return self.ray_info_df.index.values

    def _merged_snapshots_to_file(self, ray_df_sum: pd.DataFrame, dir_out: str, integration_range: dict) -> None:
        """rayramses.py:329-352."""
        if not integration_range["z"]:
            if integration_range["box"][0] == 0:
                fout = dir_out + "Ray_maps_lc.h5"
            else:
                fout = dir_out + "Ray_maps_box%d.h5" % integration_range["box"][0]
        else:
            fout = dir_out + "Ray_maps_zrange_%.2f_%.2f.h5" % (integration_range["z"][0], integration_range["z"][1])
        print("Save in %s" % fout)
        ray_df_sum.to_hdf(fout, key="df", mode="w")
