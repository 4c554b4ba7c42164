"""``SimulationCollection.sum_raytracing_snapshots`` (src/astrild/simcoll.py:238-443):
collection-level kappa-map stack over ``.h5`` DataFrames or ``.npy`` planes, summed
on the GPU.  ``self.sim`` maps simulation names to objects with ``.dirs["sim"]``
and ``.file_dsc``; ``self.config`` is the (box_nr, ray_nr)-indexed snapshot table."""
import glob
import os
from typing import Optional

import numpy as np
import pandas as pd

from .rays.rayramses import PlaneStacker


class SimulationCollectionWarning(BaseException):
    pass


class SimulationCollection(PlaneStacker):
    def __init__(self, config: pd.DataFrame, sims: dict, cosmology=None):
        self.config = config
        self.sim = sims
        self.cosmology = cosmology

    @staticmethod
    def _boxnr_from_simname(sim_name: str) -> int:
        digits = "".join(ch for ch in sim_name if ch.isdigit())
        return int(digits)

    def _load_ray_map(self, ray_file: str):
        if ray_file.split(".")[-1] == "h5":
            return pd.read_hdf(ray_file, key="df", mode="r")
        elif ray_file.split(".")[-1] == "npy":
            return np.load(ray_file)
        raise SimulationCollectionWarning("This file type is not supported.")

    def sum_raytracing_snapshots(self, dir_out: str, columns: list, columns_z_shift: list,
                                 integration_range: dict, ray_file_root: str = "Ray_maps_output%05d.h5",
                                 sim_folder_root: str = "box%d", z_src: Optional[float] = None,
                                 z_src_shift: Optional[float] = None, rm_ray: Optional[dict] = None):
        """simcoll.py:238-341.  Returns the summed DataFrame (``.h5`` planes) or
        ndarray (``.npy`` planes)."""
        box_ray_nrs = self._get_box_and_ray_nrs_for_integration_range(integration_range, rm_ray)
        maps, wnum, wden = [], [], []
        for sim_name in self.sim.keys():
            _sim = self.sim[sim_name]
            box_nr = self._boxnr_from_simname(sim_name)
            if box_nr not in list(box_ray_nrs.keys()):
                continue
            ray_nrs = box_ray_nrs[box_nr]
            for ii, ray_nr in enumerate(ray_nrs):
                sim_info_df = self.config.loc[(box_nr, ray_nr)]
                ray_file = glob.glob(
                    _sim.dirs["sim"] + f"{_sim.file_dsc['root']}_*{ray_nr}." + f"{_sim.file_dsc['extension']}"
                )[0]
                maps.append(self._load_ray_map(ray_file))
                if z_src_shift is not None:
                    z_next = (self.config.loc[(box_nr, ray_nrs[ii + 1])]["redshift"]
                              if ii + 1 < len(ray_nrs) else sim_info_df["redshift"])
                    n, d = self._translate_redshift_weight(sim_info_df["redshift"], z_next, z_src, z_src_shift)
                    wnum.append(n)
                    wden.append(d)
        if not maps:
            raise SimulationCollectionWarning("no ray-tracing snapshot in the requested range")
        if isinstance(maps[0], pd.DataFrame):
            weights = {}
            if z_src_shift is not None:
                weights = {c: (wnum, wden) for c in (columns_z_shift or ["kappa_2"]) if c in columns}
            return self._stack_columns(maps, columns, weights)
        return self._stack_arrays(maps, (wnum, wden) if z_src_shift is not None else None)

    def _get_box_and_ray_nrs_for_integration_range(self, integration_range: dict,
                                                   rm_ray: Optional[dict] = None) -> dict:
        """simcoll.py:343-388."""
        if not integration_range["z"]:
            if integration_range["box"][0] != 0 and integration_range["ray"][0] == 0:
                self.config = self.config[self.config.index.get_level_values(0).isin(integration_range["box"])]
        else:
            z_range = np.asarray(integration_range["z"])
            self.config = self.config[
                (z_range.min() < self.config["redshift"]) & (self.config["redshift"] < z_range.max())
            ]
        box_and_ray_nrs = {}
        for box_nr, ray_nr in self.config.index.values:
            box_and_ray_nrs.setdefault(box_nr, []).append(ray_nr)
        if rm_ray:
            for box_nr in rm_ray.keys():
                for ray_nr in rm_ray[box_nr]:
                    box_and_ray_nrs[box_nr].remove(ray_nr)
        return box_and_ray_nrs
