"""``SimulationCollection.sum_raytracing_snapshots`` (src/astrild/simcoll.py:238-443):
collection-level kappa-map stack over ``.h5`` DataFrames or ``.npy`` planes, summed
on the GPU.  ``self.sim`` maps simulation names to objects with ``.dirs["sim"]``
and ``.file_dsc``; ``self.config`` is the (box_nr, ray_nr)-indexed snapshot table."""
import glob
import os
from typing import Optional

import numpy as np
import pandas as pd

from .rays.rayramses import PlaneStacker, _shard_group


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
                                 z_src_shift: Optional[float] = None, rm_ray: Optional[dict] = None,
                                 reweight: bool = False, group=None):
        """simcoll.py:238-341.  Returns the summed DataFrame (``.h5`` planes) or
        ndarray (``.npy`` planes).  ``reweight`` / ``group``: see rays/rayramses.py.
        ``z_src_shift`` may be a SEQUENCE of source redshifts: the planes are read and uploaded once, stay resident in
        HBM and are re-weighted once per entry; a list comes back (one DataFrame / ndarray per source redshift; under a
        process group map m is reduced onto rank m mod P and is None elsewhere)."""
        box_ray_nrs = self._get_box_and_ray_nrs_for_integration_range(integration_range, rm_ray)
        selection = []                                      # (sim, box_nr, ray_nr) in the reference's loop order
        for sim_name in self.sim.keys():
            box_nr = self._boxnr_from_simname(sim_name)
            if box_nr not in list(box_ray_nrs.keys()):
                continue
            selection += [(self.sim[sim_name], box_nr, ray_nr) for ray_nr in box_ray_nrs[box_nr]]
        if not selection:
            raise SimulationCollectionWarning("no ray-tracing snapshot in the requested range")
        group = _shard_group(group)
        rank, world = 0, 1
        if group is not None:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        def ray_file_of(sim, ray_nr):
            return glob.glob(sim.dirs["sim"] + f"{sim.file_dsc['root']}_*{ray_nr}." + f"{sim.file_dsc['extension']}")[0]

        if isinstance(z_src_shift, (list, tuple, np.ndarray)):
            # several source redshifts over the same planes (the re-weighting loop of simcoll.py:302-320, once per source
            # plane): every file is read and uploaded ONCE, the planes stay resident in HBM, one output map per entry
            per_z = [[self._plane_weight(self.config, box_nr, ray_nr, z_src, float(zs), reweight, SimulationCollectionWarning)
                      for _, box_nr, ray_nr in selection] for zs in z_src_shift]
            mine = [ii for ii in range(len(selection)) if ii % world == rank]
            maps = [self._load_ray_map(ray_file_of(selection[ii][0], selection[ii][2])) for ii in mine]
            first = maps[0] if maps else self._load_ray_map(ray_file_of(selection[0][0], selection[0][2]))
            is_frame = isinstance(first, pd.DataFrame)
            weights_list = []
            for pw in per_z:
                if any(w is not None for w in pw):
                    wn = [(pw[ii] or (1.0, 1.0))[0] for ii in mine]
                    wd = [(pw[ii] or (1.0, 1.0))[1] for ii in mine]
                    weights_list.append({c: (wn, wd) for c in (columns_z_shift or ["kappa_2"]) if c in columns} if is_frame else (wn, wd))
                else:
                    weights_list.append({} if is_frame else None)
            if is_frame:
                return self._stack_columns_many(maps, columns, weights_list, group, None if maps else first)
            return self._stack_arrays_many(maps, weights_list, group, np.shape(first))
        plane_w = [self._plane_weight(self.config, box_nr, ray_nr, z_src, z_src_shift, reweight,
                                      SimulationCollectionWarning) for _, box_nr, ray_nr in selection]

        maps, wnum, wden = [], [], []
        for ii, (_sim, box_nr, ray_nr) in enumerate(selection):
            if ii % world != rank:
                continue                                    # plane ii lives on rank ii mod P
            maps.append(self._load_ray_map(ray_file_of(_sim, ray_nr)))
            n, d = plane_w[ii] or (1.0, 1.0)
            wnum.append(n)
            wden.append(d)
        weighted = any(w is not None for w in plane_w)
        first = maps[0] if maps else self._load_ray_map(ray_file_of(selection[0][0], selection[0][2]))
        if isinstance(first, pd.DataFrame):
            weights = {}
            if weighted:
                weights = {c: (wnum, wden) for c in (columns_z_shift or ["kappa_2"]) if c in columns}
            return self._stack_columns(maps, columns, weights, group, None if maps else first)
        out = self._stack_arrays(maps, (wnum, wden) if weighted else None, group)
        return out.reshape(np.shape(first))

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
