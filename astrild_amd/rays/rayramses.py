"""``RayRamses.sum_snapshots`` (src/astrild/rays/rayramses.py:151-326): the kappa-map
stack.  Plane selection follows the reference; the summation (and the optional
lensing-kernel re-weighting of ``kappa_2``) runs on the GPU as one streaming
kernel over all selected planes instead of a Python ``+=`` loop.

Deviations from the reference, all of them defects listed in SURVEY.md
Appendix B: ``_get_box_and_ray_nrs`` is called as a method (rayramses.py:186
lacks ``self.``).  The re-weighting block is unreachable there (:205-222 sits
behind an unconditional ``raise``): by default this class behaves as the
reference EXECUTES - ``z_src_shift`` with a selected plane at
``redshift <= z_src_shift`` raises the same warning, otherwise the planes are
summed unweighted.  ``reweight=True`` opts into what the dead block intends:
planes with ``redshift <= z_src_shift`` have ``kappa_2`` multiplied by
g(x_mid, x_s') / g(x_mid, x_s), z_next by its box rule (:207-210).

Multi-GPU: under an initialised ``torch.distributed`` group of more than one
rank, plane p of the selection is loaded and summed by rank p mod P and the
partial maps are combined by ``kappa_shard.kappa_stack_sharded`` (SURVEY.md
§8e); every rank returns the full result.
"""
import os
from typing import Optional

import numpy as np
import pandas as pd

from .. import lensing
from ..device import as_device, to_numpy, upload_planes


def _shard_group(group):
    """The process group to shard planes over, or None (single process / one rank)."""
    import torch.distributed as dist
    if group is None and dist.is_available() and dist.is_initialized():
        group = dist.group.WORLD
    if group is not None and dist.get_world_size(group) > 1:
        return group
    return None


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
        return to_numpy(lensing.kappa_stack([t], [num], [den]))

    @staticmethod
    def _z_next(table, box_nr, ray_nr):
        """Redshift of the next snapshot along the cone, the rule of the reference's dead block
        (rayramses.py:207-210, simcoll.py:306-309): at the last output of a light-cone box (the smallest ray
        number of the box, boxes 1..3) the first output of the next box, else the next row of the box's table."""
        rows = table.loc[box_nr]
        ray_nrs = np.asarray(rows.index.values)
        if ray_nr == ray_nrs.min() and box_nr < 4 and (box_nr + 1, 1) in table.index:
            return float(table.loc[(box_nr + 1, 1)]["redshift"])
        ii = int(np.nonzero(ray_nrs == ray_nr)[0][0])
        if ii + 1 < len(rows):
            return float(rows.iloc[ii + 1]["redshift"])
        return float(rows.iloc[ii]["redshift"])

    def _plane_weight(self, table, box_nr, ray_nr, z_src, z_src_shift, reweight, warning):
        """(numerator, denominator) of one plane's kappa_2 weight, or None when the plane is summed as it is."""
        if z_src_shift is None:
            return None
        z = float(table.loc[(box_nr, ray_nr)]["redshift"])
        if z > z_src_shift:
            return (1.0, 1.0) if reweight else None
        if not reweight:
            raise warning("Redshift shift has not correct data structure")       # what the reference executes
        return self._translate_redshift_weight(z, self._z_next(table, box_nr, ray_nr), z_src, z_src_shift)

    @staticmethod
    def _sum_planes(planes, wn, wd, group):
        """Device planes -> summed map as numpy; sharded over `group` when given (every rank gets the result)."""
        if group is None:
            return to_numpy(lensing.kappa_stack(planes, wn, wd))
        from ..kappa_shard import kappa_stack_sharded
        return to_numpy(kappa_stack_sharded(planes, wn, wd, group=group, all_ranks=True))

    @classmethod
    def _stack_columns(cls, frames, columns, weights, group=None, template=None):
        """frames: list of DataFrames (same index) - with `group`, this rank's share of them;
        weights: {column: (num[], den[])} or {}.  The first frame is the accumulator, like the reference
        (rayramses.py:224-232); a rank without frames fills `template`'s copy."""
        total = frames[0] if frames else template
        for column in columns:
            planes = [as_device(np.ascontiguousarray(f[column].values, dtype=np.float64)) for f in frames]
            wn, wd = weights.get(column, (None, None))
            total[column] = cls._sum_planes(planes, wn, wd, group).reshape(np.shape(total[column].values))
        return total

    # ---- several output maps from ONE pass over the files -------------------------------------------------------
    # The reference's loops re-read every plane for every call (rayramses.py:186-232, simcoll.py:267-336); a survey
    # of source redshifts re-weights the SAME planes map after map (simcoll.py:302-320).  With `z_src_shift` a
    # sequence the planes are loaded and uploaded once, stay resident in HBM (per rank when sharded) and every
    # source redshift costs one pass of the stacking kernel - a host copy of the planes per map would cost 50 x
    # the kernel (134 MB per 4096^2 plane over PCIe against 1.5 ms for 64 of them from HBM).
    @staticmethod
    def _stack_many(planes, weights_list, group=None):
        """planes: this rank's device planes (flat, resident); weights_list: per output map (wnum[], wden[]) or None
        (plain sum).  Returns one numpy map per entry - under a process group the maps reduced onto THIS rank (map m
        lands on rank m mod P, kappa_shard.MapStream: the root rotates and the next map is being stacked while the last
        one travels) and None for the others.  Maps with equal weights are computed once."""
        import torch
        keys = [None if w is None else (tuple(float(v) for v in w[0]), tuple(float(v) for v in w[1])) for w in weights_list]
        if group is None:
            done = {}
            for key, w in zip(keys, weights_list):
                if key not in done:
                    wn, wd = w if w is not None else (None, None)
                    done[key] = lensing.kappa_stack(planes, wn, wd)          # queued back to back: no host round trip between maps
            host = {key: to_numpy(t) for key, t in done.items()}
            return [host[key] for key in keys]
        from ..kappa_shard import MapStream
        import torch.distributed as dist
        npix2 = torch.tensor([planes[0].numel() if planes else 0], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            npix2 = npix2.to(lensing.device())
        dist.all_reduce(npix2, op=dist.ReduceOp.MAX, group=group)
        stream = MapStream(int(npix2.item()), group)

        def tail(m, flat):
            # the summed map sits in a rotating buffer: copy it out on the tail stream, asynchronously into pinned memory
            if flat.is_cuda:
                out = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
                out.copy_(flat, non_blocking=True)
                return out
            return flat.clone()
        for w in weights_list:
            wn, wd = w if w is not None else (None, None)
            stream.push(planes, wn, wd, tail=tail)
        res = stream.finish()
        if torch.cuda.is_available() and planes and planes[0].is_cuda:
            torch.cuda.synchronize()
        return [res[m].numpy() if m in res else None for m in range(len(weights_list))]

    @classmethod
    def _stack_columns_many(cls, frames, columns, weights_list, group=None, template=None):
        """_stack_columns for several output maps: weights_list[m] = {column: (num[], den[])} or {}.  Every column's planes
        are uploaded once; a column no map re-weights is summed once and shared.  Returns one DataFrame per map (None
        where a sharded map was reduced onto another rank)."""
        base = frames[0] if frames else template
        outs = [base.copy() for _ in weights_list]
        missing = [False] * len(weights_list)
        for column in columns:
            planes = [as_device(np.ascontiguousarray(f[column].values, dtype=np.float64)) for f in frames]
            per_map = [w.get(column) for w in weights_list]
            maps = cls._stack_many(planes, per_map, group)
            for m, res in enumerate(maps):
                if res is None:
                    missing[m] = True
                else:
                    outs[m][column] = res.reshape(np.shape(base[column].values))
            del planes
        return [None if missing[m] else outs[m] for m in range(len(weights_list))]

    @classmethod
    def _stack_arrays_many(cls, arrays, weights_list, group=None, shape=None):
        planes = upload_planes(arrays)                 # one device allocation, plane after plane
        maps = cls._stack_many(planes, weights_list, group)
        shape = np.shape(arrays[0]) if arrays else shape
        return [None if r is None else r.reshape(shape) for r in maps]

    @classmethod
    def _stack_arrays(cls, arrays, weights=None, group=None):
        planes = upload_planes(arrays, flat=False)
        wn, wd = weights if weights else (None, None)
        out = cls._sum_planes(planes, wn, wd, group)
        return out.reshape(np.shape(arrays[0])) if arrays else out


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

    def compress_snapshot(self, fields: list, dir_out: str = None, convert: bool = False, cosmo=None,
                          save: bool = True, cpu_files: Optional[dict] = None):
        """Combine the ray-tracing outputs of the individual CPUs of each snapshot into one pandas .h5 file
        (rayramses.py:69-148).  ``cpu_files``: {ray_nr: [per-CPU ASCII files]}; by default every
        ``<root>_<ray_nr>*`` file group found in ``dirs["sim"]`` (the reference asks its Simulation base class,
        which is outside the hot path).  Returns {ray_nr: DataFrame} when ``save`` is False."""
        import glob
        import re
        from ..formats import compress_rayramses_outputs
        self.cosmology = cosmo if cosmo is not None else self.cosmology
        root = self.file_dsc["root"]
        if cpu_files is None:
            cpu_files = {}
            for path in sorted(glob.glob(self.dirs["sim"] + f"{root}_*")):
                m = re.search(r"(\d+)", os.path.basename(path)[len(root):])
                if m and not path.endswith(".h5"):
                    cpu_files.setdefault(int(m.group(1)), []).append(path)
        h = None
        if convert:
            h = float(self.cosmology.H0.value) / 100 if hasattr(self.cosmology.H0, "value") else float(self.cosmology.H0) / 100
        out = {}
        for ray_nr in sorted(cpu_files):
            table = compress_rayramses_outputs(cpu_files[ray_nr], fields, convert, h)
            if save:
                table.to_hdf((dir_out or self.dirs["sim"]) + "%s_output%05d.h5" % (root, ray_nr), key="df", mode="w")
            else:
                out[ray_nr] = table
        return None if save else out

    def sum_snapshots(self, dir_out: str, columns: list, columns_z_shift: list, integration_range: dict,
                      ray_file_root: str = "Ray_maps_output%05d.h5", sim_folder_root: str = "box%d",
                      z_src: float = None, z_src_shift: float = None, reweight: bool = False,
                      group=None) -> pd.DataFrame:
        """Add ray-tracing outputs between arbitrary redshifts along the light-cone
        (rayramses.py:151-234).  Returns the summed DataFrame (and writes it like the
        reference when ``dir_out`` is not None; rank 0 writes under a process group).
        ``z_src_shift`` may be a SEQUENCE of source redshifts: every plane is then read and uploaded once, stays
        resident in HBM and is re-weighted once per entry - a list of DataFrames comes back (nothing is written;
        under a process group map m is reduced onto rank m mod P and is None on the other ranks)."""
        if self.ray_info_df is None:
            self.ray_info_df = self._load_ray_info()
        sim_folder_root = self.dirs["lc"] + sim_folder_root
        box_ray_nrs = self._get_box_and_ray_nrs(integration_range)
        if len(box_ray_nrs) == 0:
            raise RayRamsesWarning("no ray-tracing snapshot in the requested range")
        group = _shard_group(group)
        rank, world = 0, 1
        if group is not None:
            import torch.distributed as dist
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        info = self.ray_info_df
        many = isinstance(z_src_shift, (list, tuple, np.ndarray))
        if many:
            # several source redshifts: every plane is read and uploaded ONCE, one output map per entry
            per_z = [[self._plane_weight(info, box_nr, ray_nr, z_src, float(zs), reweight, RayRamsesWarning)
                      for box_nr, ray_nr in box_ray_nrs] for zs in z_src_shift]
            mine = [ii for ii in range(len(box_ray_nrs)) if ii % world == rank]
            frames = []
            for ii in mine:
                box_nr, ray_nr = box_ray_nrs[ii]
                self.dirs["sim"] = sim_folder_root % box_nr + "/"
                frames.append(self._load_ray_map(self.dirs["sim"] + ray_file_root % ray_nr))
            weights_list = []
            for pw in per_z:
                if any(w is not None for w in pw):
                    wn = [(pw[ii] or (1.0, 1.0))[0] for ii in mine]
                    wd = [(pw[ii] or (1.0, 1.0))[1] for ii in mine]
                    weights_list.append({c: (wn, wd) for c in (columns_z_shift or ["kappa_2"]) if c in columns})
                else:
                    weights_list.append({})
            template = None
            if not frames:
                self.dirs["sim"] = sim_folder_root % box_ray_nrs[0][0] + "/"
                template = self._load_ray_map(self.dirs["sim"] + ray_file_root % box_ray_nrs[0][1])
            return self._stack_columns_many(frames, columns, weights_list, group, template)
        # the weights are decided for every plane on every rank (so all ranks raise alike)
        plane_w = [self._plane_weight(info, box_nr, ray_nr, z_src, z_src_shift, reweight, RayRamsesWarning)
                   for box_nr, ray_nr in box_ray_nrs]
        frames, wnum, wden = [], [], []
        for ii, (box_nr, ray_nr) in enumerate(box_ray_nrs):
            if ii % world != rank:
                continue                                    # plane ii lives on rank ii mod P
            self.dirs["sim"] = sim_folder_root % box_nr + "/"
            frames.append(self._load_ray_map(self.dirs["sim"] + ray_file_root % ray_nr))
            n, d = plane_w[ii] or (1.0, 1.0)
            wnum.append(n)
            wden.append(d)
        weights = {}
        if any(w is not None for w in plane_w):
            # "only of kappa but not of iswrs" (rayramses.py:311)
            weights = {c: (wnum, wden) for c in (columns_z_shift or ["kappa_2"]) if c in columns}
        template = None
        if not frames:                                      # more ranks than planes: shape from the first plane
            self.dirs["sim"] = sim_folder_root % box_ray_nrs[0][0] + "/"
            template = self._load_ray_map(self.dirs["sim"] + ray_file_root % box_ray_nrs[0][1])
        ray_df_sum = self._stack_columns(frames, columns, weights, group, template)
        if dir_out is not None and rank == 0:
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
