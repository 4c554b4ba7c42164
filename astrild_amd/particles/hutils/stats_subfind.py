"""``SubFind.power_spectrum`` with astrild's API
(src/astrild/particles/hutils/stats_subfind.py:109-153): particles -> TSC paint
-> /dx^3 -> FFTPower, on the GPU."""
import numpy as np
import torch

from ... import device as dev


class SubFind:
    dtype = torch.float64

    @staticmethod
    def power_spectrum(snapshot, objects: str = "subhalo", limits: tuple = None, nbins: int = 512,
                       boxsize: float = 500.0):
        """Real-space halo power spectrum.  ``snapshot`` exposes ``.cat[...]`` and
        ``.header.hubble/.boxsize`` like astrild's read_hdf5.snapshot."""
        if boxsize is None:
            boxsize = snapshot.header.boxsize / 1e3  # [Mpc/h]
        if objects == "subhalo":
            pos_field = snapshot.cat["SubhaloPos"][:] * snapshot.header.hubble / 1e3  # [Mpc/h]
            mass_field = snapshot.cat["SubhaloMass"][:] * snapshot.header.hubble / 1e10
        else:
            raise ValueError(f"objects={objects!r} is not supported")
        dx = boxsize / nbins
        pos = dev.as_device(np.ascontiguousarray(pos_field), SubFind.dtype)
        mass = dev.as_device(np.ascontiguousarray(mass_field), SubFind.dtype)
        # pm.paint(pos, mass=mass, resampler="tsc") / dx**3   (stats_subfind.py:130-132)
        # ... then FFTPower(ArrayMesh(value_map), mode="1d")                 (stats_subfind.py:134-150)
        r = dev.paint_power_1d(pos, mass, nbins, boxsize, "tsc", scale=1.0 / dx ** 3)
        k = np.array(r["k"])
        Pk = np.array(r["power"] - r["shotnoise"])
        return k, Pk
