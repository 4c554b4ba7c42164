"""On-disk formats either side of the hot path: files go straight into / come straight
out of device memory, without the reference's intermediate numpy / pandas copies.

* DTFE density-grid binaries -> device grid (file -> page-locked staging buffer -> HBM, dtype conversion on the device).  Layout, from the reference's reader
  (particles/hutils/density.py:100-233 ``DensityHeader``, :345-442 ``readDensityData``; the same file
  format is read at rays/voids/tunnels/density.py): every block is framed by two uint64 byte counts;
  block 1 is the 1024-byte header, block 2 the payload of ``totalGrid * components`` values (f4, or i4
  for watershed files), x slowest.  A result split over several files is ``<root>.0 .. <root>.N-1``,
  each with its own header.
* Ray-Ramses per-CPU ASCII outputs -> one HDF5 table per snapshot (rays/rayramses.py:69-148
  ``compress_snapshot``).
* P(k) tables: ``PowerSpectrum3D._save_results`` (power_spectrum_3d.py:228-249) lives with the class.
"""
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

HEADER_BYTES = 1024
_FILL = HEADER_BYTES - 13 * 8 - 18 * 8 - 2 * 8

# the header record of density.py:173-196, field for field (it is a file format)
DENSITY_HEADER_DTYPE = np.dtype([
    ("gridSize", np.uint64, 3), ("totalGrid", np.uint64), ("fileType", np.int32), ("noDensityFiles", np.uint32),
    ("densityFileGrid", np.uint32, 3), ("indexDensityFile", np.uint32), ("box", np.float64, 6),
    ("npartTotal", np.uint64, 6), ("mass", np.float64, 6), ("time", np.float64), ("redshift", np.float64),
    ("BoxSize", np.float64), ("Omega0", np.float64), ("OmegaLambda", np.float64), ("HubbleParam", np.float64),
    ("method", np.uint64), ("fill", "S1", _FILL), ("FILE_ID", np.int64),
])
assert DENSITY_HEADER_DTYPE.itemsize == HEADER_BYTES

# fileType -> (payload dtype, components per grid point): density.py:10-22,64-96 (density 1, velocity 11, its
# gradient 12, divergence 13, shear 14, vorticity 15, std 16, scalar field 20 (6 components) and its gradient 21,
# gravitational potential 50, watershed void index 101 (int32), generic n-component 1000n)
_NO_SCALAR_COMPONENTS = 6
_FILE_TYPES = {1: ("f4", 1), 11: ("f4", 3), 12: ("f4", 9), 13: ("f4", 1), 14: ("f4", 5), 15: ("f4", 3), 16: ("f4", 1),
               20: ("f4", _NO_SCALAR_COMPONENTS), 21: ("f4", 3 * _NO_SCALAR_COMPONENTS), 50: ("f4", 1), 101: ("i4", 1),
               -1: ("f4", 1), 10001: ("f4", 1), 10002: ("f4", 2), 10003: ("f4", 3)}


class DensityFileError(IOError):
    pass


def _framed(f, nbytes_expected=None, what="block"):
    """Position after the leading uint64 byte count of a framed block; returns the count."""
    lead = np.fromfile(f, np.uint64, 1)
    if lead.size != 1:
        raise DensityFileError(f"unexpected end of file before the {what}")
    if nbytes_expected is not None and int(lead[0]) != nbytes_expected:
        raise DensityFileError(f"{what}: leading byte count {int(lead[0])}, expected {nbytes_expected}")
    return int(lead[0])


def read_density_header(path: str) -> Tuple[Dict, int]:
    """Header of a density file as a dict of numpy scalars / arrays, plus the payload's byte count."""
    name = path if os.path.isfile(path) else path + ".0"
    if not os.path.isfile(name):
        raise DensityFileError(f"Cannot find the density binary file. There are no '{path}' or '{name}' files.")
    with open(name, "rb") as f:
        _framed(f, HEADER_BYTES, "header")
        rec = np.fromfile(f, DENSITY_HEADER_DTYPE, 1)[0]
        trail = np.fromfile(f, np.uint64, 1)
        if trail.size != 1 or int(trail[0]) != HEADER_BYTES:
            raise DensityFileError("Error reading the header of the density file: the framing byte counts differ")
        nbytes = np.fromfile(f, np.uint64, 1)
    header = {k: rec[k] for k in DENSITY_HEADER_DTYPE.names}
    return header, (int(nbytes[0]) if nbytes.size else 0)


def data_layout(header: Dict) -> Tuple[str, int]:
    return _FILE_TYPES.get(int(header["fileType"]), ("f4", 1))


def read_density_grid(path: str, to_device: bool = True, dtype=None):
    """``readDensityData`` + the reshape of dtfe.py:70-74 / powmes.py:21-23: returns ``(header, grid)`` with the
    grid ``(gx, gy, gz)`` (or ``(gx, gy, gz, components)``), x slowest.  ``to_device``: the payload is read from
    the file straight into page-locked host memory, copied to the GPU from there and converted to ``dtype`` ON the
    device (no pageable numpy intermediate, no host-side astype); the result is a CUDA tensor.  Otherwise a numpy
    array.

    Results split over several files (``noDensityFiles > 1``) are refused: the reference's own reader places every
    piece at offset 0 of its array (density.py:437, ``startPosition + dataSize`` discards the sum), so there is no
    behaviour of a consumer to be compatible with, and the header does not say along which axis the pieces join."""
    header, nbytes = read_density_header(path)
    kind, comps = data_layout(header)
    if int(header["noDensityFiles"]) > 1:
        raise DensityFileError(f"'{path}': a result split over {int(header['noDensityFiles'])} files is not supported "
                               f"(the reference's reader overwrites offset 0 with every piece, density.py:437)")
    count = int(header["totalGrid"]) * comps
    np_kind = np.dtype(kind)
    if nbytes != count * np_kind.itemsize:
        raise DensityFileError(f"payload of {nbytes} bytes, the header announces {count * np_kind.itemsize}")
    name = path if os.path.isfile(path) else path + ".0"
    host = None
    if to_device:
        import torch
        host = torch.empty(count, dtype=torch.float32 if kind == "f4" else torch.int32, pin_memory=True)
        buf = host.numpy()
    else:
        buf = np.empty(count, dtype=np_kind)
    with open(name, "rb") as f:
        _framed(f, HEADER_BYTES, "header")
        f.seek(HEADER_BYTES, os.SEEK_CUR)
        _framed(f, HEADER_BYTES, "header trailer")
        lead = _framed(f, None, "data block")
        if lead != buf.nbytes:
            raise DensityFileError(f"'{name}': data block of {lead} bytes does not match its header")
        got = f.readinto(memoryview(buf).cast("B"))
        if got != buf.nbytes:
            raise DensityFileError(f"'{name}': truncated data block")
        trail = np.fromfile(f, np.uint64, 1)
        if trail.size != 1 or int(trail[0]) != lead:
            raise DensityFileError(f"'{name}': the byte counts before and after the data differ")
    shape = tuple(int(v) for v in header["gridSize"]) + ((comps,) if comps > 1 else ())
    if not to_device:
        grid = buf.reshape(shape)
        return header, (grid if dtype is None else grid.astype(dtype))
    from . import device as dev
    grid = host.to(dev.device(), non_blocking=True).reshape(shape)
    if dtype is not None and grid.dtype != dtype:
        grid = grid.to(dtype)                      # on the device
    import torch
    torch.cuda.current_stream().synchronize()      # the pinned staging buffer is released on return
    return header, grid


def write_density_grid(path: str, grid, box_size: float, file_type: int = 1, redshift: float = 0.0, **extra) -> None:
    """One-file density binary in the same framing (``writeDensityData``); ``grid``: numpy array or tensor,
    ``(gx, gy, gz[, components])``."""
    if hasattr(grid, "is_cuda") and grid.is_cuda:
        from .device import to_numpy
        arr = to_numpy(grid)                       # through page-locked memory (a pageable destination: a tenth of the rate)
    else:
        arr = grid.cpu().numpy() if hasattr(grid, "cpu") else np.asarray(grid)
    kind, comps = _FILE_TYPES.get(int(file_type), ("f4", 1))
    if (arr.ndim == 4) != (comps > 1) or (arr.ndim == 4 and arr.shape[3] != comps):
        raise DensityFileError(f"file type {file_type} stores {comps} component(s) per grid point")
    rec = np.zeros(1, DENSITY_HEADER_DTYPE)
    rec["gridSize"] = arr.shape[:3]
    rec["totalGrid"] = int(np.prod(arr.shape[:3]))
    rec["fileType"], rec["noDensityFiles"], rec["densityFileGrid"], rec["indexDensityFile"] = file_type, 1, 1, 0
    rec["box"] = [0.0, box_size, 0.0, box_size, 0.0, box_size]
    rec["BoxSize"], rec["redshift"], rec["FILE_ID"] = box_size, redshift, 1
    for key, val in extra.items():
        rec[key] = val
    payload = np.ascontiguousarray(arr, dtype=kind)
    with open(path, "wb") as f:
        for block in (rec, payload):
            np.array([block.nbytes], np.uint64).tofile(f)
            block.tofile(f)
            np.array([block.nbytes], np.uint64).tofile(f)


# --------------------------------------------------------------------------- Ray-Ramses
def compress_rayramses_outputs(cpu_files: Sequence[str], fields: List[str], convert: bool = False,
                               hubble_h: Optional[float] = None) -> pd.DataFrame:
    """The per-CPU ASCII outputs of ONE ray-tracing snapshot -> one table sorted and indexed by ``rayid``
    (rays/rayramses.py:101-143).  ``convert``: comoving distance to [Gpc/h] and the shear combination of
    :121-133, kept term for term (``hubble_h`` = H0/100)."""
    frames = []
    for cpu_file in cpu_files:
        frames.append(pd.read_csv(cpu_file, sep=r"\s+", skipinitialspace=True, names=fields, header=None,
                                  lineterminator="\n"))
    ray_df = pd.concat(frames) if len(frames) > 1 else frames[0]
    if convert:
        if hubble_h is None:
            raise ValueError("convert=True needs hubble_h (H0 / 100)")
        ray_df["chi_co"] /= hubble_h
        ray_df["shear_y"] *= 2.0 * np.sin(ray_df["the_co"])
        gamm1_corr = -ray_df["shear_x"] * np.cos(2.0 * ray_df["phi_co"]) - ray_df["shear_y"] * np.sin(2.0 * ray_df["phi_co"])
        gamm2_corr = -ray_df["shear_x"] * np.sin(2.0 * ray_df["phi_co"]) + ray_df["shear_y"] * np.sin(2.0 * ray_df["phi_co"])
        ray_df["shear_x"] = gamm1_corr
        ray_df["shear_y"] = gamm2_corr
    return ray_df.sort_values(by=["rayid"], axis=0, ascending=True).set_index("rayid")
