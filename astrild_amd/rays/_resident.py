"""``SkyArray.data`` with device residency.

The reference keeps its maps as numpy arrays in ``self.data`` (rays/skys/sky_array.py:119-131) and every method reads and
writes them there.  On the GPU that shape costs a PCIe round trip per method: a 4096^2 float64 map is 134 MB each way,
several times what any of the kernels takes.  ``MapStore`` is a dict that may hold a map as a CUDA tensor instead: what a
device operation produces stays in HBM until somebody asks for it as an array.

* ``store[name]`` (and ``get`` / ``items`` / ``values`` / ``pop``) return numpy arrays like the reference's dict.  A map that
  lives on the device is fetched once; from then on the ARRAY is the map (the caller may write into it), the device copy is
  dropped.
* ``store.device(name)`` is what the methods of SkyArray use: the CUDA tensor (float64, contiguous) - the resident one, or an
  upload of the host array (not cached: the array's owner may change it).
* ``store[name] = value`` takes arrays and CUDA tensors alike.
"""
import numpy as np
import torch

from ..device import as_device, to_numpy as fetch


def is_device_map(v) -> bool:
    return isinstance(v, torch.Tensor) and v.is_cuda


def to_host(v):
    """numpy view of a map wherever it lives."""
    return fetch(v) if isinstance(v, torch.Tensor) else v


def to_device(v):
    """float64 contiguous CUDA tensor of a map wherever it lives (a resident tensor is returned as it is: do not write into it)."""
    if isinstance(v, torch.Tensor):
        return v.to(device="cuda", dtype=torch.float64).contiguous()
    return as_device(np.ascontiguousarray(v, dtype=np.float64))


class MapStore(dict):
    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if is_device_map(v):
            v = fetch(v)
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in list(self.keys())]

    def values(self):
        return [self[k] for k in list(self.keys())]

    def pop(self, key, *default):
        if key in self:
            v = self[key]
            dict.__delitem__(self, key)
            return v
        if default:
            return default[0]
        raise KeyError(key)

    def copy(self):
        return MapStore(dict.copy(self))

    def device(self, key):
        return to_device(dict.__getitem__(self, key))

    def resident(self, key) -> bool:
        """True while the map lives in HBM only (nobody has asked for the array yet)."""
        return is_device_map(dict.__getitem__(self, key))

    def __deepcopy__(self, memo):
        import copy
        return MapStore({k: (v.clone() if isinstance(v, torch.Tensor) else copy.deepcopy(v, memo)) for k, v in dict.items(self)})
