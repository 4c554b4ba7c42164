"""Numpy double of kappa_shard.HipStackOps for the CPU (gloo) tests of the sharded kappa stack.
Test infrastructure: built on the oracle."""
import numpy as np
import torch

from oracle import kappa as ok


class NumpyStackOps:
    device = torch.device("cpu")

    def to_device(self, a):
        return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def empty(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def stack(self, planes, wnum=None, wden=None, out=None):
        total = None
        for p, plane in enumerate(planes):
            q = plane.numpy().astype(np.float64)
            if wnum is not None:
                q = q * wnum[p] / wden[p]                 # ast_kappa_stack's v * num / den
            total = q.copy() if total is None else total + q
        res = torch.from_numpy(total)
        if out is not None:
            out.copy_(res)
            return out
        return res
