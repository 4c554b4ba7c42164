"""Numpy double of HipSlabOps for the CPU (gloo) tests of the slab pipeline's
collective logic.  Test infrastructure: built on the oracle."""
import numpy as np
import torch

from oracle import fftpower as offt, mesh as omesh


class NumpySlabOps:
    dtype = torch.float64
    cdtype = torch.complex128
    device = torch.device("cpu")

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype)

    def empty(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype)

    def paint(self, pos, mass, n, boxsize, window, out, x_start, nx_alloc, check=False):
        full = omesh.paint(pos.numpy(), None if mass is None else mass.numpy(), n, boxsize, window)
        planes = (x_start + np.arange(nx_alloc)) % n
        out.copy_(torch.from_numpy(full[planes]))
        if check and not np.isclose(full.sum(), full[planes].sum(), rtol=1e-12):
            raise RuntimeError("deposits fell outside the slab buffer")
        return out

    def add_into(self, dst, src):
        dst += src

    def fft2d_planes(self, planes, out):
        out.copy_(torch.from_numpy(np.fft.rfft2(planes.numpy(), axes=(1, 2))))
        return out

    def pack(self, spec, out, parts):
        n0, n1, n2 = spec.shape
        out.copy_(spec.reshape(n0, parts, n1 // parts, n2).permute(1, 0, 2, 3).contiguous().reshape(out.shape))
        return out

    def fft1d_axis0(self, block, scale):
        block.copy_(torch.from_numpy(np.fft.fft(block.numpy(), axis=0) * scale))
        return block

    def shell_geometry(self, n, boxsize, i0, i1):
        z = np.zeros((i0[1], i1[1], n // 2 + 1))
        ks, _, nm = offt.project_block(z, n, boxsize, i0[0], i1[0])
        return torch.from_numpy(ks), torch.from_numpy(nm)

    def power_bin(self, block, n, boxsize, i0, i1, psum):
        b = block.numpy()
        p3d = (b * b.conj()).real * boxsize ** 3
        _, ps, _ = offt.project_block(p3d, n, boxsize, i0[0], i1[0])
        psum.copy_(torch.from_numpy(ps))
        return psum

    @staticmethod
    def _dest(pos, n, boxsize, window, parts):
        sgrid = pos.numpy()[:, 0] * (n / boxsize)
        base = np.floor(sgrid if window == "cic" else sgrid + 0.5)
        return (np.mod(base, n).astype(np.int64)) // (n // parts)

    def route_count(self, pos, n, boxsize, window, parts):
        return torch.from_numpy(np.bincount(self._dest(pos, n, boxsize, window, parts), minlength=parts).astype(np.int64))

    def route_scatter(self, pos, mass, n, boxsize, window, parts, counts):
        order = np.argsort(self._dest(pos, n, boxsize, window, parts), kind="stable")
        return pos[torch.from_numpy(order)].contiguous(), None if mass is None else mass[torch.from_numpy(order)].contiguous()
