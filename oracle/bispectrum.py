"""Oracle: 3D bispectrum by the FFT (Scoccimarro) estimator, plus a brute-force
triangle sum for tiny grids.

TEST INFRASTRUCTURE — see oracle/__init__.py.  PARITY UNPINNED: the reference's
``Bispectrum3D`` (bispectra/bispectrum_3d.py:165-215) computes a power spectrum
and holds no bispectrum arithmetic or test; its docstring (:42-44) cites
arXiv:1512.07295 / 1506.02729, whose estimator is restated here:

    shells   [m_lo, m_hi) in units of k_F on integer |m| (exact m^2 comparison)
    d_i(x) = sum_{k in i} delta_k e^{ikx}     (delta_k = rfftn/Ng, pmesh norm)
    I_i(x) = sum_{k in i} e^{ikx}
    N_tri(i,j,l) = sum_x I_i I_j I_l / Ng     (integer: closed triangles)
    B(i,j,l)     = L^6 sum_x d_i d_j d_l / sum_x I_i I_j I_l
"""
import numpy as np


def _freq_int(n):
    m = np.arange(n)
    m[m > n // 2] -= n
    return m


def shell_edges(n, width=1, m_min=1, m_max=None):
    m_max = n // 2 if m_max is None else m_max
    return np.arange(m_min, m_max + 1, width)


def _m2_half(n):
    mx = _freq_int(n)
    mz = np.arange(n // 2 + 1)
    return mx[:, None, None] ** 2 + mx[None, :, None] ** 2 + mz[None, None, :] ** 2


def shell_fields(field, edges):
    """(d_i(x), I_i(x)) for every shell [edges[i], edges[i+1])."""
    f = np.asarray(field, dtype=np.float64)
    n = f.shape[0]
    dk = np.fft.rfftn(f) / f.size
    m2 = _m2_half(n)
    ds, iis = [], []
    for lo, hi in zip(edges[:-1], edges[1:]):
        mask = (m2 >= lo * lo) & (m2 < hi * hi)
        ds.append(np.fft.irfftn(dk * mask, s=f.shape, axes=(0, 1, 2)) * f.size)
        iis.append(np.fft.irfftn(mask.astype(np.complex128), s=f.shape, axes=(0, 1, 2)) * f.size)
    return ds, iis


def bispectrum_fft(field, boxsize, edges, triangles):
    """B and N_tri for a list of (i, j, l) shell triplets."""
    ds, iis = shell_fields(field, edges)
    ng = float(np.asarray(field).size)
    out_b, out_n = [], []
    for (i, j, l) in triangles:
        num = np.sum(ds[i] * ds[j] * ds[l])
        den = np.sum(iis[i] * iis[j] * iis[l])
        out_n.append(den / ng)
        out_b.append(boxsize ** 6 * num / den if abs(den) > 0.5 else np.nan)
    return np.array(out_b), np.array(out_n)


def bispectrum_brute_force(field, boxsize, edges, triangles):
    """O(N^6) direct sum over closed triangles k1 + k2 + k3 = 0 on the FULL lattice
    (tiny n only)."""
    f = np.asarray(field, dtype=np.float64)
    n = f.shape[0]
    dk = np.fft.fftn(f) / f.size
    m = _freq_int(n)
    mm = np.stack(np.meshgrid(m, m, m, indexing="ij"), axis=-1).reshape(-1, 3)
    vals = dk.reshape(-1)
    m2 = (mm ** 2).sum(axis=1)
    shell = np.full(len(mm), -1)
    for s, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
        shell[(m2 >= lo * lo) & (m2 < hi * hi)] = s

    def idx_of(v):           # lattice vector -> flat index, aliasing like the DFT
        w = np.mod(v, n)
        return (w[..., 0] * n + w[..., 1]) * n + w[..., 2]

    out_b, out_n = [], []
    for (i, j, l) in triangles:
        a = np.nonzero(shell == i)[0]
        b = np.nonzero(shell == j)[0]
        k3 = -(mm[a][:, None, :] + mm[b][None, :, :])
        i3 = idx_of(k3)
        ok = shell[i3] == l
        ntri = int(ok.sum())
        tot = np.sum((vals[a][:, None] * vals[b][None, :] * vals[i3])[ok])
        out_n.append(ntri)
        out_b.append(boxsize ** 6 * tot.real / ntri if ntri else np.nan)
    return np.array(out_b), np.array(out_n)
