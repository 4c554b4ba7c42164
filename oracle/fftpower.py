"""Oracle: 3D r2c + ``FFTPower(mode="1d")`` shell binning.

TEST INFRASTRUCTURE — see oracle/__init__.py.

Reference anchors
-----------------
/root/reference/src/astrild/power_spectra/power_spectrum_3d.py:181-224 calls
``ArrayMesh(value_map, Nmesh=, BoxSize=, compensated=False)`` and
``FFTPower(mesh, mode="1d", kmin=2*pi/L)`` and returns
``k = r.power["k"]``, ``Pk = r.power["power"].real - attrs["shotnoise"]``.
nbodykit 0.3.14 (poetry.lock:333-336) is not vendored; its published
algorithm (nbodykit/algorithms/fftpower.py: ``FFTPower.run``,
``_compute_3d_power``, ``project_to_basis``) is restated:

* ``ArrayMesh`` hands the array over unchanged (no 1+delta normalisation,
  ``compensated``/``interlaced``/``window`` are inert attrs); shotnoise = 0.
* pmesh ``r2c`` is normalised by 1/Ng: delta_k = (1/Ng) sum_x f(x) e^{-ikx}.
* P3D = delta1_k * conj(delta2_k) * L^3, DC mode set to 0.
* dk = 2*pi/L, kedges = arange(kmin, pi*N/L + dk/2, dk) = k_F*{1..N/2}
  -> N/2-1 bins, left-closed; modes with |k| < k_F (DC) or >= (N/2) k_F dropped.
* only the half spectrum (last axis 0..N/2) is visited; modes with
  0 < i_z < N/2 get Hermitian weight 2, the others weight 1.
* per bin: k = sum(w*|k|)/sum(w), power = sum(w*P3D)/sum(w), modes = sum(w).

Bin membership: nbodykit compares float64 ``kx^2+ky^2+kz^2`` against float64
``kedges**2``.  Lattice vectors whose integer |m|^2 is a perfect square sit
exactly on an edge and land either side by one ulp there, depending on L.
``binning="float64"`` (the default, like the HIP library's) restates that
comparison - ``digitize(k2, edges**2)`` with ``k_i = k_F * m_i`` and
``edges = arange(k_F, pi*N/L + k_F/2, k_F)``; ``binning="integer"`` is the exact
rule ``bin = isqrt(mx^2+my^2+mz^2) - 1``.  The two differ only for those
edge vectors (tests/test_oracle_mesh.py shows where).
"""
import numpy as np


def r2c(field):
    """pmesh-normalised forward transform: rfftn / Ng."""
    f = np.asarray(field, dtype=np.float64)
    return np.fft.rfftn(f) / f.size


def _freq_int(n):
    m = np.arange(n)
    m[m > n // 2] -= n          # index N/2 stays +N/2 (sign irrelevant for |k|)
    return m


def isqrt_array(a):
    r = np.floor(np.sqrt(a.astype(np.float64))).astype(np.int64)
    r = np.where(r * r > a, r - 1, r)
    r = np.where((r + 1) * (r + 1) <= a, r + 1, r)
    return r


DEFAULT_BINNING = "float64"


def project_1d(p3d_half, n, boxsize, binning=None):
    """Shell-bin a half-spectrum ``(n, n, n//2+1)`` of P3D values.

    Returns (ksum, psum, modes) — raw weighted sums, float64/complex/int64,
    each of length n//2-1 — so partial results can be added across slabs.
    """
    nb = n // 2 - 1
    kf = 2.0 * np.pi / boxsize
    mx = _freq_int(n)
    mz = np.arange(n // 2 + 1)
    m2 = (mx[:, None, None] ** 2 + mx[None, :, None] ** 2 + mz[None, None, :] ** 2).astype(np.int64)
    w = np.where((mz > 0) & (mz < n // 2), 2, 1)[None, None, :] * np.ones_like(m2)
    b = shell_index(mx, mx, mz, m2, n, boxsize, binning)
    ok = (b >= 0) & (b < nb)
    bb = b[ok]
    ww = w[ok].astype(np.float64)
    kk = kf * np.sqrt(m2[ok].astype(np.float64))
    pv = p3d_half[ok]
    ksum = np.bincount(bb, weights=ww * kk, minlength=nb)
    psum = np.bincount(bb, weights=ww * pv.real, minlength=nb).astype(np.complex128)
    if np.iscomplexobj(pv):
        psum += 1j * np.bincount(bb, weights=ww * pv.imag, minlength=nb)
    modes = np.bincount(bb, weights=ww, minlength=nb).astype(np.int64)
    return ksum, psum, modes


def shell_index(m0, m1, mz, m2, n, boxsize, binning=None):
    """Shell of every mode of a block (-1: dropped).  m0, m1, mz: integer frequencies of the three axes,
    m2 = their squared norm on the block."""
    binning = binning or DEFAULT_BINNING
    nb = n // 2 - 1
    if binning == "integer":
        return isqrt_array(m2) - 1
    if binning != "float64":
        raise ValueError(binning)
    kf = 2.0 * np.pi / boxsize
    k0, k1, kz = kf * m0.astype(np.float64), kf * m1.astype(np.float64), kf * mz.astype(np.float64)
    k2 = k0[:, None, None] ** 2 + k1[None, :, None] ** 2 + kz[None, None, :] ** 2
    edges = np.arange(kf, np.pi * n / boxsize + kf / 2, kf)[:n // 2]
    b = np.digitize(k2.ravel(), edges ** 2).reshape(k2.shape) - 1
    return np.where(b >= nb, -1, b)


def project_block(p3d_block, n, boxsize, i0_start, i1_start, binning=None):
    """Shell sums of a (c0, c1, n//2+1) block of the half spectrum whose first two
    axes start at global indices i0_start / i1_start (integer binning) — the
    per-rank piece of a slab-decomposed spectrum."""
    nb = n // 2 - 1
    kf = 2.0 * np.pi / boxsize
    c0, c1, nz = p3d_block.shape
    assert nz == n // 2 + 1
    m0 = _freq_int(n)[i0_start:i0_start + c0]
    m1 = _freq_int(n)[i1_start:i1_start + c1]
    mz = np.arange(nz)
    m2 = (m0[:, None, None] ** 2 + m1[None, :, None] ** 2 + mz[None, None, :] ** 2).astype(np.int64)
    w = np.where((mz > 0) & (mz < n // 2), 2, 1)[None, None, :] * np.ones_like(m2)
    b = shell_index(m0, m1, mz, m2, n, boxsize, binning)
    ok = (b >= 0) & (b < nb)
    ww = w[ok].astype(np.float64)
    ksum = np.bincount(b[ok], weights=ww * kf * np.sqrt(m2[ok].astype(np.float64)), minlength=nb)
    psum = np.bincount(b[ok], weights=ww * p3d_block[ok].real, minlength=nb)
    modes = np.bincount(b[ok], weights=ww, minlength=nb).astype(np.int64)
    return ksum, psum, modes


def fftpower_1d(field1, boxsize, field2=None, binning=None):
    """``FFTPower(first, mode="1d", kmin=2*pi/L[, second])`` on in-memory grids.

    Returns dict(k, power (complex), modes (int64), shotnoise=0.0); empty bins
    are NaN like nbodykit's 0/0.
    """
    f1 = np.asarray(field1, dtype=np.float64)
    n = f1.shape[0]
    assert f1.shape == (n, n, n) and n % 2 == 0
    c1 = r2c(f1)
    c2 = c1 if field2 is None else r2c(field2)
    p3d = c1 * np.conj(c2)
    p3d[0, 0, 0] = 0.0
    p3d *= float(boxsize) ** 3
    ksum, psum, modes = project_1d(p3d, n, boxsize, binning)
    with np.errstate(invalid="ignore", divide="ignore"):
        k = ksum / modes
        power = psum / modes
    return {"k": k, "power": power, "modes": modes, "shotnoise": 0.0}


def power_spectrum_3d(value_map1, boxsize, value_map2=None):
    """(k, Pk) exactly as ``PowerSpectrum3D._power_spectrum_3d`` returns them
    (power_spectrum_3d.py:223-226): ``Pk = power.real - shotnoise``."""
    r = fftpower_1d(value_map1, boxsize, value_map2)
    return np.array(r["k"]), np.array(r["power"].real - r["shotnoise"])


def brute_force_mode_counts(n):
    """Mode count per shell from the FULL (not half) integer k-lattice, for
    the bit-exact histogram test.  Nyquist planes
    appear once (index N/2 only), like the DFT."""
    nb = n // 2 - 1
    m = _freq_int(n)
    m2 = m[:, None, None] ** 2 + m[None, :, None] ** 2 + m[None, None, :] ** 2
    counts = np.zeros(nb, dtype=np.int64)
    for v in m2.ravel():
        r = int(np.floor(np.sqrt(float(v))))
        while r * r > v:
            r -= 1
        while (r + 1) * (r + 1) <= v:
            r += 1
        if 1 <= r <= nb:
            counts[r - 1] += 1
    return counts


# ---------------------------------------------------------------- f-1: catalogue meshes
def _window_factor(w, window, interlaced):
    """nbodykit CatalogMesh's compensation kernels (nbodykit/source/mesh/catalog.py, un-vendored, restated):
    interlaced -> CompensateCIC / CompensateTSC = sinc(w/2)^p; otherwise the aliased-shot-noise forms
    CompensateCICShotnoise / CompensateTSCShotnoise."""
    if interlaced:
        return np.sinc(0.5 * w / np.pi) ** (3 if window == "tsc" else 2)
    s = np.sin(0.5 * w) ** 2
    return (1 - s + 2.0 / 15 * s ** 2) ** 0.5 if window == "tsc" else (1 - 2.0 / 3 * s) ** 0.5


def catalog_mesh_complex(pos, mass, nmesh, boxsize, window="tsc", interlaced=True, compensated=True):
    """``CatalogMesh(cat, Nmesh=, BoxSize=, window=, interlaced=, compensated=).compute(mode="complex")`` -
    what the keywords of power_spectrum_3d.py:197-212 do to a PARTICLE source (for the reference's ArrayMesh they
    are inert): paint -> 1 + delta (divide by the mean weight per cell) -> r2c (1/Ng) -> interlacing with a second
    paint shifted by half a cell, c = (c1 + c2 exp(i (wx+wy+wz)/2)) / 2 -> divide by the window.
    Returns (half spectrum, shotnoise = L^3 sum w^2 / (sum w)^2)."""
    from . import mesh as omesh
    n = int(nmesh)
    pos = np.asarray(pos, dtype=np.float64)
    m = np.ones(len(pos)) if mass is None else np.asarray(mass, dtype=np.float64)
    norm = n ** 3 / m.sum()
    c = r2c(omesh.paint(pos, m, n, boxsize, window) * norm)
    w0 = 2 * np.pi * _freq_int(n) / n
    wz = 2 * np.pi * np.arange(n // 2 + 1) / n
    wx, wy, wz = w0[:, None, None], w0[None, :, None], wz[None, None, :]
    if interlaced:
        c2 = r2c(omesh.paint(pos, m, n, boxsize, window, shift=0.5) * norm)
        c = 0.5 * c + 0.5 * c2 * np.exp(0.5j * (wx + wy + wz))
    if compensated:
        c = c / (_window_factor(wx, window, interlaced) * _window_factor(wy, window, interlaced)
                 * _window_factor(wz, window, interlaced))
    return c, float(boxsize) ** 3 * (m ** 2).sum() / m.sum() ** 2


def catalog_power_1d(pos1, mass1, nmesh, boxsize, window="tsc", interlaced=True, compensated=True, pos2=None, mass2=None):
    """``FFTPower(first=CatalogMesh(...), mode="1d", kmin=2 pi / L[, second=CatalogMesh(...)])``."""
    n = int(nmesh)
    c1, sn = catalog_mesh_complex(pos1, mass1, n, boxsize, window, interlaced, compensated)
    c2 = c1
    if pos2 is not None:
        c2, _ = catalog_mesh_complex(pos2, mass2, n, boxsize, window, interlaced, compensated)
        sn = 0.0
    p3d = c1 * np.conj(c2)
    p3d[0, 0, 0] = 0.0
    p3d *= float(boxsize) ** 3
    ksum, psum, modes = project_1d(p3d, n, boxsize)
    with np.errstate(invalid="ignore", divide="ignore"):
        return {"k": ksum / modes, "power": psum / modes, "modes": modes, "shotnoise": sn}
