"""Oracle: particle -> grid mass assignment (NGP assign, NGP/CIC/TSC paint).

TEST INFRASTRUCTURE — see oracle/__init__.py.  float64 throughout, like the
reference (``np.zeros`` default at power_spectrum_3d.py:142; pmesh default
``f8``).

Reference anchors
-----------------
* NGP *assign* (last write wins):
  /root/reference/src/astrild/power_spectra/power_spectrum_3d.py:140-148
  (twin: bispectra/bispectrum_3d.py:141-154).
* TSC paint + ``/dx**3``:
  /root/reference/src/astrild/particles/hutils/stats_subfind.py:129-132
  -> ``pmesh.pm.ParticleMesh.paint(pos, mass=, resampler="tsc")``.
  pmesh 0.1.55 is not vendored; its published convention (pmesh/_window_imp.c,
  ``_fill_k``) is restated here: grid point ``i`` sits at ``x = i*dx`` (no
  half-cell offset); for a window of support ``S`` the leftmost touched
  point is ``floor(s + shift) - left`` with ``s = x/dx``,
  ``left = (S-1)//2`` and ``shift = 0.5`` for odd ``S`` else ``0``; the
  separable weights are the window evaluated at the distance to each point,
  renormalised to sum to one; indices wrap periodically.
"""
import numpy as np

WINDOWS = ("ngp", "cic", "tsc")


def ngp_assign(x, y, z, values, npar):
    """``value_map[(ix,iy,iz)] = values`` — power_spectrum_3d.py:142-148.

    ``ix = (npar * x).astype(int)`` truncates toward zero; duplicates keep
    the LAST particle's value (numpy fancy assignment, sequential order).
    """
    grid = np.zeros((npar, npar, npar))
    ix = (npar * np.asarray(x, dtype=np.float64)).astype(int)
    iy = (npar * np.asarray(y, dtype=np.float64)).astype(int)
    iz = (npar * np.asarray(z, dtype=np.float64)).astype(int)
    grid[(ix, iy, iz)] = np.asarray(values, dtype=np.float64)
    return grid


def window_1d(s, window):
    """Leftmost index (unwrapped) and the per-point weights along one axis.

    s: positions in grid units (float64).  Returns (i0, w) with
    w.shape == (support, len(s)).
    """
    s = np.asarray(s, dtype=np.float64)
    if window == "ngp":
        # support 1: left = 0, shift = 0.5 -> nearest grid point
        i0 = np.floor(s + 0.5)
        w = np.ones((1, s.size))
    elif window == "cic":
        i0 = np.floor(s)
        f = s - i0
        w = np.stack([1.0 - f, f])
    elif window == "tsc":
        ic = np.floor(s + 0.5)          # nearest grid point
        d = s - ic                       # in [-0.5, 0.5)
        w = np.stack([0.5 * (0.5 - d) ** 2, 0.75 - d * d, 0.5 * (0.5 + d) ** 2])
        i0 = ic - 1.0
    else:
        raise ValueError(window)
    w = w / w.sum(axis=0, keepdims=True)   # pmesh renormalises (sum is 1 analytically)
    return i0.astype(np.int64), w


def paint(pos, mass, nmesh, boxsize, window="cic", out=None, shift=0.0):
    """Mass-weighted scatter-add of particles onto a periodic ``nmesh**3`` grid.

    pos: (Np, 3) in box units [0, boxsize) (any real value wraps);
    mass: (Np,) or None (unit mass).  Returns the float64 grid (C order,
    axis 0 slowest), NOT divided by the cell volume — the caller does the
    ``/dx**3`` of stats_subfind.py:132.  ``shift`` is added to the coordinates in
    grid units (pmesh's affine transform ``s = pos * Nmesh/BoxSize + shift``; nbodykit's
    interlacing paints its second mesh with shift = 0.5).
    """
    pos = np.asarray(pos, dtype=np.float64)
    n = int(nmesh)
    npart = pos.shape[0]
    m = np.ones(npart) if mass is None else np.asarray(mass, dtype=np.float64)
    inv_dx = n / float(boxsize)
    i0 = []
    w = []
    for d in range(3):
        a, b = window_1d(pos[:, d] * inv_dx + shift, window)
        i0.append(a)
        w.append(b)
    support = w[0].shape[0]
    grid = np.zeros(n * n * n) if out is None else out.reshape(-1)
    for a in range(support):
        ia = np.mod(i0[0] + a, n)
        for b in range(support):
            ib = np.mod(i0[1] + b, n)
            wab = m * w[0][a] * w[1][b]
            for c in range(support):
                ic = np.mod(i0[2] + c, n)
                flat = (ia * n + ib) * n + ic
                grid += np.bincount(flat, weights=wab * w[2][c], minlength=n * n * n)
    return grid.reshape(n, n, n)


def paint_loop(pos, mass, nmesh, boxsize, window="cic"):
    """Pure-Python per-particle loop (tiny cases only): independent check of
    :func:`paint` that does not share its vectorised indexing."""
    n = int(nmesh)
    grid = np.zeros((n, n, n))
    inv_dx = n / float(boxsize)
    for p in range(len(pos)):
        mp = 1.0 if mass is None else float(mass[p])
        ax = [window_1d(np.array([pos[p][d] * inv_dx]), window) for d in range(3)]
        for a in range(ax[0][1].shape[0]):
            for b in range(ax[1][1].shape[0]):
                for c in range(ax[2][1].shape[0]):
                    grid[(ax[0][0][0] + a) % n, (ax[1][0][0] + b) % n, (ax[2][0][0] + c) % n] += (
                        mp * ax[0][1][a, 0] * ax[1][1][b, 0] * ax[2][1][c, 0]
                    )
    return grid


def lattice_particles(npside, nmesh, boxsize, seed=20240601, sigma_cells=0.5,
                      shuffle=False, dtype=np.float64):
    """Synthetic "Gaussian-random" particle set of SURVEY.md §8(d):
    q = (i+1/2, j+1/2, k+1/2) * L/npside on an npside**3 lattice (k fastest),
    x = (q + sigma*xi) mod L, xi ~ N(0,1) from Generator(PCG64(seed)),
    sigma = sigma_cells * L/nmesh.  ``shuffle`` applies the fixed permutation
    drawn from seed+1.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    g = (np.arange(npside) + 0.5) * (boxsize / npside)
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    xi = rng.standard_normal(q.shape)
    x = np.mod(q + sigma_cells * (boxsize / nmesh) * xi, boxsize)
    if shuffle:
        perm = np.random.Generator(np.random.PCG64(seed + 1)).permutation(x.shape[0])
        x = x[perm]
    # a float32 cast may round up to exactly boxsize; paint wraps it to cell 0
    return x.astype(dtype)
