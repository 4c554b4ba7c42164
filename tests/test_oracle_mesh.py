"""Known-answer tests for the oracle's 3D path.

The reference has no test for paint / FFTPower (parity unpinned), so the
oracle is anchored by analytic results here.  CPU only.
"""
import numpy as np
import pytest

from oracle import mesh, fftpower


@pytest.mark.parametrize("window", ["ngp", "cic", "tsc"])
def test_mass_conservation(window):
    rng = np.random.default_rng(1)
    pos = rng.uniform(-3.0, 25.0, size=(2000, 3))      # also exercises wrap of out-of-box values
    mass = rng.uniform(0.5, 2.0, size=2000)
    g = mesh.paint(pos, mass, 16, 20.0, window)
    assert g.sum() == pytest.approx(mass.sum(), rel=1e-13)
    assert g.min() >= 0.0


@pytest.mark.parametrize("window", ["ngp", "cic", "tsc"])
def test_vectorised_paint_matches_particle_loop(window):
    rng = np.random.default_rng(2)
    pos = rng.uniform(0.0, 10.0, size=(300, 3))
    mass = rng.uniform(0.5, 2.0, size=300)
    a = mesh.paint(pos, mass, 8, 10.0, window)
    b = mesh.paint_loop(pos, mass, 8, 10.0, window)
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-13)


def test_cic_stencil_single_particle_with_wrap():
    # particle at s = (7.25, 0.5, 3.0) on an 8-grid: x wraps 7 -> 0
    n, L = 8, 8.0
    g = mesh.paint(np.array([[7.25, 0.5, 3.0]]), None, n, L, "cic")
    exp = np.zeros((n, n, n))
    for ix, wx in ((7, 0.75), (0, 0.25)):
        for iy, wy in ((0, 0.5), (1, 0.5)):
            exp[ix, iy, 3] += wx * wy
    np.testing.assert_allclose(g, exp, atol=1e-15)


def test_tsc_stencil_single_particle_with_wrap():
    # s = 0.2 -> nearest point 0, d = 0.2 -> weights (0.045, 0.71, 0.245) on (-1, 0, 1)
    n, L = 8, 8.0
    g = mesh.paint(np.array([[0.2, 4.0, 7.5]]), None, n, L, "tsc")
    wx = {7: 0.5 * 0.3 ** 2, 0: 0.75 - 0.04, 1: 0.5 * 0.7 ** 2}
    wy = {3: 0.125, 4: 0.75, 5: 0.125}
    # s = 7.5 -> floor(8.0) = 8 (half rounds up), d = -0.5 -> (0.5, 0.5, 0) on (7, 0, 1)
    wz = {7: 0.5, 0: 0.5, 1: 0.0}
    exp = np.zeros((n, n, n))
    for ix, a in wx.items():
        for iy, b in wy.items():
            for iz, c in wz.items():
                exp[ix, iy, iz] += a * b * c
    np.testing.assert_allclose(g, exp, atol=1e-15)


@pytest.mark.parametrize("window", ["cic", "tsc"])
def test_integer_cell_translation_invariance(window):
    rng = np.random.default_rng(3)
    n, L = 16, 32.0
    pos = rng.uniform(0, L, size=(500, 3))
    g0 = mesh.paint(pos, None, n, L, window)
    shift = np.array([3, 5, 15]) * (L / n)
    g1 = mesh.paint(pos + shift, None, n, L, window)
    np.testing.assert_allclose(g1, np.roll(g0, (3, 5, 15), axis=(0, 1, 2)), atol=1e-12)


def test_ngp_assign_last_write_wins_and_truncation():
    x = np.array([0.10, 0.99, 0.10, 0.49999])
    y = np.array([0.20, 0.00, 0.20, 0.5])
    z = np.array([0.30, 0.50, 0.30, 0.75])
    v = np.array([1.0, 2.0, 3.0, 4.0])
    g = mesh.ngp_assign(x, y, z, v, 4)
    assert g[0, 0, 1] == 3.0          # duplicate cell: the later particle wins
    assert g[3, 0, 2] == 2.0
    assert g[1, 2, 3] == 4.0
    assert np.count_nonzero(g) == 3


@pytest.mark.parametrize("mvec", [(1, 0, 0), (0, 2, 1), (3, 3, 2), (0, 0, 5), (2, -3, 4)])
def test_plane_wave_power(mvec):
    # delta = A cos(2 pi m.x / L)  ->  |delta_k|^2 = A^2/4 at +-m  ->  P = A^2 L^3 / 4
    # in exactly the shell floor(|m|), zero elsewhere (checks 1/Ng, L^3, Hermitian doubling)
    n, L, A = 16, 100.0, 0.3
    x = np.arange(n) / n
    ph = 2 * np.pi * (mvec[0] * x[:, None, None] + mvec[1] * x[None, :, None] + mvec[2] * x[None, None, :])
    f = 5.0 + A * np.cos(ph)           # DC offset must not leak anywhere
    r = fftpower.fftpower_1d(f, L)
    shell = int(np.floor(np.sqrt(sum(c * c for c in mvec)))) - 1
    expected = np.zeros(n // 2 - 1)
    # the shell average spreads the two modes' power over all modes of the shell
    expected[shell] = 2 * (A * A / 4) * L ** 3 / r["modes"][shell]
    np.testing.assert_allclose(r["power"].real, expected, atol=1e-9 * L ** 3)
    np.testing.assert_allclose(r["power"].imag, 0, atol=1e-9 * L ** 3)


def test_parseval():
    rng = np.random.default_rng(4)
    n, L = 16, 10.0
    f = rng.standard_normal((n, n, n))
    c = fftpower.r2c(f)
    w = np.full(n // 2 + 1, 2.0)
    w[0] = w[-1] = 1.0
    assert (w * np.abs(c) ** 2).sum() == pytest.approx((f ** 2).mean(), rel=1e-12)


@pytest.mark.parametrize("n", [16, 32])
def test_mode_counts_bit_exact_vs_full_lattice(n):
    f = np.zeros((n, n, n))
    r = fftpower.fftpower_1d(f, 1.0, binning="integer")
    np.testing.assert_array_equal(r["modes"], fftpower.brute_force_mode_counts(n))
    assert len(r["modes"]) == n // 2 - 1
    # the float64 rule (the default) moves edge vectors only: same total except for the |m| = N/2 vectors it may admit
    rf = fftpower.fftpower_1d(f, 1.0)
    m = fftpower._freq_int(n)
    m2 = m[:, None, None] ** 2 + m[None, :, None] ** 2 + np.arange(n // 2 + 1)[None, None, :] ** 2
    w = np.where((np.arange(n // 2 + 1) > 0) & (np.arange(n // 2 + 1) < n // 2), 2, 1)[None, None, :] * np.ones_like(m2)
    on_last_edge = int(w[m2 == (n // 2) ** 2].sum())
    assert r["modes"].sum() <= rf["modes"].sum() + int(w[m2 == 1].sum()) and rf["modes"].sum() <= r["modes"].sum() + on_last_edge


def test_k_values_and_white_noise_level():
    rng = np.random.default_rng(5)
    n, L = 32, 50.0
    f = rng.standard_normal((n, n, n))
    r = fftpower.fftpower_1d(f, L)
    kf = 2 * np.pi / L
    assert np.all(r["k"] >= kf * np.arange(1, n // 2)) and np.all(r["k"] < kf * np.arange(2, n // 2 + 1))
    # unit-variance white noise: P = L^3/Ng in every shell within sampling error
    err = np.abs(r["power"].real / (L ** 3 / n ** 3) - 1.0) * np.sqrt(r["modes"] / 2.0)
    assert err.max() < 5.0


def test_cross_spectrum_is_hermitian_weighted_real_part():
    rng = np.random.default_rng(6)
    n, L = 16, 10.0
    f1 = rng.standard_normal((n, n, n))
    f2 = rng.standard_normal((n, n, n))
    a = fftpower.fftpower_1d(f1, L, f2)["power"]
    b = fftpower.fftpower_1d(f2, L, f1)["power"]
    np.testing.assert_allclose(a.real, b.real, rtol=1e-12, atol=1e-12)
    auto = fftpower.fftpower_1d(f1, L, f1)["power"]
    np.testing.assert_allclose(auto, fftpower.fftpower_1d(f1, L)["power"], rtol=1e-14)


def test_integer_vs_float64_binning_differ_only_on_exact_edges():
    # documents the 1-ulp edge ambiguity of the float64 digitize (SURVEY.md §7 hard part 3)
    n, L = 32, 1000.0
    z = np.zeros((n, n, n // 2 + 1))
    _, _, mi = fftpower.project_1d(z, n, L, "integer")
    _, _, mf = fftpower.project_1d(z, n, L, "float64")
    assert mi.sum() == fftpower.brute_force_mode_counts(n).sum()
    # any disagreement is confined to lattice vectors with perfect-square |m|^2
    assert abs(int(mi.sum()) - int(mf.sum())) <= 6 * 2      # outermost edge modes only


def _sinusoidal_catalogue(n, L, m, amp, per_cell=4):
    """A density 1 + amp cos(2 pi m.x / L) carried by a regular lattice of `per_cell`^3 particles per cell with
    sinusoidal weights: fine enough that only the window, not the sampling, shapes the measured mode."""
    g = (np.arange(n * per_cell) + 0.5) * (L / (n * per_cell))
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    pos = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    mass = 1.0 + amp * np.cos(2 * np.pi * (m[0] * pos[:, 0] + m[1] * pos[:, 1] + m[2] * pos[:, 2]) / L)
    return pos, mass


@pytest.mark.parametrize("window", ["cic", "tsc"])
def test_compensation_restores_a_plane_wave_amplitude(window):
    """delta = A cos(k.x): painting multiplies the mode by the window, sinc(k_i H / 2)^p per axis; dividing by it
    (compensated + interlaced) gives back P = A^2 L^3 / 4 in the mode's shell; uncompensated the power is W^2 lower."""
    n, L, amp, m = 16, 100.0, 0.2, (3, -2, 4)
    pos, mass = _sinusoidal_catalogue(n, L, m, amp, per_cell=8)      # images of the particle lattice: (u / (u - 8))^p
    shell = int(np.floor(np.sqrt(sum(v * v for v in m)))) - 1
    full = fftpower.catalog_power_1d(pos, mass, n, L, window, interlaced=True, compensated=True)
    raw = fftpower.catalog_power_1d(pos, mass, n, L, window, interlaced=False, compensated=False)
    exact = 2 * (amp ** 2 / 4) * L ** 3 / full["modes"][shell]
    # what is left is the particle lattice's own images, (u / (u - 8))^p of the amplitude per axis: 0.6 % (CIC)
    tol = 8e-3 if window == "cic" else 1e-3
    assert full["power"][shell].real == pytest.approx(exact, rel=tol)
    p = 3 if window == "tsc" else 2
    w = np.prod([np.sinc(v / n) ** p for v in m])
    assert raw["power"][shell].real == pytest.approx(exact * w * w, rel=tol)
    others = np.delete(full["power"].real, shell)
    assert np.abs(others).max() < 1e-3 * exact


def test_interlacing_cancels_the_first_alias():
    """A weight pattern at m = (N - 3, 0, 0) - beyond the Nyquist frequency - aliases onto m = -3 on the grid.  The
    image comes with the sign (-1)^(sum of the alias vector) relative to the half-cell shifted paint, so the
    interlaced combination cancels it (Sefusatti et al. 2016)."""
    n, L, amp = 16, 100.0, 0.2
    pos, mass = _sinusoidal_catalogue(n, L, (n - 3, 0, 0), amp, per_cell=4)
    plain = fftpower.catalog_power_1d(pos, mass, n, L, "tsc", interlaced=False, compensated=False)
    inter = fftpower.catalog_power_1d(pos, mass, n, L, "tsc", interlaced=True, compensated=False)
    shell = 3 - 1
    w = np.sinc((n - 3) / n) ** 3                                   # the TSC window at the pattern's own frequency
    expect = 2 * (amp ** 2 / 4) * L ** 3 * w * w / plain["modes"][shell]
    assert plain["power"][shell].real == pytest.approx(expect, rel=0.05)     # the alias is there ...
    assert inter["power"][shell].real < 1e-6 * plain["power"][shell].real     # ... and interlacing removes it


def test_catalogue_shot_noise_level():
    """Poisson catalogue: P(k) = L^3 / N on every shell once the window is divided out (interlaced + compensated)."""
    rng = np.random.default_rng(2)
    n, L, npart = 32, 200.0, 200000
    pos = rng.uniform(0, L, size=(npart, 3))
    r = fftpower.catalog_power_1d(pos, None, n, L, "tsc", interlaced=True, compensated=True)
    assert r["shotnoise"] == pytest.approx(L ** 3 / npart)
    ratio = r["power"].real / r["shotnoise"]
    # exponentially distributed mode powers: the shell mean scatters by 1 / sqrt(independent modes)
    assert np.all(np.abs(ratio - 1) < 4.5 / np.sqrt(r["modes"] / 2.0))
    assert abs(np.average(ratio, weights=r["modes"]) - 1) < 0.02
