"""Pins the oracle's rays sub-path against the reference's own known-answer
test values (tests/golden/reference_known_answers.json).  CPU only."""
import json
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kappa as ok

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def _halo():
    g = GOLD["nfw_halo"]
    return {k: np.array(v) for k, v in g["halo"].items()}, g


@pytest.fixture(scope="module")
def dt_map():
    halo, g = _halo()
    return ok.analytic_halo_signal_map(halo, g["extent"], g["direction"], g["suppress"], g["suppression_R"],
                                       g["npix"], "dT")


def test_nfw_dT_map_matches_reference_test(dt_map):
    e = GOLD["nfw_halo"]["dT"]
    assert np.unravel_index(dt_map.argmax(), dt_map.shape) == tuple(e["argmax"])
    npt.assert_almost_equal(dt_map.min(), e["min"], decimal=e["decimals"]["min"])
    npt.assert_almost_equal(dt_map.mean(), e["mean"], decimal=e["decimals"]["mean"])
    npt.assert_almost_equal(dt_map.max(), e["max"], decimal=e["decimals"]["max"])


def test_nfw_alpha_map_matches_reference_test():
    halo, g = _halo()
    m = ok.analytic_halo_signal_map(halo, g["extent"], g["direction"], g["suppress"], g["suppression_R"],
                                    g["npix"], "alpha")
    e = g["alpha"]
    assert np.unravel_index(m.argmax(), m.shape) == tuple(e["argmax"])
    npt.assert_almost_equal(m.min(), e["min"], decimal=e["decimals"]["min"])
    npt.assert_almost_equal(m.mean(), e["mean"], decimal=e["decimals"]["mean"])
    npt.assert_almost_equal(m.max(), e["max"], decimal=e["decimals"]["max"])


def test_kappa0_to_alphas_matches_reference_test():
    g = GOLD["kappa_to_alphas"]
    gg = ok.general_gaussian(100, 1, 10)
    a1, a2 = ok.kappa0_to_alphas(np.outer(gg, gg), g["npix"], np.deg2rad(g["opening_angle_deg"]))
    e = g["alpha_1"]
    npt.assert_almost_equal(a1.min(), e["min"], decimal=e["decimal"])
    npt.assert_almost_equal(a1.mean(), e["mean"], decimal=e["decimal"])
    npt.assert_almost_equal(a1.max(), e["max"], decimal=e["decimal"])
    # the input is symmetric under transposition, so alpha2 must be alpha1 transposed
    npt.assert_allclose(a2, a1.T, rtol=1e-12, atol=1e-15)


def test_gaussian_smoothing_matches_reference_test(dt_map):
    g = GOLD["gaussian_smoothing"]
    for case in g["cases"]:
        sm = ok.gaussian_smooth(dt_map, g["theta_deg"], ok.fwhm_to_sigma(case["fwhm_arcmin"]))
        npt.assert_almost_equal(sm.max() * 1e8, case["max_times_1e8"], decimal=case["decimal"])


def test_unit_conversion_matches_reference_test():
    g = GOLD["unit_conversion"]
    assert ok.C_LIGHT_KMS == g["c_light_km_s"]
    for case in g["cases"]:
        v = ok.convert_code_to_phy_units(case["quantity"], [g["c_light_km_s"] ** case["power"]] * 10)
        assert v[0] == g["expected"]


def test_phi_and_alpha_are_consistent():
    # alpha ~ grad(phi) up to the half-pixel offset the kernels are sampled with
    # ((i + 1/2) * dsx, lensing_funcs.c:52-53): loose finite-difference check only
    nc, bsz = 64, 0.1
    x = np.arange(nc) - nc / 2 + 0.5
    kap = np.exp(-0.5 * (x[:, None] ** 2 + x[None, :] ** 2) / 4.0 ** 2)
    a1, a2 = ok.kappa0_to_alphas(kap, nc, bsz)
    phi = ok.kappa0_to_phi(kap, nc, bsz)
    d = bsz / nc
    sl = slice(8, -8)
    for grad, alpha in ((np.gradient(phi, d, axis=0), a1), (np.gradient(phi, d, axis=1), a2)):
        err = np.sqrt(np.mean((grad[sl, sl] - alpha[sl, sl]) ** 2)) / np.sqrt(np.mean(alpha[sl, sl] ** 2))
        assert err < 0.15
        assert np.corrcoef(grad[sl, sl].ravel(), alpha[sl, sl].ravel())[0, 1] > 0.99


def test_fft_and_real_space_smoothing_agree_in_the_interior():
    rng = np.random.default_rng(0)
    img = np.zeros((256, 256))
    img[96:160, 96:160] = rng.standard_normal((64, 64))
    a = ok.gaussian_smooth(img, 1.0, 0.6, kind="gaussian")
    b = ok.gaussian_smooth(img, 1.0, 0.6, kind="gaussianFFT")
    npt.assert_allclose(a, b, atol=2e-4 * abs(a).max())      # real-space kernel is truncated at 4 sigma


def test_stack_is_sequential_sum_and_weights():
    rng = np.random.default_rng(1)
    planes = [rng.standard_normal((8, 8)) for _ in range(5)]
    tot = ok.kappa_stack(planes)
    ref = planes[0].copy()
    for p in planes[1:]:
        ref = ref + p
    assert np.array_equal(tot, ref)
    xn = np.array([0.0, 100, 200, 300, 400]); xf = xn + 100
    w = ok.kappa_stack(planes, xn, xf, 1100.0, 450.0)
    # last plane lies beyond the shifted source: x_s' is clamped to x_far
    g = ok.kernel_function
    exp = sum(p * g(0.5 * (a + b), max(b, 450.0) if b > 450.0 else 450.0) / g(0.5 * (a + b), 1100.0)
              for p, a, b in zip(planes, xn, xf))
    npt.assert_allclose(w, exp, rtol=1e-13)


def test_gsn_and_pdf_and_reshape():
    g = ok.galaxy_shape_noise(64, 34077)
    assert g.shape == (64, 64) and abs(g.std() - 0.007) < 3e-4
    assert np.array_equal(g, ok.galaxy_shape_noise(64, 34077))
    vals, edges = ok.pdf(g, 50)
    assert len(vals) == 50 and np.isclose((vals * np.diff(edges)).sum(), 1.0)
    v = np.arange(16.0)
    assert np.array_equal(ok.rays_to_map(v), v.reshape(4, 4))


def test_dgd3_window_matches_reference_test(dt_map):
    g = GOLD["dgd3"]
    for case in g["cases"]:
        f = ok.dgd_filter(dt_map, g["theta_deg"], g["theta_i_deg"], case["direction"], order=3)
        x_slice, y_slice = f[:, len(f) // 2], f[len(f) // 2, :]
        assert x_slice.max() == case["x_slice_max"]
        npt.assert_almost_equal(y_slice.max() * 1e7, case["y_slice_max_times_1e7"], decimal=case["decimal"])


def test_locate_peaks_hand_made_map():
    """Strict 8-neighbour maxima of the interior only; plateaus and border pixels are not peaks."""
    a = np.zeros((6, 6))
    a[2, 2] = 1.0                     # isolated peak
    a[4, 3] = a[4, 4] = 0.7           # plateau: neither is strictly larger than the other
    a[0, 5] = 9.0                     # on the border: never a peak
    a[1, 4] = 0.5                     # interior, but its neighbour (0, 5) is larger
    vals, pos = ok.locate_peaks(a, np.array([-1.0, 0.0, 2.0]))
    assert vals.tolist() == [1.0] and pos.tolist() == [[2, 2]]
    vals, _ = ok.locate_peaks(a, np.array([-1.0, 1.0]))       # upper threshold is exclusive
    assert vals.size == 0
    rng = np.random.default_rng(3)
    img = rng.standard_normal((40, 40))
    vals, pos = ok.locate_peaks(img, np.array([-10.0, 10.0]))
    for v, (y, x) in zip(vals, pos):
        nb = img[y - 1:y + 2, x - 1:x + 2].copy()
        nb[1, 1] = -np.inf
        assert v == img[y, x] and v > nb.max()
    centres, counts = ok.wl_peak_counts(img, 8, "normalize")
    assert counts.sum() <= vals.size and len(centres) == 8 and np.all(np.diff(centres) > 0)


def test_flat_power_spectrum_known_answers():
    """White noise of variance s^2 per pixel: P_l = s^2 * pixel area on every annulus; a single plane wave of
    amplitude A lands in one annulus with P = A^2 * area / 4 / (pixels of the annulus) per occupied pixel."""
    rng = np.random.default_rng(6)
    n, theta = 128, 5.0
    img = rng.standard_normal((n, n)) * 0.3
    area = np.deg2rad(theta) ** 2
    lf = 2 * np.pi / np.deg2rad(theta)
    edges = lf * np.arange(4.5, 60.0, 5.0)
    l, p = ok.flat_power_spectrum(img, theta, edges)
    npt.assert_allclose(p, 0.09 * area / n ** 2, rtol=0.12)
    assert np.allclose(l, 0.5 * (edges[:-1] + edges[1:]))
    y, x = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    wave = 0.2 * np.cos(2 * np.pi * (7 * y + 3 * x) / n)
    l, p = ok.flat_power_spectrum(wave, theta, edges)
    k = np.searchsorted(edges, lf * np.sqrt(58.0)) - 1
    pix = ((ok._pixel_l(n, np.deg2rad(theta)) > edges[k]) & (ok._pixel_l(n, np.deg2rad(theta)) <= edges[k + 1])).sum()
    npt.assert_allclose(p[k], (0.2 / 2 * n * n) ** 2 / pix * (np.deg2rad(theta) / n ** 2) ** 2, rtol=1e-10)
    assert np.abs(np.delete(p, k)).max() < 1e-20


def test_flat_bispectrum_brute_force_counts_triangles():
    """Three plane waves closing a triangle give a bispectrum only in the bin holding all three sides."""
    n, theta = 24, 4.0
    y, x = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m1, m2 = (5, 0), (-2, 4)
    m3 = (-(m1[0] + m2[0]), -(m1[1] + m2[1]))                              # (-3, -4): |m| = 5, 4.47, 5
    img = sum(np.cos(2 * np.pi * (a * y + b * x) / n) for a, b in (m1, m2, m3))
    lf = 2 * np.pi / np.deg2rad(theta)
    l, b, ntri = ok.flat_bispectrum_equilateral_brute(img, theta, lf * np.array([2.5, 4.2, 5.5, 8.0]))
    assert ntri[1] > 0 and b[1] > 0 and abs(b[0]) < 1e-12 * b[1] and abs(b[2]) < 1e-12 * b[1]
    # every ordered closed triangle built from +-(m1, m2, m3): 3! orderings x 2 signs, each (n^2/2)^3
    expect = 12 * (n * n / 2.0) ** 3 / ntri[1] * np.deg2rad(theta) ** 4 / float(n) ** 6
    npt.assert_allclose(b[1], expect, rtol=1e-10)


def test_resize_antialiased_properties():
    """The oracle's restatement of SkyArray.resize (parity unpinned: no scikit-image here): a constant map stays
    constant, the same size is the identity, and a factor-2 reduction without the prefilter is the 2 x 2 block mean -
    bilinear samples at pixel centres sit in the middle of each block."""
    from scipy import ndimage
    c = np.full((64, 64), 0.37)
    npt.assert_allclose(ok.resize_antialiased(c, 16), 0.37, rtol=1e-14)
    rng = np.random.default_rng(3)
    m = rng.standard_normal((64, 64))
    assert np.array_equal(ok.resize_antialiased(m, 64), m)
    z = ndimage.zoom(m, (0.5, 0.5), order=1, mode="reflect", grid_mode=True)
    npt.assert_allclose(z, m.reshape(32, 2, 32, 2).mean(axis=(1, 3)), rtol=0, atol=1e-14)
    # the prefilter of a factor-2 reduction has sigma = 1/2
    want = ndimage.zoom(ndimage.gaussian_filter(m, 0.5, mode="mirror"), (0.5, 0.5), order=1, mode="mirror", grid_mode=True)
    assert np.array_equal(ok.resize_antialiased(m, 32), want)


def mirror_resize_by_hand(img, npix):
    """skimage.transform.resize(img, (npix, npix), anti_aliasing=True) written out without ndimage: numpy.pad's
    "reflect" (= ndimage "mirror": d c b | a b c d | c b a) around the map, the truncated, normalised Gaussian of
    sigma = (nin / npix - 1) / 2 as explicit tap sums, then bilinear samples at (o + 1/2) nin / npix - 1/2."""
    nin = img.shape[0]
    sigma = (nin / npix - 1.0) / 2.0
    r = int(4.0 * sigma + 0.5)
    w = np.exp(-0.5 * (np.arange(-r, r + 1) / sigma) ** 2)
    w /= w.sum()
    pad = np.pad(img, r, mode="reflect")
    rows = sum(w[k] * pad[k:k + nin, :] for k in range(2 * r + 1))
    sm = sum(w[k] * rows[:, k:k + nin] for k in range(2 * r + 1))
    c = (np.arange(npix) + 0.5) * nin / npix - 0.5
    i0 = np.clip(np.floor(c).astype(int), 0, nin - 2)
    f = c - i0
    a = sm[i0] * (1 - f)[:, None] + sm[i0 + 1] * f[:, None]
    return a[:, i0] * (1 - f)[None, :] + a[:, i0 + 1] * f[None, :]


def test_resize_antialiased_border_is_mirror_not_reflect():
    """resize's default mode="reflect" is numpy.pad's naming -> ndimage "mirror".  Border pixels of the
    oracle equal a hand-computed mirror-padded convolution and differ from ndimage's "reflect" (edge repeated)."""
    from scipy import ndimage
    rng = np.random.default_rng(12)
    m = rng.standard_normal((96, 96)) + np.linspace(0, 3, 96)[:, None]
    got = ok.resize_antialiased(m, 24)
    npt.assert_allclose(got, mirror_resize_by_hand(m, 24), rtol=0, atol=1e-13)
    wrong = ndimage.zoom(ndimage.gaussian_filter(m, 1.5, mode="reflect"), (0.25, 0.25), order=1, mode="reflect", grid_mode=True)
    assert np.abs(got - wrong)[0].max() > 1e-3 and np.abs(got - wrong)[8:16, 8:16].max() < 1e-12


def test_deflection_to_shear_known_answer():
    """sky_utils.py:342-362 restated: for alpha = (a x, b y) + c (x y, 0) on the pixel lattice np.gradient is exact in the
    interior (linear and bilinear fields), so gamma1 = 0.5 ((1 - d0 a1) - (1 - d1 a2)) and gamma2 = 0.5 (-d0 a2 - d1 a1)
    are known in closed form; axis 0 is the first index."""
    n, h = 24, 0.01
    i, j = np.meshgrid(np.arange(n) * h, np.arange(n) * h, indexing="ij")
    a1 = 0.3 * i + 0.2 * i * j          # d0 a1 = 0.3 + 0.2 j, d1 a1 = 0.2 i
    a2 = -0.1 * j + 0.05 * i            # d0 a2 = 0.05,        d1 a2 = -0.1
    g1, g2 = ok.deflection_to_shear(a1, a2, h)
    npt.assert_allclose(g1, 0.5 * ((1 - (0.3 + 0.2 * j)) - (1 + 0.1)), rtol=0, atol=1e-12)
    npt.assert_allclose(g2, 0.5 * (-0.05 - 0.2 * i), rtol=0, atol=1e-12)
